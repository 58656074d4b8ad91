#!/usr/bin/env python3
"""Launch the bench.py roofline kernels for rocprofv3 passes: the fusion FFN fc1 GEMM (bias + GELU + saved 8-bit GELU' code;
M = 32 clips x (1 + 4) sequences x 432 tokens = 69120, N = 3072, K = 768) and fused clip + AdamW over a 225 M parameter arena."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_empirical_mvm_amd import kernels as K
M, N, Kd = 32 * 5 * 432, 3072, 768
A = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
B = torch.randn(N, Kd, device="cuda").to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
pre = torch.empty(M, N, device="cuda", dtype=torch.uint8)          # 8-bit GELU' codes (vmvm_gemm_desc.aux_code8)
for _ in range(5):
    K.gemm(A, B, bias=bias, act=1, out_preact=pre, code8=True)
n = 225_086_976
p, g = torch.randn(n, device="cuda") * 0.02, torch.randn(n, device="cuda") * 1e-3
m, v, sh = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda", dtype=torch.bfloat16)
ss = torch.ones(1, device="cuda")
for i in range(3):
    K.adamw(p, g, m, v, sh, lr=1e-5, weight_decay=1e-3, beta1=0.9, beta2=0.98, eps=1e-8, step=i + 1, sumsq_t=ss, max_grad_norm=1.0, grad_scale=1.0)
torch.cuda.synchronize()
