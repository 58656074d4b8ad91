#!/usr/bin/env python3
"""Launch the bench.py roofline kernel (fusion FFN fc1 + GELU GEMM, M=69120 N=3072 K=768) for rocprofv3 --pmc passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_empirical_mvm_amd import kernels as K
M, N, Kd = 32 * 5 * 432, 3072, 768
A = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
B = torch.randn(N, Kd, device="cuda").to(torch.bfloat16)
bias = torch.zeros(N, device="cuda")
pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    K.gemm(A, B, bias=bias, act=1, out_preact=pre)
torch.cuda.synchronize()
