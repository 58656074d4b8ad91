#!/usr/bin/env python3
"""launch the step's dominant weight-gradient GEMMs for rocprofv3 counter passes (round 6)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_empirical_mvm_amd import kernels as K
dev = "cuda"
ws = torch.empty(192 << 20, device=dev, dtype=torch.uint8)
K.set_workspace(ws)
for (Mo, No, Kt) in [(3072, 768, 69120), (768, 3072, 69120), (2048, 512, 47040)]:
    dy = (torch.randn(Kt, Mo, device=dev) * 0.1).to(torch.bfloat16)
    x = (torch.randn(Kt, No, device=dev) * 0.1).to(torch.bfloat16)
    gw = torch.zeros(Mo, No, device=dev)
    gb = torch.zeros(Mo, device=dev)
    for _ in range(4):
        K.gemm(dy, x, a_kmajor=False, b_kmajor=False, M=Mo, N=No, K=Kt, out=gw, accumulate=True, colsum=gb, workspace=ws)
# and the k-major x k-major plain class for comparison (fusion fc1 dgrad shape)
A = torch.randn(69120, 3072, device=dev).to(torch.bfloat16); B = torch.randn(768, 3072, device=dev).to(torch.bfloat16)
for _ in range(4):
    K.gemm(A, B)
torch.cuda.synchronize()
