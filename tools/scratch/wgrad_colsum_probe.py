#!/usr/bin/env python3
"""what does the fused bias gradient (vmvm_gemm_desc.colsum) cost the weight-gradient GEMM?  (round 6)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_empirical_mvm_amd import kernels as K
dev = "cuda"
ws = torch.empty(192 << 20, device=dev, dtype=torch.uint8)
K.set_workspace(ws)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (Mo, No, Kt) in [(3072, 768, 69120), (768, 3072, 69120), (2304, 768, 69120), (768, 768, 69120), (2048, 512, 47040), (512, 2048, 47040), (1536, 512, 47040), (512, 512, 47040)]:
    dy = (torch.randn(Kt, Mo, device=dev) * 0.1).to(torch.bfloat16)
    x = (torch.randn(Kt, No, device=dev) * 0.1).to(torch.bfloat16)
    gw = torch.zeros(Mo, No, device=dev)
    gb = torch.zeros(Mo, device=dev)
    a = t(lambda: K.gemm(dy, x, a_kmajor=False, b_kmajor=False, M=Mo, N=No, K=Kt, out=gw, accumulate=True, workspace=ws))
    b = t(lambda: K.gemm(dy, x, a_kmajor=False, b_kmajor=False, M=Mo, N=No, K=Kt, out=gw, accumulate=True, colsum=gb, workspace=ws))
    fl = 2.0 * Mo * No * Kt
    print(f"dW [{Mo} x {No}], K = {Kt}: without colsum {a:7.1f} us ({fl / a / 1e6:6.0f} TF)   with {b:7.1f} us ({fl / b / 1e6:6.0f} TF)   +{(b / a - 1) * 100:.1f} %")
