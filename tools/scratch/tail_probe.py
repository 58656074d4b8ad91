#!/usr/bin/env python3
"""How much of the 128x128 persistent kernel's time on the Swin stage-3 shapes is the partial last round (1568 tiles on 512 workgroups)?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_empirical_mvm_amd import kernels as K
dev = "cuda"
def rnd(*s): return torch.randn(*s, device=dev).to(torch.bfloat16)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, Kd) in [(512, 2048), (512, 512), (2048, 512), (1536, 512)]:
    B = rnd(N, Kd)
    for M in (49152, 50176, 65536):
        A = rnd(M, Kd)
        us = t(lambda: K.gemm(A, B))
        tiles = (M // 128) * (N // 128)
        print(f"M={M} N={N} K={Kd}: {us:7.1f} us  {2.0*M*N*Kd/us/1e6:7.1f} TF  tiles={tiles} rounds={tiles/512:.2f}  us/round-up={us/-(-tiles//512):.1f}")
