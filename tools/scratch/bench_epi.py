#!/usr/bin/env python3
"""Microbenchmark of the epilogue-heavy GEMM forms of the step (fc1 + bias + GELU + pre-activation; fc2 dgrad x GELU')."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_empirical_mvm_amd import kernels as K
dev = "cuda:0"
def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
for (M, N, Kd) in [(55296, 3072, 768), (50176, 2048, 512), (200704, 1024, 256), (802816, 512, 128)]:
    A, W = rnd(M, Kd), rnd(N, Kd)
    bias = torch.randn(N, device=dev)
    pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    aux = rnd(M, N)
    forms = (("plain", lambda: K.gemm(A, W)), ("bias", lambda: K.gemm(A, W, bias=bias)), ("bias+gelu+pre", lambda: K.gemm(A, W, bias=bias, act=1, out_preact=pre)),
             ("gelu' * aux", lambda: K.gemm(A, W, act=3, aux=aux)))
    for name, fn in forms:
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"  {name:14s} {M}x{N}x{Kd}: {ms:.3f} ms {2.0 * M * N * Kd / ms / 1e9:7.1f} TF")
