#!/usr/bin/env python3
"""Small-output weight gradients (few tiles, very long K): dispatcher's choice vs forced kernels, stand-alone."""
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
ws = torch.empty(256 << 20, device=dev, dtype=torch.uint8)
K.set_workspace(ws)
for (Nout, Kin, tok) in [(512, 512, 50176), (128, 128, 802816), (512, 128, 802816), (128, 512, 802816), (384, 128, 802816), (1024, 256, 200704), (256, 256, 200704)]:
    dy, x = rnd(tok, Nout), rnd(tok, Kin)
    g = torch.zeros(Nout, Kin, device=dev)
    for name, kw in (("auto", {}), ("pers", dict(variant=6)), ("pp", dict(variant=7))):
        try:
            us = t(lambda: K.gemm(dy, x, a_kmajor=False, b_kmajor=False, M=Nout, N=Kin, K=tok, out=g, accumulate=True, **kw))
            print(f"{Nout}x{Kin}x{tok} {name:5s}: {us:7.1f} us  {2.0*Nout*Kin*tok/us/1e6:6.0f} TF  {(Nout+Kin)*tok*2/us/1e3:6.0f} GB/s")
        except RuntimeError as e:
            print(f"{Nout}x{Kin}x{tok} {name}: unsupported")
