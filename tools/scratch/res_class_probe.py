"""bias + residual (+ dropout) class at the fusion attention-output shape: auto dispatch (128x128 persistent kernel) against the
ping-pong kernel (variant 7) and the plain GEMM of the shape"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from pytorch_empirical_mvm_amd import kernels as K
dev = torch.device("cuda:0")


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (M, N, Kd) in [(69120, 768, 768), (69120, 768, 3072), (55296, 768, 768)]:
    A, B = rnd(M, Kd), rnd(N, Kd, scale=0.05)
    bias, r = torch.randn(N, device=dev), rnd(M, N)
    for name, kw in (("plain", {}), ("bias+res", dict(bias=bias, resid=r)), ("bias+res+drop", dict(bias=bias, resid=r, dropout_p=0.1, seed=3, offset=77))):
        row = []
        for v in (0, 6, 7):
            try:
                row.append(f"variant {v}: {timed(lambda: K.gemm(A, B, variant=v, **kw)):7.1f} us")
            except Exception as e:
                row.append(f"variant {v}: n/a")
        print(f"{M} x {N} x {Kd} {name:14s} " + "   ".join(row))
