#!/usr/bin/env python3
"""What the residual / map / row-scale epilogue of the Swin stage-3 fc2 GEMM costs, term by term (47 040 x 512 x 2 048, the
`bias+res+map+rs` class of tools/gemm_shapes.py at 0.51-0.58 of the vendor GEMM)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_empirical_mvm_amd import kernels as K
dev = "cuda:0"
def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
B, L = 30, 1568
for (M, N, Kd) in [(B * L, 512, 2048), (B * L, 512, 512), (69120, 768, 768)]:
    A, W = rnd(M, Kd), rnd(N, Kd)
    bias = torch.randn(N, device=dev)
    res = rnd(M, N)
    rs = torch.rand(B, device=dev) + 0.5
    perm = torch.from_numpy(np.random.RandomState(0).permutation(L).astype(np.int32)).to(dev)
    rpb = M // B
    forms = [("plain", lambda: K.gemm(A, W)), ("bias", lambda: K.gemm(A, W, bias=bias)), ("res", lambda: K.gemm(A, W, resid=res)),
             ("bias+res", lambda: K.gemm(A, W, bias=bias, resid=res)),
             ("bias+res+rs", lambda: K.gemm(A, W, bias=bias, resid=res, row_scale=rs, rows_per_scale=rpb, scale_bias_only=True))]
    if M == B * L:
        forms.append(("bias+res+rs+map", lambda: K.gemm(A, W, bias=bias, resid=res, row_scale=rs, rows_per_scale=L, scale_bias_only=True, row_map=perm, map_len=L, map_stride=L, out_rows=M)))
    for name, fn in forms:
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"  {name:16s} {M}x{N}x{Kd}: {ms * 1e3:7.1f} us {2.0 * M * N * Kd / ms / 1e9:7.1f} TF")
    t0 = torch.matmul(A, W.t()); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): torch.matmul(A, W.t())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"  {'hipBLASLt plain':16s} {M}x{N}x{Kd}: {ms * 1e3:7.1f} us {2.0 * M * N * Kd / ms / 1e9:7.1f} TF")
