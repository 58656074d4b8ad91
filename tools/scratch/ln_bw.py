#!/usr/bin/env python3
"""LayerNorm kernels alone at the step's shapes: achieved HBM GB/s (algorithmic bytes / HIP-event time).  Development probe."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from pytorch_empirical_mvm_amd import kernels as K, swin_index as SI
dev = "cuda"
K.set_workspace(torch.empty(256 << 20, device=dev, dtype=torch.uint8))
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def rnd(*s): return torch.randn(*s, device=dev).to(torch.bfloat16)
B = 32
rows = []
# fusion layers: M = 160 * 432
for (M, C, drop) in [(69120, 768, 0.1), (69120, 768, 0.0)]:
    x, dy = rnd(M, C), rnd(M, C)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    y, mean, rstd = K.layernorm_fwd(x, g, b, 1e-12)
    us = t(lambda: K.layernorm_fwd(x, g, b, 1e-12))
    print(f"ln_fwd  fusion M={M} C={C}: {us:7.1f} us  {2 * M * C * 2 / us / 1e3:7.0f} GB/s")
    us = t(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dg, db, want_dX2=drop > 0, dropout_p=drop, seed=1, offset=0))
    nb = (3 + (drop > 0)) * M * C * 2
    print(f"ln_bwd  fusion M={M} C={C} drop={drop}: {us:7.1f} us  {nb / us / 1e3:7.0f} GB/s")
# Swin blocks: norm1 through the window map (gather), norm2 plain, with dX_add
for (dims, C) in [((8, 56, 56), 128), ((8, 28, 28), 256), ((8, 14, 14), 512), ((8, 7, 7), 1024)]:
    D, H, W = dims
    L = D * H * W
    for shifted in (False, True):
        ws, ss = SI.get_window_size(dims, (8, 7, 7), (4, 3, 3) if shifted else (0, 0, 0))
        wm, _ = SI.window_map(D, H, W, ws, ss)
        Lp = wm.size
        src = torch.from_numpy(wm.astype(np.int32)).to(dev)
        x, dy, add = rnd(B * L, C), rnd(B * Lp, C), rnd(B * L, C)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        kw = dict(M=B * Lp, C_=C, nseg=1, src=src, rows_out_per_batch=Lp, rows_in_per_batch=L, pad_mode=0)
        y, mean, rstd = K.layernorm_fwd(x, g, b, 1e-5, **kw)
        us = t(lambda: K.layernorm_fwd(x, g, b, 1e-5, **kw))
        print(f"ln_fwd  swin C={C} L={L} Lp={Lp} gather: {us:7.1f} us  {(B * L + B * Lp) * C * 2 / us / 1e3:7.0f} GB/s")
        us = t(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dg, db, rows_in=B * L, nseg=1, src=src, rows_out_per_batch=Lp, rows_in_per_batch=L, pad_mode=0, dX_add=add))
        print(f"ln_bwd  swin C={C} L={L} Lp={Lp} gather+add: {us:7.1f} us  {(B * Lp + 3 * B * L) * C * 2 / us / 1e3:7.0f} GB/s")
    x, dy, add = rnd(B * L, C), rnd(B * L, C), rnd(B * L, C)
    y, mean, rstd = K.layernorm_fwd(x, g, b, 1e-5)
    us = t(lambda: K.layernorm_fwd(x, g, b, 1e-5))
    print(f"ln_fwd  swin C={C} L={L} plain: {us:7.1f} us  {2 * B * L * C * 2 / us / 1e3:7.0f} GB/s")
    us = t(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dX_add=add))
    print(f"ln_bwd  swin C={C} L={L} plain+add: {us:7.1f} us  {4 * B * L * C * 2 / us / 1e3:7.0f} GB/s")
