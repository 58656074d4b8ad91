"""LayerNorm backward (+ residual add) at the Swin stage-3 shape (50 176 x 512) and the fusion shape (69 120 x 768): operands warm in the
Infinity Cache (one buffer set, repeated) against cold (8 buffer sets in rotation, 1.6 GB), in order and through the window map."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from pytorch_empirical_mvm_amd import kernels as K, swin_index as SI

dev = torch.device("cuda:0")
K.set_workspace(torch.empty(64 << 20, device=dev, dtype=torch.uint8))


def rnd(*s):
    return torch.randn(*s, device=dev).to(torch.bfloat16)


def timed(fns, reps=40):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (B, L, C) in [(32, 1568, 512), (160, 432, 768), (32, 6272, 256)]:
    M = B * L
    g = torch.randn(C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    src = None
    if L in (1568, 6272):
        dims = (8, 14, 14) if L == 1568 else (8, 28, 28)
        wm, _ = SI.window_map(*dims, (8, 7, 7), (0, 3, 3))
        src = torch.from_numpy(wm.astype(np.int32)).to(dev)
    sets = []
    for _ in range(8):
        x, dy, add = rnd(M, C), rnd(M, C), rnd(M, C)
        y, mean, rstd = K.layernorm_fwd(x, g, g, 1e-5)
        sets.append((x, dy, add, mean, rstd, torch.empty_like(x)))
    def plain(s):
        return lambda: K.layernorm_bwd(s[1], s[0], g, s[3], s[4], dg, db, dX_add=s[2], dX=s[5])
    def mapped(s):
        return lambda: K.layernorm_bwd(s[1], s[0], g, s[3], s[4], dg, db, dX_add=s[2], dX=s[5], rows_in=M, nseg=1, pad_mode=0, src=src, rows_out_per_batch=L, rows_in_per_batch=L)
    def fwd(s):
        return lambda: K.layernorm_fwd(s[0], g, g, 1e-5)
    byt = 4 * M * C * 2
    for name, mk, nb in (("bwd+add in order", plain, byt), ("bwd+add window map", mapped if src is not None else None, byt), ("fwd", fwd, 2 * M * C * 2)):
        if mk is None:
            continue
        tw = timed([mk(sets[0])])
        tc = timed([mk(s) for s in sets])
        print(f"{M} x {C} {name:20s}: warm {tw * 1e3:7.1f} us {nb / tw / 1e9:6.2f} TB/s   cold {tc * 1e3:7.1f} us {nb / tc / 1e9:6.2f} TB/s")
