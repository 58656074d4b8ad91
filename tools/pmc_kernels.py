#!/usr/bin/env python3
"""Tiny driver for rocprofv3 --pmc passes: launches each interesting kernel shape a few times (no timing)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_empirical_mvm_amd import kernels as K, swin_index as SI
dev = "cuda"
def rnd(*s): return torch.randn(*s, device=dev).to(torch.bfloat16)
which = sys.argv[1:] or ["gemm", "attn"]
if "gemm" in which:
    M, N, Kd = 69120, 3072, 768
    A, B = rnd(M, Kd), rnd(N, Kd)
    At, Bt = A.t().contiguous(), B.t().contiguous()
    dy = rnd(M, N); gw = torch.zeros(N, Kd, device=dev)
    ws = torch.empty(256 << 20, device=dev, dtype=torch.uint8)
    for _ in range(3):
        K.gemm(A, B, variant=3)                                   # NT 128^2
        K.gemm(A, Bt, b_kmajor=False, variant=3)                  # NN
        K.gemm(At, Bt, a_kmajor=False, b_kmajor=False, variant=3) # TN
        K.gemm(A, B, variant=4)                                   # NT 256^2
        K.gemm(dy, A, a_kmajor=False, b_kmajor=False, M=N, N=Kd, K=M, out=gw, accumulate=True, workspace=ws, variant=3)
if "attn" in which:
    B_, heads, N = 32, 16, 392
    nW = 4; nseq = B_ * nW; C_ = heads * 32
    qkv = rnd(nseq * N, 3 * C_)
    rc, rc0 = SI.rc_codes(N, (8, 7, 7)); rc_t = torch.from_numpy(rc).to(dev)
    table = torch.randn(2535, heads, device=dev) * 0.1
    reg_np = SI.region_ids(8, 14, 14, (8, 7, 7), (0, 3, 3))
    layout = int(os.environ.get("VMVM_PMC_LAYOUT", "1"))
    if layout:                                                    # the win_layout = 1 token order (win3 kernels)
        import numpy as np
        pm = SI.win3_perm()
        rc_t = torch.from_numpy(np.ascontiguousarray(rc[pm])).to(dev)
        reg_np = np.ascontiguousarray(reg_np[:, pm])
    reg = torch.from_numpy(reg_np).to(dev) if os.environ.get("VMVM_PMC_SHIFTED", "1") == "1" else None
    kw = dict(q_off=0, k_off=C_, v_off=2 * C_, bias_table=table, rc=rc_t, rc0=rc0, region=reg, n_win=nW, win_layout=layout)
    dtab = torch.zeros_like(table)
    for _ in range(3):
        out, lse = K.attention_fwd(qkv, nseq, N, heads, 32, 0, 32 ** -0.5, **kw)
        K.attention_bwd(rnd(nseq * N, C_), qkv, out, lse, nseq, N, heads, 32, 0, 32 ** -0.5, dbias_table=dtab, **kw)
if "bert" in which:                                               # fusion-encoder attention at the C2 shape (160 sequences x 12 heads x 432, head_dim 64, dropout 0.1)
    nseq, Lq, heads, Hd = 160, 432, 12, 768
    qkv = rnd(nseq * Lq, 3 * Hd)
    km = torch.ones(nseq, Lq, dtype=torch.uint8, device=dev)
    kw = dict(q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km, dropout_p=0.1, seed=1)
    if os.environ.get("VMVM_PMC_DROPMASK", "0") == "1":            # the forward records its dropout decisions, the backward reads them
        kw["drop_mask"] = K.attention_drop_mask(nseq, Lq, heads, 64, 1, 0.1, dev)
    for _ in range(3):
        out, lse = K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, **kw)
        K.attention_bwd(rnd(nseq * Lq, Hd), qkv, out, lse, nseq, Lq, heads, 64, 1, 0.125, **kw)
torch.cuda.synchronize()
