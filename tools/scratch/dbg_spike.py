#!/usr/bin/env python3
"""which rows of the win4 forward are wrong under the spike input (round 6 debugging of check_attn_window_spike, un-shifted)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gpu_check as G
from pytorch_empirical_mvm_amd import kernels as K
torch.manual_seed(0)
dims, B, heads = (8, 14, 14), 2, 2
pairs_all = [(0, 0, 5, 390), (1, 1, 200, 201), (7, 0, 391, 388), (2, 1, 17, 3), (3, 0, 300, 310), (2, 0, 40, 13 * 16 + 5), (3, 1, 14 * 16 + 2, 19 * 16 + 9), (1, 0, 9 * 16 + 1, 24 * 16 + 3)]
for sel in ([0, 1, 2, 3, 4], [5], [6], [7], list(range(8))):
    torch.manual_seed(0)
    N, nW, rc_t, rc0, reg_t = G._win_problem(dims, B, heads, False)
    C_ = heads * 32
    nseq = B * nW
    qkv = G.rnd(nseq * N, 3 * C_, scale=1.0)
    q3 = qkv.view(nseq, N, 3, heads, 32)
    for i in sel:
        sq, hh, qi, kj = pairs_all[i]
        q3[sq, kj, 1, hh] = (q3[sq, qi, 0, hh].float() * 8.0).to(G.BF)
    table = (torch.randn((2 * 8 - 1) * 13 * 13, heads, device="cuda") * 0.5)
    out, lse = K.attention_fwd(qkv, nseq, N, heads, 32, 0, 32 ** -0.5, q_off=0, k_off=C_, v_off=2 * C_, bias_table=table, rc=rc_t, rc0=rc0, region=reg_t, n_win=nW, win_layout=1)
    _, _, s_, ref = G._win_ref(qkv, table, nseq, N, heads, rc_t, rc0, reg_t, B)
    err = (out.float() - ref.detach()).view(nseq, N, heads, 32).abs().amax(-1)          # (seq, slot, head)
    bad = torch.nonzero(err > 0.05)
    print(f"pairs {sel}: {bad.shape[0]} bad (seq, slot, head) rows; max err {float(err.max()):.3f}")
    for b_ in bad[:12].tolist():
        sq, sl, hh = b_
        row = s_[sq, hh, sl].detach()
        print(f"   seq {sq} head {hh} query slot {sl} (tile {sl // 16}): err {float(err[sq, sl, hh]):.3f}  row max {float(row.max()):.1f} at key {int(row.argmax())} (tile {int(row.argmax()) // 16}), second {float(row.topk(2).values[1]):.1f}")
