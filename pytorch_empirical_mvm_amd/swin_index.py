"""Host-side index tables for the Video-Swin kernels (tiny int tables, built once per shape and cached).

The reference materialises pad -> roll -> window_partition -> attention -> window_reverse -> roll -> crop
(video_swin.py:206-245) and 2x2 strided slices + cat (video_swin.py:273-284) as full tensor copies; here they
are gather maps consumed by the LayerNorm / GEMM-epilogue kernels:
  window_map   : window slot -> source token (or -1 for a pad slot)
  region_ids   : per-window region label; shift mask = (region[i] != region[j]) ? -100 : 0  (video_swin.py:292-307)
  rc_codes     : relative-position index as rc[i] - rc[j] + rc0                                (video_swin.py:123-137,155)
  merge_map    : PatchMerging 2x2 gather in (h0w0, h1w0, h0w1, h1w1) order                     (video_swin.py:280-284)
  win3_perm    : the token order INSIDE an (8,7,7) window that the `win_layout = 1` attention kernels assume (the order is free:
                 the same gather map serves LayerNorm, the un-gather epilogue and the backward; rc / region are per-slot tables)
"""
from functools import lru_cache

import numpy as np


def get_window_size(x_size, window_size, shift_size=None):
    """video_swin.py:95-108: clamp window (and zero the shift) on axes not larger than the window."""
    ws = list(window_size)
    ss = list(shift_size) if shift_size is not None else None
    for i, x in enumerate(x_size):
        if x <= window_size[i]:
            ws[i] = x
            if ss is not None:
                ss[i] = 0
    return tuple(ws) if ss is None else (tuple(ws), tuple(ss))


def _partition(a, ws):
    D, H, W = a.shape
    a = a.reshape(D // ws[0], ws[0], H // ws[1], ws[1], W // ws[2], ws[2])
    return a.transpose(0, 2, 4, 1, 3, 5).reshape(-1, ws[0] * ws[1] * ws[2])


@lru_cache(maxsize=None)
def window_map(D, H, W, ws, ss):
    """int32 [nW*N]: source token d*H*W+h*W+w of each window slot after pad+roll(-ss)+partition, -1 = pad."""
    Dp, Hp, Wp = (-(-D // ws[0]) * ws[0], -(-H // ws[1]) * ws[1], -(-W // ws[2]) * ws[2])
    idx = np.full((Dp, Hp, Wp), -1, dtype=np.int64)
    idx[:D, :H, :W] = np.arange(D * H * W).reshape(D, H, W)
    if any(ss):
        idx = np.roll(idx, shift=(-ss[0], -ss[1], -ss[2]), axis=(0, 1, 2))
    return np.ascontiguousarray(_partition(idx, ws).reshape(-1).astype(np.int32)), (Dp, Hp, Wp)


@lru_cache(maxsize=None)
def region_ids(Dp, Hp, Wp, ws, ss):
    """uint8 [nW][N] region label of every window slot (None when the block is not shifted)."""
    if not any(ss):
        return None
    lab = np.zeros((Dp, Hp, Wp), dtype=np.int64)
    cnt = 0
    for d in (slice(-ws[0]), slice(-ws[0], -ss[0]), slice(-ss[0], None)):
        for h in (slice(-ws[1]), slice(-ws[1], -ss[1]), slice(-ss[1], None)):
            for w in (slice(-ws[2]), slice(-ws[2], -ss[2]), slice(-ss[2], None)):
                lab[d, h, w] = cnt
                cnt += 1
    return np.ascontiguousarray(_partition(lab, ws).astype(np.uint8))


@lru_cache(maxsize=None)
def rc_codes(n_tokens, win_cfg):
    """rc[i] (int32 [N]) and rc0 with relative_position_index[i,j] == rc[i]-rc[j]+rc0 for i,j < N.
    Tokens are numbered in the CONFIGURED window's d-major order, exactly like the reference's [:N,:N] slice."""
    Wd, Wh, Ww = win_cfg
    i = np.arange(n_tokens)
    d, h, w = i // (Wh * Ww), (i // Ww) % Wh, i % Ww
    a, b = (2 * Wh - 1) * (2 * Ww - 1), (2 * Ww - 1)
    rc = (d * a + h * b + w).astype(np.int32)
    rc0 = (Wd - 1) * a + (Wh - 1) * b + (Ww - 1)
    return rc, int(rc0)


@lru_cache(maxsize=None)
def merge_map(D, H, W):
    """int32 [D*H2*W2*4]: source token of each of the 4 concatenated segments (-1 = zero pad row/col)."""
    H2, W2 = (H + 1) // 2, (W + 1) // 2
    out = np.full((D, H2, W2, 4), -1, dtype=np.int64)
    for s, (dh, dw) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        hh = np.arange(H2) * 2 + dh
        ww = np.arange(W2) * 2 + dw
        for d in range(D):
            src = d * H * W + hh[:, None] * W + ww[None, :]
            valid = (hh[:, None] < H) & (ww[None, :] < W)
            out[d, :, :, s] = np.where(valid, src, -1)
    return np.ascontiguousarray(out.reshape(-1).astype(np.int32)), (D, H2, W2)


# ---- token order inside an (8,7,7) window for the win_layout = 1 attention kernels ------------------------------------------
# Slot t = tile * 16 + l * 8 + d: a 16-token MFMA tile holds TWO (h, w) positions x the 8 temporal slices, d fastest.
#  * d fastest makes the relative-position bias of a tile pair four 8 x 8 TOEPLITZ blocks (the table index is
#    (dq - dk + 7) * 169 + rho with rho = 13 (hq - hk + 6) + (wq - wk + 6), video_swin.py:123-137): a lane's four consecutive keys
#    are four consecutive entries of the 15-entry row rho, i.e. ONE 16-byte LDS read per score tile instead of a register-resident
#    bias block per wave.
#  * the (h, w) positions are ordered REGION-major for the (0,3,3) shift (video_swin.py:292-307: the mask separates h < 4 | h >= 4
#    and w < 4 | w >= 4 in the windows of the last window row / column): classes A (h<4,w<4: 16 positions = 8 tiles), B (h<4,w>=4:
#    12 = 6 tiles), C (h>=4,w<4: 12 = 6 tiles), D (h>=4,w>=4: 9 = 4.5 tiles) start at tokens 0 / 128 / 224 / 320 -- every tile lies
#    in ONE region of every window type, so a (query tile, key tile) pair is either fully live or fully masked (-100: probability
#    exactly 0 in f32) and the masked pairs are skipped: 49 % of the tile pairs of an edge window, 74 % of a corner window.
#  * the two positions of a tile are w-neighbours (rho step 1) or h-neighbours (rho step 13), so a lane's table row is one of two
#    lane-constant bases plus a per-tile immediate.
def _win3_positions():
    pos = [(h, w) for h in range(4) for w in range(4)]                                   # A
    pos += [(h, w) for w in (4, 5, 6) for h in range(4)]                                 # B (pairs along h)
    pos += [(h, w) for h in (4, 5, 6) for w in range(4)]                                 # C
    pos += [(4, 4), (4, 5), (5, 4), (5, 5), (6, 4), (6, 5), (4, 6), (5, 6), (6, 6)]      # D
    return pos


WIN3_POS = _win3_positions()          # 49 (h, w) positions in slot order; csrc/attn_win3.h holds the same table (checked by a test)


def win3_ok(ws, ss):
    """the geometry the win_layout = 1 kernels are instantiated for: window (8,7,7), un-shifted or the (0,3,3) shift"""
    return tuple(ws) == (8, 7, 7) and tuple(ss) in ((0, 0, 0), (0, 3, 3))


@lru_cache(maxsize=None)
def win3_perm():
    """int64 [392]: slot of the d-major window order (d*49 + h*7 + w) that lands in slot t of the win_layout = 1 order"""
    perm = np.empty(392, dtype=np.int64)
    for i, (h, w) in enumerate(WIN3_POS):
        for d in range(8):
            perm[i * 8 + d] = d * 49 + h * 7 + w
    return perm
