"""Host-side index tables for the Video-Swin kernels (tiny int tables, built once per shape and cached).

The reference materialises pad -> roll -> window_partition -> attention -> window_reverse -> roll -> crop
(video_swin.py:206-245) and 2x2 strided slices + cat (video_swin.py:273-284) as full tensor copies; here they
are gather maps consumed by the LayerNorm / GEMM-epilogue kernels:
  window_map   : window slot -> source token (or -1 for a pad slot)
  region_ids   : per-window region label; shift mask = (region[i] != region[j]) ? -100 : 0  (video_swin.py:292-307)
  rc_codes     : relative-position index as rc[i] - rc[j] + rc0                                (video_swin.py:123-137,155)
  merge_map    : PatchMerging 2x2 gather in (h0w0, h1w0, h0w1, h1w1) order                     (video_swin.py:280-284)
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
