"""Host index tables vs the oracle's tensor-level restatement of the reference ops (CPU)."""
import numpy as np
import pytest
import torch

from oracle import violet_ref as R
from pytorch_empirical_mvm_amd import swin_index as SI

CASES = [((8, 14, 14), (8, 7, 7), (0, 3, 3)), ((8, 56, 56), (8, 7, 7), (0, 0, 0)), ((12, 24, 20), (8, 7, 7), (4, 3, 3)),
         ((4, 7, 7), (8, 7, 7), (4, 3, 3)), ((16, 12, 12), (8, 12, 12), (4, 6, 6)), ((3, 5, 6), (2, 3, 3), (1, 1, 1))]


@pytest.mark.parametrize("dims,win,shift", CASES)
def test_window_map_and_regions(dims, win, shift):
    D, H, W = dims
    ws, ss = SI.get_window_size(dims, win, shift)
    assert (ws, ss) == R.get_window_size(dims, win, shift)
    m, (Dp, Hp, Wp) = SI.window_map(D, H, W, ws, ss)
    x = torch.arange(D * H * W, dtype=torch.float32).view(1, D, H, W, 1) + 1.0        # 0 = pad after F.pad
    xp = torch.nn.functional.pad(x, (0, 0, 0, Wp - W, 0, Hp - H, 0, Dp - D))
    if any(ss):
        xp = torch.roll(xp, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
    ref = R.window_partition(xp, ws).reshape(-1).long().numpy() - 1
    np.testing.assert_array_equal(m, ref)
    real = m[m >= 0]
    assert sorted(real.tolist()) == list(range(D * H * W))                            # every token exactly once
    reg = SI.region_ids(Dp, Hp, Wp, ws, ss)
    if any(ss):
        mask = R.compute_mask(Dp, Hp, Wp, ws, ss)
        mine = np.where(reg[:, :, None] != reg[:, None, :], -100.0, 0.0)
        np.testing.assert_array_equal(mine, mask.numpy())
    else:
        assert reg is None


@pytest.mark.parametrize("win,n", [((8, 7, 7), 392), ((8, 7, 7), 196), ((8, 12, 12), 1152), ((2, 3, 3), 18), ((8, 7, 7), 100)])
def test_rc_codes(win, n):
    rc, rc0 = SI.rc_codes(n, win)
    ref = R.relative_position_index(win)[:n, :n].numpy()
    np.testing.assert_array_equal(rc[:, None] - rc[None, :] + rc0, ref)


@pytest.mark.parametrize("dims", [(8, 56, 56), (2, 7, 5), (3, 4, 4)])
def test_merge_map(dims):
    D, H, W = dims
    m, (D2, H2, W2) = SI.merge_map(D, H, W)
    x = torch.arange(D * H * W, dtype=torch.float32).view(1, D, H, W, 1) + 1.0
    if H % 2 or W % 2:
        x = torch.nn.functional.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    cat = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1)
    np.testing.assert_array_equal(m, cat.reshape(-1).long().numpy() - 1)


def test_win3_layout_tables():
    """win_layout = 1 (include/vmvm.h): the slot order inside an (8,7,7) window is a permutation, every 16-token tile lies in ONE mask
    region of every window of the (0,3,3) shift (video_swin.py:292-307) with the class boundaries at tiles 8 / 14 / 20, the two (h, w)
    positions of a tile are w- or h-neighbours, and the C++ geometry table (csrc/attn_win3.h) lists the same positions."""
    import os
    import re
    from pytorch_empirical_mvm_amd import swin_index as SI
    pm = SI.win3_perm()
    assert sorted(pm.tolist()) == list(range(392))
    assert SI.win3_ok((8, 7, 7), (0, 3, 3)) and SI.win3_ok((8, 7, 7), (0, 0, 0)) and not SI.win3_ok((8, 7, 7), (4, 3, 3)) and not SI.win3_ok((4, 7, 7), (0, 3, 3))
    for (H, W) in ((14, 14), (28, 28), (56, 56)):
        reg = SI.region_ids(8, H, W, (8, 7, 7), (0, 3, 3))[:, pm]
        for w in range(reg.shape[0]):
            tiles = [set(reg[w, 16 * t:16 * t + 16].tolist()) for t in range(25)]
            assert all(len(x) == 1 for x in tiles), (H, W, w)
            ids = [next(iter(x)) for x in tiles]
            for lo, hi in ((0, 8), (8, 14), (14, 20), (20, 25)):
                assert len(set(ids[lo:hi])) == 1, (H, W, w, ids)
    pos = SI.WIN3_POS
    for t in range(24):
        (h0, w0), (h1, w1) = pos[2 * t], pos[2 * t + 1]
        assert (h1 - h0, w1 - w0) in ((0, 1), (1, 0)), (t, pos[2 * t], pos[2 * t + 1])
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytorch_empirical_mvm_amd", "csrc", "attn_win3.h")).read()
    nums = lambda name: [int(x) for x in re.findall(r"\d+", re.sub(r"//[^\n]*", "", re.search(name + r"\[50\] = \{([^}]*)\}", hdr, re.S).group(1)))]
    assert list(zip(nums("PH"), nums("PW")))[:49] == pos
