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
