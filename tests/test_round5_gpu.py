"""Round-5 parity cases (VERDICT r04 "next round" item 2).

* `test_full_width_c2_batch32_equals_eight_batches_of_four`: THE BENCHMARKED BATCH.  Every earlier full-width C2 test ran B = 2 or 4; B = 32
  alone exercises O = 4 with 128 VTM rows through `_bert_layer_qrow`, the M = 69 120 merged fusion pass and its whole-round / remainder
  GEMM split, clip-chunked attention workgroups, DropPath kept-lists with K-tile padding at realistic counts, and the 537 MB
  dropout-decision record.  A B = 32 oracle run is out of reach of a CPU test, so the reference is the library itself at B = 4, which IS
  pinned to the oracle (tests/test_parity_gpu.py, tests/test_round3_gpu.py): one forward-backward at B = 32 with block-diagonal VTM
  negatives against eight B = 4 runs on the same clips (tools/dp_check.py::concat_check's bookkeeping: the eight groups' MLM-target and
  covered-patch counts are equalised, so the B = 32 mean losses are the means of the groups' means and the gradient arena is the mean of
  the groups' arenas).  Eval mode, then train mode with fixed DropPath scales (dense, compact and empty kept-lists all occur) and hidden
  / attention dropout off.  Asserted: per-clip `out_mtm` / `out_mvm` / `out_vtm` equal within bf16 noise (cosine >= 0.9995 per output,
  max |diff| <= 5e-2 of the output scale: the two batch sizes take different GEMM tilings / split-K plans, so 36 layers of bf16 rounding
  differ -- measured 0.99989 / 2.1e-2 on out_mvm), losses within 1e-3 relative, gradient arena cosine >= 0.9995 and norm within 1 %.
* `test_window_order_dx1_path_equals_gather_path`: the Video-Swin block backward with `norm2`'s backward writing d(x1) in window order
  (`vmvm_ln_bwd_desc.dx_map` / `add_by_out`, `Switches.dx1_window`) against the form with the `gather_rows` pass: same kernels' arithmetic,
  rows only routed differently -- the gradient arenas agree to f32 summation order (cosine >= 0.999999, every optimizer group).  Eval mode
  (every block takes the path) and train mode with DropPath draws that leave some blocks whole and compact others.
* `test_full_width_c5_forward_losses_vs_oracle`: BASELINE config 5 at FULL width (Swin-L, 16 x 384^2, 1152-token windows, 2352-token
  fusion sequences), B = 1, forward only, against the CPU oracle (earlier rounds pinned that geometry at reduced width only): losses
  within 2e-2 relative (vtm: 8e-2 absolute at temp 0.05), out_mvm / out_mtm cosine >= 0.999.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _engine(cfg_args):
    from pytorch_empirical_mvm_amd import config as CFG
    from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain
    args = CFG.get_args(**cfg_args)
    return VIOLET_Pretrain(args, None, device="cuda"), args


@pytest.mark.timeout(1500)
def test_full_width_c2_batch32_equals_eight_batches_of_four():
    from oracle import violet_ref as R                    # (batch / weight generators only: the reference here is the library at B = 4)
    B, G = 32, 8
    cfg = R.make_cfg("base", T=8, temp=1.0)
    model, args = _engine(dict(vis_backbone_size="base", size_frame=8, max_size_frame=8, temp=1.0))
    model.load_state_dict(R.make_state_dict(cfg))
    eng, S = model.engine, model.engine.store
    img, txt, mask = R.make_batch(cfg, B)
    mb = R.default_masking(cfg, img, txt, mask, seed=9)
    ans = mb["ans_mtm"].clone()
    cov = mb["mvm_mask"][:, :, 0, ::32, ::32].to(torch.uint8).clone()
    # equalise the loss denominators of the eight groups of four clips (extra MLM targets / covered patches are dropped)
    n_ans = min(int((ans[4 * g:4 * g + 4] != -1).sum()) for g in range(G))
    n_cov = min(int(cov[4 * g:4 * g + 4].sum()) for g in range(G))
    assert n_ans > 0 and n_cov > 0
    for g in range(G):
        a_, c_ = ans[4 * g:4 * g + 4], cov[4 * g:4 * g + 4]
        idx = (a_ != -1).flatten().nonzero().flatten()
        a_.view(-1)[idx[n_ans:]] = -1
        idc = c_.flatten().nonzero().flatten()
        c_.view(-1)[idc[n_cov:]] = 0
    full = dict(img=img.cuda(), cov=cov.cuda().contiguous(), txt=mb["txt"].cuda(), mask=mask.cuda(), ans_mtm=ans.cuda())
    negs = [eng.sample_negatives(4, np.random.RandomState(31 + g)) for g in range(G)]
    neg_cat = np.concatenate([n_ + 4 * g for g, n_ in enumerate(negs)], 0)
    # DropPath draws for the train-mode pass (video_swin.py:49-54), fixed: every block draws; a few branches lose every clip of a group
    rng = np.random.RandomState(5)
    n_blk = sum(cfg["depths"])
    dpr = np.linspace(0, 0.2, n_blk)
    scales = np.ones((n_blk, 2, B), np.float32)
    for blk in range(1, n_blk):
        keep = 1.0 - dpr[blk]
        scales[blk] = np.floor(keep + rng.rand(2, B)) / keep
    scales[7, 0, 4:8] = 0.0                                # a whole group's attention branch dropped (empty kept-list in the B = 4 run)
    scales[15, 1, 20:24] = 0.0
    scales[10, 0, :] = 0.0                                 # ... and one attention branch with every clip of the batch dropped
    kept = (scales != 0).sum(-1)
    assert kept.min() == 0 and kept.max() == B and 0 < kept[3:].min(initial=B, where=kept[3:] > 0) < B

    def run(batch, neg, sl, train):
        S.grad.zero_()
        kw = dict(train=True, dp_all=[(torch.from_numpy(scales[i, 0, sl]).cuda().contiguous(), torch.from_numpy(scales[i, 1, sl]).cuda().contiguous())
                                      for i in range(n_blk)], dropout=False) if train else dict(train=False)
        losses, outs = eng.forward_backward(batch, negatives=neg, backward=True, want_outputs=True, **kw)
        torch.cuda.synchronize()
        return ({k: float(losses[k].item()) for k in ("mtm", "mvm", "vtm")}, {k: outs[k].float().clone() for k in ("out_mtm", "out_mvm", "out_vtm")},
                S.grad[:S.n_trainable].clone())

    for train in (False, True):
        l32, o32, g32 = run(full, neg_cat, slice(0, B), train)
        lsum = {k: 0.0 for k in l32}
        gsum = torch.zeros_like(g32)
        outs4 = {k: [] for k in o32}
        for g in range(G):
            sl = slice(4 * g, 4 * g + 4)
            l4, o4, g4 = run({k: v[sl].contiguous() for k, v in full.items()}, negs[g], sl, train)
            for k in lsum:
                lsum[k] += l4[k] / G
            gsum += g4 / G
            for k in outs4:
                outs4[k].append(o4[k])
        for k in l32:
            assert abs(l32[k] - lsum[k]) <= 1e-3 * abs(lsum[k]) + 1e-4, (train, k, l32[k], lsum[k])
        for k in o32:
            ref = torch.cat(outs4[k], 0)
            assert ref.shape == o32[k].shape, (k, ref.shape, o32[k].shape)
            c = _cos(o32[k], ref)
            d = float((o32[k] - ref).abs().max() / (ref.abs().max() + 1e-12))
            assert c >= 0.9995 and d <= 5e-2, (train, k, c, d)
        c = _cos(g32, gsum)
        ratio = float(g32.double().norm() / gsum.double().norm())
        assert c >= 0.9995 and abs(ratio - 1.0) <= 1e-2, (train, c, ratio)
        # per reduction group as well (a wrong Swin stage would hide in the arena-wide cosine behind the fusion encoder's 137 M parameters)
        for gi in range(4):
            a, e = S.segments[gi]
            if e > a:
                cg = _cos(g32[a:e], gsum[a:e])
                assert cg >= 0.999, (train, gi, cg)


@pytest.mark.timeout(900)
def test_window_order_dx1_path_equals_gather_path():
    from oracle import violet_ref as R                    # (batch / weight generators only)
    B = 4
    cfg = R.make_cfg("base", T=8, temp=1.0)
    model, args = _engine(dict(vis_backbone_size="base", size_frame=8, max_size_frame=8, temp=1.0))
    model.load_state_dict(R.make_state_dict(cfg))
    eng, S = model.engine, model.engine.store
    img, txt, mask = R.make_batch(cfg, B)
    mb = R.default_masking(cfg, img, txt, mask, seed=3)
    cov = mb["mvm_mask"][:, :, 0, ::32, ::32].to(torch.uint8).cuda().contiguous()
    batch = dict(img=img.cuda(), cov=cov, txt=mb["txt"].cuda(), mask=mask.cuda(), ans_mtm=mb["ans_mtm"].cuda())
    neg = eng.sample_negatives(B, np.random.RandomState(7))
    n_blk = sum(cfg["depths"])
    rng = np.random.RandomState(11)
    scales = np.ones((n_blk, 2, B), np.float32)
    for blk in range(2, n_blk, 3):                         # every third block: one branch loses a clip (compacted: the gather path either way)
        scales[blk, blk & 1, rng.randint(B)] = 0.0
        scales[blk, blk & 1][scales[blk, blk & 1] != 0] = 1.25

    def run(train, on):
        eng.sw.dx1_window = on
        S.grad.zero_()
        kw = dict(train=True, dp_all=[(torch.from_numpy(scales[i, 0]).cuda(), torch.from_numpy(scales[i, 1]).cuda()) for i in range(n_blk)],
                  dropout=False) if train else dict(train=False)
        losses, _ = eng.forward_backward(batch, negatives=neg, backward=True, **kw)
        torch.cuda.synchronize()
        return {k: float(losses[k].item()) for k in ("mtm", "mvm", "vtm")}, S.grad[:S.n_trainable].clone()

    try:
        for train in (False, True):
            l0, g0 = run(train, False)
            l1, g1 = run(train, True)
            for k in l0:                                    # the forward is untouched (loss sums use atomics: last-bit differences run to run)
                assert abs(l0[k] - l1[k]) <= 1e-6 * abs(l0[k]) + 1e-7, (train, k, l0[k], l1[k])
            assert torch.isfinite(g1).all()
            for gi in range(4):
                a, e = S.segments[gi]
                if e > a:
                    c = _cos(g0[a:e], g1[a:e])
                    assert c >= 0.999999, (train, gi, c)
            d = float((g0 - g1).abs().max() / (g0.abs().max() + 1e-30))
            assert d <= 1e-3, (train, d)
    finally:
        eng.sw.dx1_window = True


@pytest.mark.timeout(2400)
def test_full_width_c5_forward_losses_vs_oracle():
    from oracle import violet_ref as R
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    arch = dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48), window=(8, 12, 12))       # swin_large_384 (visbackbone/swin_large_384_*.py:4)
    cfg = R.make_cfg("large", T=16, img=384, temp=1.0, max_size_frame=16, arch=arch)
    model, args = _engine(dict(vis_backbone_size="large", size_frame=16, max_size_frame=16, size_img=384, temp=1.0))
    sd = R.make_state_dict(cfg)
    model.load_state_dict(sd)
    B = 1
    img, txt, mask = R.make_batch(cfg, B)
    mb = R.default_masking(cfg, img, txt, mask, seed=1)           # (3 MLM targets at B = 1)
    assert int((mb["ans_mtm"] != -1).sum()) > 0 and float(mb["mvm_mask"].sum()) > 0
    neg = R.vtm_negatives_default(B)                       # B = 1: O = min(B, 4) = 1, no negative pairings (main_pretrain.py:243-259) -- the VTM loss is 0 on both sides
    with torch.no_grad():
        ref = R.pretrain_losses(sd, cfg, mb, negatives=neg)
    cov = mb["mvm_mask"][:, :, 0, ::32, ::32].to(torch.uint8).cuda().contiguous()
    batch = dict(img=img.cuda(), cov=cov, txt=mb["txt"].cuda(), mask=mask.cuda(), ans_mtm=mb["ans_mtm"].cuda())
    losses, outs = model.engine.forward_backward(batch, negatives=neg, train=False, want_outputs=True, backward=False)
    torch.cuda.synchronize()
    for k in ("mtm", "mvm"):
        got, want = float(losses[k].item()), float(ref[k])
        assert abs(got - want) <= 2e-2 * abs(want) + 1e-3, (k, got, want)
    assert abs(float(losses["vtm"].item()) - float(ref["vtm"])) <= 8e-2, (float(losses["vtm"].item()), float(ref["vtm"]))
    assert torch.isfinite(outs["out_mvm"].float()).all() and torch.isfinite(outs["out_mtm"].float()).all()
    assert _cos(outs["out_mvm"].float().cpu(), ref["out"]["out_mvm"]) >= 0.999
    assert _cos(outs["out_mtm"].float().cpu(), ref["out"]["out_mtm"]) >= 0.999
