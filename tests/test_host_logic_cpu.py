"""Host-side logic of the product package against the oracle / reference goldens (CPU only)."""
import os
import random

import numpy as np
import torch

from oracle import violet_ref as R
from pytorch_empirical_mvm_amd import config as CFG

G = os.path.join(os.path.dirname(__file__), "golden")


def test_param_inventory_matches_reference_key_list():
    for size, T in (("tiny", 4), ("base", 8), ("large", 8)):
        args = CFG.get_args(vis_backbone_size=size, size_frame=T, max_size_frame=max(T, 6))
        mine = CFG.param_shapes(CFG.model_cfg(args))
        ref = R.param_shapes(R.make_cfg(size, T=T))
        assert list(mine.keys()) == list(ref.keys())
        assert all(tuple(mine[k]) == tuple(ref[k]) for k in ref)
    n_base = sum(int(np.prod(s)) for s in CFG.param_shapes(CFG.model_cfg(CFG.get_args(vis_backbone_size="base", max_size_frame=8))).values())
    assert n_base == 225_086_979                      # SURVEY.md section 8(a) a15 probe (Swin-B, T=8)


def test_param_groups_and_lr_schedule_match_reference_goldens():
    d = np.load(os.path.join(G, "optimizer.npz"))
    for n, row in zip([str(x) for x in d["names"]], d["groups"]):
        g = CFG.param_group(n)                      # 0 decay+swin, 1 decay+other, 2 nodecay+swin, 3 nodecay+other
        assert int(np.argmax(row)) == g, n
    for k, lr in d["lr_table"]:
        assert abs(max(1e-8, 5e-5 * CFG.lr_factor(int(k), 1000)) - lr) <= 1e-12 + 1e-9 * lr


def test_agent_masking_reproduces_reference_draws():
    """Agent_Pretrain.masking consumes `random`, `torch.rand`, `np.random` in the reference's order (main_pretrain.py:303-352)."""
    from pytorch_empirical_mvm_amd.agent import Agent_Pretrain
    d = np.load(os.path.join(G, "masking.npz"))
    cfg = R.make_cfg("tiny", T=4)
    img, txt, mask = R.make_batch(cfg, 3)

    class _M:
        patch_size = 32
        engine = type("E", (), {"device": torch.device("cpu")})()
    for name in ("rm", "bm"):
        args = CFG.get_args(pretrain_masks=[name])
        ag = Agent_Pretrain.__new__(Agent_Pretrain)
        ag.args, ag.patch_size = args, 32
        ag.cls_token_id, ag.sep_token_id, ag.pad_token_id, ag.mask_token_id = 101, 102, 0, 103
        random.seed(5); np.random.seed(5); torch.manual_seed(5)
        o = ag.masking(img.clone(), txt.clone(), mask.clone(), None, materialize=True)
        np.testing.assert_array_equal(o["cov"].numpy(), d[f"{name}.cov"])
        np.testing.assert_array_equal(o["txt"].numpy(), d[f"{name}.txt"])
        np.testing.assert_array_equal(o["ans_mtm"].numpy(), d[f"{name}.ans_mtm"])
        np.testing.assert_array_equal(o["ans_mvm"].numpy(), d[f"{name}.ans_mvm"])
        np.testing.assert_allclose(float(o["img"].double().sum()), float(d[f"{name}.img_sum"]), rtol=1e-9)
        assert float(o["mvm_mask"].double().sum()) == float(d[f"{name}.mask_sum"])


def test_arena_layout_groups_and_fused_qkv():
    """ParamStore: optimizer groups are contiguous segments, every view is 8-element aligned, BERT q/k/v are adjacent."""
    from pytorch_empirical_mvm_amd.engine import ParamStore
    args = CFG.get_args(vis_backbone_size="tiny", arch_override=dict(embed_dim=32, depths=(1, 1, 1, 1), num_heads=(1, 2, 4, 8), window=(8, 7, 7)),
                        bert_layers=2)
    shapes = CFG.param_shapes(CFG.model_cfg(args))
    S = ParamStore.__new__(ParamStore)
    # layout only (no device allocation): replicate __init__'s index computation through a CPU store
    S.__init__(shapes, torch.device("cpu"))
    seen = 0
    for n, (o, c, shp) in S.index.items():
        assert o % 8 == 0 and o >= seen
        seen = o + c
        g = 4 if n in S.FROZEN else CFG.param_group(n)
        a, e = S.segments[g]
        assert a <= o and o + c <= e, (n, g)
    assert S.segments[4][1] - S.segments[4][0] == 768          # frozen enc_img.emb_odr
    H = 768
    qn = [f"trsfr.layer.1.attention.self.{x}.weight" for x in ("query", "key", "value")]
    w = S.fused(S.flat, qn, (3 * H, H))
    S.p(qn[1]).fill_(2.0)
    assert float(w[H:2 * H].sum()) == 2.0 * H * H and float(w[:H].sum()) == 0.0
    assert S.n_trainable == S.segments[3][1]


def test_vq_index_matches_reference_answer_map():
    """SURVEY a13 host side: covered vq positions (max_pool2d(mvm_mask, 8) != 0, main_pretrain.py:485-488) <-> index lists"""
    import numpy as np
    import torch
    from oracle import violet_ref as R
    from pytorch_empirical_mvm_amd import config as CFG
    from pytorch_empirical_mvm_amd.agent import Agent_Pretrain

    class _M:                                   # masking()/vq_index() only need these attributes
        patch_size = 32
        class engine: device = "cpu"
    args = CFG.get_args(mvm_target=["vq"], size_frame=4, size_txt=32)
    ag = Agent_Pretrain.__new__(Agent_Pretrain)
    ag.args, ag.patch_size = args, 32
    cfg = R.make_cfg("tiny", T=4, mvm_target=["vq"], dvae_hid=64, dvae_vocab=512)
    img, txt, mask = R.make_batch(cfg, 3)
    mb = R.default_masking(cfg, img, txt, mask, seed=11)
    cov = mb["mvm_mask"][:, :, 0, ::32, ::32].to(torch.uint8)
    vqi = ag.vq_index(cov)
    tok = torch.arange(3 * 4 * 28 * 28).view(12, 28, 28)
    ans = R.vq_answers(tok, mb["mvm_mask"]).flatten()
    assert sorted(vqi["vq_tok_index"].tolist()) == torch.nonzero(ans != -1).flatten().tolist()
    assert vqi["vq_patch_rows"].numel() * 16 == vqi["vq_tok_index"].numel() == int(cov.sum()) * 16
    Lq = 4 * 50 + 32
    b, t, hh, ww = np.nonzero(cov.numpy())
    np.testing.assert_array_equal(vqi["vq_patch_rows"].numpy(), b * Lq + t * 50 + 1 + hh * 7 + ww)


def test_masking_from_uniform_matches_reference_distributions():
    """SURVEY 8f.2 contract of the device-side masking kernel, stated by the oracle: the explicit-draw mapping reproduces the
    reference's ranges (numpy randint bounds of main_pretrain.py:308-313, cls slot never covered, special tokens never masked)
    including u at the top of [0,1), and its statistics (15 % Bernoulli fields, uniform cuboid sizes)."""
    import numpy as np
    import torch
    from oracle import violet_ref as R
    cfg = R.make_cfg("tiny", T=8)
    B, T, h, w, X = 64, 8, 7, 7, 32
    img, txt, mask = R.make_batch(cfg, 2)
    img = img[:1].expand(B, -1, -1, -1, -1)
    txt = txt[:1].repeat(B, 1); mask = mask[:1].repeat(B, 1)
    g = np.random.RandomState(5)
    top = np.float32(1.0) - np.float32(2.0 ** -24)
    u_type = g.rand(B).astype(np.float32)
    u_txt = g.rand(B, X).astype(np.float32)
    u_rm = g.rand(B, T, 1 + h * w).astype(np.float32)
    u_bm = g.rand(B, T, 6).astype(np.float32)
    u_type[0], u_bm[0] = top, top                        # extreme draws: clamped to the last value, never out of range
    u_type[1], u_bm[1] = top, 0.0
    o = R.masking_from_uniform(cfg, img, txt, mask, u_type, u_txt, u_rm, u_bm)
    assert o["kinds"][0] == "bm" and o["kinds"][1] == "bm" and set(o["kinds"]) == {"rm", "bm"}
    # extreme cuboids: every draw at the top -> t = T-1, hh = ww = 3 at the last admissible corner; all zero -> 1x1x1 at the origin
    c0 = o["cov"][0]
    assert float(c0.sum()) == (T - 1) * 3 * 3 and float(c0[1:, 4:, 4:].sum()) == (T - 1) * 9
    assert float(o["cov"][1].sum()) == 1 and float(o["cov"][1][0, 0, 0]) == 1
    spc = (txt == 101) | (txt == 102) | (txt == 0) | (txt == 103)
    assert bool(((o["ans_mtm"] != -1) & spc).any()) is False
    assert torch.equal(o["txt"][o["ans_mtm"] != -1], torch.full((int((o["ans_mtm"] != -1).sum()),), 103))
    rm = [b for b in range(B) if o["kinds"][b] == "rm"]
    frac = float(o["cov"][rm].mean())
    assert abs(frac - 0.15) < 0.02, frac
    for b in rm:                                         # the field is exactly the draw, cls slot dropped
        np.testing.assert_array_equal(o["cov"][b].numpy().reshape(T, -1), (u_rm[b, :, 1:] < np.float32(0.15)).astype(np.float32))
    # mvm_mask is the x32 expansion of cov and img is zeroed under it (main_pretrain.py:362-364)
    assert torch.equal(o["mvm_mask"][:, :, 0, ::32, ::32], o["cov"])
    assert float((o["img"] * o["mvm_mask"]).abs().max()) == 0.0


def test_agent_am_masking_reproduces_reference_branch():
    """The attention-guided 'am' branch of Agent_Pretrain.masking (main_pretrain.py:320-343) for fixed attention weights and a
    fixed torch seed, against what the reference's masking() produced (tests/golden/am.npz); then the fallback to 'rm' when the
    draw contains no text position (the reference's sticky `failed_masking`)."""
    from pytorch_empirical_mvm_amd.agent import Agent_Pretrain
    d = np.load(os.path.join(G, "am.npz"))
    cfg = R.make_cfg("tiny", T=4)
    img, txt, mask = R.make_batch(cfg, 2)
    fake = torch.from_numpy(d["fake"])

    class _Model:
        def __init__(self, att):
            self.att = att

        def get_att(self, *a, **k):
            return None, self.att.clone()
    args = CFG.get_args(pretrain_masks=["am"])
    ag = Agent_Pretrain.__new__(Agent_Pretrain)
    ag.args, ag.patch_size, ag.model = args, 32, _Model(fake)
    ag.cls_token_id, ag.sep_token_id, ag.pad_token_id, ag.mask_token_id = 101, 102, 0, 103
    random.seed(7); np.random.seed(7); torch.manual_seed(7)
    o = ag.masking(img.clone(), txt.clone(), mask.clone(), None)
    np.testing.assert_array_equal(o["txt"].numpy(), d["am_txt"])
    np.testing.assert_array_equal(o["ans_mtm"].numpy(), d["am_ans_mtm"])
    np.testing.assert_array_equal(o["cov"].numpy(), d["am_cov"])
    n = int(((1 + 49) * 4 + 32) * 0.15)
    for i in range(2):
        assert int(o["cov"][i].sum()) + int((o["ans_mtm"][i] != -1).sum()) == n
    # all weight on the visual positions -> no text position can be drawn -> 'rm' fallback (Bernoulli field, ~15 % cover)
    vis_only = fake.clone(); vis_only[:, 200:] = 0.0
    ag.model = _Model(vis_only)
    torch.manual_seed(3)
    o2 = ag.masking(img.clone(), txt.clone(), mask.clone(), None)
    assert 0.05 < float(o2["cov"].float().mean()) < 0.3


def test_checkpoint_key_lists_of_targets_teachers_and_downstream_heads():
    """Checkpoint surface: for every MVM target / task the package's key -> shape list equals the oracle's, which is
    the list the reference's own state_dict had when the fixtures were generated (gen_goldens asserts oracle keys == reference keys)."""
    from pytorch_empirical_mvm_amd import teacher as TCH
    for kw in (dict(mvm_target=["pixel"]), dict(mvm_target=["vq"]), dict(mvm_target=["hog"]), dict(mvm_target=["3d_feature"]),
               dict(mvm_target=["2d_feature"]), dict(mvm_target=["pixel", "hog"]), dict(task="retrieval", mvm_target=[]),
               dict(task="qaoe", mvm_target=[], size_vocab=1000)):
        args = CFG.get_args(vis_backbone_size="tiny", size_frame=4, max_size_frame=6, **kw)
        mine = CFG.param_shapes(CFG.model_cfg(args))
        cfg = R.make_cfg("tiny", T=4, mvm_target=list(kw.get("mvm_target", ["pixel"])))
        if "task" in kw:
            cfg["task"] = kw["task"]; cfg["size_vocab"] = kw.get("size_vocab", 0)
        ref = R.param_shapes(cfg)
        assert {k: tuple(v) for k, v in mine.items()} == {k: tuple(v) for k, v in ref.items()}, kw
    arch = dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window=(8, 7, 7))
    for kind, target in (("3d", "3d_feature"), ("2d", "2d_feature")):
        cfg = R.make_cfg("tiny", T=4, mvm_target=[target])
        ref = R.teacher_param_shapes(cfg)
        if kind == "3d":
            mine = {"feature_model." + k[len("enc_img.swin."):]: v for k, v in CFG.swin_param_shapes(arch, (8, 7, 7)).items()}
        else:
            mine = TCH.hf_swin2d_param_shapes(arch, "feature_model.", 7)
        assert {k: tuple(v) for k, v in mine.items()} == {k: tuple(v) for k, v in ref.items()}, kind
        assert len(ref) > 300


def test_swinbert_checkpoint_key_renames():
    """model.py:355-386 (`load_SwinBERT_weight`): the rename table of a SwinBERT checkpoint, rule priority included (a key holding
    `swin.backbone` is a backbone key even when it starts with `fc.`-like text), dropped keys, the duplicated MLM decoder bias."""
    from pytorch_empirical_mvm_amd.model import swinbert_renames
    t = lambda i: torch.full((1,), float(i))
    src = {"swin.backbone.layers.0.blocks.0.attn.qkv.weight": t(0), "trans_encoder.bert.encoder.layer.3.output.dense.bias": t(1),
           "trans_encoder.bert.embeddings.word_embeddings.weight": t(2), "fc.weight": t(3), "trans_encoder.bert.img_embedding.weight": t(4),
           "trans_encoder.cls.predictions.bias": t(5), "trans_encoder.cls.predictions.transform.dense.weight": t(6),
           "learn_mask_enabled": t(7), "module.fc.weight": t(8)}
    out = swinbert_renames(src)
    want = {"enc_img.swin.layers.0.blocks.0.attn.qkv.weight": 0, "trsfr.layer.3.output.dense.bias": 1, "enc_txt.emb_txt.word_embeddings.weight": 2,
            "enc_img.fc.weight": 3, "enc_img.img_embedding.weight": 4, "fc_mtm.predictions.bias": 5, "fc_mtm.predictions.decoder.bias": 5,
            "fc_mtm.predictions.transform.dense.weight": 6}
    assert set(out) == set(want), sorted(out)
    for k, v in want.items():
        assert float(out[k]) == v, k
    # the renamed names are this model's names
    inv = set(R.param_shapes(R.make_cfg("base", T=8)))
    assert {"enc_img.swin.layers.0.blocks.0.attn.qkv.weight", "trsfr.layer.3.output.dense.bias", "enc_txt.emb_txt.word_embeddings.weight",
            "fc_mtm.predictions.bias", "fc_mtm.predictions.transform.dense.weight"} <= inv


def test_2d_to_3d_weight_inflation_matches_reference_outputs():
    """visbackbone.inflate_2d_state against SwinTransformer3D.inflate_weights' own results (video_swin.py:484-535; inflate2d.npz holds a synthetic
    image-Swin checkpoint's tensors and what the reference module held after loading it): patch kernel tiled over the temporal taps / tap count,
    7x7 tables tiled 15x, 6x6 tables resized bicubically to 13x13 first, index / mask buffers dropped, everything else untouched."""
    import os
    import numpy as np
    import torch
    from pytorch_empirical_mvm_amd.visbackbone import inflate_2d_state
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "inflate2d.npz"))
    arch = dict(window=(8, 7, 7), patch=(2, 4, 4))
    for tag in ("w7", "w6"):
        keys = sorted({k[len(tag) + 4:] for k in d.files if k.startswith(tag + ".in.")})
        assert len(keys) == 5
        sd2 = {k: torch.from_numpy(d[f"{tag}.in.{k}"]) for k in keys}
        sd2["layers.0.blocks.0.attn.relative_position_index"] = torch.zeros(36, 36, dtype=torch.long)
        sd2["layers.0.blocks.0.attn_mask"] = torch.zeros(4, 49, 49)
        out = inflate_2d_state(sd2, arch)
        assert sorted(out) == keys                                         # the two buffers are gone
        for k in keys:
            want = d[f"{tag}.out.{k}"]
            assert tuple(out[k].shape) == want.shape, (tag, k)
            np.testing.assert_allclose(out[k].numpy(), want, rtol=1e-6, atol=1e-7, err_msg=f"{tag} {k}")
        assert out["patch_embed.proj.weight"].shape == (32, 3, 2, 4, 4)
        assert out["layers.2.blocks.1.attn.relative_position_bias_table"].shape == (15 * 169, 4)
    # a table whose head count differs from the model's stays as it is (the loader skips it by shape)
    t = torch.randn(169, 3)
    out = inflate_2d_state({"layers.0.blocks.0.attn.relative_position_bias_table": t}, arch, {"layers.0.blocks.0.attn.relative_position_bias_table": (2535, 1)})
    assert out["layers.0.blocks.0.attn.relative_position_bias_table"] is t


def test_dropscale_kept_lists_padding_and_single_scale():
    """engine.DropScale (host side of the DropPath dead-clip elimination): kept / dropped clip lists with -1 behind their entries, the kept clips'
    scales followed by zeros, ONE scale per draw (None for hand-made non-uniform scales), and `take`: the clip count a branch runs on = the kept
    count rounded up so that count * rows_per_clip is a multiple of 64 (the K tile of the weight-gradient GEMMs), capped at the batch."""
    import numpy as np
    import torch
    from pytorch_empirical_mvm_amd.engine import DropScale
    s = 1.0 / 0.8
    d = DropScale(torch.tensor([s, 0, s, s, 0, s, s, 0], dtype=torch.float32))
    assert d.n_kept == 5 and abs(d.scale - s) < 1e-6
    assert d.kept.tolist() == [0, 2, 3, 5, 6, -1, -1, -1] and d.dropped.tolist() == [1, 4, 7, -1, -1, -1, -1, -1]
    np.testing.assert_allclose(d.dev_kept.numpy(), [s] * 5 + [0, 0, 0], rtol=1e-6)
    assert d.take(6272, 8)[0] == 5            # Swin-B stage 2: 6 272 rows per clip, any count fills whole K tiles
    assert d.take(1568, 8)[0] == 6            # stage 3: 1 568 = 24.5 x 64 -> even counts (one padding clip)
    assert d.take(392, 8)[0] == 8             # stage 4: 392 = 6.125 x 64 -> multiples of 8 (here: the whole batch, nothing to eliminate)
    assert d.take(392, 32)[0] == 8 and d.take(1568, 5)[0] == 5          # ... and never more than the batch
    for n in range(0, 9):
        for rows in (25088, 6272, 1568, 392, 4608, 1152, 100):
            k = DropScale(torch.tensor([s] * n + [0.0] * (8 - n))).take(rows, 8)[0]
            assert k >= n and (k == 8 or (k * rows) % 64 == 0) and (k - n) * rows < 64 * rows
    assert DropScale(torch.zeros(4)).n_kept == 0 and DropScale(torch.zeros(4)).scale is None
    assert DropScale(torch.tensor([1.25, 0.0, 1.5])).scale is None      # hand-made non-uniform scales: the engine keeps the scaled formulation
    assert DropScale(torch.ones(3)).take(1568, 3)[0] == 3
