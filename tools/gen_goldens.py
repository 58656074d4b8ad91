#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (CPU, fp32).

Runs ONLY in the build container (needs /root/reference); the reference never
travels to the GPU box -- only the small output fixtures written here do.
Import recipe = SURVEY.md section 8(c).  Weights/inputs are the closed-form tensors of
oracle/violet_ref.py, so fixtures hold outputs only.

    python tools/gen_goldens.py            # writes tests/golden/*.npz
"""
import json
import os
import random
import sys
import types

import numpy as np
import torch
import transformers  # noqa: F401  (must be imported before the stubs)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import violet_ref as R  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


# ------------------------------------------------------------------ stubs
class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _stub(name, **attrs):
    import importlib.machinery
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    class Dict(dict):  # addict.Dict: nested attr dict
        def __init__(self, *a, **k):
            super().__init__()
            for kk, vv in dict(*a, **k).items():
                self[kk] = vv

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, Dict):
                v = Dict(v)
            super().__setitem__(k, v)

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

        def to_dict(self):
            return {k: (v.to_dict() if isinstance(v, Dict) else v) for k, v in self.items()}

    _stub("addict", Dict=Dict)
    _stub("yapf"); _stub("yapf.yapflib"); _stub("yapf.yapflib.yapf_api", FormatCode=lambda s, **k: (s, True))
    _stub("easydict", EasyDict=_AttrDict)
    _stub("skimage"); _stub("skimage.feature", hog=None); _stub("skimage.transform")
    class Normalize:                      # torchvision.transforms.Normalize (absent from this image): (x - mean) / std per channel
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32)

        def __call__(self, x):
            return (x - self.mean.view(-1, 1, 1).to(x.dtype)) / self.std.view(-1, 1, 1).to(x.dtype)

    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms", Normalize=Normalize, Compose=object)
    tv.models = _stub("torchvision.models")
    _stub("torchvision.models.optical_flow", raft_large=None)
    _stub("cv2")
    _stub("fairscale"); _stub("fairscale.nn"); _stub("fairscale.nn.misc", checkpoint_wrapper=lambda m, **k: m)
    _stub("toolz"); _stub("toolz.sandbox", unzip=None)
    _stub("deepspeed")
    _stub("tensorboardX", SummaryWriter=object)
    _stub("timm"); _stub("timm.models"); _stub("timm.models.layers", DropPath=object, to_2tuple=None, trunc_normal_=None)
    _stub("wandb")
    _stub("dataset", Dataset_Base=object, get_dl=None, move_to_cuda=lambda b: b, get_tsv_dls=None,
          MetaLoader=object, PrefetchLoader=object, TsvCompositeDataset=object, make_data_loader=None)
    os.environ["WANDB_ENABLE"] = "0"


def import_reference():
    import transformers as tr
    from transformers import BertConfig, BertForMaskedLM, BertModel  # noqa: F401
    _ = (tr.AutoModel, tr.AutoModelForMaskedLM, tr.AutoTokenizer, tr.AutoConfig, tr.RobertaForMaskedLM)
    install_stubs()
    os.chdir(REF)
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    from transformers import BertConfig, BertForMaskedLM, BertModel
    import transformers as tr
    mk = lambda cls: (lambda *a, **k: cls(BertConfig(attn_implementation="eager")))
    tr.AutoModel.from_pretrained = staticmethod(mk(BertModel))
    tr.AutoModelForMaskedLM.from_pretrained = staticmethod(mk(BertForMaskedLM))


def ref_args(size, T, mvm_target="pixel"):
    a = _AttrDict(json.load(open(os.path.join(REF, "_args/args_pretrain.json"))))
    a.update(mvm_target=[mvm_target], vis_backbone_size=size, vis_backbone_init="random", size_patch=32,
             max_size_frame=max(T, 6), max_size_patch=14, temporal_fusion="vidswin", fusion_encoder_rand_init=False,
             use_checkpoint=False, deepspeed=False, tokenizer="bert-base-uncased", dalle_model_path="",
             max_iter=100, enable_task_token=False, enable_prompt=False, size_frame=T, kinetics=600)
    return a


def samp(t, n=64):
    """n deterministic sample positions of a tensor + its sum and abs-sum."""
    f = t.detach().double().flatten()
    idx = np.unique((np.arange(n, dtype=np.int64) * 7919 + 13) % f.numel())
    return dict(idx=idx, val=f[idx].numpy(), sum=float(f.sum()), asum=float(f.abs().sum()), shape=np.array(t.shape))


def put(d, name, t, n=64):
    for k, v in samp(t, n).items():
        d[f"{name}.{k}"] = v


# ------------------------------------------------------------------ golden sets
def gold_helpers(vs):
    d = {}
    x = torch.arange(1 * 8 * 14 * 14 * 2, dtype=torch.float32).view(1, 8, 14, 14, 2)
    w = vs.window_partition(x, (8, 7, 7))
    d["wp_shape"] = np.array(w.shape); d["wp_1_0"] = w[1, 0].numpy(); d["wp_2_5"] = w[2, 5].numpy()
    d["wp_full"] = w.numpy().astype(np.int32)
    r = vs.window_reverse(w, (8, 7, 7), 1, 8, 14, 14)
    d["wr_equal"] = np.array(int(torch.equal(r, x)))
    for name, (D, H, W, ws, ss) in dict(a=(8, 14, 14, (8, 7, 7), (0, 3, 3)), b=(16, 12, 12, (8, 12, 12), (4, 0, 0)),
                                        c=(16, 28, 21, (8, 7, 7), (4, 3, 3))).items():
        m = vs.compute_mask(D, H, W, ws, ss, "cpu")
        d[f"mask_{name}_shape"] = np.array(m.shape)
        d[f"mask_{name}_n100"] = np.array(int((m == -100).sum()))
        d[f"mask_{name}_rowsum"] = (m == -100).sum(-1).numpy().astype(np.int32)
    d["gws_a"] = np.array(vs.get_window_size((4, 56, 56), (8, 7, 7), (4, 3, 3)))
    d["gws_b"] = np.array(vs.get_window_size((16, 12, 12), (8, 12, 12), (4, 6, 6)))
    att = vs.WindowAttention3D(32, (8, 7, 7), 1)
    d["rpi_877"] = att.relative_position_index.numpy().astype(np.int32)
    att = vs.WindowAttention3D(32, (2, 3, 3), 1)
    d["rpi_233"] = att.relative_position_index.numpy().astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "helpers.npz"), **d)
    print("helpers ok")


def gold_reduced_swin(vs):
    """Reduced Swin: exercises D-pad 12->16, temporal shift 4, H/W pad 24x20 -> 28x21."""
    arch = dict(embed_dim=32, depths=(2, 2, 2, 2), num_heads=(1, 2, 4, 8), window=(8, 7, 7))
    cfg = R.make_cfg("tiny", T=12, arch=arch)
    m = vs.SwinTransformer3D(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=(8, 7, 7),
                             patch_size=(2, 4, 4)).eval()
    shapes = {k: v for k, v in R.param_shapes(cfg).items() if k.startswith("enc_img.swin.")}
    sd = {k[len("enc_img.swin."):]: R.closed_form(k, s) for k, s in shapes.items()}
    own = {k: v.shape for k, v in m.state_dict().items() if "relative_position_index" not in k}
    assert {k: tuple(v) for k, v in own.items()} == {k: tuple(v.shape) for k, v in sd.items()}, "key/shape mismatch"
    m.load_state_dict(sd, strict=False)
    n = 1 * 3 * 12 * 96 * 80
    x = R.make_batch(dict(T=12, img=96, n_txt=32, vocab=30522), 1)[0][:, :, :, :, :80].transpose(1, 2).contiguous()
    y = m(x)
    d = {}
    put(d, "y", y.permute(0, 2, 3, 4, 1).contiguous(), 256)
    (y * torch.cos(torch.arange(y.numel(), dtype=torch.float32).view_as(y) * 0.01)).sum().backward()
    for k, p in m.named_parameters():
        put(d, "g." + k, p.grad, 16)
    np.savez_compressed(os.path.join(OUT, "reduced_swin.npz"), **d)
    print("reduced swin ok", tuple(y.shape))


def build_ref_model(size, T):
    import main_pretrain as mp
    args = ref_args(size, T)
    model = mp.VIOLET_Pretrain(args, None).eval()
    # 4.26 semantics for the two API-drift points
    def mask_ext(m, shape=None, device=None):
        m = m[:, None, None, :] if m.dim() == 2 else m[:, None, :, :]
        return (1.0 - m.float()) * torch.finfo(torch.float32).min
    model.mask_ext = mask_ext
    enc = model.trsfr
    class _Wrap(torch.nn.Module):
        def __init__(s, e):
            super().__init__(); s.e = e
        def forward(s, feat, mask, output_attentions=False):
            o = s.e(feat, attention_mask=mask)
            return {"last_hidden_state": o[0] if not hasattr(o, "last_hidden_state") else o.last_hidden_state, "attentions": ()}
    object.__setattr__(model, "_enc_wrap", _Wrap(enc))
    model.__dict__["trsfr_call"] = model._enc_wrap
    return mp, args, model


def gold_c1(size="tiny", T=4, B=2, tag="c1"):
    mp, args, model = build_ref_model(size, T)
    cfg = R.make_cfg(size, T=T)
    sd = R.make_state_dict(cfg)
    own = model.state_dict()
    miss = [k for k in sd if k not in own]
    assert not miss, miss
    extra = [k for k in own if k not in sd and "relative_position_index" not in k and "position_ids" not in k
             and "token_type_ids" not in k and k != "fc_mtm.predictions.decoder.bias"]
    assert not extra, extra
    for k, v in sd.items():
        assert tuple(own[k].shape) == tuple(v.shape), (k, own[k].shape, v.shape)
    model.load_state_dict(sd, strict=False)
    nparam = sum(p.numel() for p in model.parameters())
    # route trsfr through the wrapper without renaming parameters
    wrap = model._enc_wrap
    orig_go_cross = model.go_cross
    def go_cross(feat_img, mask_img, feat_txt, mask_txt, **kw):
        feat = torch.cat([feat_img, feat_txt], dim=1)
        mask = model.get_attn_mask(mask_img, mask_txt)
        mask = model.mask_ext(mask)
        out = wrap(feat, mask)
        return out["last_hidden_state"], out["attentions"]
    model.go_cross = go_cross

    img, txt, mask = R.make_batch(cfg, B)
    mb = R.default_masking(cfg, img, txt, mask, seed=3)
    neg = R.vtm_negatives_default(B)
    # make np.random.permutation return our explicit negatives (main_pretrain.py:250)
    calls = {"i": 0}
    def fake_perm(lst):
        i = calls["i"]; calls["i"] += 1
        rest = [j for j in lst if j not in list(neg[i])]
        return np.array(list(neg[i]) + rest)
    mp.np.random.permutation = fake_perm
    agent = mp.Agent_Pretrain.__new__(mp.Agent_Pretrain)
    agent.args, agent.model, agent.patch_size = args, model, 32
    agent.loss_func = torch.nn.CrossEntropyLoss(ignore_index=-1)
    batch = dict(mb)
    out = model(batch)
    ls_mtm = agent.loss_func(out["out_mtm"].flatten(0, 1), out["ans_mtm"].flatten())
    ls_vtm = agent.loss_func(out["out_vtm"], out["ans_vtm"])
    ls_mvm = agent.calc_mvm_loss(batch, out["out_mvm"], is_train=True)
    ls = ls_mtm + ls_vtm + ls_mvm
    ls.backward()
    d = dict(nparam=np.array(nparam), ls_mtm=np.array(float(ls_mtm)), ls_vtm=np.array(float(ls_vtm)),
             ls_mvm=np.array(float(ls_mvm)), out_vtm=out["out_vtm"].detach().numpy(), neg=neg)
    put(d, "out_mtm", out["out_mtm"], 256)
    put(d, "out_mvm", out["out_mvm"], 256)
    gsq = 0.0
    nograd = []
    for k, p in model.named_parameters():
        if p.grad is None:
            nograd.append(k); continue
        gsq += float((p.grad.double() ** 2).sum())
        put(d, "g." + k, p.grad, 8)
    d["grad_norm"] = np.array(gsq ** 0.5)
    d["nograd"] = np.array(nograd)
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **d)
    print(tag, "ok nparam", nparam, "losses", float(ls_mtm), float(ls_vtm), float(ls_mvm), "gn", gsq ** 0.5, "nograd", nograd)
    return mp, agent


def gold_vq(size="tiny", T=4, B=2, S=224):
    """MVM 'vq' target = frozen dVAE tokenizer (visbackbone/dalle) + decoder_vq / fc_mvm head + CE.
    A reduced reference Encoder (n_hid 64, 512 codes) with closed-form weights is pickled the way DalleModel expects."""
    import main_pretrain as mp
    from visbackbone.dalle.encoder import Encoder
    cfg = R.make_cfg(size, T=T, img=S, mvm_target=["vq"], dvae_hid=64, dvae_vocab=512)
    sd = R.make_state_dict(cfg)
    enc = Encoder(n_hid=64, vocab_size=512)
    esd = enc.state_dict()
    dsh = R.dvae_param_shapes(cfg)
    assert set("dalle.encoder." + k for k in esd) == set(dsh), (sorted(esd)[:5], sorted(dsh)[:5])
    enc.load_state_dict({k: sd["dalle.encoder." + k] for k in esd})
    path = "/tmp/dvae_tiny_ref.pkl"
    torch.save(enc, path)
    args = ref_args(size, T, mvm_target="vq")
    args.update(dalle_model_path=path, size_img=S)
    _orig_load = torch.load                 # API drift: torch >= 2.6 defaults to weights_only=True; the file is the module pickled above
    torch.load = lambda f, *a, **k: _orig_load(f, *a, **dict(k, weights_only=False))
    try:
        model = mp.VIOLET_Pretrain(args, None).eval()
    finally:
        torch.load = _orig_load
    def mask_ext(m, shape=None, device=None):
        m = m[:, None, None, :] if m.dim() == 2 else m[:, None, :, :]
        return (1.0 - m.float()) * torch.finfo(torch.float32).min
    enc_t = model.trsfr
    def go_cross(feat_img, mask_img, feat_txt, mask_txt, **kw):
        feat = torch.cat([feat_img, feat_txt], dim=1)
        mask = mask_ext(model.get_attn_mask(mask_img, mask_txt))
        o = enc_t(feat, attention_mask=mask)
        return (o[0] if not hasattr(o, "last_hidden_state") else o.last_hidden_state), ()
    model.go_cross = go_cross
    own = model.state_dict()
    msd = {k: v for k, v in sd.items() if not k.startswith("dalle.")}
    miss = [k for k in msd if k not in own]
    assert not miss, miss
    model.load_state_dict(msd, strict=False)
    img, txt, mask = R.make_batch(cfg, B)
    mb = R.default_masking(cfg, img, txt, mask, seed=5)
    neg = R.vtm_negatives_default(B)
    calls = {"i": 0}
    def fake_perm(lst):
        i = calls["i"]; calls["i"] += 1
        rest = [j for j in lst if j not in list(neg[i])]
        return np.array(list(neg[i]) + rest)
    mp.np.random.permutation = fake_perm
    agent = mp.Agent_Pretrain.__new__(mp.Agent_Pretrain)
    agent.args, agent.model, agent.patch_size = args, model, 32
    agent.loss_func = torch.nn.CrossEntropyLoss(ignore_index=-1)
    batch = dict(mb)
    out = model(batch)
    ls_mtm = agent.loss_func(out["out_mtm"].flatten(0, 1), out["ans_mtm"].flatten())
    ls_vtm = agent.loss_func(out["out_vtm"], out["ans_vtm"])
    ls_mvm = agent.calc_mvm_loss(batch, out["out_mvm"], is_train=True)
    (ls_mtm + ls_vtm + ls_mvm).backward()
    with torch.no_grad():
        tok = model.dalle.extract_vq_token(batch["unmask_img"].view(B * T, 3, S, S))
        zl = model.dalle.encoder(model.dalle.preprocess(batch["unmask_img"].view(B * T, 3, S, S)))
    top2 = zl.topk(2, dim=1).values
    d = dict(ls_mtm=np.array(float(ls_mtm)), ls_vtm=np.array(float(ls_vtm)), ls_mvm=np.array(float(ls_mvm)),
             tokens=tok.numpy().astype(np.int16), min_margin=np.array(float((top2[:, 0] - top2[:, 1]).min())), neg=neg)
    put(d, "z_logits", zl, 256)
    gsq = 0.0
    for k, p_ in model.named_parameters():
        if p_.grad is None:
            continue
        gsq += float((p_.grad.double() ** 2).sum())
        if k.startswith("decoder_vq") or k.startswith("fc_mvm"):
            put(d, "g." + k, p_.grad, 16)
    d["grad_norm"] = np.array(gsq ** 0.5)
    np.savez_compressed(os.path.join(OUT, "vq.npz"), **d)
    print("vq ok losses", float(ls_mtm), float(ls_vtm), float(ls_mvm), "gn", gsq ** 0.5, "tokens", tok.shape, "codes used", tok.unique().numel(),
          "min margin", float(d["min_margin"]))


def gold_feature(kind, size="tiny", T=4, B=2, S=224):
    """MVM '3d_feature' / '2d_feature' targets -- frozen Swin-B teachers (VideoSwin via get_vidswin_model, HF
    SwinModel via get_swin_model; no checkpoints offline, so both get the oracle's closed-form weights) + fc_mvm head + masked L1
    through the REFERENCE's VIOLET_Pretrain / calc_mvm_loss."""
    import main_pretrain as mp
    import transformers as tr
    from visbackbone import video_swin as vs
    cfg = R.make_cfg(size, T=T, img=S, mvm_target=[kind])
    sd = R.make_state_dict(cfg)
    args = ref_args(size, T, mvm_target=kind)
    args.update(size_img=S, imagenet=22)
    vs.load_checkpoint_3d = lambda path: {}            # no Kinetics checkpoint here: the teacher keeps init weights until loaded below
    _orig = getattr(tr.SwinModel, "from_pretrained")
    tr.SwinModel.from_pretrained = classmethod(lambda cls, name, *a, **k: tr.SwinModel(tr.SwinConfig(
        image_size=224, patch_size=4, num_channels=3, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=7)))
    try:
        model = mp.VIOLET_Pretrain(args, None).eval()
    finally:
        tr.SwinModel.from_pretrained = _orig
    def mask_ext(m, shape=None, device=None):
        m = m[:, None, None, :] if m.dim() == 2 else m[:, None, :, :]
        return (1.0 - m.float()) * torch.finfo(torch.float32).min
    enc_t = model.trsfr
    def go_cross(feat_img, mask_img, feat_txt, mask_txt, **kw):
        feat = torch.cat([feat_img, feat_txt], dim=1)
        mask = mask_ext(model.get_attn_mask(mask_img, mask_txt))
        o = enc_t(feat, attention_mask=mask)
        return (o[0] if not hasattr(o, "last_hidden_state") else o.last_hidden_state), ()
    model.go_cross = go_cross
    own = model.state_dict()
    if kind == "2d_feature":
        # API drift (installed transformers 5.15 vs the README's 4.26): SwinModel's attention / MLP sub-modules were renamed; the
        # oracle and the build keep the 4.26 names the reference's checkpoints carry, this shim renames for the installed class
        def to_515(k):
            if not k.startswith("feature_model."):
                return k
            for a, b in ((".attention.self.query.", ".attention.q_proj."), (".attention.self.key.", ".attention.k_proj."),
                         (".attention.self.value.", ".attention.v_proj."), (".attention.output.dense.", ".attention.o_proj."),
                         (".attention.self.relative_position_bias_table", ".attention.relative_position_bias.relative_position_bias_table"),
                         (".intermediate.dense.", ".mlp.fc1."), (".output.dense.", ".mlp.fc2.")):
                k = k.replace(a, b)
            return k
        sd = {to_515(k): v for k, v in sd.items()}
    miss = [k for k in sd if k not in own]
    assert not miss, miss[:8]
    extra = [k for k in own if k.startswith("feature_model.") and k not in sd and "relative_position_index" not in k]
    assert not extra, extra[:8]
    for k in sd:
        assert tuple(own[k].shape) == tuple(sd[k].shape), (k, own[k].shape, sd[k].shape)
    model.load_state_dict(sd, strict=False)
    img, txt, mask = R.make_batch(cfg, B)
    mb = R.default_masking(cfg, img, txt, mask, seed=5)
    neg = R.vtm_negatives_default(B)
    calls = {"i": 0}
    def fake_perm(lst):
        i = calls["i"]; calls["i"] += 1
        rest = [j for j in lst if j not in list(neg[i])]
        return np.array(list(neg[i]) + rest)
    mp.np.random.permutation = fake_perm
    agent = mp.Agent_Pretrain.__new__(mp.Agent_Pretrain)
    agent.args, agent.model, agent.patch_size = args, model, 32
    agent.loss_func = torch.nn.CrossEntropyLoss(ignore_index=-1)
    batch = dict(mb)
    out = model(batch)
    ls_mtm = agent.loss_func(out["out_mtm"].flatten(0, 1), out["ans_mtm"].flatten())
    ls_vtm = agent.loss_func(out["out_vtm"], out["ans_vtm"])
    ls_mvm = agent.calc_mvm_loss(batch, out["out_mvm"], is_train=True)
    (ls_mtm + ls_vtm + ls_mvm).backward()
    with torch.no_grad():
        if kind == "3d_feature":
            tgt = model.feature_model(batch["unmask_img"].transpose(1, 2)).transpose(1, 2).permute(0, 1, 3, 4, 2).reshape(B, T, 49, -1)
        else:
            f = model.feature_model(batch["unmask_img"].flatten(0, 1), output_hidden_states=True)["hidden_states"][-1]
            tgt = f.permute(0, 2, 1).reshape(B, T, -1, 49).permute(0, 1, 3, 2)
    d = dict(ls_mtm=np.array(float(ls_mtm)), ls_vtm=np.array(float(ls_vtm)), ls_mvm=np.array(float(ls_mvm)), neg=neg)
    put(d, "target", tgt, 512)
    gsq = 0.0
    for k, p_ in model.named_parameters():
        if p_.grad is None:
            assert k.startswith("feature_model.") or k == "enc_img.emb_odr", k
            continue
        gsq += float((p_.grad.double() ** 2).sum())
        if k.startswith("fc_mvm"):
            put(d, "g." + k, p_.grad, 32)
    d["grad_norm"] = np.array(gsq ** 0.5)
    np.savez_compressed(os.path.join(OUT, f"feature{kind[:2]}.npz"), **d)
    print(kind, "ok losses", float(ls_mtm), float(ls_vtm), float(ls_mvm), "gn", gsq ** 0.5, "target", tuple(tgt.shape), float(tgt.abs().mean()))


def gold_hog(size="tiny", T=4, B=2, S=224):
    """MVM 'hog' target head + loss through the reference's calc_mvm_loss (HOG maps: closed-form stand-in for the
    data loader's skimage output)."""
    import main_pretrain as mp
    cfg = R.make_cfg(size, T=T, img=S, mvm_target=["hog"])
    sd = R.make_state_dict(cfg)
    args = ref_args(size, T, mvm_target="hog")
    model = mp.VIOLET_Pretrain(args, None).eval()
    def mask_ext(m, shape=None, device=None):
        m = m[:, None, None, :] if m.dim() == 2 else m[:, None, :, :]
        return (1.0 - m.float()) * torch.finfo(torch.float32).min
    enc_t = model.trsfr
    def go_cross(feat_img, mask_img, feat_txt, mask_txt, **kw):
        feat = torch.cat([feat_img, feat_txt], dim=1)
        mask = mask_ext(model.get_attn_mask(mask_img, mask_txt))
        o = enc_t(feat, attention_mask=mask)
        return (o[0] if not hasattr(o, "last_hidden_state") else o.last_hidden_state), ()
    model.go_cross = go_cross
    own = model.state_dict()
    miss = [k for k in sd if k not in own]
    assert not miss, miss[:8]
    model.load_state_dict(sd, strict=False)
    img, txt, mask = R.make_batch(cfg, B)
    mb = R.default_masking(cfg, img, txt, mask, seed=5)
    neg = R.vtm_negatives_default(B)
    calls = {"i": 0}
    def fake_perm(lst):
        i = calls["i"]; calls["i"] += 1
        rest = [j for j in lst if j not in list(neg[i])]
        return np.array(list(neg[i]) + rest)
    mp.np.random.permutation = fake_perm
    agent = mp.Agent_Pretrain.__new__(mp.Agent_Pretrain)
    agent.args, agent.model, agent.patch_size = args, model, 32
    agent.loss_func = torch.nn.CrossEntropyLoss(ignore_index=-1)
    batch = dict(mb)
    batch["hog"] = R.make_hog(cfg, B)
    out = model(batch)
    ls_mtm = agent.loss_func(out["out_mtm"].flatten(0, 1), out["ans_mtm"].flatten())
    ls_vtm = agent.loss_func(out["out_vtm"], out["ans_vtm"])
    ls_mvm = agent.calc_mvm_loss(batch, out["out_mvm"], is_train=True)
    (ls_mtm + ls_vtm + ls_mvm).backward()
    d = dict(ls_mtm=np.array(float(ls_mtm.detach())), ls_vtm=np.array(float(ls_vtm.detach())), ls_mvm=np.array(float(ls_mvm.detach())), neg=neg)
    gsq = 0.0
    for k, p_ in model.named_parameters():
        if p_.grad is None:
            continue
        gsq += float((p_.grad.double() ** 2).sum())
        if k.startswith("decoder_hog"):
            put(d, "g." + k, p_.grad, 32)
    d["grad_norm"] = np.array(gsq ** 0.5)
    np.savez_compressed(os.path.join(OUT, "hog.npz"), **d)
    print("hog ok losses", float(ls_mtm), float(ls_vtm), float(ls_mvm), "gn", gsq ** 0.5)


def gold_smtm(size="tiny", T=4, B=2):
    """the smtm task -- a third fusion pass under the seq2seq attention mask (get_smtm_output main_pretrain.py:217-224,
    get_attn_mask model.py:191-199) + the MLM head, through the reference's own VIOLET_Pretrain.forward / step arithmetic."""
    import main_pretrain as mp
    cfg = R.make_cfg(size, T=T, pretrain_tasks=("vtm", "mlm", "mvm", "smtm"))
    sd = R.make_state_dict(cfg)
    args = ref_args(size, T, mvm_target="pixel")
    args.update(pretrain_tasks=["vtm", "mlm", "mvm", "smtm"])
    model = mp.VIOLET_Pretrain(args, None).eval()
    # API-drift shims (SURVEY 8c import recipe, step 6): Transformers-4.26 mask_ext semantics and BertEncoder call signature
    def mask_ext(m, shape=None, device=None):
        m = m[:, None, None, :] if m.dim() == 2 else m[:, None, :, :]
        return (1.0 - m.float()) * torch.finfo(torch.float32).min
    model.mask_ext = mask_ext
    enc_fwd = model.trsfr.forward
    def trsfr_forward(feat, mask=None, output_attentions=False, **kw):
        o = enc_fwd(feat, attention_mask=mask)
        return {"last_hidden_state": (o[0] if not hasattr(o, "last_hidden_state") else o.last_hidden_state), "attentions": ()}
    model.trsfr.forward = trsfr_forward
    own = model.state_dict()
    miss = [k for k in sd if k not in own]
    assert not miss, miss[:8]
    model.load_state_dict(sd, strict=False)
    img, txt, mask = R.make_batch(cfg, B)
    mb = R.default_masking(cfg, img, txt, mask, seed=5)
    neg = R.vtm_negatives_default(B)
    calls = {"i": 0}
    def fake_perm(lst):
        i = calls["i"]; calls["i"] += 1
        rest = [j for j in lst if j not in list(neg[i])]
        return np.array(list(neg[i]) + rest)
    mp.np.random.permutation = fake_perm
    agent = mp.Agent_Pretrain.__new__(mp.Agent_Pretrain)
    agent.args, agent.model, agent.patch_size = args, model, 32
    agent.loss_func = torch.nn.CrossEntropyLoss(ignore_index=-1)
    batch = dict(mb)
    out = model(batch)
    assert out["out_smtm"] is not None
    ls_mtm = agent.loss_func(out["out_mtm"].flatten(0, 1), out["ans_mtm"].flatten())
    ls_vtm = agent.loss_func(out["out_vtm"], out["ans_vtm"])
    ls_mvm = agent.calc_mvm_loss(batch, out["out_mvm"], is_train=True)
    ls_smtm = agent.loss_func(out["out_smtm"].flatten(0, 1), out["ans_mtm"].flatten())
    (ls_mtm + ls_vtm + ls_mvm + ls_smtm).backward()
    d = dict(ls_mtm=np.array(float(ls_mtm.detach())), ls_vtm=np.array(float(ls_vtm.detach())), ls_mvm=np.array(float(ls_mvm.detach())),
             ls_smtm=np.array(float(ls_smtm.detach())), neg=neg)
    put(d, "out_smtm", out["out_smtm"], 256)
    gsq = 0.0
    for k, p_ in model.named_parameters():
        if p_.grad is None:
            continue
        gsq += float((p_.grad.double() ** 2).sum())
        if k.startswith("trsfr.layer.0.attention.self") or k.startswith("fc_mtm.predictions.transform.dense"):
            put(d, "g." + k, p_.grad, 32)
    d["grad_norm"] = np.array(gsq ** 0.5)
    np.savez_compressed(os.path.join(OUT, "smtm.npz"), **d)
    print("smtm ok losses", float(ls_mtm), float(ls_vtm), float(ls_mvm), float(ls_smtm), "gn", gsq ** 0.5)


def gold_am(size="tiny", T=4, B=2):
    """attention-guided masking.  (1) VIOLET_Pretrain.get_att (eval mode) on the C1 batch with the attention
    probabilities taken from HF's BertSelfAttention modules (Transformers 5.15 no longer returns them from BertEncoder: forward
    hooks on `layer.attention.self`, eager attention); (2) Agent_Pretrain.masking with every sample drawn as 'am' and get_att
    replaced by a closed-form weight vector, torch seeded -- pins the position arithmetic and RNG use of the 'am' branch."""
    import main_pretrain as mp
    cfg = R.make_cfg(size, T=T)
    sd = R.make_state_dict(cfg)
    args = ref_args(size, T, mvm_target="pixel")
    args.update(pretrain_masks=["am"])
    model = mp.VIOLET_Pretrain(args, None).eval()
    def mask_ext(m, shape=None, device=None):
        m = m[:, None, None, :] if m.dim() == 2 else m[:, None, :, :]
        return (1.0 - m.float()) * torch.finfo(torch.float32).min
    model.mask_ext = mask_ext
    caps = []
    for lyr in model.trsfr.layer:
        lyr.attention.self.register_forward_hook(lambda m_, i_, o_: caps.append(o_[1]))
    enc_fwd = model.trsfr.forward
    def trsfr_forward(feat, mask=None, output_attentions=False, **kw):
        caps.clear()
        o = enc_fwd(feat, attention_mask=mask)
        return {"last_hidden_state": (o[0] if not hasattr(o, "last_hidden_state") else o.last_hidden_state), "attentions": tuple(caps)}
    model.trsfr.forward = trsfr_forward
    model.load_state_dict(sd, strict=False)
    img, txt, mask = R.make_batch(cfg, B)
    with torch.no_grad():
        _, att = model.get_att(img, txt, mask)
    d = dict(att=att.numpy().astype(np.float64))
    # (2) the 'am' branch of masking() with fixed weights
    Lv, X = (1 + 49) * T, txt.shape[1]
    fake = (1.0 + torch.sin(torch.arange(B * (Lv + X), dtype=torch.float64) * 0.37)).float().view(B, Lv + X)
    model.get_att = lambda *a, **k: (None, fake.clone())
    agent = mp.Agent_Pretrain.__new__(mp.Agent_Pretrain)
    agent.args, agent.model, agent.patch_size = args, model, 32
    agent.cls_token_id, agent.sep_token_id, agent.pad_token_id, agent.mask_token_id = 101, 102, 0, 103
    agent.prepare_batch = lambda b: b
    random.seed(7); np.random.seed(7); torch.manual_seed(7)
    o = agent.masking(img.clone(), txt.clone(), mask.clone(), None)
    d.update(fake=fake.numpy(), am_txt=o["txt"].numpy(), am_ans_mtm=o["ans_mtm"].numpy(),
             am_cov=o["mvm_mask"][:, :, 0, ::32, ::32].numpy().astype(np.uint8))
    np.savez_compressed(os.path.join(OUT, "am.npz"), **d)
    print("am ok att", tuple(att.shape), float(att.sum()), "masked txt", int((o["ans_mtm"] != -1).sum()), "covered", int(d["am_cov"].sum()))


def gold_retrieval(size="tiny", T=4, B=3):
    """VIOLET_Retrieval.forward (B x B pairs through the fusion encoder, fc head) + NormSoftmaxLoss + backward through
    the reference's own classes (main_retrieval.py, agent.py:34-50)."""
    import main_retrieval as mr
    import agent as ag
    cfg = R.make_cfg(size, T=T)
    cfg["task"] = "retrieval"
    sd = R.make_state_dict(cfg)
    args = ref_args(size, T, mvm_target="pixel")
    model = mr.VIOLET_Retrieval(args, None).eval()
    def mask_ext(m, shape=None, device=None):
        m = m[:, None, None, :] if m.dim() == 2 else m[:, None, :, :]
        return (1.0 - m.float()) * torch.finfo(torch.float32).min
    model.mask_ext = mask_ext
    enc_fwd = model.trsfr.forward
    def trsfr_forward(feat, mask=None, output_attentions=False, **kw):
        o = enc_fwd(feat, attention_mask=mask)
        return {"last_hidden_state": (o[0] if not hasattr(o, "last_hidden_state") else o.last_hidden_state), "attentions": ()}
    model.trsfr.forward = trsfr_forward
    own = model.state_dict()
    miss = [k for k in sd if k not in own]
    assert not miss, miss[:8]
    model.load_state_dict(sd, strict=False)
    img, txt, mask = R.make_batch(cfg, B)
    out, ans = model(img, txt, mask, None)
    ls = ag.NormSoftmaxLoss(temperature=args.temp)(out)
    ls.backward()
    d = dict(out=out.detach().numpy().astype(np.float64), loss=np.array(float(ls.detach())), ans=ans.numpy())
    gsq = 0.0
    for k, p_ in model.named_parameters():
        if p_.grad is None:
            continue
        gsq += float((p_.grad.double() ** 2).sum())
        if k.startswith("fc."):
            put(d, "g." + k, p_.grad, 32)
    d["grad_norm"] = np.array(gsq ** 0.5)
    np.savez_compressed(os.path.join(OUT, "retrieval.npz"), **d)
    print("retrieval ok out", out.detach().numpy().round(4).tolist(), "loss", float(ls), "gn", gsq ** 0.5)


def gold_qaoe(size="tiny", T=4, B=3, NV=1000):
    """VIOLET_QAOE.forward + CrossEntropyLoss through the reference's classes (main_qaoe.py:42-76)."""
    import main_qaoe as mq
    cfg = R.make_cfg(size, T=T)
    cfg["task"], cfg["size_vocab"] = "qaoe", NV
    sd = R.make_state_dict(cfg)
    args = ref_args(size, T, mvm_target="pixel")
    args.update(size_vocab=NV)
    model = mq.VIOLET_QAOE(args, None).eval()
    def mask_ext(m, shape=None, device=None):
        m = m[:, None, None, :] if m.dim() == 2 else m[:, None, :, :]
        return (1.0 - m.float()) * torch.finfo(torch.float32).min
    model.mask_ext = mask_ext
    enc_fwd = model.trsfr.forward
    def trsfr_forward(feat, mask=None, output_attentions=False, **kw):
        o = enc_fwd(feat, attention_mask=mask)
        return {"last_hidden_state": (o[0] if not hasattr(o, "last_hidden_state") else o.last_hidden_state), "attentions": ()}
    model.trsfr.forward = trsfr_forward
    own = model.state_dict()
    miss = [k for k in sd if k not in own]
    assert not miss, miss[:8]
    model.load_state_dict(sd, strict=False)
    img, txt, mask = R.make_batch(cfg, B)
    ans = torch.tensor([17, -1, 903])
    out, _ = model(img, txt, mask, ans)
    ls = torch.nn.CrossEntropyLoss(ignore_index=-1)(out, ans)
    ls.backward()
    d = dict(loss=np.array(float(ls.detach())), ans=ans.numpy())
    put(d, "out", out, 256)
    gsq = 0.0
    for k, p_ in model.named_parameters():
        if p_.grad is None:
            continue
        gsq += float((p_.grad.double() ** 2).sum())
        if k.startswith("fc."):
            put(d, "g." + k, p_.grad, 32)
    d["grad_norm"] = np.array(gsq ** 0.5)
    np.savez_compressed(os.path.join(OUT, "qaoe.npz"), **d)
    print("qaoe ok loss", float(ls), "gn", gsq ** 0.5)


def gold_masking(mp, agent):
    """Masking geometry: drive the reference's masking() with seeded global RNGs, record draws' effect."""
    import random
    cfg = R.make_cfg("tiny", T=4)
    img, txt, mask = R.make_batch(cfg, 3)
    agent.cls_token_id, agent.sep_token_id, agent.pad_token_id, agent.mask_token_id = 101, 102, 0, 103
    d = {}
    for name, masks in (("rm", ["rm"]), ("bm", ["bm"])):
        agent.args.pretrain_masks = masks
        random.seed(5); np.random.seed(5); torch.manual_seed(5)
        o = agent.masking(img.clone(), txt.clone(), mask.clone(), None)
        cov = o["mvm_mask"][:, :, 0, ::32, ::32]
        d[f"{name}.cov"] = cov.numpy().astype(np.uint8)
        d[f"{name}.txt"] = o["txt"].numpy(); d[f"{name}.ans_mtm"] = o["ans_mtm"].numpy()
        d[f"{name}.ans_mvm"] = o["ans_mvm"].numpy()
        d[f"{name}.img_sum"] = np.array(float(o["img"].double().sum()))
        d[f"{name}.mask_sum"] = np.array(float(o["mvm_mask"].double().sum()))
        d[f"{name}.unmask_equal"] = np.array(int(torch.equal(o["unmask_img"], img)))
    np.savez_compressed(os.path.join(OUT, "masking.npz"), **d)
    print("masking ok")


def gold_optimizer():
    """3 AdamW + WarmupLinearLR steps on a toy with one name from each of the 4 groups (agent.py:84-113)."""
    import agent as ag
    names = ["enc_img.swin.layers.0.blocks.0.mlp.fc1.weight", "trsfr.layer.0.output.dense.weight",
             "enc_img.swin.layers.0.blocks.0.attn.relative_position_bias_table", "trsfr.layer.0.output.LayerNorm.weight",
             "enc_img.swin.layers.0.blocks.0.norm1.weight"]
    class Toy(torch.nn.Module):
        def __init__(s):
            super().__init__()
            s.ps = torch.nn.ParameterDict()
        def named_parameters(s, *a, **k):
            return [(n, s.ps[n.replace(".", "_")]) for n in names]
        def parameters(s, *a, **k):
            return [p for _, p in s.named_parameters()]
    toy = Toy()
    for n in names:
        toy.ps[n.replace(".", "_")] = torch.nn.Parameter(R.closed_form(n, (6,)) * 10)
    A = ag.Agent_Base.__new__(ag.Agent_Base)
    A.args = _AttrDict(decay=1e-3, vis_backbone_lr_mul=2.0, lr=5e-5, max_grad_norm=1.0, deepspeed=False)
    A.model = toy
    A.optzr = A.build_optimizer()
    sched = ag.WarmupLinearLR(A.optzr, 20)
    d = {"names": np.array(names)}
    d["groups"] = np.array([[any(p is q for q in g["params"]) for g in A.optzr.param_groups] for _, p in toy.named_parameters()])
    lrs = []
    for step in range(1, 4):
        for i, (n, p) in enumerate(toy.named_parameters()):
            p.grad = R.closed_form(n + f".g{step}", (6,)) * 30
        tot = torch.nn.utils.clip_grad_norm_(toy.parameters(), 1.0)
        lrs.append([g["lr"] for g in A.optzr.param_groups])
        A.optzr.step(); sched.step(); A.optzr.zero_grad()
        d[f"norm{step}"] = np.array(float(tot))
        d[f"p{step}"] = np.stack([p.detach().numpy() for _, p in toy.named_parameters()])
    d["lrs"] = np.array(lrs)
    s2 = ag.WarmupLinearLR(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=5e-5), 1000)
    tab = []
    for k in range(0, 1001):
        if k in (0, 1, 50, 100, 101, 500, 999, 1000):
            tab.append((k, s2.get_last_lr()[0]))
        s2.optimizer.step(); s2.step()
    d["lr_table"] = np.array(tab)
    np.savez_compressed(os.path.join(OUT, "optimizer.npz"), **d)
    print("optimizer ok")


def write_tsv_fixture_files(root):
    """the deterministic TSV files both the generator and the test write: two image files (key, caption, base64 payload columns of
    varying length, one line with surrounding blanks) + list / sequence files of the composite view"""
    import base64
    os.makedirs(root, exist_ok=True)
    names = ["img_a.tsv", "img_b.tsv"]
    for fi, name in enumerate(names):
        with open(os.path.join(root, name), "w") as f:
            for r in range(5 + 2 * fi):
                payload = [base64.b64encode(bytes((7 * r + 3 * c + fi + k) % 251 for k in range(40 + 17 * r + c))).decode() for c in range(1 + (r % 3))]
                key = f"vid{fi}_{r:03d}" + ("x" * (30 * (r % 2)))
                f.write("\t".join([key, f" caption {r} of file {fi} "] + payload) + "\n")
    with open(os.path.join(root, "files.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    with open(os.path.join(root, "seq.tsv"), "w") as f:
        for src, row in [(0, 2), (1, 6), (1, 0), (0, 4), (0, 0), (1, 3)]:
            f.write(f"{src}\t{row}\n")
    return names


def gold_tsv():
    """the reference's TSVFile / CompositeTSVFile / create_lineidx (utils/tsv_file.py) run on the fixture files; the
    locking helper of utils/qd_common.py (it imports cv2 / matplotlib / progressbar, absent here) is replaced by plain open()."""
    import importlib
    import tempfile
    import types
    qd = types.ModuleType("utils.qd_common")
    qd.exclusive_open_to_read = lambda fname, mode="r": open(fname, mode)
    sys.modules["utils.qd_common"] = qd
    tf = importlib.import_module("utils.tsv_file")
    root = tempfile.mkdtemp()
    names = write_tsv_fixture_files(root)
    out = {}
    for name in names:
        t = tf.TSVFile(os.path.join(root, name), generate_lineidx=True)
        out[name] = dict(lineidx=[int(x) for x in open(os.path.splitext(os.path.join(root, name))[0] + ".lineidx").read().split()],
                         num_rows=t.num_rows(), rows=[t.seek(i) for i in range(t.num_rows())], keys=[t.seek_first_column(i) for i in range(t.num_rows())],
                         getitem_last=t[t.num_rows() - 1], length=len(t))
    c = tf.CompositeTSVFile(os.path.join(root, "files.txt"), os.path.join(root, "seq.tsv"), root=root)
    out["composite"] = dict(num_rows=c.num_rows(), rows=[c[i] for i in range(len(c))], keys=[c.get_key(i) for i in range(len(c))],
                            source_idx=c.get_composite_source_idx(), file_list=tf.load_list_file(os.path.join(root, "files.txt")))
    # Dataset_Base.sampling (dataset.py:142-146): dataset.py itself is stubbed out (cv2 / torchvision video transforms), so the one method
    # is compiled from the reference's source text at generation time
    import ast
    import textwrap
    src = open(os.path.join(REF, "dataset.py")).read()
    fn = [n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "sampling"][0]
    ns = {}
    exec(textwrap.dedent(ast.get_source_segment(src, fn)), ns)
    smp = ns["sampling"]
    out["sampling"] = {f"{a},{b},{n}": smp(None, a, b, n) for (a, b, n) in [(0, 31, 8), (0, 5, 6), (3, 40, 4), (0, 9, 1), (2, 2, 3), (0, 100, 16)]}
    json.dump(out, open(os.path.join(OUT, "tsv.json"), "w"))
    print("tsv ok", {k: (v["num_rows"] if isinstance(v, dict) and "num_rows" in v else len(v)) for k, v in out.items()})


def qamc_batch(cfg, B, O):
    """deterministic (B, O, X) multiple-choice batch: the text / mask rows of make_batch(B*O) with one [MASK] (id 103) per sequence,
    labelled `true` (2995) for the answer option and `false` (6270) for the others (main_qamc_tsv_mlm_head.py:26-37)"""
    img, _, _ = R.make_batch(cfg, B)
    _, txt, mask = R.make_batch(cfg, B * O)
    X = txt.shape[1]
    txt, mask = txt.clone().view(B, O, X), mask.clone().view(B, O, X)
    mask_ans = torch.full((B, O, X), -1, dtype=torch.long)
    ans_idx = torch.tensor([(2 * i + 1) % O for i in range(B)])
    for i in range(B):
        for o in range(O):
            n = int(mask[i, o].sum())
            pos = max(1, n - 2)                                   # the appended [MASK] sits before [SEP]
            txt[i, o, pos] = 103
            mask_ans[i, o, pos] = 2995 if o == int(ans_idx[i]) else 6270
    return img, txt, mask, mask_ans, ans_idx


def gold_qamc(size="tiny", T=4, B=2, O=3):
    """VIOLET_QAMC_MLM_Head.forward + Agent_QAMC_MLM_Head.step's loss / eval arithmetic through the reference's classes
    (main_qamc_tsv_mlm_head.py:61-123)."""
    import main_qamc_tsv_mlm_head as mq
    cfg = R.make_cfg(size, T=T)
    cfg["task"] = "qamc_mlm"
    sd = R.make_state_dict(cfg)
    args = ref_args(size, T, mvm_target="pixel")
    args.update(num_video_tokens=-1, size_option=O)
    model = mq.VIOLET_QAMC_MLM_Head(args, None).eval()
    def mask_ext(m, shape=None, device=None):
        m = m[:, None, None, :] if m.dim() == 2 else m[:, None, :, :]
        return (1.0 - m.float()) * torch.finfo(torch.float32).min
    model.mask_ext = mask_ext
    enc_fwd = model.trsfr.forward
    def trsfr_forward(feat, mask=None, output_attentions=False, **kw):
        o = enc_fwd(feat, attention_mask=mask)
        return {"last_hidden_state": (o[0] if not hasattr(o, "last_hidden_state") else o.last_hidden_state), "attentions": ()}
    model.trsfr.forward = trsfr_forward
    own = model.state_dict()
    miss = [k for k in sd if k not in own]
    assert not miss, miss[:8]
    extra = [k for k in own if k not in sd and "position_ids" not in k and "decoder.bias" not in k and "relative_position_index" not in k]
    assert not extra, extra[:8]
    model.load_state_dict(sd, strict=False)
    img, txt, mask, mask_ans, ans_idx = qamc_batch(cfg, B, O)
    out, ans = model(dict(img=img, txt=txt, mask=mask, mask_ans=mask_ans))
    ls = torch.nn.CrossEntropyLoss(ignore_index=-1)(out.flatten(0, len(out.shape) - 2), ans.flatten())
    ls.backward()
    # eval arithmetic of Agent_QAMC_MLM_Head.step (:111-123)
    p_true, p_false = out[:, :, 2995], out[:, :, 6270]
    sc = (p_true / (p_true + p_false))[ans.view(B * O, -1) != -1].view(B, O)
    d = dict(loss=np.array(float(ls.detach())), txt=txt.numpy(), mask=mask.numpy(), mask_ans=mask_ans.numpy(), ans_idx=ans_idx.numpy(),
             scores=sc.detach().numpy().astype(np.float64), pred=torch.argmax(sc, -1).numpy())
    put(d, "out", out, 512)
    gsq = 0.0
    for k, p_ in model.named_parameters():
        if p_.grad is None:
            continue
        gsq += float((p_.grad.double() ** 2).sum())
        if k.startswith("fc_mtm.") or k == "enc_txt.emb_txt.LayerNorm.weight":
            put(d, "g." + k, p_.grad, 32)
    d["grad_norm"] = np.array(gsq ** 0.5)
    d["no_grad"] = np.array([k for k, p_ in model.named_parameters() if p_.grad is None])
    np.savez_compressed(os.path.join(OUT, "qamc.npz"), **d)
    print("qamc ok loss", float(ls), "gn", gsq ** 0.5, "pred", d["pred"].tolist(), "ans", ans_idx.tolist(), "no_grad", d["no_grad"].tolist())


def mlm_qa_batch(cfg, B, n_ans=5):
    """deterministic (B, X) single-sequence QA batch for the MLM-head variants: make_batch's text with one [MASK] (103) before [SEP],
    labelled with the answer's vocabulary id (main_qaoe_tsv_mlm_head.py:26-60 / main_qamc_tsv_mlm_gen_ans_idx.py:30-82); the
    generative multiple-choice form scores `n_ans` candidate answer tokens, `ans_idx` = position of the right one among them."""
    img, txt, mask = R.make_batch(cfg, B)
    txt, mask = txt.clone(), mask.clone()
    X = txt.shape[1]
    ans_tok_ids = [2000 + 37 * j for j in range(n_ans)]
    ans_idx = torch.tensor([(3 * i + 1) % n_ans for i in range(B)])
    mask_ans = torch.full((B, X), -1, dtype=torch.long)
    for i in range(B):
        pos = max(1, int(mask[i].sum()) - 2)
        txt[i, pos] = 103
        mask_ans[i, pos] = ans_tok_ids[int(ans_idx[i])]
    return img, txt, mask, mask_ans, ans_idx, ans_tok_ids


def _patch_ref_fusion(model):
    def mask_ext(m, shape=None, device=None):
        m = m[:, None, None, :] if m.dim() == 2 else m[:, None, :, :]
        return (1.0 - m.float()) * torch.finfo(torch.float32).min
    model.mask_ext = mask_ext
    enc_fwd = model.trsfr.forward
    def trsfr_forward(feat, mask=None, output_attentions=False, **kw):
        o = enc_fwd(feat, attention_mask=mask)
        return {"last_hidden_state": (o[0] if not hasattr(o, "last_hidden_state") else o.last_hidden_state), "attentions": ()}
    model.trsfr.forward = trsfr_forward


def gold_mlm_qa(size="tiny", T=4, B=3):
    """SURVEY 8f.4 tail: the two remaining MLM-head variants through the reference's own classes --
    VIOLET_QAMC_MLM_Head_GEN.forward + Agent_QAMC_MLM_Head_GEN.step (main_qamc_tsv_mlm_gen_ans_idx.py:83-125: answer-token eval) and
    VIOLET_QAOE_LSMDC.forward + Agent_QAOE_LSMDC.step / get_top_k_acc (main_qaoe_lsmdc_fib.py:55-115), which
    Agent_QAOE_MLM_Head (main_qaoe_tsv_mlm_head.py:101-130) inherits unchanged."""
    import main_qamc_tsv_mlm_gen_ans_idx as mg
    import main_qaoe_lsmdc_fib as ml
    cfg = R.make_cfg(size, T=T)
    cfg["task"] = "qamc_mlm"
    sd = R.make_state_dict(cfg)
    args = ref_args(size, T, mvm_target="pixel")
    args.update(num_video_tokens=-1, size_option=5, size_vocab=-1, deepspeed=False, freeze_violet=False)
    img, txt, mask, mask_ans, ans_idx, ans_tok_ids = mlm_qa_batch(cfg, B)
    batch = dict(img=img, txt=txt, mask=mask, mask_ans=mask_ans, ans_idx=ans_idx)
    d = dict(txt=txt.numpy(), mask=mask.numpy(), mask_ans=mask_ans.numpy(), ans_idx=ans_idx.numpy(), ans_tok_ids=np.array(ans_tok_ids))
    outs = {}
    for tag, mod, mcls, acls in (("gen", mg, "VIOLET_QAMC_MLM_Head_GEN", "Agent_QAMC_MLM_Head_GEN"), ("oe", ml, "VIOLET_QAOE_LSMDC", "Agent_QAOE_LSMDC")):
        model = getattr(mod, mcls)(args, None).eval()
        _patch_ref_fusion(model)
        own = model.state_dict()
        assert not [k for k in sd if k not in own]
        model.load_state_dict(sd, strict=False)
        agent = getattr(mod, acls).__new__(getattr(mod, acls))
        agent.args, agent.model, agent.ans_tok_ids = args, model, ans_tok_ids
        agent.forward_step = lambda b, m=model: m(b)
        agent.loss_func = torch.nn.CrossEntropyLoss(ignore_index=-1)
        with torch.no_grad():
            r = agent.step(dict(batch), False)                      # the reference's own eval arithmetic
        out, ans = model(dict(batch))
        ls = agent.loss_func(out.flatten(0, len(out.shape) - 2), ans.flatten())
        outs[tag] = out.detach()
        d[f"{tag}.loss"] = np.array(float(ls.detach()))
        if tag == "gen":
            d["gen.ac"] = np.array(r, dtype=np.float64)
            p = out[:, :, ans_tok_ids][ans != -1]
            d["gen.scores"] = (p / p.sum(-1, keepdim=True)).detach().numpy().astype(np.float64)
        else:
            d["oe.ac_1"], d["oe.ac_5"] = np.array(r["ac_1"], dtype=np.float64), np.array(r["ac_5"], dtype=np.float64)
            # top-k on a crafted (logits, answers) pair incl. a row without any answer: get_top_k_acc's padding rule
            g = torch.Generator().manual_seed(3)
            lo = torch.randn(4, 6, 50, generator=g)
            an = torch.full((4, 6), -1, dtype=torch.long)
            an[0, 2], an[1, 4], an[3, 1] = int(lo[0, 2].argmax()), int(lo[1, 4].topk(4).indices[3]), int(lo[3, 1].argmin())
            d["oe.toy_logits"], d["oe.toy_ans"] = lo.numpy(), an.numpy()
            d["oe.toy_ac1"] = np.array(agent.get_top_k_acc(lo, an, k=1), dtype=np.float64)
            d["oe.toy_ac5"] = np.array(agent.get_top_k_acc(lo, an, k=5), dtype=np.float64)
        put(d, f"{tag}.out", out, 512)
    assert torch.equal(outs["gen"], outs["oe"])                     # the two variants share one forward
    np.savez_compressed(os.path.join(OUT, "mlm_qa.npz"), **d)
    print("mlm_qa ok", {k: (v.tolist() if v.size < 8 else v.shape) for k, v in d.items() if k.split(".")[-1] in ("loss", "ac", "ac_1", "ac_5", "toy_ac1", "toy_ac5")})


def gold_loop():
    """tests/golden/loop.json: the control flow of the reference's training driver -- `MetaLoader` (dataset.py:511-547; the class body is
    exec'd from the reference's file here, the module itself needs the data stack), `RunningMeter` (utils/logger.py) and
    `Agent_Pretrain_YAML.run_meta_loader` / `go_ep` (main_pretrain_yaml.py:123-194) with the step / evaluate / save hooks replaced by
    recorders: which (task, batch) comes when, when evaluation and checkpoints happen, what the smoothed losses and the log dict hold."""
    import re
    from utils.logger import RunningMeter
    import main_pretrain_yaml as MY
    src = open(os.path.join(REF, "dataset.py")).read()
    m = re.search(r"^class MetaLoader\(object\):.*?(?=^class |\Z)", src, re.S | re.M)
    ns = dict(random=random, T=torch, DIST=None)
    exec(m.group(0), ns)
    RefMeta = ns["MetaLoader"]
    mk = lambda n, tag: torch.utils.data.DataLoader([f"{tag}{i}" for i in range(n)], batch_size=1, collate_fn=lambda x: x[0])
    out = {}
    for acc in (1, 2):
        random.seed(5)
        ml = RefMeta({"a": (mk(3, "a"), 2), "b": mk(2, "b")}, accum_steps=acc)
        it = iter(ml)
        out[f"meta_accum{acc}"] = [list(next(it)) for _ in range(14)]
    rm = RunningMeter("x")
    vals = []
    for v in (2.0, 1.0, 4.0, -1.0):
        rm(v); vals.append(rm.val)
    out["running_meter"] = vals

    class Rec(MY.Agent_Pretrain_YAML):
        def __init__(self, args):
            self.args, self.trace = args, []
            self.task2loss, self.log, self.ds_tr_steps, self.global_step = {}, MY.defaultdict(list), MY.defaultdict(int), 0
            self.n = 0
        def masking(self, img, txt, mask, vq): return {"masked": True}
        def prepare_batch(self, b): return b
        def step(self, batch, is_train):
            self.n += 1
            self.trace.append(["step", batch["id"]])
            return {"mtm": 1.0 + 0.5 * self.n, "vtm": 0.25 * self.n, "mvm": -1}
        def evaluate(self, dl):
            self.trace.append(["eval", dl])
            return {"mtm": 0.5, "vtm": 0.75}
        def save_model(self, ep, ds, step): self.trace.append(["save", ep, ds, step])
        def log_memory(self, ep=-1, step=-1): return "mem"
        def log_dict_to_wandb(self, d, step=-1): pass
    dump = lambda a: dict(trace=a.trace, meters={k: v.val for k, v in a.task2loss.items()}, log={k: v for k, v in a.log.items()},
                          ds_tr_steps=dict(a.ds_tr_steps), global_step=a.global_step)
    for max_iter, eval_step in ((7, 3), (6, 3), (4, 10)):
        a = Rec(_AttrDict(iter_per_ep=4, logging_steps=2, eval_step=eval_step, max_iter=max_iter))
        stream = [("ds%d" % (i % 2), {"id": i, "img": None, "txt": None, "mask": None, "vq": None}) for i in range(50)]
        a.run_meta_loader(stream, {"val": "VL"})
        out[f"run_meta_loader_{max_iter}_{eval_step}"] = dump(a)
    for iter_per_ep, eval_step in ((5, 2), (4, 2)):
        a = Rec(_AttrDict(iter_per_ep={"d": iter_per_ep}, logging_steps=2, eval_step={"d": eval_step}))
        class DL(list):
            pass
        a.go_ep({"d": DL({"id": i, "img": None, "txt": None, "mask": None, "vq": None, "vid": ["v"]} for i in range(20))}, {"val": "VL"}, 2)
        out[f"go_ep_{iter_per_ep}_{eval_step}"] = dump(a)
    json.dump(out, open(os.path.join(OUT, "loop.json"), "w"), indent=1)
    print("wrote loop.json")


def gold_inflate(vs):
    """SwinTransformer3D.inflate_weights (video_swin.py:484-535) on a synthetic image-Swin checkpoint: one with 7x7-window tables (tiled
    only) and one with 6x6-window tables (bicubic resize to 13x13, then tiled); the reference module's state_dict afterwards"""
    import tempfile
    arch = dict(embed_dim=32, depths=(1, 1, 2, 1), num_heads=(1, 2, 4, 8), window_size=(8, 7, 7), patch_size=(2, 4, 4))
    d = {}
    for tag, side in (("w7", 7), ("w6", 6)):
        g = torch.Generator().manual_seed(31 + side)
        probe = vs.SwinTransformer3D(pretrained=None, pretrained2d=True, **arch)
        sd2 = {}
        for k, v in probe.state_dict().items():
            if "relative_position_index" in k:
                sd2[k] = torch.zeros(side * side, side * side, dtype=torch.long)             # a 2-D model's buffer: must be dropped
            elif "relative_position_bias_table" in k:
                sd2[k] = torch.randn((2 * side - 1) ** 2, v.shape[1], generator=g) * 0.02
            elif k == "patch_embed.proj.weight":
                sd2[k] = torch.randn(v.shape[0], 3, 4, 4, generator=g) * 0.1
            else:
                sd2[k] = torch.randn(v.shape, generator=g) * 0.05
        sd2["layers.0.blocks.0.attn_mask"] = torch.zeros(4, 49, 49)
        path = os.path.join(tempfile.mkdtemp(), f"swin2d_{tag}.pth")
        torch.save({"model": sd2}, path)
        m = vs.SwinTransformer3D(pretrained=path, pretrained2d=True, **arch)
        m.init_weights()
        out = m.state_dict()
        for k in ("patch_embed.proj.weight", "layers.0.blocks.0.attn.relative_position_bias_table", "layers.2.blocks.1.attn.relative_position_bias_table",
                  "layers.1.blocks.0.mlp.fc1.bias", "norm.weight"):
            d[f"{tag}.in.{k}"] = sd2[k].numpy().astype(np.float32)          # whole tensors: the test feeds them to the product's inflation
            d[f"{tag}.out.{k}"] = out[k].numpy().astype(np.float32)
        d[f"{tag}.dropped"] = np.array(sorted(k for k in sd2 if k not in out or tuple(sd2[k].shape) != tuple(out[k].shape) and "table" not in k and "patch_embed.proj" not in k))
    np.savez_compressed(os.path.join(OUT, "inflate2d.npz"), **d)
    print("inflate2d:", len(d), "entries")


def gold_encvideo_odr(size="tiny", T=4, B=3):
    """EncVideo.forward with a frame order and a visual-token mask (model.py:61-67,75): outputs of the reference module itself"""
    mp, args, model = build_ref_model(size, T)
    cfg = R.make_cfg(size, T=T)
    sd = R.make_state_dict(cfg)
    model.load_state_dict(sd, strict=False)
    img, txt, mask = R.make_batch(cfg, B)
    h = w = img.shape[-1] // 32
    odr = [[0, 1, 2, 3], [1, 0, 2, 3], [3, 2, 1, 0]]
    vt = torch.ones(B, T, 1 + h * w, dtype=torch.long)
    vt[1, 2, 5:] = 0
    vt[2, 0] = 0
    with torch.no_grad():
        f, m = model.enc_img(img, odr, vt)
        f0, m0 = model.enc_img(img)
    d = {"odr": np.asarray(odr, np.int64), "vt_mask": vt.numpy(), "mask": m.numpy(), "mask_plain": m0.numpy()}
    put(d, "feat", f, 256)
    put(d, "feat_plain", f0, 256)
    np.savez_compressed(os.path.join(OUT, "encvideo_odr.npz"), **d)
    print("encvideo_odr: feat", tuple(f.shape), "mask zeros", int((m == 0).sum()))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import_reference()
    from visbackbone import video_swin as vs
    if "--tsv-only" in sys.argv:
        gold_tsv()
        sys.exit(0)
    if "--loop-only" in sys.argv:
        gold_loop()
        sys.exit(0)
    if "--inflate-only" in sys.argv:
        gold_inflate(vs)
        sys.exit(0)
    if "--encvideo-odr-only" in sys.argv:
        gold_encvideo_odr()
        sys.exit(0)
    if "--qamc-only" in sys.argv:
        gold_qamc()
        sys.exit(0)
    if "--mlm-qa-only" in sys.argv:
        gold_mlm_qa()
        sys.exit(0)
    if "--qaoe-only" in sys.argv:
        gold_qaoe()
        sys.exit(0)
    if "--retrieval-only" in sys.argv:
        gold_retrieval()
        sys.exit(0)
    if "--am-only" in sys.argv:
        gold_am()
        sys.exit(0)
    if "--smtm-only" in sys.argv:
        gold_smtm()
        sys.exit(0)
    if "--hog-only" in sys.argv:
        gold_hog()
        sys.exit(0)
    if "--feature-only" in sys.argv:
        if "--2d" not in sys.argv:
            gold_feature("3d_feature")
        gold_feature("2d_feature")
        sys.exit(0)
    if "--vq-only" not in sys.argv:
        gold_helpers(vs)
        gold_reduced_swin(vs)
    if "--vq-only" not in sys.argv:
        mp, agent = gold_c1("tiny", 4, 2, "c1")
        gold_masking(mp, agent)
        gold_optimizer()
    gold_vq()
    gold_feature("3d_feature")
    gold_feature("2d_feature")
    gold_hog()
    gold_smtm()
    gold_am()
    gold_retrieval()
    gold_qaoe()
    gold_qamc()
    gold_mlm_qa()
    gold_encvideo_odr()
    gold_inflate(vs)
    gold_tsv()
    gold_loop()
