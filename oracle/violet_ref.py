"""CPU oracle for the VIOLETv2 pretraining step  --  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch *functional* restatement (plain torch CPU, fp32 or
fp64) of the reference algorithm on the hot path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the
product package (`pytorch_empirical_mvm_amd`) never does and fails loudly when
its HIP library is missing.

Pinning: the reference ships no tests / golden vectors, so
the oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, produced in the
build container by `tools/gen_goldens.py` (imports /root/reference with module
stubs) and committed as `tests/golden/*.npz`; `tests/test_oracle_golden.py`
re-checks the oracle against them on every run.  The BERT arithmetic lives in
the third-party `transformers` package (README pins 4.26; container has 5.x,
layer math unchanged) -- pinned through the same fixtures.

Every function cites the reference file:line it follows (paths relative to
/root/reference).
"""
import math
import zlib
from functools import lru_cache

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# architecture table  (visbackbone/swin_tiny.py:2-24, swin_base.py:3-6,
# swin_large.py:3-6, swin_*_patch244_*.py:4 ; get_vidswin_model video_swin.py:573-639)
# ----------------------------------------------------------------------------
ARCH = {
    "tiny": dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window=(8, 7, 7)),
    "small": dict(embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24), window=(8, 7, 7)),
    "base": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window=(8, 7, 7)),
    "large": dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48), window=(8, 7, 7)),
    "large384": dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48), window=(8, 12, 12)),
}
PATCH = (2, 4, 4)
BERT = dict(vocab=30522, hidden=768, layers=12, heads=12, ffn=3072, max_pos=512, types=2, eps=1e-12)


def make_cfg(size="base", T=8, img=224, n_txt=32, max_size_frame=None, max_size_patch=14,
             size_patch=32, temp=0.05, bert_layers=12, mvm_target="pixel", arch=None, vocab=None,
             size_vq=8192, dvae_hid=256, dvae_vocab=8192, teacher_arch=None, pretrain_tasks=("vtm", "mlm", "mvm")):
    a = dict(ARCH[size]) if arch is None else dict(arch)
    cfg = dict(a)
    cfg["pretrain_tasks"] = tuple(pretrain_tasks)        # args_pretrain.json:19-23 ; "smtm" adds the seq2seq MLM pass
    cfg["teacher_arch"] = dict(ARCH["base"]) if teacher_arch is None else dict(teacher_arch)   # main_pretrain.py:157,168: always "base"
    cfg.update(size=size, T=T, img=img, n_txt=n_txt, max_size_frame=max_size_frame or max(T, 6),
               max_size_patch=max_size_patch, size_patch=size_patch, temp=temp,
               bert_layers=bert_layers, mvm_target=mvm_target, hidden=BERT["hidden"],
               vocab=vocab or BERT["vocab"], size_vq=size_vq, dvae_hid=dvae_hid, dvae_vocab=dvae_vocab)
    return cfg


# ----------------------------------------------------------------------------
# parameter inventory with the reference's state_dict key names
# ----------------------------------------------------------------------------
def param_shapes(cfg):
    """Ordered {key: shape} of every learnable tensor of VIOLET_Pretrain (pixel target)."""
    E, depths, heads, win = cfg["embed_dim"], cfg["depths"], cfg["num_heads"], cfg["window"]
    H, V = cfg["hidden"], cfg["vocab"]
    s = {}
    # EncTxt (model.py:81-94) -> HF BertEmbeddings
    s["enc_txt.emb_txt.word_embeddings.weight"] = (V, H)
    s["enc_txt.emb_txt.position_embeddings.weight"] = (BERT["max_pos"], H)
    s["enc_txt.emb_txt.token_type_embeddings.weight"] = (BERT["types"], H)
    s["enc_txt.emb_txt.LayerNorm.weight"] = (H,)
    s["enc_txt.emb_txt.LayerNorm.bias"] = (H,)
    # fusion encoder (model.py:124-133) -> HF BertEncoder
    for l in range(cfg["bert_layers"]):
        p = f"trsfr.layer.{l}."
        for n in ("query", "key", "value"):
            s[p + f"attention.self.{n}.weight"] = (H, H)
            s[p + f"attention.self.{n}.bias"] = (H,)
        s[p + "attention.output.dense.weight"] = (H, H)
        s[p + "attention.output.dense.bias"] = (H,)
        s[p + "attention.output.LayerNorm.weight"] = (H,)
        s[p + "attention.output.LayerNorm.bias"] = (H,)
        s[p + "intermediate.dense.weight"] = (BERT["ffn"], H)
        s[p + "intermediate.dense.bias"] = (BERT["ffn"],)
        s[p + "output.dense.weight"] = (H, BERT["ffn"])
        s[p + "output.dense.bias"] = (H,)
        s[p + "output.LayerNorm.weight"] = (H,)
        s[p + "output.LayerNorm.bias"] = (H,)
    # EncVideo (model.py:9-30)
    C_out = E * 8
    s["enc_img.emb_cls"] = (1, 1, 1, H)
    s["enc_img.emb_pos"] = (1, 1, 1 + cfg["max_size_patch"] ** 2, H)
    s["enc_img.emb_len"] = (1, cfg["max_size_frame"], 1, H)
    s["enc_img.emb_odr"] = (1, 1, 1, H)
    if C_out != H:
        s["enc_img.fc.weight"] = (H, C_out)
        s["enc_img.fc.bias"] = (H,)
    s["enc_img.norm.weight"] = (H,)
    s["enc_img.norm.bias"] = (H,)
    # Video-Swin (video_swin.py:410-468)
    sw = "enc_img.swin."
    s[sw + "patch_embed.proj.weight"] = (E, 3) + PATCH
    s[sw + "patch_embed.proj.bias"] = (E,)
    s[sw + "patch_embed.norm.weight"] = (E,)
    s[sw + "patch_embed.norm.bias"] = (E,)
    ntab = (2 * win[0] - 1) * (2 * win[1] - 1) * (2 * win[2] - 1)
    for i, (d, nh) in enumerate(zip(depths, heads)):
        C = E * 2 ** i
        for b in range(d):
            p = sw + f"layers.{i}.blocks.{b}."
            s[p + "norm1.weight"] = (C,)
            s[p + "norm1.bias"] = (C,)
            s[p + "attn.relative_position_bias_table"] = (ntab, nh)
            s[p + "attn.qkv.weight"] = (3 * C, C)
            s[p + "attn.qkv.bias"] = (3 * C,)
            s[p + "attn.proj.weight"] = (C, C)
            s[p + "attn.proj.bias"] = (C,)
            s[p + "norm2.weight"] = (C,)
            s[p + "norm2.bias"] = (C,)
            s[p + "mlp.fc1.weight"] = (4 * C, C)
            s[p + "mlp.fc1.bias"] = (4 * C,)
            s[p + "mlp.fc2.weight"] = (C, 4 * C)
            s[p + "mlp.fc2.bias"] = (C,)
        if i < len(depths) - 1:
            p = sw + f"layers.{i}.downsample."
            s[p + "reduction.weight"] = (2 * C, 4 * C)
            s[p + "norm.weight"] = (4 * C,)
            s[p + "norm.bias"] = (4 * C,)
    s[sw + "norm.weight"] = (C_out,)
    s[sw + "norm.bias"] = (C_out,)
    # heads (main_pretrain.py:146-151,178-179)
    s["fc.1.weight"] = (2 * H, H)
    s["fc.1.bias"] = (2 * H,)
    s["fc.3.weight"] = (1, 2 * H)
    s["fc.3.bias"] = (1,)
    if cfg.get("task", "pretrain") == "retrieval":       # VIOLET_Retrieval main_retrieval.py:56-61: VIOLET_Base + fc
        return s
    if cfg.get("task", "pretrain") == "qaoe":            # VIOLET_QAOE main_qaoe.py:42-47
        s["fc.3.weight"] = (int(cfg["size_vocab"]), 2 * H)
        s["fc.3.bias"] = (int(cfg["size_vocab"]),)
        return s
    if cfg.get("task", "pretrain") == "qamc_mlm":        # VIOLET_QAMC_MLM_Head main_qamc_tsv_mlm_head.py:61-71: `del self.fc`, + fc_mtm + emb_task
        for k in ("fc.1.weight", "fc.1.bias", "fc.3.weight", "fc.3.bias"):
            del s[k]
        s["emb_task"] = (10, H)
    s["fc_mtm.predictions.bias"] = (V,)
    s["fc_mtm.predictions.transform.dense.weight"] = (H, H)
    s["fc_mtm.predictions.transform.dense.bias"] = (H,)
    s["fc_mtm.predictions.transform.LayerNorm.weight"] = (H,)
    s["fc_mtm.predictions.transform.LayerNorm.bias"] = (H,)
    s["fc_mtm.predictions.decoder.weight"] = (V, H)
    if cfg.get("task", "pretrain") == "qamc_mlm":
        return s
    if "pixel" in cfg["mvm_target"]:
        s["decoder_pixel.0.weight"] = (cfg["size_patch"] ** 2 * 3, H, 1, 1)
        s["decoder_pixel.0.bias"] = (cfg["size_patch"] ** 2 * 3,)
    if "hog" in cfg["mvm_target"]:          # main_pretrain.py:180-183
        s["decoder_hog.0.weight"] = (cfg["size_patch"] ** 2, H, 1, 1)
        s["decoder_hog.0.bias"] = (cfg["size_patch"] ** 2,)
    if "vq" in cfg["mvm_target"]:
        # main_pretrain.py:194-209 (on-the-fly dVAE branch): Conv2d 1x1 H -> 2H, PixelShuffle(32/8), Dropout, Linear, ReLU, Linear
        up = cfg["size_patch"] // 8
        c = 2 * H // (up * up)
        s["decoder_vq.0.weight"] = (2 * H, H, 1, 1)
        s["decoder_vq.0.bias"] = (2 * H,)
        s["fc_mvm.1.weight"] = (2 * c, c)
        s["fc_mvm.1.bias"] = (2 * c,)
        s["fc_mvm.3.weight"] = (cfg["size_vq"], 2 * c)
        s["fc_mvm.3.bias"] = (cfg["size_vq"],)
    for kind in ("3d_feature", "2d_feature"):
        if kind in cfg["mvm_target"]:      # main_pretrain.py:153-174: Dropout, Linear(H, 2H), ReLU, Linear(2H, feat_size)
            feat = cfg["teacher_arch"]["embed_dim"] * 8
            s["fc_mvm.1.weight"] = (2 * H, H)
            s["fc_mvm.1.bias"] = (2 * H,)
            s["fc_mvm.3.weight"] = (feat, 2 * H)
            s["fc_mvm.3.bias"] = (feat,)
    return s


def swin_param_shapes(arch, win, prefix):
    """{key: shape} of a SwinTransformer3D (video_swin.py:410-468) under `prefix`."""
    E, depths, heads = arch["embed_dim"], arch["depths"], arch["num_heads"]
    s = {}
    s[prefix + "patch_embed.proj.weight"] = (E, 3) + PATCH
    s[prefix + "patch_embed.proj.bias"] = (E,)
    s[prefix + "patch_embed.norm.weight"] = (E,)
    s[prefix + "patch_embed.norm.bias"] = (E,)
    ntab = (2 * win[0] - 1) * (2 * win[1] - 1) * (2 * win[2] - 1)
    for i, (d, nh) in enumerate(zip(depths, heads)):
        C = E * 2 ** i
        for b in range(d):
            p = prefix + f"layers.{i}.blocks.{b}."
            s[p + "norm1.weight"] = (C,); s[p + "norm1.bias"] = (C,)
            s[p + "attn.relative_position_bias_table"] = (ntab, nh)
            s[p + "attn.qkv.weight"] = (3 * C, C); s[p + "attn.qkv.bias"] = (3 * C,)
            s[p + "attn.proj.weight"] = (C, C); s[p + "attn.proj.bias"] = (C,)
            s[p + "norm2.weight"] = (C,); s[p + "norm2.bias"] = (C,)
            s[p + "mlp.fc1.weight"] = (4 * C, C); s[p + "mlp.fc1.bias"] = (4 * C,)
            s[p + "mlp.fc2.weight"] = (C, 4 * C); s[p + "mlp.fc2.bias"] = (C,)
        if i < len(depths) - 1:
            p = prefix + f"layers.{i}.downsample."
            s[p + "reduction.weight"] = (2 * C, 4 * C)
            s[p + "norm.weight"] = (4 * C,); s[p + "norm.bias"] = (4 * C,)
    s[prefix + "norm.weight"] = (E * 8,)
    s[prefix + "norm.bias"] = (E * 8,)
    return s


def hf_swin2d_param_shapes(arch, prefix, ws=7):
    """{key: shape} of HF `transformers.SwinModel` (third-party; built at visbackbone/swin.py:16-35) under `prefix`."""
    E, depths, heads = arch["embed_dim"], arch["depths"], arch["num_heads"]
    s = {}
    s[prefix + "embeddings.patch_embeddings.projection.weight"] = (E, 3, 4, 4)
    s[prefix + "embeddings.patch_embeddings.projection.bias"] = (E,)
    s[prefix + "embeddings.norm.weight"] = (E,)
    s[prefix + "embeddings.norm.bias"] = (E,)
    for i, (d, nh) in enumerate(zip(depths, heads)):
        C = E * 2 ** i
        for b in range(d):
            p = prefix + f"encoder.layers.{i}.blocks.{b}."
            s[p + "layernorm_before.weight"] = (C,); s[p + "layernorm_before.bias"] = (C,)
            s[p + "attention.self.relative_position_bias_table"] = ((2 * ws - 1) ** 2, nh)
            for n in ("query", "key", "value"):
                s[p + f"attention.self.{n}.weight"] = (C, C); s[p + f"attention.self.{n}.bias"] = (C,)
            s[p + "attention.output.dense.weight"] = (C, C); s[p + "attention.output.dense.bias"] = (C,)
            s[p + "layernorm_after.weight"] = (C,); s[p + "layernorm_after.bias"] = (C,)
            s[p + "intermediate.dense.weight"] = (4 * C, C); s[p + "intermediate.dense.bias"] = (4 * C,)
            s[p + "output.dense.weight"] = (C, 4 * C); s[p + "output.dense.bias"] = (C,)
        if i < len(depths) - 1:
            p = prefix + f"encoder.layers.{i}.downsample."
            s[p + "reduction.weight"] = (2 * C, 4 * C)
            s[p + "norm.weight"] = (4 * C,); s[p + "norm.bias"] = (4 * C,)
    s[prefix + "layernorm.weight"] = (E * 8,)
    s[prefix + "layernorm.bias"] = (E * 8,)
    return s


def teacher_param_shapes(cfg):
    """frozen feature teacher `feature_model.*` (main_pretrain.py:153-174): VideoSwin-B for '3d_feature', HF Swin-B for '2d_feature'"""
    if "3d_feature" in cfg["mvm_target"]:
        return swin_param_shapes(cfg["teacher_arch"], tuple(cfg["teacher_arch"]["window"]), "feature_model.")
    if "2d_feature" in cfg["mvm_target"]:
        return hf_swin2d_param_shapes(cfg["teacher_arch"], "feature_model.", cfg["teacher_arch"]["window"][-1])
    return {}


def dvae_param_shapes(cfg):
    """Frozen DALL-E dVAE encoder (visbackbone/dalle/encoder.py:41-93): keys as in its state_dict, prefixed `dalle.encoder.`"""
    nh, V = cfg["dvae_hid"], cfg["dvae_vocab"]
    s = {}
    pre = "dalle.encoder.blocks."
    s[pre + "input.w"] = (nh, 3, 7, 7); s[pre + "input.b"] = (nh,)
    n_in = nh
    for gi, mult in enumerate((1, 2, 4, 8)):
        n_out = mult * nh
        for bi in range(2):
            q = pre + f"group_{gi + 1}.block_{bi + 1}."
            if n_in != n_out:
                s[q + "id_path.w"] = (n_out, n_in, 1, 1); s[q + "id_path.b"] = (n_out,)
            hid = n_out // 4
            s[q + "res_path.conv_1.w"] = (hid, n_in, 3, 3); s[q + "res_path.conv_1.b"] = (hid,)
            s[q + "res_path.conv_2.w"] = (hid, hid, 3, 3); s[q + "res_path.conv_2.b"] = (hid,)
            s[q + "res_path.conv_3.w"] = (hid, hid, 3, 3); s[q + "res_path.conv_3.b"] = (hid,)
            s[q + "res_path.conv_4.w"] = (n_out, hid, 1, 1); s[q + "res_path.conv_4.b"] = (n_out,)
            n_in = n_out
    s[pre + "output.conv.w"] = (V, 8 * nh, 1, 1); s[pre + "output.conv.b"] = (V,)
    return s


def _hash_uniform(n, seed):
    """n uniforms in [0,1): splitmix64 of (index, seed) in pure integer numpy -> bit-identical on every platform."""
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x632BE59BD9B4E019)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def closed_form(key, shape, dtype=torch.float32):
    """Deterministic closed-form tensor (integer hash -> uniform, scaled like the reference's own initialisation:
    Linear / embeddings std .02, LayerNorm ~1 / ~0, conv fan-in uniform) so the network is as well conditioned as a
    freshly initialised one.  Re-creatable bit-identically anywhere; golden fixtures therefore store outputs only
    (SURVEY 8c 'Golden-vector policy')."""
    n = int(np.prod(shape))
    k = zlib.crc32(key.encode())
    u = 2.0 * _hash_uniform(n, k) - 1.0                       # uniform(-1,1), std 1/sqrt(3)
    r3 = math.sqrt(3.0)
    is_norm_w = key.endswith("weight") and ("norm" in key.lower()) and len(shape) == 1
    is_bias = key.endswith("bias") and len(shape) == 1
    if is_norm_w:
        v = 1.0 + 0.1 * u
    elif is_bias:
        v = 0.02 * u
    elif "emb_" in key or "embeddings" in key:
        v = 0.02 * r3 * u
    elif "relative_position_bias_table" in key:
        v = 0.2 * r3 * u
    elif key.startswith("dalle.") and key.endswith(".w"):
        fan_in = int(np.prod(shape[1:]))
        v = r3 * u / math.sqrt(fan_in)                         # dalle/utils.py:28 : normal(std = 1/sqrt(n_in*kw^2)), same variance
    elif key.startswith("dalle.") and key.endswith(".b"):
        v = 0.05 * u                                           # (zeros in the reference; non-zero here so the bias path is exercised)
    elif "patch_embed.proj.weight" in key or key.startswith("decoder_pixel") or key.startswith("fc.") or key.startswith("decoder_vq") \
            or key.startswith("fc_mvm") or key.startswith("decoder_hog"):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else n
        v = u / math.sqrt(fan_in)
    else:
        v = 0.02 * r3 * u
    return torch.from_numpy(v.reshape(shape)).to(dtype)


def make_state_dict(cfg, dtype=torch.float32):
    sd = {k: closed_form(k, shp, dtype) for k, shp in param_shapes(cfg).items()}
    if "vq" in cfg["mvm_target"]:
        sd.update({k: closed_form(k, shp, dtype) for k, shp in dvae_param_shapes(cfg).items()})
    sd.update({k: closed_form(k, shp, dtype) for k, shp in teacher_param_shapes(cfg).items()})
    return sd


def make_batch(cfg, B, dtype=torch.float32):
    """Closed-form synthetic batch (collate schema main_pretrain_yaml.py:69-80)."""
    T, S, X = cfg["T"], cfg["img"], cfg["n_txt"]
    n = B * T * 3 * S * S
    smooth = np.sin(np.arange(n, dtype=np.float64) * 0.0137) * 0.8
    noise = (2.0 * _hash_uniform(n, 77) - 1.0) * 1.2
    img = torch.from_numpy((smooth + noise).reshape(B, T, 3, S, S)).to(dtype)
    txt = torch.zeros(B, X, dtype=torch.long)
    for b in range(B):
        ln = 6 + (5 * b) % (X - 8)
        ids = 1000 + (np.arange(ln) * 7919 + b * 104729) % (cfg["vocab"] - 1000)
        txt[b, 0] = 101
        txt[b, 1:1 + ln] = torch.from_numpy(ids)
        txt[b, 1 + ln] = 102
    mask = (txt != 0).long()
    return img, txt, mask


# ----------------------------------------------------------------------------
# Video-Swin helpers
# ----------------------------------------------------------------------------
def get_window_size(x_size, window_size, shift_size=None):
    """video_swin.py:95-108"""
    uw = list(window_size)
    us = list(shift_size) if shift_size is not None else None
    for i in range(len(x_size)):
        if x_size[i] <= window_size[i]:
            uw[i] = x_size[i]
            if us is not None:
                us[i] = 0
    return tuple(uw) if us is None else (tuple(uw), tuple(us))


def window_partition(x, ws):
    """video_swin.py:84-88   (B,D,H,W,C) -> (B*nW, wd*wh*ww, C)"""
    B, D, H, W, C = x.shape
    x = x.reshape(B, D // ws[0], ws[0], H // ws[1], ws[1], W // ws[2], ws[2], C)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, ws[0] * ws[1] * ws[2], C)


def window_reverse(w, ws, B, D, H, W):
    """video_swin.py:90-93"""
    x = w.reshape(B, D // ws[0], H // ws[1], W // ws[2], ws[0], ws[1], ws[2], -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, D, H, W, -1)


def compute_mask(D, H, W, ws, ss, dtype=torch.float32):
    """video_swin.py:292-307   (nW, N, N) with 0 / -100"""
    img_mask = torch.zeros((1, D, H, W, 1), dtype=dtype)
    cnt = 0
    for d in (slice(-ws[0]), slice(-ws[0], -ss[0]), slice(-ss[0], None)):
        for h in (slice(-ws[1]), slice(-ws[1], -ss[1]), slice(-ss[1], None)):
            for w in (slice(-ws[2]), slice(-ws[2], -ss[2]), slice(-ss[2], None)):
                img_mask[:, d, h, w, :] = cnt
                cnt += 1
    mw = window_partition(img_mask, ws).squeeze(-1)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(am != 0, torch.full_like(am, -100.0), torch.zeros_like(am))


@lru_cache(maxsize=None)
def relative_position_index(win):
    """video_swin.py:123-137  (built for the CONFIGURED window, sliced [:N,:N] at use :155)"""
    cd, ch, cw = torch.arange(win[0]), torch.arange(win[1]), torch.arange(win[2])
    coords = torch.stack(torch.meshgrid(cd, ch, cw, indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += win[0] - 1
    rel[:, :, 1] += win[1] - 1
    rel[:, :, 2] += win[2] - 1
    rel[:, :, 0] *= (2 * win[1] - 1) * (2 * win[2] - 1)
    rel[:, :, 1] *= 2 * win[2] - 1
    return rel.sum(-1)


def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def window_attention(sd, p, x, mask, nh, win):
    """WindowAttention3D.forward video_swin.py:147-172"""
    B_, N, C = x.shape
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).reshape(B_, N, 3, nh, C // nh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * (C // nh) ** -0.5
    attn = q @ k.transpose(-2, -1)
    idx = relative_position_index(tuple(win))[:N, :N].reshape(-1)
    bias = sd[p + "relative_position_bias_table"][idx].reshape(N, N, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, nh, N, N) + mask.to(attn.dtype).unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, nh, N, N)
    attn = attn.softmax(-1)
    x = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return F.linear(x, sd[p + "proj.weight"], sd[p + "proj.bias"])


def swin_block(sd, p, x, mask_matrix, nh, win_cfg, shift_cfg, dp_scale=None):
    """SwinTransformerBlock3D.forward video_swin.py:206-263 ; dp_scale: None (eval), one (B,) DropPath scale vector used by both
    branches, or a pair ((B,), (B,)) -- the reference calls drop_path twice per block with independent draws (:256 attention, :248 MLP)."""
    dp_a, dp_m = (dp_scale if isinstance(dp_scale, (tuple, list)) else (dp_scale, dp_scale))
    B, D, H, W, C = x.shape
    ws, ss = get_window_size((D, H, W), win_cfg, shift_cfg)
    shortcut = x
    x = layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    pd = (ws[0] - D % ws[0]) % ws[0]
    pb = (ws[1] - H % ws[1]) % ws[1]
    pr = (ws[2] - W % ws[2]) % ws[2]
    x = F.pad(x, (0, 0, 0, pr, 0, pb, 0, pd))
    _, Dp, Hp, Wp, _ = x.shape
    if any(i > 0 for i in ss):
        sx = torch.roll(x, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
        am = mask_matrix
    else:
        sx, am = x, None
    xw = window_partition(sx, ws)
    aw = window_attention(sd, p + "attn.", xw, am, nh, win_cfg)
    sx = window_reverse(aw.view(-1, *(ws + (C,))), ws, B, Dp, Hp, Wp)
    if any(i > 0 for i in ss):
        x = torch.roll(sx, shifts=(ss[0], ss[1], ss[2]), dims=(1, 2, 3))
    else:
        x = sx
    x = x[:, :D, :H, :W, :]
    if dp_a is not None:
        x = x * dp_a.view(B, 1, 1, 1, 1)
    x = shortcut + x
    y = layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    y = F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    y = F.gelu(y)
    y = F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    if dp_m is not None:
        y = y * dp_m.view(B, 1, 1, 1, 1)
    return x + y


def patch_merging(sd, p, x):
    """PatchMerging.forward video_swin.py:273-289"""
    B, D, H, W, C = x.shape
    if H % 2 == 1 or W % 2 == 1:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x0 = x[:, :, 0::2, 0::2, :]
    x1 = x[:, :, 1::2, 0::2, :]
    x2 = x[:, :, 0::2, 1::2, :]
    x3 = x[:, :, 1::2, 1::2, :]
    x = torch.cat([x0, x1, x2, x3], -1)
    x = layer_norm(x, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
    return F.linear(x, sd[p + "reduction.weight"])


def patch_embed(sd, p, x):
    """PatchEmbed3D.forward video_swin.py:390-407 ; x (B,3,D,H,W) -> channels-last (B,D,H/4,W/4,E)"""
    _, _, D, H, W = x.shape
    if W % PATCH[2] != 0:
        x = F.pad(x, (0, PATCH[2] - W % PATCH[2]))
    if H % PATCH[1] != 0:
        x = F.pad(x, (0, 0, 0, PATCH[1] - H % PATCH[1]))
    x = F.pad(x, (0, 0, 0, 0, 0, 1))                     # one zero frame at the END of D (:398)
    x = F.conv3d(x, sd[p + "proj.weight"], sd[p + "proj.bias"], stride=(1, 4, 4))
    x = x.permute(0, 2, 3, 4, 1)
    return layer_norm(x, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)


def swin_forward(sd, cfg, x, dp_scales=None, prefix="enc_img.swin.", final_norm=True):
    """SwinTransformer3D.forward video_swin.py:470-482 + BasicLayer.forward :352-370.
    x (B,3,T,H,W) -> channels-last (B,T,H/32,W/32,8E).  dp_scales: list per block of (B,) or None."""
    win = tuple(cfg["window"])
    shift = tuple(i // 2 for i in win)
    x = patch_embed(sd, prefix + "patch_embed.", x)
    blk = 0
    for i, (d, nh) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
        B, D, H, W, C = x.shape
        ws, ss = get_window_size((D, H, W), win, shift)
        Dp = int(np.ceil(D / ws[0])) * ws[0]
        Hp = int(np.ceil(H / ws[1])) * ws[1]
        Wp = int(np.ceil(W / ws[2])) * ws[2]
        am = compute_mask(Dp, Hp, Wp, ws, ss, x.dtype)
        for b in range(d):
            x = swin_block(sd, prefix + f"layers.{i}.blocks.{b}.", x, am, nh, win,
                           (0, 0, 0) if b % 2 == 0 else shift,
                           None if dp_scales is None else dp_scales[blk])
            blk += 1
        if i < len(cfg["depths"]) - 1:
            x = patch_merging(sd, prefix + f"layers.{i}.downsample.", x)
    if not final_norm:
        return x
    return layer_norm(x, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 1e-5)


# ----------------------------------------------------------------------------
# EncVideo / EncTxt / fusion encoder / heads
# ----------------------------------------------------------------------------
def enc_video(sd, cfg, img, dp_scales=None, odr=None, vt_mask=None):
    """EncVideo.forward model.py:32-78 ; img (B,T,3,H,W) -> feat (B,T*(1+hw),768), mask (ones, times vt_mask (B,T,1+hw) if given).
    odr (B x T frame orders, model.py:61-67): frame slot i of clip b gets emb_len[i] when odr[b][i] == i, else emb_odr."""
    B, T, _, H, W = img.shape
    h, w = H // 32, W // 32
    f = swin_forward(sd, cfg, img.transpose(1, 2), dp_scales)          # (B,T,h,w,8E)
    f = f.reshape(B, T, h * w, -1)
    if "enc_img.fc.weight" in sd:
        f = F.linear(f, sd["enc_img.fc.weight"], sd["enc_img.fc.bias"])
    f = torch.cat([sd["enc_img.emb_cls"].expand(B, T, -1, -1), f], dim=2)
    f = f + sd["enc_img.emb_pos"][:, :, :1 + h * w, :]
    if odr is not None:
        rows = [torch.cat([sd["enc_img.emb_len"][:, i:i + 1] if i == int(p_) else sd["enc_img.emb_odr"] for i, p_ in enumerate(odr[b])], dim=1)
                for b in range(B)]
        f = f + torch.cat(rows, dim=0)
    else:
        f = f + sd["enc_img.emb_len"][:, :T, :, :]
    f = layer_norm(f, sd["enc_img.norm.weight"], sd["enc_img.norm.bias"], 1e-5).reshape(B, T * (1 + h * w), -1)
    m = torch.ones(B, T, 1 + h * w, dtype=torch.long)
    if vt_mask is not None:
        m = m * vt_mask
    return f, m.reshape(B, T * (1 + h * w))


def enc_txt(sd, txt):
    """EncTxt.forward model.py:106-115 (embed_only) == HF BertEmbeddings (eval: no dropout)"""
    B, X = txt.shape
    p = "enc_txt.emb_txt."
    e = sd[p + "word_embeddings.weight"][txt] + sd[p + "token_type_embeddings.weight"][0] \
        + sd[p + "position_embeddings.weight"][:X].unsqueeze(0)
    return layer_norm(e, sd[p + "LayerNorm.weight"], sd[p + "LayerNorm.bias"], BERT["eps"])


def bert_layer(sd, p, x, add_mask, probs_out=None, drop=None):
    """HF BertLayer (post-LN); call site model.py:213.  probs_out: list receiving the attention probabilities
    (B, heads, L, L) -- HF's `attentions` (output_attentions=True, model.py:213).
    drop=None: eval mode.  drop=dict(attn (B,heads,L,L), h1 (B,L,H), h2 (B,L,H)): train mode with EXPLICIT dropout multipliers
    (0 or 1/keep) at HF's three sites -- BertSelfAttention.dropout on the probabilities, BertSelfOutput.dropout and
    BertOutput.dropout on the dense outputs before the residual add."""
    B, L, H = x.shape
    nh, hd = BERT["heads"], H // BERT["heads"]
    q = F.linear(x, sd[p + "attention.self.query.weight"], sd[p + "attention.self.query.bias"])
    k = F.linear(x, sd[p + "attention.self.key.weight"], sd[p + "attention.self.key.bias"])
    v = F.linear(x, sd[p + "attention.self.value.weight"], sd[p + "attention.self.value.bias"])
    q, k, v = (t.view(B, L, nh, hd).transpose(1, 2) for t in (q, k, v))
    s = q @ k.transpose(-1, -2) / math.sqrt(hd) + add_mask
    pr = s.softmax(-1)
    if probs_out is not None:
        probs_out.append(pr)
    if drop is not None:
        pr = pr * drop["attn"]
    a = (pr @ v).transpose(1, 2).reshape(B, L, H)
    a = F.linear(a, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
    if drop is not None:
        a = a * drop["h1"]
    x = layer_norm(a + x, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], BERT["eps"])
    y = F.gelu(F.linear(x, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
    y = F.linear(y, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
    if drop is not None:
        y = y * drop["h2"]
    return layer_norm(y + x, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], BERT["eps"])


def go_cross(sd, cfg, feat_img, mask_img, feat_txt, mask_txt, seq2seq=False):
    """VIOLET_Base.go_cross model.py:204-214 ; mask -> (1-m)*finfo.min, broadcast (B,1,1,L).
    seq2seq=True: get_smtm_output main_pretrain.py:217-224 with get_attn_mask(attn_mask_type="seq2seq") model.py:191-199 --
    (B,L,L): every query sees the visual keys (mask_img), text queries see text keys lower-triangularly (mask_txt unused)."""
    feat = torch.cat([feat_img, feat_txt], dim=1)
    if seq2seq:
        B, Lv = mask_img.shape
        Lt = mask_txt.shape[1]
        m = torch.zeros(B, Lv + Lt, Lv + Lt, dtype=feat.dtype)
        m[:, :, :Lv] = mask_img[:, None, :].to(feat.dtype)
        m[:, Lv:, Lv:] = torch.tril(torch.ones(Lt, Lt, dtype=feat.dtype))
        add = (1.0 - m[:, None, :, :]) * torch.finfo(feat.dtype).min
        for l in range(cfg["bert_layers"]):
            feat = bert_layer(sd, f"trsfr.layer.{l}.", feat, add)
        return feat
    mask = torch.cat([mask_img, mask_txt], dim=1)
    add = (1.0 - mask[:, None, None, :].to(feat.dtype)) * torch.finfo(feat.dtype).min
    for l in range(cfg["bert_layers"]):
        feat = bert_layer(sd, f"trsfr.layer.{l}.", feat, add)
    return feat


def get_att(sd, cfg, img, txt, mask):
    """VIOLET_Pretrain.get_att main_pretrain.py:211-215 (eval mode): layer-summed, head-averaged attention column sums (B, L) --
    the sampling weights of the attention-guided 'am' masking (:320-343)."""
    feat_img, mask_img = enc_video(sd, cfg, img)
    feat_txt = enc_txt(sd, txt)
    feat = torch.cat([feat_img, feat_txt], dim=1)
    m = torch.cat([mask_img, mask], dim=1)
    add = (1.0 - m[:, None, None, :].to(feat.dtype)) * torch.finfo(feat.dtype).min
    probs = []
    for l in range(cfg["bert_layers"]):
        feat = bert_layer(sd, f"trsfr.layer.{l}.", feat, add, probs)
    att = torch.cat([a.mean(dim=1, keepdim=True) for a in probs], dim=1).sum(dim=(1, 2))
    return feat_img, att


def am_masking(cfg, img, txt, mask, att, p_mask=0.15, generator=None):
    """The 'am' branch of Agent_Pretrain.masking (main_pretrain.py:320-343,354-364) for every sample, given the attention weights:
    zero the special positions, draw int(L * p) positions without replacement (torch.multinomial), split them into patch and
    token positions.  Returns the batch dict of apply_masking plus `failed` (no text position drawn -> the reference falls back
    to 'rm' for that sample and, by its sticky flag, every later one)."""
    B, T, _, H, W = img.shape
    h, w = H // cfg["size_patch"], W // cfg["size_patch"]
    X = txt.shape[1]
    Lv = (1 + h * w) * T
    spc_txt = (txt == SPECIAL["cls"]) | (txt == SPECIAL["sep"]) | (txt == SPECIAL["pad"]) | (txt == SPECIAL["mask"])
    spc_v = torch.tensor(sum([[True] + [False] * (h * w) for _ in range(T)], []))
    sel = torch.zeros(B, X, dtype=torch.bool)
    cov = torch.zeros(B, T, h, w)
    failed = []
    for i in range(B):
        a = att[i].clone().float()
        a[torch.cat([spc_v, spc_txt[i]])] = 0.0
        pos = torch.multinomial(a, int((Lv + X) * p_mask), generator=generator).numpy()
        n_txt = 0
        for p_ in pos:
            if p_ < Lv:
                i_t, q = p_ // (1 + h * w), p_ % (1 + h * w) - 1
                cov[i, i_t, q // w, q % w] = 1.0
            else:
                sel[i, p_ - Lv] = True
                n_txt += 1
        failed.append(n_txt == 0)
    out = apply_masking(img, txt, mask, sel, cov, cfg["size_patch"])
    out["cov"], out["failed"] = cov, failed
    return out


def mlm_head(sd, x):
    """HF BertOnlyMLMHead ; call site main_pretrain.py:236"""
    p = "fc_mtm.predictions."
    x = F.gelu(F.linear(x, sd[p + "transform.dense.weight"], sd[p + "transform.dense.bias"]))
    x = layer_norm(x, sd[p + "transform.LayerNorm.weight"], sd[p + "transform.LayerNorm.bias"], BERT["eps"])
    return F.linear(x, sd[p + "decoder.weight"], sd[p + "bias"])


def vtm_head(sd, x, temp):
    """main_pretrain.py:146-147,260  Dropout(eval)->Linear->ReLU->Linear, /temp"""
    x = F.relu(F.linear(x, sd["fc.1.weight"], sd["fc.1.bias"]))
    return F.linear(x, sd["fc.3.weight"], sd["fc.3.bias"]) / temp


def vtm_negatives_default(B):
    """Deterministic stand-in for np.random.permutation (main_pretrain.py:250): negatives (B, O-1)."""
    O = min(B, 4)
    neg = np.zeros((B, max(O - 1, 0)), dtype=np.int64)
    for i in range(B):
        others = [j for j in range(B) if j != i]
        for k in range(O - 1):
            neg[i, k] = others[(i + k) % len(others)]
    return neg


def pretrain_forward(sd, cfg, img, txt, mask, negatives=None, dp_scales=None, drop=None):
    """VIOLET_Pretrain.forward main_pretrain.py:226-267 (eval mode / explicit negatives).
    drop (train mode with EXPLICIT dropout multipliers, 0 or 1/keep): dict(emb (B,X,H) -- BertEmbeddings.dropout on the text features
    (HF BertEmbeddings.forward, call site model.py:107), vtm (B*O,H) -- the VTM head's Dropout(0.1) on the text-[CLS] states
    (main_pretrain.py:146,260))."""
    B, T, _, H, W = img.shape
    h, w = H // cfg["size_patch"], W // cfg["size_patch"]
    O = min(B, 4)
    Lv = (1 + h * w) * T
    feat_img, mask_img = enc_video(sd, cfg, img, dp_scales)
    feat_txt = enc_txt(sd, txt)
    if drop is not None and drop.get("emb") is not None:
        feat_txt = feat_txt * drop["emb"]
    out = go_cross(sd, cfg, feat_img, mask_img, feat_txt, mask)
    out_mtm = mlm_head(sd, out[:, Lv:])
    out_mvm = out[:, :Lv]
    out_smtm = None
    if "smtm" in cfg.get("pretrain_tasks", ()):              # main_pretrain.py:238-240
        out_smtm = mlm_head(sd, go_cross(sd, cfg, feat_img, mask_img, feat_txt, mask, seq2seq=True)[:, Lv:])
    if negatives is None:
        negatives = vtm_negatives_default(B)
    ii, jj = [], []
    for i in range(B):
        ii.append(i); jj.append(i)
        for k in range(O - 1):
            ii.append(i); jj.append(int(negatives[i][k]))
    out2 = go_cross(sd, cfg, feat_img[ii], mask_img[ii], feat_txt[jj], mask[jj])
    cls2 = out2[:, Lv, :]
    if drop is not None and drop.get("vtm") is not None:
        cls2 = cls2 * drop["vtm"]
    out_vtm = vtm_head(sd, cls2, cfg["temp"]).reshape(B, O)
    return dict(out_vtm=out_vtm, out_mvm=out_mvm, out_mtm=out_mtm, out_smtm=out_smtm, feat_img=feat_img, feat_txt=feat_txt,
                vtm_cls=out2[:, Lv, :], ans_vtm=torch.zeros(B, dtype=torch.long))


def pixel_loss(sd, cfg, out_mvm, unmask_img, mvm_mask):
    """calc_mvm_loss pixel branch main_pretrain.py:420-432"""
    B, T, Cin, H, W = unmask_img.shape
    ps = cfg["size_patch"]
    h, w = H // ps, W // ps
    _, L, C = out_mvm.shape
    l = L // T
    x = torch.cat([out_mvm[:, l * t + 1:l * (t + 1), :] for t in range(T)], dim=1)     # drop per-frame cls
    x = x.permute(0, 2, 1).reshape(B, C, T, h, w).permute(0, 2, 1, 3, 4).reshape(B * T, C, h, w)
    x = F.conv2d(x, sd["decoder_pixel.0.weight"], sd["decoder_pixel.0.bias"])
    x = F.pixel_shuffle(x, ps).view(B, T, Cin, H, W)
    ls = (x - unmask_img).abs()
    return (ls.float() * mvm_mask.float()).sum() / (mvm_mask.float().sum() + 1e-5) / Cin


# ----------------------------------------------------------------------------
# MVM 'vq' target: frozen dVAE tokenizer + vq head
# ----------------------------------------------------------------------------
IMNET_MEAN = (0.485, 0.456, 0.406)
IMNET_STD = (0.229, 0.224, 0.225)


def dvae_encoder(sd, cfg, x):
    """Encoder.forward visbackbone/dalle/encoder.py:41-93 (fp32): 7x7 conv, 4 groups x 2 bottleneck blocks
    (id_path + post_gain * [relu,3x3,relu,3x3,relu,3x3,relu,1x1], :12-39), max-pool 2 after groups 1-3, relu + 1x1 conv."""
    pre = "dalle.encoder.blocks."
    conv = lambda t, k: F.conv2d(t, sd[k + ".w"], sd[k + ".b"], padding=(sd[k + ".w"].shape[-1] - 1) // 2)
    post_gain = 1.0 / (4 * 2) ** 2
    x = conv(x, pre + "input")
    for gi in range(4):
        for bi in range(2):
            q = pre + f"group_{gi + 1}.block_{bi + 1}."
            idp = conv(x, q + "id_path") if (q + "id_path.w") in sd else x
            r = conv(F.relu(x), q + "res_path.conv_1")
            r = conv(F.relu(r), q + "res_path.conv_2")
            r = conv(F.relu(r), q + "res_path.conv_3")
            r = conv(F.relu(r), q + "res_path.conv_4")
            x = idp + post_gain * r
        if gi < 3:
            x = F.max_pool2d(x, 2)
    return conv(F.relu(x), pre + "output.conv")


def vq_tokens(sd, cfg, unmask_img):
    """DalleModel.extract_vq_token dalle/__init__.py:38-54 : un-normalise (ImageNet), map_pixels 0.8x+0.1 (utils.py:46-52),
    encoder, argmax over the vocabulary -> (B*T, H/8, W/8) int64"""
    B, T, C, H, W = unmask_img.shape
    x = unmask_img.reshape(B * T, C, H, W).float()
    mean = torch.tensor(IMNET_MEAN).view(1, 3, 1, 1); std = torch.tensor(IMNET_STD).view(1, 3, 1, 1)
    x = x * std + mean                                         # Normalize(-mean/std, 1/std)
    x = 0.8 * x + 0.1
    return torch.argmax(dvae_encoder(sd, cfg, x), dim=1)


def vq_answers(tokens, mvm_mask):
    """main_pretrain.py:485-488 : ans = where(max_pool2d(mask, 8).sum(channels) == 0, -1, token)"""
    B, T, C, H, W = mvm_mask.shape
    mk = F.max_pool2d(mvm_mask.reshape(B * T, C, H, W).float(), 8).sum(dim=1)
    ans = torch.where(mk == 0, torch.full_like(tokens, -1), tokens)
    return ans.view(B, T * tokens.shape[-1] * tokens.shape[-2])


def vq_logits(sd, cfg, out_mvm, T, h, w):
    """main_pretrain.py:474-477,490-496 : drop per-frame cls, 1x1 conv H -> 2H, PixelShuffle(4), per-position MLP (eval: no dropout)"""
    B, L, C = out_mvm.shape
    l = L // T
    x = torch.cat([out_mvm[:, l * t + 1:l * (t + 1), :] for t in range(T)], dim=1)
    x = x.permute(0, 2, 1).reshape(B, C, T, h, w).permute(0, 2, 1, 3, 4).reshape(B * T, C, h, w)
    x = F.conv2d(x, sd["decoder_vq.0.weight"], sd["decoder_vq.0.bias"])
    up = cfg["size_patch"] // 8
    x = F.pixel_shuffle(x, up)                                 # (B*T, 2H/up^2, h*up, w*up)
    vs = h * up
    x = x.view(B, T, -1, vs, w * up).permute(0, 1, 3, 4, 2).reshape(B, T * vs * w * up, -1)
    x = F.relu(F.linear(x, sd["fc_mvm.1.weight"], sd["fc_mvm.1.bias"]))
    return F.linear(x, sd["fc_mvm.3.weight"], sd["fc_mvm.3.bias"])


def vq_loss(sd, cfg, out_mvm, unmask_img, mvm_mask, tokens=None):
    """calc_mvm_loss vq branch main_pretrain.py:469-502 ; `tokens` overrides the teacher (used to pin the head alone)"""
    B, T, _, H, W = unmask_img.shape
    h, w = H // cfg["size_patch"], W // cfg["size_patch"]
    if tokens is None:
        tokens = vq_tokens(sd, cfg, unmask_img)
    ans = vq_answers(tokens, mvm_mask)
    lg = vq_logits(sd, cfg, out_mvm, T, h, w)
    return cross_entropy_ignore(lg.flatten(0, 1), ans.flatten()), lg, ans


def make_hog(cfg, B, dtype=torch.float32):
    """Closed-form stand-in for the data loader's HOG maps (dataset.py:197-206 -> batch["hog"], (B,T,H,W), non-negative)."""
    T, S = cfg["T"], cfg["img"]
    i = torch.arange(B * T * S * S, dtype=torch.float64)
    return (0.5 * torch.sin(i * 0.0173).abs() * (1.0 + torch.cos(i * 0.00031))).reshape(B, T, S, S).to(dtype)


def hog_loss(sd, cfg, out_mvm, hog, mvm_mask):
    """calc_mvm_loss 'hog' branch main_pretrain.py:453-468"""
    B, T, H, W = hog.shape
    ps = cfg["size_patch"]
    h, w = H // ps, W // ps
    _, L, C = out_mvm.shape
    l = L // T
    x = torch.cat([out_mvm[:, l * t + 1:l * (t + 1), :] for t in range(T)], dim=1)
    x = x.permute(0, 2, 1).reshape(B, C, T, h, w).permute(0, 2, 1, 3, 4).reshape(B * T, C, h, w)
    x = F.pixel_shuffle(F.conv2d(x, sd["decoder_hog.0.weight"], sd["decoder_hog.0.bias"]), ps).view(B, T, H, W)
    ls = (x - hog).abs()
    m = (mvm_mask.sum(dim=2) > 0)
    return (ls.float() * m.float()).sum() / (m.float().sum() + 1e-5)


# ----------------------------------------------------------------------------
# MVM feature targets: frozen Swin teachers + fc_mvm head + masked L1
# ----------------------------------------------------------------------------
def hf_swin2d_to_3d_keys(sd, arch, prefix="feature_model."):
    """HF `transformers.SwinModel` tensors (third-party dependency, README 'Transformers 4.26'; SwinModel.forward with
    output_hidden_states) renamed onto the SwinTransformer3D layout so swin_forward() can run them with window (1, ws, ws) on
    single-frame clips: a 2-D Swin block IS the 3-D block with D = 1 (same partition / shift / -100 mask / bias index order
    (dh + ws-1)(2ws-1) + (dw + ws-1); q k^T / sqrt(hd) == (q * hd^-0.5) k^T; PatchMerging concat order identical).  The 4x4
    patch conv becomes the (2,4,4) conv whose second temporal slice is zero (the 3-D embed pads one zero frame, :398).
    Pinned by tests/golden/feature2d.npz, produced by SwinModel itself."""
    o = {}
    w = sd[prefix + "embeddings.patch_embeddings.projection.weight"]
    o["t.patch_embed.proj.weight"] = torch.stack([w, torch.zeros_like(w)], dim=2)
    o["t.patch_embed.proj.bias"] = sd[prefix + "embeddings.patch_embeddings.projection.bias"]
    o["t.patch_embed.norm.weight"] = sd[prefix + "embeddings.norm.weight"]
    o["t.patch_embed.norm.bias"] = sd[prefix + "embeddings.norm.bias"]
    for i, d in enumerate(arch["depths"]):
        for b in range(d):
            p, q = prefix + f"encoder.layers.{i}.blocks.{b}.", f"t.layers.{i}.blocks.{b}."
            for wb in ("weight", "bias"):
                o[q + "norm1." + wb] = sd[p + "layernorm_before." + wb]
                o[q + "norm2." + wb] = sd[p + "layernorm_after." + wb]
                o[q + "attn.qkv." + wb] = torch.cat([sd[p + f"attention.self.{n}." + wb] for n in ("query", "key", "value")], 0)
                o[q + "attn.proj." + wb] = sd[p + "attention.output.dense." + wb]
                o[q + "mlp.fc1." + wb] = sd[p + "intermediate.dense." + wb]
                o[q + "mlp.fc2." + wb] = sd[p + "output.dense." + wb]
            o[q + "attn.relative_position_bias_table"] = sd[p + "attention.self.relative_position_bias_table"]
        if i < len(arch["depths"]) - 1:
            p, q = prefix + f"encoder.layers.{i}.downsample.", f"t.layers.{i}.downsample."
            o[q + "reduction.weight"] = sd[p + "reduction.weight"]
            o[q + "norm.weight"], o[q + "norm.bias"] = sd[p + "norm.weight"], sd[p + "norm.bias"]
    o["t.norm.weight"], o["t.norm.bias"] = sd[prefix + "layernorm.weight"], sd[prefix + "layernorm.bias"]
    return o


def teacher_features(sd, cfg, unmask_img):
    """no-grad targets (B, T, h*w, F).  '3d_feature' main_pretrain.py:516-518: feature_model(img.transpose(1,2)) (final norm
    applied, video_swin.py:480) ; '2d_feature' :535-537: SwinModel(img.flatten(0,1)).hidden_states[-1] = last stage output
    BEFORE SwinModel.layernorm, tokens (B*T, h*w, F)."""
    B, T, _, H, W = unmask_img.shape
    ta = cfg["teacher_arch"]
    with torch.no_grad():
        if "3d_feature" in cfg["mvm_target"]:
            tc = dict(ta)
            f = swin_forward(sd, tc, unmask_img.transpose(1, 2), None, prefix="feature_model.")       # (B, T, h, w, F)
            return f.reshape(B, T, -1, f.shape[-1])
        ws = ta["window"][-1]
        tsd = hf_swin2d_to_3d_keys(sd, ta)
        tc = dict(ta); tc["window"] = (1, ws, ws)
        f = swin_forward(tsd, tc, unmask_img.flatten(0, 1).unsqueeze(2), None, prefix="t.", final_norm=False)   # (B*T, 1, h, w, F)
        return f.reshape(B, T, -1, f.shape[-1])


def feature_loss(sd, cfg, out_mvm, unmask_img, mvm_mask, target=None):
    """calc_mvm_loss '3d_feature' / '2d_feature' branches main_pretrain.py:508-545 (eval mode: fc_mvm's Dropout is identity)"""
    B, T, Cin, H, W = unmask_img.shape
    ps = cfg["size_patch"]
    h, w = H // ps, W // ps
    _, L, C = out_mvm.shape
    l = L // T
    x = torch.cat([out_mvm[:, l * t + 1:l * (t + 1), :] for t in range(T)], dim=1)
    x = F.relu(F.linear(x, sd["fc_mvm.1.weight"], sd["fc_mvm.1.bias"]))
    pred = F.linear(x, sd["fc_mvm.3.weight"], sd["fc_mvm.3.bias"]).reshape(B, T, h * w, -1)
    if target is None:
        target = teacher_features(sd, cfg, unmask_img)
    m = F.max_pool2d(mvm_mask.reshape(B * T, Cin, H, W).float(), ps).sum(dim=1) / 3.0
    m = m.view(B, T, h * w, 1)
    ls = (pred - target).abs()
    return (ls.float() * m).sum() / (m.sum() + 1e-5) / Cin, pred, target


# ----------------------------------------------------------------------------
# downstream: text-to-video retrieval
# ----------------------------------------------------------------------------
def retrieval_forward(sd, cfg, img, txt, mask):
    """VIOLET_Retrieval.forward main_retrieval.py:63-85 (eval mode): out[i][j] = fc(fusion([img_i ; txt_j])[text CLS])"""
    B, T, _, H, W = img.shape
    h, w = H // 32, W // 32
    feat_img, mask_img = enc_video(sd, cfg, img)
    feat_txt = enc_txt(sd, txt)
    ii = [i for i in range(B) for _ in range(B)]
    jj = [j for _ in range(B) for j in range(B)]
    out = go_cross(sd, cfg, feat_img[ii], mask_img[ii], feat_txt[jj], mask[jj])
    x = out[:, (1 + h * w) * T, :]
    x = F.relu(F.linear(x, sd["fc.1.weight"], sd["fc.1.bias"]))
    return F.linear(x, sd["fc.3.weight"], sd["fc.3.bias"]).squeeze(-1).view(B, B)


def qaoe_forward(sd, cfg, img, txt, mask):
    """VIOLET_QAOE.forward main_qaoe.py:49-58 (eval mode): logits (B, size_vocab) from the text [CLS] state of one fusion pass"""
    B, T, _, H, W = img.shape
    h, w = H // 32, W // 32
    feat_img, mask_img = enc_video(sd, cfg, img)
    feat_txt = enc_txt(sd, txt)
    out = go_cross(sd, cfg, feat_img, mask_img, feat_txt, mask)
    x = F.relu(F.linear(out[:, (1 + h * w) * T, :], sd["fc.1.weight"], sd["fc.1.bias"]))
    return F.linear(x, sd["fc.3.weight"], sd["fc.3.bias"])


def qamc_mlm_forward(sd, cfg, img, txt, mask):
    """VIOLET_QAMC_MLM_Head.forward main_qamc_tsv_mlm_head.py:76-94 (eval mode, no task token / prompt: `prepro_txt_inputs` is the
    identity then, model.py:252-258): txt / mask (B, O, X) = one tokenised "question + option + [MASK]" per option; the video
    tokens are shared by a clip's O sequences; MLM-head logits (B*O, X, vocab) of the text positions."""
    B, T, _, H, W = img.shape
    O, X = txt.shape[1], txt.shape[2]
    h, w = H // 32, W // 32
    feat_img, mask_img = enc_video(sd, cfg, img)
    feat_txt = enc_txt(sd, txt.reshape(B * O, X))
    ii = [i for i in range(B) for _ in range(O)]
    out = go_cross(sd, cfg, feat_img[ii], mask_img[ii], feat_txt, mask.reshape(B * O, X))
    return mlm_head(sd, out[:, (1 + h * w) * T:])


def qamc_mlm_loss(logits, mask_ans):
    """Agent_QAMC_MLM_Head.step, train branch (:104-109): CrossEntropyLoss(ignore_index=-1) over every text position"""
    return cross_entropy_ignore(logits.reshape(-1, logits.shape[-1]), mask_ans.reshape(-1))


def qamc_mlm_predict(logits, mask_ans, true_id, false_id):
    """eval branch (:111-123): at each option's [MASK] position p_true / (p_true + p_false) of the RAW logits, arg-max over the options;
    returns (predicted option (B,), answer option (B,))"""
    B, O, L = mask_ans.shape
    pt, pf = logits[:, :, true_id], logits[:, :, false_id]
    sc = pt / (pt + pf)
    m = mask_ans.reshape(B * O, L)
    sc = sc[m != -1].view(B, O)
    am = m[m != -1].view(B, O)
    return torch.argmax(sc, dim=-1), (am == true_id).nonzero()[:, 1]


def mlm_qa_forward(sd, cfg, img, txt, mask):
    """VIOLET_QAMC_MLM_Head_GEN.forward (main_qamc_tsv_mlm_gen_ans_idx.py:87-101) == VIOLET_QAOE_LSMDC.forward (main_qaoe_lsmdc_fib.py:71-84):
    ONE (video, question + [MASK]) sequence per clip, txt / mask (B, X); MLM-head logits (B, X, vocab) of the text positions."""
    return qamc_mlm_forward(sd, cfg, img, txt[:, None], mask[:, None])


def qamc_gen_predict(logits, mask_ans, ans_tok_ids):
    """Agent_QAMC_MLM_Head_GEN.step, eval branch (main_qamc_tsv_mlm_gen_ans_idx.py:116-125): the RAW logits of the candidate answer tokens at
    each clip's [MASK] position, divided by their sum, arg-max -> (scores (B, n_ans), predicted candidate (B,))"""
    B = mask_ans.shape[0]
    p = logits[:, :, ans_tok_ids][mask_ans != -1]
    p = (p / p.sum(dim=-1).view(B, 1)).view(B, -1)
    return p, torch.argmax(p, dim=-1)


def top_k_acc(out, ans, k=5):
    """Agent_QAOE_LSMDC.get_top_k_acc (main_qaoe_lsmdc_fib.py:100-112): per labelled position 1.0 when the label is among the k largest
    logits; the list is padded with 0.0 up to the batch size (clips without a label count as wrong)."""
    B = out.shape[0]
    ac = []
    if bool((ans != -1).any()):
        lab = ans[ans != -1].view(-1, 1)
        top = torch.topk(out[ans != -1].view(lab.shape[0], -1), k=k, dim=-1).indices
        ac = (top == lab).any(dim=-1).float().tolist()
    return ac + [0.0] * (B - len(ac))


def norm_softmax_loss(x, temperature):
    """NormSoftmaxLoss agent.py:34-50"""
    i_logsm = F.log_softmax(x / temperature, dim=1)
    j_logsm = F.log_softmax(x.t() / temperature, dim=1)
    return -torch.diag(i_logsm).mean() - torch.diag(j_logsm).mean()


def cross_entropy_ignore(logits, target):
    """T.nn.CrossEntropyLoss(ignore_index=-1) agent.py:57 (mean over non-ignored; NaN if none)"""
    return F.cross_entropy(logits, target, ignore_index=-1)


def pretrain_losses(sd, cfg, batch, negatives=None, dp_scales=None, drop=None):
    """Agent_Pretrain.step main_pretrain.py:555-567 : ls = mtm + vtm + mvm"""
    out = pretrain_forward(sd, cfg, batch["img"], batch["txt"], batch["mask"], negatives, dp_scales, drop)
    ls_mtm = cross_entropy_ignore(out["out_mtm"].flatten(0, 1), batch["ans_mtm"].flatten())
    ls_vtm = cross_entropy_ignore(out["out_vtm"], out["ans_vtm"])
    ls_mvm = 0.0
    if "pixel" in cfg["mvm_target"]:
        ls_mvm = ls_mvm + pixel_loss(sd, cfg, out["out_mvm"], batch["unmask_img"], batch["mvm_mask"])
    if "vq" in cfg["mvm_target"]:                             # the step sums the terms of calc_mvm_loss (main_pretrain.py:563-564)
        lv, _, _ = vq_loss(sd, cfg, out["out_mvm"], batch["unmask_img"], batch["mvm_mask"], batch.get("vq_tokens"))
        ls_mvm = ls_mvm + lv
    if "hog" in cfg["mvm_target"]:
        ls_mvm = ls_mvm + hog_loss(sd, cfg, out["out_mvm"], batch["hog"], batch["mvm_mask"])
    if "3d_feature" in cfg["mvm_target"] or "2d_feature" in cfg["mvm_target"]:
        lf, pred_f, tgt_f = feature_loss(sd, cfg, out["out_mvm"], batch["unmask_img"], batch["mvm_mask"], batch.get("feature_target"))
        ls_mvm = ls_mvm + lf
        out["pred_feature"], out["feature_target"] = pred_f, tgt_f
    total = ls_mtm + ls_vtm + ls_mvm
    res = dict(mtm=ls_mtm, vtm=ls_vtm, mvm=ls_mvm, out=out)
    if out.get("out_smtm") is not None:                       # main_pretrain.py:566-568
        res["smtm"] = cross_entropy_ignore(out["out_smtm"].flatten(0, 1), batch["ans_mtm"].flatten())
        total = total + res["smtm"]
    res["total"] = total
    return res


# ----------------------------------------------------------------------------
# masking with explicit draws (Agent_Pretrain.masking main_pretrain.py:276-372)
# ----------------------------------------------------------------------------
SPECIAL = dict(cls=101, sep=102, pad=0, mask=103)


def apply_masking(img, txt, mask, txt_sel, cov, size_patch=32):
    """Deterministic core of masking(): given the random draws, build the batch dict.

    txt_sel : (B,X) bool  -- rand(_X)<p_mask draw (special tokens are excluded here, :305/:346)
    cov     : (B,T,h,w) {0,1} -- covered patches (from 'bm' cuboids :308-317 or 'rm' Bernoulli :348-352)
    The x32 expansion is hard-coded in the reference (:362)."""
    img = img.clone(); txt = txt.clone()
    B, T, _, H, W = img.shape
    orig = img.clone()
    spc = (txt == SPECIAL["cls"]) | (txt == SPECIAL["sep"]) | (txt == SPECIAL["pad"]) | (txt == SPECIAL["mask"])
    sel = txt_sel & ~spc
    ans_mtm = torch.where(sel, txt, torch.full_like(txt, -1))
    txt = torch.where(sel, torch.full_like(txt, SPECIAL["mask"]), txt)
    c = cov.to(img.dtype)
    full = c[:, :, None, :, None, :, None].expand(-1, -1, 3, -1, 32, -1, 32).reshape(B, T, 3, c.shape[2] * 32, c.shape[3] * 32)
    img = img * (1.0 - full)
    h, w = H // size_patch, W // size_patch
    ans_mvm = torch.full((B, T * (1 + h * w)), -1, dtype=torch.long)
    return dict(img=img, txt=txt, mask=mask, ans_mtm=ans_mtm, ans_mvm=ans_mvm, mvm_mask=full, unmask_img=orig)


def bm_cover(T, h, w, draws):
    """'bm' block masking: draws = list of T tuples (t,hh,ww,t1,h1,w1) (:308-317)."""
    cov = torch.zeros(T, h, w)
    for (t, hh, ww, t1, h1, w1) in draws:
        cov[t1:t1 + t, h1:h1 + hh, w1:w1 + ww] = 1.0
    return cov


def masking_from_uniform(cfg, img, txt, mask, u_type, u_txt, u_rm, u_bm, types=("rm", "bm"), p_mask=0.15):
    """masking() driven by explicit uniform draws (float32 in [0,1)) -- the CPU statement of the device-side masking kernel's
    contract (include/vmvm.h vmvm_masking): mask type = types[floor(u*len)] (random.choice :303), MLM selection u < p (:305),
    'rm' field u[t][1+i] < p with the cls slot dropped (:348-352), 'bm' cuboids with numpy's randint bounds (:308-313) mapped as
    lo + floor(u*(hi-lo)) in float32.  The geometry itself goes through bm_cover / apply_masking (pinned by masking.npz)."""
    B, T, _, H, W = img.shape
    h, w = H // cfg["size_patch"], W // cfg["size_patch"]
    f32 = np.float32

    def ri(u, n):
        return min(int(f32(u) * f32(n)), n - 1)
    u_type, u_txt, u_rm, u_bm = (np.asarray(a, dtype=np.float32) for a in (u_type, u_txt, u_rm, u_bm))
    sel = torch.from_numpy(u_txt.reshape(B, -1) < f32(p_mask))
    cov = torch.zeros(B, T, h, w)
    u_rm = u_rm.reshape(B, T, 1 + h * w)
    u_bm = u_bm.reshape(B, T, 6)
    kinds = []
    for b in range(B):
        kind = types[ri(u_type[b], len(types))]
        kinds.append(kind)
        if kind == "rm":
            cov[b] = torch.from_numpy((u_rm[b, :, 1:] < f32(p_mask)).astype(np.float32).reshape(T, h, w))
        else:
            draws = []
            for k in range(T):
                u = u_bm[b, k]
                t = 1 + ri(u[0], T - 1) if T > 1 else 1
                hh, ww = 1 + ri(u[1], h * 2 // 3 - 1), 1 + ri(u[2], w * 2 // 3 - 1)
                draws.append((t, hh, ww, ri(u[3], T - t + 1), ri(u[4], h - hh + 1), ri(u[5], w - ww + 1)))
            cov[b] = bm_cover(T, h, w, draws)
    out = apply_masking(img, txt, mask, sel, cov, cfg["size_patch"])
    out["cov"] = cov
    out["kinds"] = kinds
    return out


def default_masking(cfg, img, txt, mask, seed=0, p_mask=0.15):
    """Seeded masking with the reference's distributions ('rm' for even samples, 'bm' for odd)."""
    g = np.random.RandomState(seed)
    B, T, _, H, W = img.shape
    h, w = H // cfg["size_patch"], W // cfg["size_patch"]
    X = txt.shape[1]
    sel = torch.from_numpy(g.rand(B, X) < p_mask)
    cov = torch.zeros(B, T, h, w)
    for b in range(B):
        if b % 2 == 0:
            cov[b] = torch.from_numpy((g.rand(T, h, w) < p_mask).astype(np.float32))
        else:
            draws = []
            for _ in range(T):
                t = g.randint(1, T) if T > 1 else 1
                hh, ww = g.randint(1, h * 2 // 3), g.randint(1, w * 2 // 3)
                draws.append((t, hh, ww, g.randint(0, T - t + 1), g.randint(0, h - hh + 1), g.randint(0, w - ww + 1)))
            cov[b] = bm_cover(T, h, w, draws)
    return apply_masking(img, txt, mask, sel, cov, cfg["size_patch"])


# ----------------------------------------------------------------------------
# optimizer / schedule (agent.py:13-32, 84-113, 181-193)
# ----------------------------------------------------------------------------
def param_group_of(name):
    """agent.py:86-95 : (is_swin, no_decay) by SUBSTRING match."""
    no_decay = any(nd in name for nd in ("bias", "LayerNorm.bias", "LayerNorm.weight"))
    return ("swin." in name), no_decay


def lr_factor(step, max_iter, warmup_ratio=0.1):
    """WarmupLinearLR.get_lr_factor agent.py:22-28 (step = scheduler.last_epoch)"""
    warm = int(warmup_ratio * max_iter)
    if step < warm:
        return max(0.0, step / warm)
    step = min(step, max_iter)
    return max(0.0, (max_iter - step) / (max_iter - warm))


def lr_at(step, base_lr, max_iter, min_lr=1e-8):
    return max(min_lr, base_lr * lr_factor(step, max_iter))


def clip_coef(total_norm, max_norm):
    """torch.nn.utils.clip_grad_norm_ (agent.py:188): coef = clamp(max_norm/(norm+1e-6), max=1)"""
    return min(1.0, max_norm / (total_norm + 1e-6))


def adamw_step(p, g, m, v, step, lr, wd, b1=0.9, b2=0.98, eps=1e-8):
    """torch.optim.AdamW single-tensor math (agent.py:111-112); step is 1-based. In-place on p,m,v."""
    p.mul_(1.0 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    return p


def train_step(sd, cfg, batch, opt_state, step, max_iter, negatives=None, lr=5e-5, decay=1e-3,
               lr_mul=1.0, max_grad_norm=1.0):
    """One full optimizer step in the reference's order (agent.py:181-193, fp32, no GradScaler):
    forward -> loss -> backward -> clip -> AdamW at lr(step-1 scheduler state) -> returns losses."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    ls = pretrain_losses(params, cfg, batch, negatives)
    ls["total"].backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items()}
    tot = math.sqrt(sum(float((g.double() ** 2).sum()) for k, g in grads.items() if params[k].grad is not None))
    coef = clip_coef(tot, max_grad_norm) if max_grad_norm > 0 else 1.0
    cur = lr_at(step - 1, lr, max_iter)            # scheduler ctor did step 0
    for k in sd:
        if params[k].grad is None:
            continue
        is_swin, nd = param_group_of(k)
        st = opt_state.setdefault(k, dict(m=torch.zeros_like(sd[k]), v=torch.zeros_like(sd[k])))
        adamw_step(sd[k], grads[k] * coef, st["m"], st["v"], step, cur * (lr_mul if is_swin else 1.0), 0.0 if nd else decay)
    return dict(mtm=float(ls["mtm"]), vtm=float(ls["vtm"]), mvm=float(ls["mvm"]), grad_norm=tot, grads=grads)
