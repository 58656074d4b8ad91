"""Architecture table and argument defaults of the reference surface.

Arch values: visbackbone/swin_tiny.py:2-24, swin_base.py:3-6, swin_large.py:3-6, swin_*_patch244_*.py:4
(only patch_size/embed_dim/depths/num_heads/window_size/patch_norm are read, video_swin.py:621-639).
Argument names/defaults: utils/args.py:24-150 and _args/args_pretrain.json."""
import json

ARCH = {
    "tiny": dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window=(8, 7, 7)),
    "small": dict(embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24), window=(8, 7, 7)),
    "base": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window=(8, 7, 7)),
    "large": dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48), window=(8, 7, 7)),
    "large384": dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48), window=(8, 12, 12)),
}
PATCH = (2, 4, 4)
DROP_PATH_RATE = 0.2                      # hard-coded for every size (video_swin.py:635)
BERT = dict(vocab=30522, hidden=768, layers=12, heads=12, ffn=3072, max_pos=512, types=2, eps=1e-12,
            hidden_dropout=0.1, attn_dropout=0.1)
TOKENS = dict(cls=101, sep=102, pad=0, mask=103, unk=100)      # bert-base-uncased

DEFAULT_ARGS = dict(  # utils/args.py defaults overlaid by _args/args_pretrain.json
    type="pretrain", task="pretrain", temp=0.05, pretrain_tasks=["vtm", "mlm", "mvm"], pretrain_masks=["rm", "bm"],
    mvm_target=["pixel"], size_img=224, size_frame=4, size_txt=32, size_batch=20, size_epoch=10, lr=5e-5, decay=1e-3,
    max_grad_norm=1.0, vis_backbone="vidswin", vis_backbone_size="base", vis_backbone_init="random", vis_backbone_lr_mul=1,
    txt_backbone="bert-base-uncased", txt_backbone_embed_only=True, fusion_encoder="bert-base-uncased",
    temporal_fusion="vidswin", size_patch=32, max_size_frame=6, max_size_patch=14, p_mask=0.15, seed=88,
    path_ckpt="", path_output="_snapshot/pretrain", logging_steps=20, max_iter=1000, deepspeed=False, use_checkpoint=False,
    dataset=["synthetic"],
    size_vq=8192, dvae_hid=256, dvae_vocab=8192, dalle_model_path="",      # MVM 'vq' target (main_pretrain.py:144,194-209)
)


class Args(dict):
    """EasyDict-like (utils/args.py:246): attribute access over a dict, JSON round-trippable (agent.py:131)."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def get_args(config_path=None, **overrides):
    """utils/args.py:14-22,235-246 : defaults <- JSON config <- explicit overrides (CLI wins)."""
    a = Args(DEFAULT_ARGS)
    if config_path:
        a.update(json.load(open(config_path)))
    a.update(overrides)
    if isinstance(a.mvm_target, str):
        a.mvm_target = [a.mvm_target]
    return a


def swin_arch(size, size_img=224, override=None):
    """video_swin.py:573-599 : the Video-Swin architecture table by (size_img, vis_backbone_size)"""
    if int(size_img) == 384 and size == "large":
        size = "large384"                 # video_swin.py:574-580
    if size not in ARCH:
        raise ValueError(f"unknown vis_backbone_size {size}")
    arch = dict(ARCH[size])
    if override:
        arch.update(override)
    return arch, size


def model_cfg(args):
    cfg, size = swin_arch(args.vis_backbone_size, args.size_img, args.get("arch_override"))
    cfg.update(size=size, hidden=BERT["hidden"], vocab=BERT["vocab"], bert_layers=args.get("bert_layers", BERT["layers"]),
               max_size_frame=args.max_size_frame, max_size_patch=args.max_size_patch, size_patch=args.size_patch,
               temp=args.temp, mvm_target=list(args.mvm_target), size_vq=args.get("size_vq", 8192),
               dvae_hid=args.get("dvae_hid", 256), dvae_vocab=args.get("dvae_vocab", 8192))
    # frozen feature teachers of the '3d_feature' / '2d_feature' targets are always the "base" size (main_pretrain.py:157,168)
    cfg["teacher_arch"] = dict(args["teacher_arch_override"]) if args.get("teacher_arch_override") else dict(ARCH["base"])
    cfg["pretrain_tasks"] = tuple(args.get("pretrain_tasks", ("vtm", "mlm", "mvm")))      # "smtm" adds the seq2seq MLM pass
    cfg["task"] = args.get("task", "pretrain")
    cfg["size_vocab"] = args.get("size_vocab", 0)        # open-ended QA answer vocabulary (main_qaoe.py:47)
    cfg["fp8_forward"] = bool(args.get("fp8_forward", False))     # config 5 ("fp8 MFMA path"): e4m3 forward GEMMs of the fusion qkv / FFN-in, opt-in
    return cfg


def param_shapes(cfg):
    """Ordered {state_dict key: shape} of VIOLET_Pretrain (SURVEY.md section 8b.2 key list)."""
    E, depths, heads, win = cfg["embed_dim"], cfg["depths"], cfg["num_heads"], cfg["window"]
    H, V = cfg["hidden"], cfg["vocab"]
    s = {}
    s["enc_txt.emb_txt.word_embeddings.weight"] = (V, H)
    s["enc_txt.emb_txt.position_embeddings.weight"] = (BERT["max_pos"], H)
    s["enc_txt.emb_txt.token_type_embeddings.weight"] = (BERT["types"], H)
    s["enc_txt.emb_txt.LayerNorm.weight"] = (H,)
    s["enc_txt.emb_txt.LayerNorm.bias"] = (H,)
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
    C_out = E * 8
    s["enc_img.emb_cls"] = (1, 1, 1, H)
    s["enc_img.emb_pos"] = (1, 1, 1 + cfg["max_size_patch"] ** 2, H)
    s["enc_img.emb_len"] = (1, cfg["max_size_frame"], 1, H)
    s["enc_img.emb_odr"] = (1, 1, 1, H)
    if C_out != H:                                     # model.py:20-21
        s["enc_img.fc.weight"] = (H, C_out)
        s["enc_img.fc.bias"] = (H,)
    s["enc_img.norm.weight"] = (H,)
    s["enc_img.norm.bias"] = (H,)
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
    s["fc.1.weight"] = (2 * H, H)
    s["fc.1.bias"] = (2 * H,)
    s["fc.3.weight"] = (1, 2 * H)
    s["fc.3.bias"] = (1,)
    if cfg.get("task", "pretrain") == "retrieval":       # VIOLET_Retrieval (main_retrieval.py:56-61): VIOLET_Base + fc only
        return s
    if cfg.get("task", "pretrain") == "qaoe":            # VIOLET_QAOE (main_qaoe.py:42-47): fc ends in Linear(2H, size_vocab)
        s["fc.3.weight"] = (int(cfg["size_vocab"]), 2 * H)
        s["fc.3.bias"] = (int(cfg["size_vocab"]),)
        return s
    if cfg.get("task", "pretrain") == "qamc_mlm":        # VIOLET_QAMC_MLM_Head (main_qamc_tsv_mlm_head.py:61-71): no fc; fc_mtm + emb_task
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
    if "hog" in cfg["mvm_target"]:         # main_pretrain.py:180-183
        s["decoder_hog.0.weight"] = (cfg["size_patch"] ** 2, H, 1, 1)
        s["decoder_hog.0.bias"] = (cfg["size_patch"] ** 2,)
    if "vq" in cfg["mvm_target"]:          # main_pretrain.py:194-209 (on-the-fly tokenizer branch)
        up = cfg["size_patch"] // 8
        c = 2 * H // (up * up)
        s["decoder_vq.0.weight"] = (2 * H, H, 1, 1)
        s["decoder_vq.0.bias"] = (2 * H,)
        s["fc_mvm.1.weight"] = (2 * c, c)
        s["fc_mvm.1.bias"] = (2 * c,)
        s["fc_mvm.3.weight"] = (cfg.get("size_vq", 8192), 2 * c)
        s["fc_mvm.3.bias"] = (cfg.get("size_vq", 8192),)
    if "3d_feature" in cfg["mvm_target"] or "2d_feature" in cfg["mvm_target"]:      # main_pretrain.py:153-174
        feat = cfg["teacher_arch"]["embed_dim"] * 8
        s["fc_mvm.1.weight"] = (2 * H, H)
        s["fc_mvm.1.bias"] = (2 * H,)
        s["fc_mvm.3.weight"] = (feat, 2 * H)
        s["fc_mvm.3.bias"] = (feat,)
    return s


def swin_param_shapes(arch, win, prefix="enc_img.swin."):
    """{key: shape} of one SwinTransformer3D (video_swin.py:410-468) under `prefix` (the frozen feature teachers)."""
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
            for n, shp in (("norm1.weight", (C,)), ("norm1.bias", (C,)), ("attn.relative_position_bias_table", (ntab, nh)),
                           ("attn.qkv.weight", (3 * C, C)), ("attn.qkv.bias", (3 * C,)), ("attn.proj.weight", (C, C)), ("attn.proj.bias", (C,)),
                           ("norm2.weight", (C,)), ("norm2.bias", (C,)), ("mlp.fc1.weight", (4 * C, C)), ("mlp.fc1.bias", (4 * C,)),
                           ("mlp.fc2.weight", (C, 4 * C)), ("mlp.fc2.bias", (C,))):
                s[p + n] = shp
        if i < len(depths) - 1:
            p = prefix + f"layers.{i}.downsample."
            s[p + "reduction.weight"] = (2 * C, 4 * C)
            s[p + "norm.weight"] = (4 * C,)
            s[p + "norm.bias"] = (4 * C,)
    s[prefix + "norm.weight"] = (E * 8,)
    s[prefix + "norm.bias"] = (E * 8,)
    return s


def param_group(name):
    """agent.py:86-106 : 0 decay+swin, 1 decay+other, 2 no-decay+swin, 3 no-decay+other (substring rules)."""
    nd = any(k in name for k in ("bias", "LayerNorm.bias", "LayerNorm.weight"))
    return (2 if nd else 0) + (0 if "swin." in name else 1)


def lr_factor(step, max_iter, warmup_ratio=0.1):
    """WarmupLinearLR.get_lr_factor (agent.py:22-28); step = scheduler.last_epoch."""
    warm = int(warmup_ratio * max_iter)
    if step < warm:
        return max(0.0, step / warm)
    step = min(step, max_iter)
    return max(0.0, (max_iter - step) / (max_iter - warm))
