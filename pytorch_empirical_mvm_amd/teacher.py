"""Frozen Swin feature teachers of the MVM '3d_feature' / '2d_feature' targets (main_pretrain.py:153-174, 508-545) on the
same HIP kernels as the student backbone.

* '3d_feature': VideoSwin-B (`get_vidswin_model`, video_swin.py:573-650; checkpoint keys `feature_model.patch_embed.*`,
  `feature_model.layers.*`, `feature_model.norm.*`); target = the backbone output AFTER its final LayerNorm (:480).
* '2d_feature': HF `transformers.SwinModel` Swin-B (`get_swin_model`, visbackbone/swin.py:16-35; checkpoint keys of
  Transformers 4.26: `feature_model.embeddings.*`, `feature_model.encoder.layers.*`, `feature_model.layernorm.*`); target =
  `hidden_states[-1]`, the last stage's output BEFORE `layernorm`.  A 2-D Swin block is the 3-D block with a depth-1 window, so
  the frames run as single-frame clips with window (1, 7, 7); the 4x4 patch conv becomes the (2,4,4) conv whose second temporal
  slice is zero (the 3-D embed pads one zero frame).  The oracle states the same mapping and is pinned against SwinModel itself
  (tests/golden/feature2d.npz).

The teacher is an engine instance with a frozen parameter arena (f32 + bf16 copy, no gradients / Adam state); its tape is dropped."""
import torch

from . import config as CFG
from .engine import ParamStore, VioletEngine
from .switches import Switches


class _NoTape(list):
    """the engine's blocks push their backward closures here; a frozen teacher drops them (and with them the activations)"""
    def append(self, fn):
        pass


class SwinTeacher:
    PREFIX = "feature_model."

    def __init__(self, kind, arch, device, seed=88):
        assert kind in ("3d", "2d")
        self.kind = kind
        self.arch = dict(arch)
        ws = tuple(arch["window"])
        self.win = ws if kind == "3d" else (1, ws[-1], ws[-1])
        cfg = dict(self.arch)
        cfg["window"] = self.win
        eng = VioletEngine.__new__(VioletEngine)
        eng.cfg, eng.device = cfg, torch.device(device)
        eng.store = ParamStore(CFG.swin_param_shapes(self.arch, self.win), eng.device, frozen=True)
        eng.seed, eng.rng_offset, eng._idx_cache, eng.tape = int(seed), 0, {}, []
        eng.teacher = eng.feature_teacher = eng.on_swin_tail_ready = None
        eng.dpr = [0.0] * sum(cfg["depths"])
        eng.sw = Switches.from_env()
        self.eng = eng
        self.feat_size = self.arch["embed_dim"] * 8

    # ------------------------------------------------------------------ checkpoint surface
    def external_shapes(self):
        """{checkpoint key: shape} as the reference's state_dict holds the teacher."""
        if self.kind == "3d":
            return {self.PREFIX + k[len("enc_img.swin."):]: v for k, v in CFG.swin_param_shapes(self.arch, self.win).items()}
        return hf_swin2d_param_shapes(self.arch, self.PREFIX, self.win[-1])

    def load_state_dict(self, sd):
        """Takes the tensors under `feature_model.` (missing ones keep their current values)."""
        P, I = self.PREFIX, "enc_img.swin."
        own = {}
        if self.kind == "3d":
            for k in self.eng.store.index:
                src = P + k[len(I):]
                if src in sd:
                    own[k] = sd[src]
        else:
            def take(dst, src, fn=None):
                if src in sd:
                    own[I + dst] = fn(sd[src]) if fn else sd[src]
            take("patch_embed.proj.weight", P + "embeddings.patch_embeddings.projection.weight",
                 lambda w: torch.stack([w, torch.zeros_like(w)], dim=2))
            take("patch_embed.proj.bias", P + "embeddings.patch_embeddings.projection.bias")
            take("patch_embed.norm.weight", P + "embeddings.norm.weight")
            take("patch_embed.norm.bias", P + "embeddings.norm.bias")
            for i, d in enumerate(self.arch["depths"]):
                for b in range(d):
                    p, q = P + f"encoder.layers.{i}.blocks.{b}.", f"layers.{i}.blocks.{b}."
                    for wb in ("weight", "bias"):
                        take(q + "norm1." + wb, p + "layernorm_before." + wb)
                        take(q + "norm2." + wb, p + "layernorm_after." + wb)
                        take(q + "attn.proj." + wb, p + "attention.output.dense." + wb)
                        take(q + "mlp.fc1." + wb, p + "intermediate.dense." + wb)
                        take(q + "mlp.fc2." + wb, p + "output.dense." + wb)
                        qkv = [p + f"attention.self.{n}." + wb for n in ("query", "key", "value")]
                        if all(n in sd for n in qkv):
                            own[I + q + "attn.qkv." + wb] = torch.cat([sd[n] for n in qkv], 0)
                    take(q + "attn.relative_position_bias_table", p + "attention.self.relative_position_bias_table")
                if i < len(self.arch["depths"]) - 1:
                    p, q = P + f"encoder.layers.{i}.downsample.", f"layers.{i}.downsample."
                    take(q + "reduction.weight", p + "reduction.weight")
                    take(q + "norm.weight", p + "norm.weight")
                    take(q + "norm.bias", p + "norm.bias")
            take("norm.weight", P + "layernorm.weight")
            take("norm.bias", P + "layernorm.bias")
        if own:
            self.eng.store.load_state(own)
        return sorted(own)

    @torch.no_grad()
    def init_weights(self, seed=88):
        """video_swin.py:544-551-style init (trunc-normal .02 Linear, LayerNorm 1/0): placeholder until a checkpoint is loaded."""
        gen = torch.Generator().manual_seed(int(seed) + 7)
        sd = {}
        for name, (_, _, shape) in self.eng.store.index.items():
            t = torch.zeros(shape)
            last = name.split(".")[-1]
            if "norm" in name and len(shape) == 1:
                t.fill_(1.0 if last == "weight" else 0.0)
            elif last != "bias":
                t.normal_(0.0, 0.02, generator=gen).clamp_(-2.0, 2.0)
            sd[name] = t
        if self.kind == "2d":
            sd["enc_img.swin.patch_embed.proj.weight"][:, :, 1] = 0.0
        self.eng.store.load_state(sd)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def features(self, img):
        """img f32 (B,T,3,H,W), un-masked -> bf16 [B*T*h*w, F] rows (b, t, patch): exactly the (B,T,hw,F) target layout of
        main_pretrain.py:518 / :537."""
        B, T, _, H, W = img.shape
        eng = self.eng
        eng.tape = _NoTape()
        if self.kind == "3d":
            out, dims, C = eng.swin_forward(img, None, None)
        else:
            out, dims, C = eng.swin_forward(img.reshape(B * T, 1, 3, H, W), None, None, final_norm=False)
        return out.t

    def state_dict(self):
        """{checkpoint key: f32 tensor} under `feature_model.` (Transformers-4.26 names for the 2-D teacher)."""
        S, P, I = self.eng.store, self.PREFIX, "enc_img.swin."
        if self.kind == "3d":
            return {P + k[len(I):]: S.p(k).detach().clone() for k in S.index}
        o = {}
        o[P + "embeddings.patch_embeddings.projection.weight"] = S.p(I + "patch_embed.proj.weight")[:, :, 0].detach().clone()
        o[P + "embeddings.patch_embeddings.projection.bias"] = S.p(I + "patch_embed.proj.bias").detach().clone()
        o[P + "embeddings.norm.weight"] = S.p(I + "patch_embed.norm.weight").detach().clone()
        o[P + "embeddings.norm.bias"] = S.p(I + "patch_embed.norm.bias").detach().clone()
        for i, d in enumerate(self.arch["depths"]):
            for b in range(d):
                p, q = P + f"encoder.layers.{i}.blocks.{b}.", I + f"layers.{i}.blocks.{b}."
                for wb in ("weight", "bias"):
                    o[p + "layernorm_before." + wb] = S.p(q + "norm1." + wb).detach().clone()
                    o[p + "layernorm_after." + wb] = S.p(q + "norm2." + wb).detach().clone()
                    o[p + "attention.output.dense." + wb] = S.p(q + "attn.proj." + wb).detach().clone()
                    o[p + "intermediate.dense." + wb] = S.p(q + "mlp.fc1." + wb).detach().clone()
                    o[p + "output.dense." + wb] = S.p(q + "mlp.fc2." + wb).detach().clone()
                    for n, part in zip(("query", "key", "value"), S.p(q + "attn.qkv." + wb).detach().clone().chunk(3, 0)):
                        o[p + f"attention.self.{n}." + wb] = part
                o[p + "attention.self.relative_position_bias_table"] = S.p(q + "attn.relative_position_bias_table").detach().clone()
            if i < len(self.arch["depths"]) - 1:
                p, q = P + f"encoder.layers.{i}.downsample.", I + f"layers.{i}.downsample."
                o[p + "reduction.weight"] = S.p(q + "reduction.weight").detach().clone()
                o[p + "norm.weight"], o[p + "norm.bias"] = S.p(q + "norm.weight").detach().clone(), S.p(q + "norm.bias").detach().clone()
        o[P + "layernorm.weight"], o[P + "layernorm.bias"] = S.p(I + "norm.weight").detach().clone(), S.p(I + "norm.bias").detach().clone()
        return o


def hf_swin2d_param_shapes(arch, prefix, ws=7):
    """{key: shape} of HF SwinModel (Transformers 4.26 naming) under `prefix`."""
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
            for n in ("layernorm_before", "layernorm_after"):
                s[p + n + ".weight"] = (C,)
                s[p + n + ".bias"] = (C,)
            s[p + "attention.self.relative_position_bias_table"] = ((2 * ws - 1) ** 2, nh)
            for n in ("query", "key", "value"):
                s[p + f"attention.self.{n}.weight"] = (C, C)
                s[p + f"attention.self.{n}.bias"] = (C,)
            s[p + "attention.output.dense.weight"] = (C, C)
            s[p + "attention.output.dense.bias"] = (C,)
            s[p + "intermediate.dense.weight"] = (4 * C, C)
            s[p + "intermediate.dense.bias"] = (4 * C,)
            s[p + "output.dense.weight"] = (C, 4 * C)
            s[p + "output.dense.bias"] = (C,)
        if i < len(depths) - 1:
            p = prefix + f"encoder.layers.{i}.downsample."
            s[p + "reduction.weight"] = (2 * C, 4 * C)
            s[p + "norm.weight"] = (4 * C,)
            s[p + "norm.bias"] = (4 * C,)
    s[prefix + "layernorm.weight"] = (E * 8,)
    s[prefix + "layernorm.bias"] = (E * 8,)
    return s
