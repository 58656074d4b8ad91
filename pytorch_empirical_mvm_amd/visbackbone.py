"""Standalone module surface of the visual side: `get_vidswin_model(args)` (visbackbone/video_swin.py:573-650),
`load_checkpoint_3d` (:653-659), the 2-D -> 3-D weight inflation (:484-535) and `EncVideo` (model.py:8-78) over the HIP engine.

These are the inference / feature-extraction faces of the same kernels the pretraining step runs (a frozen arena, no tape);
training the backbone goes through `VIOLET_Pretrain`, whose `enc_img.swin.*` parameters live in the optimizer's arena."""
import torch

from . import config as CFG
from .teacher import SwinTeacher


def load_checkpoint_3d(model_path):
    """video_swin.py:653-659 : upstream Video-Swin `.pth` -> state_dict with the `backbone.` prefix stripped."""
    ckpt = torch.load(model_path, map_location="cpu", weights_only=False)["state_dict"]
    return {k.replace("backbone.", ""): v for k, v in ckpt.items()}


def inflate_2d_state(state_2d, arch, current_tables=None):
    """2-D Swin checkpoint tensors -> the 3-D backbone's (what SwinTransformer3D.inflate_weights does, video_swin.py:484-535):
      * `relative_position_index` / `attn_mask` buffers are dropped (always rebuilt);
      * the patch-embedding kernel (E, 3, ph, pw) is copied to every temporal tap and divided by the tap count, so that a static clip
        gives the 2-D model's activations;
      * each relative-position-bias table ((2s-1)^2, heads) is resized to the 3-D window's spatial extent ((2wh-1)(2ww-1) rows, bicubic
        over the (2s-1) x (2s-1) grid) when the sizes differ, then tiled (2wd-1) times along the rows: the same spatial bias at every
        temporal offset.  A table whose head count differs from the model's is left as it is (the loader then skips it by shape).
    `arch`: dict with `patch` and `window`; `current_tables`: {key: (rows, heads)} of the model, for the head-count check."""
    pt = int(arch["patch"][0]) if "patch" in arch else 2
    wd, wh, ww = arch["window"]
    out = {}
    for k, v in state_2d.items():
        if "relative_position_index" in k or "attn_mask" in k:
            continue
        if k == "patch_embed.proj.weight" and v.dim() == 4:
            v = v.unsqueeze(2).repeat(1, 1, pt, 1, 1) / pt
        elif "relative_position_bias_table" in k:
            rows, heads = v.shape
            want = (2 * wh - 1) * (2 * ww - 1)
            if current_tables is not None and k in current_tables and current_tables[k][1] != heads:
                print(f"Error in loading {k}, passing")
            else:
                if rows != want:
                    side = int(rows ** 0.5)
                    grid = v.permute(1, 0).reshape(1, heads, side, side).float()
                    grid = torch.nn.functional.interpolate(grid, size=(2 * wh - 1, 2 * ww - 1), mode="bicubic")
                    v = grid.reshape(heads, want).permute(1, 0).to(v.dtype)
                v = v.repeat(2 * wd - 1, 1)
        out[k] = v
    return out


def load_checkpoint_2d(model_path, arch, current_tables=None):
    """An image-Swin `.pth` (`{'model': state_dict}`) inflated for the video backbone (video_swin.py:495-496 + inflate_2d_state)."""
    sd = torch.load(model_path, map_location="cpu", weights_only=False)["model"]
    return inflate_2d_state(sd, arch, current_tables)


class _Norm:
    def __init__(self, n):
        self.normalized_shape = (n,)


class VidSwin(torch.nn.Module):
    """`SwinTransformer3D` as the reference's callers see it: `.norm.normalized_shape[0]` (model.py:12-13), upstream key names
    (`patch_embed.*`, `layers.*`, `norm.*`), forward (B,3,T,H,W) -> (B,8E,T,H/32,W/32)."""

    def __init__(self, arch, device="cuda", seed=88):
        super().__init__()
        self.arch = dict(arch)
        self.t = SwinTeacher("3d", arch, device, seed=seed)
        self.t.init_weights(seed)
        self.norm = _Norm(self.t.feat_size)

    def init_weights(self, pretrained=None):
        self.t.init_weights(self.t.eng.seed)

    def state_dict(self, *a, **k):
        P = SwinTeacher.PREFIX
        return {key[len(P):]: v for key, v in self.t.state_dict().items()}

    def load_state_dict(self, sd, strict=False):
        P = SwinTeacher.PREFIX
        own = set(self.state_dict())
        taken = self.t.load_state_dict({P + k: v for k, v in sd.items() if k in own and tuple(v.shape) == tuple(self.t.eng.store.index["enc_img.swin." + k][2])})
        missing = sorted(own - {k[len("enc_img.swin."):] for k in taken})
        unexpected = sorted(k for k in sd if k not in own and not k.endswith("relative_position_index") and not k.endswith("attn_mask"))
        if strict and (missing or unexpected):
            raise RuntimeError(f"missing {missing[:5]} unexpected {unexpected[:5]}")
        return missing, unexpected

    @torch.no_grad()
    def forward(self, x):
        B, C, T, H, W = x.shape
        img = x.to(self.t.eng.device, torch.float32).permute(0, 2, 1, 3, 4).contiguous()          # the engine reads (B,T,3,H,W)
        rows = self.t.features(img)                                                                   # [B*T*h*w, 8E] channels-last
        F = self.t.feat_size
        return rows.view(B, T, H // 32, W // 32, F).permute(0, 4, 1, 2, 3).float()


def get_vidswin_model(args, device="cuda"):
    """video_swin.py:573-650 : architecture by (size_img, vis_backbone_size); `vis_backbone_init` "3d" loads the upstream Video-Swin
    checkpoint, "2d" an image-Swin checkpoint inflated along time (`vis_backbone_pretrained_weight` = the file), "random" keeps the init."""
    arch, _ = CFG.swin_arch(args.vis_backbone_size, int(args.size_img), args.get("arch_override"))
    m = VidSwin(arch, device=device, seed=args.get("seed", 88))
    init = args.get("vis_backbone_init", "random")
    path = args.get("vis_backbone_pretrained_weight")
    if init == "2d" and path:                                     # image-Swin weights inflated along time (video_swin.py:556-560)
        print("Inflate 2D model into 3D model.")
        tables = {k: tuple(v.shape) for k, v in m.state_dict().items() if k.endswith("relative_position_bias_table")}
        missing, unexpected = m.load_state_dict(load_checkpoint_2d(path, dict(m.arch, patch=(2, 4, 4)), tables), strict=False)
        print(f"Missing keys in the inflated swin_transformer: {missing}")
        print(f"Unexpected keys in the inflated swin_transformer: {unexpected}")
    elif init == "3d" and path:
        missing, unexpected = m.load_state_dict(load_checkpoint_3d(path), strict=False)
        print(f"Missing keys in loaded video_swin_transformer: {missing}")
        print(f"Unexpected keys in loaded video_swin_transformer: {unexpected}")
    return m


class EncVideo(torch.nn.Module):
    """model.py:8-78 on the engine of a VIOLET model: forward(img (B,T,3,H,W)) -> (feat (B, T*(1+hw), hidden), mask ones).
    Shares the model's parameters (`enc_img.*`); `odr` (frame-order embedding, :61-67) and `vt_mask` (:75) as the reference takes them."""

    def __init__(self, model):
        super().__init__()
        self._m = [model]                      # not a sub-module: the parameters belong to `model`

    @torch.no_grad()
    def forward(self, img, odr=None, vt_mask=None):
        m = self._m[0]
        feat_img, mask_img, _, _ = m.go_feat(img, torch.zeros(img.shape[0], 1, dtype=torch.long), torch.ones(img.shape[0], 1, dtype=torch.long),
                                             odr=odr, vt_mask=vt_mask)
        return feat_img, mask_img
