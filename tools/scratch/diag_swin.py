#!/usr/bin/env python3
"""Diagnostic: per-block comparison of the HIP Swin forward against the CPU oracle (fp32) on the reduced config."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import violet_ref as R
from pytorch_empirical_mvm_amd import config as CFG, kernels as K, swin_index as SI
from pytorch_empirical_mvm_amd.engine import VioletEngine, V

def cmp(name, got, ref):
    got, ref = got.float().cpu().double().flatten(), ref.double().flatten()
    cos = float(got @ ref / (got.norm() * ref.norm()))
    print(f"{name:40s} rel_fro={float((got-ref).norm()/ref.norm()):.4e} max={float((got-ref).abs().max()):.4e} refmax={float(ref.abs().max()):.3e} cos={cos:.6f}", flush=True)

arch = dict(embed_dim=32, depths=(2, 2, 2, 2), num_heads=(1, 2, 4, 8), window=(8, 7, 7))
T, Hh, Ww = (12, 96, 80) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split(","))
if len(sys.argv) > 2:
    arch = dict(CFG.ARCH[sys.argv[2]])
args = CFG.get_args(vis_backbone_size="tiny", max_size_frame=16, arch_override=arch, bert_layers=1)
cfg = CFG.model_cfg(args)
eng = VioletEngine(cfg, "cuda")
sd = {k: R.closed_form(k, s) for k, s in CFG.param_shapes(cfg).items() if k.startswith("enc_img.swin.")}
eng.store.load_state(sd)
# oracle on bf16-rounded weights (isolates activation rounding from weight rounding)
sdr = {k: v.to(torch.bfloat16).float() if v.dim() > 1 and "relative_position" not in k else v for k, v in sd.items()}
n = 1 * 3 * T * Hh * Ww
x = R.make_batch(dict(T=T, img=Hh, n_txt=32, vocab=30522), 1)[0][:, :, :, :, :Ww].transpose(1, 2).contiguous()
img = x.transpose(1, 2).contiguous().cuda()
ocfg = R.make_cfg("tiny", T=T, arch=arch)
win = tuple(arch["window"]); shift = tuple(i // 2 for i in win)
# oracle, block by block
xo = R.patch_embed(sdr, "enc_img.swin.patch_embed.", x)
eng.tape = []
xe = eng._patch_embed(img, None)
cmp("patch_embed", xe.t, xo)
dims = (T, Hh // 4, Ww // 4); C = arch['embed_dim']
for i, (d, nh) in enumerate(zip(arch["depths"], arch["num_heads"])):
    B_, D, H, W, _ = xo.shape
    ws, ss = R.get_window_size((D, H, W), win, shift)
    Dp, Hp, Wp = [int(np.ceil(a / b)) * b for a, b in zip((D, H, W), ws)]
    am = R.compute_mask(Dp, Hp, Wp, ws, ss)
    for b in range(d):
        p = f"enc_img.swin.layers.{i}.blocks.{b}."
        xo = R.swin_block(sdr, p, xo, am, nh, win, (0, 0, 0) if b % 2 == 0 else shift)
        xe = eng._swin_block(xe, 1, dims, C, nh, p, b % 2 == 1, None)
        cmp(f"stage{i} block{b} dims={dims} ws={ws} ss={ss if b%2 else (0,0,0)}", xe.t, xo)
    if i < 3:
        p = f"enc_img.swin.layers.{i}.downsample."
        xo = R.patch_merging(sdr, p, xo)
        xe, dims = eng._patch_merge(xe, 1, dims, C, p)
        C *= 2
        cmp(f"merge{i} -> {dims}", xe.t, xo)
