import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import violet_ref as R
from pytorch_empirical_mvm_amd import config as CFG, kernels as K
from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain
def cmp(name, got, ref):
    got, ref = got.float().cpu().double().flatten(), ref.detach().double().flatten()
    cos = float(got @ ref / (got.norm() * ref.norm() + 1e-30))
    print(f"{name:44s} rel_fro={float((got-ref).norm()/(ref.norm()+1e-30)):.4e} max={float((got-ref).abs().max()):.4e} refmax={float(ref.abs().max()):.3e} cos={cos:.6f}", flush=True)
size, T = "tiny", 4
args = CFG.get_args(vis_backbone_size=size, size_frame=T, max_size_frame=6)
model = VIOLET_Pretrain(args, None, device="cuda")
cfg = R.make_cfg(size, T=T)
sd = R.make_state_dict(cfg)
model.load_state_dict(sd)
eng = model.engine
for B in (1, 2):
    img, txt, mask = R.make_batch(cfg, B)
    mb = R.default_masking(cfg, img, txt, mask, seed=3)
    cov = mb["mvm_mask"][:, :, 0, ::32, ::32].to(torch.uint8).cuda().contiguous()
    for use_cov in (False, True):
        eng.tape = []
        out, dims, C8 = eng.swin_forward(img.cuda(), cov if use_cov else None, None)
        ref = R.swin_forward(sd, cfg, (mb["img"] if use_cov else img).transpose(1, 2))
        cmp(f"swin out B={B} cov={use_cov}", out.t, ref)
        for b in range(B):
            cmp(f"   clip {b}", out.t.view(B, -1, C8)[b], ref[b])
