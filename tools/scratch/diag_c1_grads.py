#!/usr/bin/env python3
"""Diagnostic (round 6): per-tensor sampled gradient parity of the C1 step against the REFERENCE's own fixtures (tests/golden/c1.npz,
temp = 0.05), by parameter family -- which tensors the VTM cancellation x 20 makes rounding-noise dominated, and which are clean."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import violet_ref as R
from pytorch_empirical_mvm_amd import config as CFG
from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain

d = np.load(os.path.join(ROOT, "tests", "golden", "c1.npz"))
cfg = R.make_cfg("tiny", T=4)
args = CFG.get_args(vis_backbone_size="tiny", size_frame=4, max_size_frame=6)
model = VIOLET_Pretrain(args, None, device="cuda")
sd = R.make_state_dict(cfg)
model.load_state_dict(sd)
img, txt, mask = R.make_batch(cfg, 2)
mb = R.default_masking(cfg, img, txt, mask, seed=3)
cov = mb["mvm_mask"][:, :, 0, ::32, ::32].to(torch.uint8)
batch = dict(img=img.cuda(), cov=cov.cuda().contiguous(), txt=mb["txt"].cuda(), mask=mask.cuda(), ans_mtm=mb["ans_mtm"].cuda())
eng = model.engine
eng.store.grad.zero_()
eng.forward_backward(batch, negatives=d["neg"], train=False, backward=True)
torch.cuda.synchronize()
rows = []
for name in eng.store.index:
    key = "g." + name
    if key + ".val" not in d.files:
        continue
    g = eng.store.g(name).detach().double().flatten().cpu().numpy()
    ref, idx = d[key + ".val"], d[key + ".idx"]
    got = g[idx]
    scale = max(np.abs(ref).max(), float(d[key + ".asum"]) / g.size, 1e-12)
    cos = float(got @ ref / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30))
    asum_ratio = float(np.abs(g).sum() / (float(d[key + ".asum"]) + 1e-30))
    rows.append((name, cos, float(np.abs(got - ref).max() / scale), asum_ratio))
fam = lambda n: "swin" if n.startswith("enc_img.swin.") else n.split(".")[0]
for f in sorted({fam(r[0]) for r in rows}):
    rr = [r for r in rows if fam(r[0]) == f]
    cs, es, ar = np.array([r[1] for r in rr]), np.array([r[2] for r in rr]), np.array([r[3] for r in rr])
    print(f"{f:14s} n={len(rr):3d} cos8 min {cs.min():.4f} p10 {np.percentile(cs, 10):.4f} med {np.median(cs):.4f} | err/scale max {es.max():.3f} p90 {np.percentile(es, 90):.3f} med {np.median(es):.3f} | asum ratio {ar.min():.3f}..{ar.max():.3f}")
rows.sort(key=lambda r: r[1])
for r in rows[:40]:
    print(f"   cos8={r[1]:.4f} err/scale={r[2]:.3f} asum ratio={r[3]:.3f} {r[0]}")
