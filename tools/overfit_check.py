#!/usr/bin/env python3
"""Sanity run: the full C2-size model (Swin-B, 8 x 224^2, pixel target) trained for N optimizer steps on ONE fixed synthetic batch
(fixed masking, fixed negatives, dropout on): the three losses must fall.  usage: python tools/overfit_check.py [steps] [B]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
from pytorch_empirical_mvm_amd import config as CFG
from pytorch_empirical_mvm_amd.agent import Agent_Pretrain
from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
args = CFG.get_args(vis_backbone_size="base", size_frame=8, max_size_frame=8, mvm_target=["pixel"], max_iter=steps * 4, lr=2e-4, seed=88)
model = VIOLET_Pretrain(args, None, device="cuda:0")
agent = Agent_Pretrain(args, model)
agent.sched_step = steps // 2
img, txt, mask = bench.synth_batch(args, B, "cuda:0", 5)
import random
random.seed(1); np.random.seed(1); torch.manual_seed(1)
mb = agent.prepare_batch(agent.masking(img, txt, mask, None))
neg = model.engine.sample_negatives(B, np.random.RandomState(3))
hist = []
for i in range(steps):
    r = agent.step(mb, is_train=True, negatives=neg, sync=True)
    hist.append((r["mtm"], r["vtm"], r["mvm"]))
    if i % 5 == 0 or i == steps - 1:
        print(f"step {i:3d}  mtm {r['mtm']:.4f}  vtm {r['vtm']:.4f}  mvm {r['mvm']:.4f}", flush=True)
a, b = np.mean(hist[:3], 0), np.mean(hist[-3:], 0)
print("first3", a.round(4).tolist(), "last3", b.round(4).tolist())
assert all(np.isfinite(b)) and b[0] < a[0] and b[2] < a[2], "losses did not fall"
print("overfit check ok")
