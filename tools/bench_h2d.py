#!/usr/bin/env python3
"""PCIe-inclusive step rate: masking on the host + prepare_batch (H2D copy of the fp32 clip) + one step, batches NOT resident."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pytorch_empirical_mvm_amd import config as CFG
from pytorch_empirical_mvm_amd.agent import Agent_Pretrain
from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain
B = 32
args = CFG.get_args(vis_backbone_size="base", size_frame=8, max_size_frame=8, mvm_target=["pixel"], max_iter=10000)
model = VIOLET_Pretrain(args, None, device="cuda:0"); agent = Agent_Pretrain(args, model); agent.sched_step = 500
host = [bench.synth_batch(args, B, "cpu", 88 + i) for i in range(2)]
host = [tuple(t.pin_memory() if t.is_floating_point() else t for t in hb) for hb in host]
def step(i, with_mask=True):
    img, txt, mask = host[i % 2]
    mb = agent.masking(img, txt, mask, None)
    return agent.step(agent.prepare_batch(mb), is_train=True, sync=False)
for i in range(3): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 8
for i in range(n): step(i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
t1 = time.perf_counter()
for i in range(n):
    img, txt, mask = host[i % 2]; mb = agent.masking(img, txt, mask, None)
tm = (time.perf_counter() - t1) / n
t2 = time.perf_counter()
for i in range(n):
    b = agent.prepare_batch(mb)
torch.cuda.synchronize(); th = (time.perf_counter() - t2) / n
print(f"host-fed step (masking + H2D + step, pinned fp32 clip of {img.numel() * 4 / 1e6:.0f} MB): {dt * 1e3:.1f} ms = {B / dt:.1f} clips/s ; masking alone {tm * 1e3:.1f} ms ; H2D alone {th * 1e3:.1f} ms")
