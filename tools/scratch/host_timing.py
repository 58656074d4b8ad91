#!/usr/bin/env python3
"""Does the host run ahead of the GPU across step boundaries?  Host-side enqueue time of each phase of a bench step (no device sync inside the loop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pytorch_empirical_mvm_amd import config as CFG
from pytorch_empirical_mvm_amd.agent import Agent_Pretrain
from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain
dev = "cuda:0"
args = CFG.get_args(vis_backbone_size="base", size_frame=8, max_size_frame=8, size_img=224, size_txt=32, mvm_target=["pixel"], max_iter=10000, seed=88)
model = VIOLET_Pretrain(args, None, device=dev)
agent = Agent_Pretrain(args, model)
agent.sched_step = 500
img, txt, mask = bench.synth_batch(args, 32, dev, 88)
raw = (img.to(dev), txt.to(dev), mask.to(dev))
gen = torch.Generator(device=dev).manual_seed(88)
for _ in range(4):
    agent.step(agent.masking_device(*raw, generator=gen), is_train=True, sync=False)
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
rows = []
for i in range(N):
    a = time.perf_counter()
    mb = agent.masking_device(*raw, generator=gen)
    b = time.perf_counter()
    agent.step(mb, is_train=True, sync=False)
    c = time.perf_counter()
    rows.append((b - a, c - b))
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue of {N} steps: {t_host * 1e3 / N:.1f} ms/step; with the device drained: {t_all * 1e3 / N:.1f} ms/step")
for m, s in rows:
    print(f"  masking_device {m * 1e3:6.2f} ms   step {s * 1e3:7.2f} ms")

# where does the host block?  cumulative host time inside a few functions per step
import pytorch_empirical_mvm_amd.engine as E
eng = model.engine
acc = {}
def wrap(obj, name, key=None):
    f = getattr(obj, name); key = key or name
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[key] = acc.get(key, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, g)
wrap(E, "_h2d"); wrap(eng, "sample_drop_path"); wrap(eng, "swin_forward"); wrap(eng, "go_cross"); wrap(agent, "backward_step"); wrap(eng, "_wgrad_join")
wrap(eng, "forward_backward"); wrap(agent, "prepare_batch")
import pytorch_empirical_mvm_amd.kernels as KK
wrap(KK, "gemm", "K.gemm"); wrap(KK, "layernorm_fwd", "K.ln_fwd"); wrap(KK, "attention_fwd", "K.attn_fwd"); wrap(KK, "expand_batch_map", "K.expand_map")
torch.cuda.synchronize()
for i in range(6):
    acc.clear()
    a = time.perf_counter()
    agent.step(agent.masking_device(*raw, generator=gen), is_train=True, sync=False)
    tot = time.perf_counter() - a
    print(f"step {i}: host {tot * 1e3:6.1f} ms | " + "  ".join(f"{k} {v * 1e3:.1f}" for k, v in sorted(acc.items(), key=lambda x: -x[1])))
