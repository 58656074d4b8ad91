"""RCCL smoke at world size 1 (the driver's GPU box has one GPU): torch.distributed backend "nccl" (= RCCL on ROCm) is
initialised, the gradient reducer is forced on (VMVM_FORCE_DIST=1) and three optimizer steps at the C2 shapes run with the
side-stream all-reduces issued next to the persistent GEMMs; the parameters must equal the run without a reducer bit for bit.
Launch: RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=<p> VMVM_FORCE_DIST=1 python tools/rccl_smoke.py [--batch B]"""
import argparse
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from pytorch_empirical_mvm_amd import config as CFG, dist as D  # noqa: E402
from pytorch_empirical_mvm_amd.agent import Agent_Pretrain  # noqa: E402
from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain  # noqa: E402


def run(with_reducer, B, steps):
    args = CFG.get_args(vis_backbone_size="base", size_frame=8, max_size_frame=8, mvm_target=["pixel"], max_iter=1000, seed=88)
    model = VIOLET_Pretrain(args, None, device="cuda:0")
    agent = Agent_Pretrain(args, model)
    if with_reducer:
        agent.prepare_dist_model()
        assert agent.comm is not None and agent.world_size == 1
    agent.sched_step = 50
    img, txt, mask = bench.synth_batch(args, B, "cuda:0", 777)
    random.seed(3); np.random.seed(3); torch.manual_seed(3)
    mb = agent.prepare_batch(agent.masking(img, txt, mask, None))
    rs = np.random.RandomState(11)
    for _ in range(steps):
        neg = model.engine.sample_negatives(B, rs)
        dp = model.engine.sample_drop_path(B, rs)
        agent.step(mb, is_train=True, negatives=neg, dp_all=dp, sync=False)
    torch.cuda.synchronize()
    S = model.engine.store
    return S.flat[:S.total].clone()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    rank, world, local = D.init_from_env("nccl")
    assert world == 1 and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl", "needs VMVM_FORCE_DIST=1 + env:// variables"
    torch.cuda.set_device(0)
    t = torch.ones(1 << 20, device="cuda:0")
    torch.distributed.all_reduce(t)                     # RCCL communicator creation + one collective
    torch.cuda.synchronize()
    assert float(t.sum().item()) == float(1 << 20)
    p1 = run(True, a.batch, a.steps)
    os.environ.pop("VMVM_FORCE_DIST")                   # D.is_initialized() -> False at world size 1: no reducer
    p0 = run(False, a.batch, a.steps)
    same = bool(torch.equal(p0, p1))
    print(f"rccl world-1: identical={same} max|diff|={float((p0 - p1).abs().max()):.3e} params={p0.numel()}", flush=True)
    torch.distributed.destroy_process_group()
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
