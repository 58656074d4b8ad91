"""RCCL smoke at world size 1 (the driver's GPU box has one GPU): torch.distributed backend "nccl" (= RCCL on ROCm) is
initialised, the gradient reducer is forced on (VMVM_FORCE_DIST=1) and three optimizer steps at the C2 shapes run with the
side-stream all-reduces issued next to the persistent GEMMs; the reduced gradient must equal the gradient of the run without a
reducer (up to the run-to-run floor of the f32 atomics).
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
    """returns (gradient arena of one forward/backward through the hooks, parameters after `steps` optimizer steps)"""
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
    eng, S = model.engine, model.engine.store
    neg, dp = eng.sample_negatives(B, rs), eng.sample_drop_path(B, rs)
    b = dict(img=mb["unmask_img"].float().contiguous(), cov=mb["cov"].contiguous(), txt=mb["txt"], mask=mb["mask"], ans_mtm=mb["ans_mtm"])
    off0 = eng.rng_offset
    S.grad.zero_()
    if with_reducer:
        eng.on_swin_tail_ready = agent.comm.reduce_swin_tail
        eng.forward_backward(b, negatives=neg, train=True, dp_all=dp, on_other_grads_ready=agent.comm.reduce_other)
        agent.comm.reduce_swin_and_wait()
    else:
        eng.forward_backward(b, negatives=neg, train=True, dp_all=dp)
    torch.cuda.synchronize()
    grad = S.grad[:S.n_trainable].clone()
    S.grad.zero_()
    eng.rng_offset = off0
    last = None
    for _ in range(steps):
        last = agent.step(mb, is_train=True, negatives=neg, dp_all=dp, sync=True)
    torch.cuda.synchronize()
    assert all(np.isfinite(v) for v in last.values()), last
    return grad, S.flat[:S.total].clone()


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
    g1, p1 = run(True, a.batch, a.steps)
    os.environ.pop("VMVM_FORCE_DIST")                   # D.is_initialized() -> False at world size 1: no reducer
    g0, p0 = run(False, a.batch, a.steps)
    g0b, _ = run(False, a.batch, 1)                     # run-to-run floor of the gradient (f32 atomics in a few reductions)
    scale = float(g0.abs().max())
    err, floor = float((g1 - g0).abs().max()) / scale, float((g0b - g0).abs().max()) / scale
    n0 = float(g0.double().norm())
    err2, floor2 = float((g1 - g0).double().norm()) / n0, float((g0b - g0).double().norm()) / n0
    # a segment reduced twice / not at all is an O(1) relative error of its entries; the atomics' reordering noise is ~1e-6 and its
    # maximum over 2e8 entries varies a few-fold from run to run, hence the generous multiple of the floor
    wire = D.grad_wire()
    if wire == "f32":
        same = err <= max(20 * floor, 1e-4) and err2 <= max(20 * floor2, 1e-5)
    else:       # 16-bit payload: every entry rounded to bf16 once (world size 1: cast, reduce = identity, cast back): 2^-9 relative per entry
        rel = float(((g1 - g0).abs() / (g0.abs() + 1e-6 * scale)).max())
        same = rel <= 2.0 ** -8 + 20 * floor and err2 <= 2.0 ** -8
    # (parameters after AdamW steps are NOT compared bit for bit: a near-zero gradient entry whose last bit differs between two runs
    #  flips the sign of its Adam update)
    print(f"rccl world-1: wire={wire} identical={same} grad rel diff max {err:.3e} l2 {err2:.3e} (run-to-run floor {floor:.3e} / {floor2:.3e}) |dparam| {float((p0 - p1).abs().max()):.3e} "
          f"params={p0.numel()}", flush=True)
    torch.distributed.destroy_process_group()
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
