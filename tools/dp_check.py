"""Multi-rank data-parallel step on the GPU path (launched by torch.distributed.run).  On a 1-GPU box: VMVM_DIST_BACKEND=gloo
VMVM_SHARE_GPU=1 puts every rank on cuda:0 and moves the collectives through gloo, which exercises everything except RCCL itself:
rank-0 broadcast, the side-stream reductions hooked into the backward (non-Swin groups, Swin tail, rest), the 1/world scale in
AdamW, replica consistency.  Checks after 3 steps: parameters bit-identical on all ranks; the reduced gradient of step 1 equals
the mean of the per-rank gradients computed without the reducer."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from pytorch_empirical_mvm_amd import config as CFG, dist as D  # noqa: E402
from pytorch_empirical_mvm_amd.agent import Agent_Pretrain  # noqa: E402
from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain  # noqa: E402


def concat_check(agent, model, mb, world, rank, dev):
    """SURVEY 8 a17's pin: the data-parallel result equals ONE process on the concatenated batch with the same negatives.  The
    reference's DDP averages per-rank MEAN losses, which is the mean over the concatenated batch exactly when the per-rank denominators
    agree -- so the ranks' MLM target counts and covered-patch counts are first equalised (extra targets / covered patches are dropped
    down to the minimum over the ranks; both sides see the same edited batch) and B_local = 4 keeps O = min(B, 4) the same on both
    sides, with the single process' negatives block-diagonal (drawn inside each rank's own clips, main_pretrain.py:250).  Eval mode
    (no dropout / DropPath draws).  Returns max |reduced / world - single| / max |single| over the gradient arena."""
    eng, S = model.engine, model.engine.store
    B = mb["txt"].shape[0]
    ans, cov = mb["ans_mtm"].clone(), mb["cov"].clone()
    n_t = torch.tensor([int((ans != -1).sum()), int(cov.sum())], device=dev)
    dist.all_reduce(n_t, op=dist.ReduceOp.MIN)
    idx = (ans != -1).flatten().nonzero().flatten()
    ans.view(-1)[idx[int(n_t[0]):]] = -1
    idc = cov.flatten().nonzero().flatten()
    cov.view(-1)[idc[int(n_t[1]):]] = 0
    assert int((ans != -1).sum()) == int(n_t[0]) > 0 and int(cov.sum()) == int(n_t[1]) > 0
    b = dict(img=mb["img"].float().contiguous(), cov=cov.contiguous(), txt=mb["txt"].contiguous(), mask=mb["mask"].contiguous(), ans_mtm=ans.contiguous())
    neg = eng.sample_negatives(B, np.random.RandomState(11 + rank))
    # data-parallel side: every rank its own clips, gradients through the reducer
    S.grad.zero_()
    eng.on_swin_tail_ready = agent.comm.reduce_swin_tail
    eng.forward_backward(b, negatives=neg, train=False, backward=True, on_other_grads_ready=agent.comm.reduce_other)
    agent.comm.reduce_swin_and_wait()
    torch.cuda.synchronize()
    own = agent.comm.own                                      # list of (lo, hi): the whole arena, or this rank's ZeRO-1 parts
    cat_own = lambda t: torch.cat([t[lo:hi] for lo, hi in own])
    red = (cat_own(S.grad) / world).clone()
    # single-process side (computed redundantly on every rank): the concatenated batch, block-diagonal negatives
    cat = {}
    for k, v in b.items():
        parts = [torch.empty_like(v) for _ in range(world)]
        dist.all_gather(parts, v)
        cat[k] = torch.cat(parts, 0).contiguous()
    negs = [torch.empty_like(torch.from_numpy(np.ascontiguousarray(neg)).to(dev)) for _ in range(world)]
    dist.all_gather(negs, torch.from_numpy(np.ascontiguousarray(neg)).to(dev))
    neg_cat = np.concatenate([n_.cpu().numpy() + r_ * B for r_, n_ in enumerate(negs)], 0)
    S.grad.zero_()
    eng.on_swin_tail_ready = None
    eng.forward_backward(cat, negatives=neg_cat, train=False, backward=True)
    torch.cuda.synchronize()
    one = cat_own(S.grad).clone()
    S.grad.zero_()
    err = float((red - one).abs().max() / (one.abs().max() + 1e-12))
    cos = float((red.double() @ one.double()) / (red.double().norm() * one.double().norm() + 1e-30))
    # (bf16 wire: each rank's term is rounded once before the sum; f32 wire: only summation order differs)
    assert cos > (0.99999 if agent.comm.wire == "f32" else 0.9999) and err < (2e-3 if agent.comm.wire == "f32" else 1e-2), (cos, err)
    return err


def main():
    rank, world, local = D.init_from_env("nccl")
    assert world >= 2, "launch with torch.distributed.run --nproc-per-node >= 2"
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    args = CFG.get_args(vis_backbone_size="tiny", size_frame=4, max_size_frame=6, mvm_target=["pixel"], max_iter=100, seed=88 + 17 * rank,
                        bert_layers=2)
    model = VIOLET_Pretrain(args, None, device=dev)          # different seeds: the broadcast must make the replicas identical
    agent = Agent_Pretrain(args, model)
    agent.prepare_dist_model()
    agent.sched_step = 5
    S = model.engine.store
    B = 4
    img, txt, mask = bench.synth_batch(args, B, dev, 500 + rank)
    import random
    random.seed(3 + rank); np.random.seed(3 + rank); torch.manual_seed(3 + rank)
    mb = agent.prepare_batch(agent.masking(img, txt, mask, None))
    neg = model.engine.sample_negatives(B, np.random.RandomState(1))
    dp = model.engine.sample_drop_path(B, np.random.RandomState(2))
    # reference gradient of this rank WITHOUT the reducer (same dropout stream: rng_offset restored)
    eng = model.engine
    off0 = eng.rng_offset
    b = dict(img=mb["img"].float(), cov=mb["cov"], txt=mb["txt"], mask=mb["mask"], ans_mtm=mb["ans_mtm"])
    S.grad.zero_()
    eng.forward_backward(b, negatives=neg, train=True, dp_all=dp)
    own = S.grad[:S.n_trainable].clone()
    gather = [torch.empty_like(own) for _ in range(world)]
    dist.all_gather(gather, own)
    mean_ref = sum(gather) / world
    # the same step through the agent (hooks + side stream); AdamW applies 1/world, so compare the SUM in the arena
    eng.rng_offset = off0
    S.grad.zero_()
    eng.on_swin_tail_ready = agent.comm.reduce_swin_tail
    eng.forward_backward(b, negatives=neg, train=True, dp_all=dp, on_other_grads_ready=agent.comm.reduce_other)
    agent.comm.reduce_swin_and_wait()
    torch.cuda.synchronize()
    own = agent.comm.own                                      # ZeRO-1 (VMVM_ZERO1=1): a rank holds the reduced gradient of its parts only
    cat_own = lambda t: torch.cat([t[lo:hi] for lo, hi in own])
    red = cat_own(S.grad) / world
    mean_ref = cat_own(mean_ref)
    gather = [cat_own(g_) for g_ in gather]
    err = float((red - mean_ref).abs().max() / (mean_ref.abs().max() + 1e-12))
    wire = agent.comm.wire
    # f32 payload: the mean to rounding; bf16 payload: two roundings of 2^-9 relative each (cast + reduction), element by element
    if wire == "f32":
        assert err < 1e-5, err
    else:
        mag = sum(g_.abs() for g_ in gather) / world           # rounding acts on each rank's term, not on the (possibly cancelling) mean
        rel = float(((red - mean_ref).abs() / (mag + 1e-6 * mag.max())).max())
        assert rel < 2.0 ** -7 and agent.comm.wire_bytes == 2 * S.n_trainable, (rel, agent.comm.wire_bytes)
    from pytorch_empirical_mvm_amd import kernels as K
    assert K.RESERVE_CUS == 0 and K.RESERVE_EVENT is None     # back to the whole chip once the reductions have been waited for
    concat_err = concat_check(agent, model, mb, world, rank, dev)
    # Round 6: the AUTOGRAD-driven step under data parallel -- out = model(batch); a torch loss; loss.backward() -- must start the same
    # exchange phases from inside its backward node (model._OpenStep installs the reducer's hooks) and leave the same reduced gradient
    # as the engine-driven step on the same batch, draws and Philox offsets.  The loss here is a fixed linear functional of the three
    # outputs (the same on both paths: what is compared is the exchange, not a loss function).
    eng.rng_offset = off0
    S.grad.zero_()
    model.train()
    rb = dict(img=b["img"], cov=b["cov"], unmask_img=b["img"], txt=b["txt"], mask=b["mask"], ans_mtm=b["ans_mtm"])
    n0 = agent.comm.collectives
    out = model(rb, negatives=neg, dp_all=dp)
    wm = torch.linspace(-1, 1, out["out_mtm"].shape[-1], device=dev) * 1e-3
    loss = (out["out_mtm"] * wm).sum() + out["out_mvm"].float().mean() + out["out_vtm"].sum() * 0.1
    loss.backward()
    agent.comm.reduce_swin_and_wait()
    torch.cuda.synchronize()
    g_auto = cat_own(S.grad).clone()
    phases_auto = agent.comm.collectives - n0
    # the same functional through the engine's open step with explicit hooks
    eng.rng_offset = off0
    S.grad.zero_()
    n0 = agent.comm.collectives
    eng.on_swin_tail_ready, eng.on_fusion_mid_ready = agent.comm.reduce_swin_tail, agent.comm.reduce_other_early
    outs, tr = eng.forward_open(dict(img=b["img"], cov=b["cov"], txt=b["txt"], mask=b["mask"]), negatives=neg, train=True, dp_all=dp)
    d_mtm = wm.expand_as(outs["out_mtm"]).contiguous()
    d_mvm = torch.full_like(outs["out_mvm"], 1.0 / outs["out_mvm"].numel())
    d_vtm = torch.full_like(outs["out_vtm"], 0.1)
    eng.backward_open(tr, d_mtm, d_mvm, d_vtm, on_other_grads_ready=agent.comm.reduce_other)
    eng.on_swin_tail_ready = eng.on_fusion_mid_ready = None
    agent.comm.reduce_swin_and_wait()
    torch.cuda.synchronize()
    g_eng = cat_own(S.grad).clone()
    auto_err = float((g_auto - g_eng).abs().max() / (g_eng.abs().max() + 1e-12))
    assert float(g_eng.abs().max()) > 0 and auto_err < 1e-3 and phases_auto == agent.comm.collectives - n0, (auto_err, phases_auto, agent.comm.collectives - n0)
    S.grad.zero_()
    wb0, nc0 = agent.comm.wire_bytes, agent.comm.collectives
    S.grad.zero_()
    for _ in range(3):
        r = agent.step(mb, is_train=True)
    torch.cuda.synchronize()
    flat = S.flat[:S.total].clone()
    ref = flat.clone()
    dist.broadcast(ref, 0)
    same = bool(torch.equal(flat, ref))
    if not same:
        d = (flat - ref).abs()
        print(f"rank {rank}: {int((d > 0).sum())} of {flat.numel()} parameters differ from rank 0, max abs diff {float(d.max()):.3e}", flush=True)
    if rank == 0:
        print(f"param-checksum {float(flat.double().abs().sum()):.9e}", flush=True)
    t = torch.tensor([1.0 if same else 0.0], device=dev)
    dist.all_reduce(t)
    if rank == 0:
        per_step = (agent.comm.wire_bytes - wb0) / 3 / S.n_trainable
        print(f"dp_check world={world} backend={dist.get_backend()} wire={wire} zero1={int(agent.comm.zero1)} wire-bytes/element/step {per_step:.2f} "
              f"reserve_cus={agent.comm.reserve_cus} grad-mean rel err {err:.2e} "
              f"concat-batch rel err {concat_err:.2e} autograd-step rel err {auto_err:.2e} replicas identical={int(t.item()) == world} losses {r}", flush=True)
    assert int(t.item()) == world
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
