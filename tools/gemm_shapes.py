#!/usr/bin/env python3
"""Per-shape GEMM time inside one real training step: wraps kernels.gemm with HIP-event timing (one sync per call, so the
step itself runs slowly; the per-call durations are what is reported)."""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pytorch_empirical_mvm_amd import config as CFG
from pytorch_empirical_mvm_amd import kernels as K
from pytorch_empirical_mvm_amd.agent import Agent_Pretrain
from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda:0"
torch.cuda.set_device(0)
args = CFG.get_args(vis_backbone_size="base", size_frame=8, max_size_frame=8, size_img=224, size_txt=32, mvm_target=["pixel"], max_iter=10000, seed=88)
model = VIOLET_Pretrain(args, None, device=dev)
agent = Agent_Pretrain(args, model)
agent.sched_step = 500
model.engine.sw.block_abi = False      # per-kernel entry points: the GEMMs of a fusion layer / Swin block are then visible to the K.gemm hook (same launches as the block-level calls)
img, txt, mask = bench.synth_batch(args, B, dev, 88)
mb = agent.prepare_batch(agent.masking(img, txt, mask, None))
for _ in range(2):
    agent.step(mb, is_train=True, sync=False)
torch.cuda.synchronize()
stats = collections.defaultdict(lambda: [0, 0.0])
orig = K.gemm


def timed(A, Bm, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if os.environ.get("VMVM_NO_PP") and not kw.get("variant"):
        kw["variant"] = 6                      # force the 128x128 persistent kernel where the dispatcher would take the 256x256 ping-pong one
    if os.environ.get("VMVM_FORCE_PP") and not kw.get("variant") and kw.get("a_kmajor", True) and kw.get("b_kmajor", True):
        try:                                   # force the 256x256 ping-pong kernel wherever it has an instantiation for the call
            kw7 = dict(kw); kw7["variant"] = 7
            e0.record()
            out = orig(A, Bm, **kw7)
            e1.record()
            torch.cuda.synchronize()
            kw = dict(kw); kw["forced_pp"] = True
        except RuntimeError:
            kw = dict(kw); kw["forced_pp"] = False
    if kw.pop("forced_pp", None):
        pass
    else:
        e0.record()
        out = orig(A, Bm, **kw)
        e1.record()
    torch.cuda.synchronize()
    ak, bk = kw.get("a_kmajor", True), kw.get("b_kmajor", True)
    M = kw.get("M") or (A.shape[0] if ak else A.shape[1])
    Kd = kw.get("K") or (A.shape[1] if ak else A.shape[0])
    N = kw.get("N") or (Bm.shape[0] if bk else Bm.shape[1])
    flags = ("T" if ak else "N") + ("T" if bk else "N")
    epi = "+".join(x for x, c in (("bias", kw.get("bias") is not None), ("act%d" % kw.get("act", 0), kw.get("act", 0)), ("aux", kw.get("aux") is not None),
                                  ("pre", kw.get("out_preact") is not None), ("c8", kw.get("code8", False)), ("res", kw.get("resid") is not None), ("map", kw.get("row_map") is not None),
                                  ("rs", kw.get("row_scale") is not None), ("drop", kw.get("dropout_p", 0) > 0), ("acc", kw.get("accumulate", False)),
                                  ("f32", kw.get("out_dtype", None) == torch.float32 or (kw.get("out") is not None and kw["out"].dtype == torch.float32))) if c)
    s = stats[(M, N, Kd, flags, epi)]
    s[0] += 1
    s[1] += e0.elapsed_time(e1)
    return out


K.gemm = timed
import pytorch_empirical_mvm_amd.engine as E
E.K.gemm = timed
agent.step(mb, is_train=True, sync=False)
torch.cuda.synchronize()
tot = sum(v[1] for v in stats.values())
print(f"GEMM total {tot:.1f} ms over {sum(v[0] for v in stats.values())} calls (B={B})")
# the vendor library on the same shapes (plain bf16 GEMM, no epilogue: torch.matmul -> hipBLASLt), as the practical ceiling of each
# shape: the same operand layouts, 5 timed launches after 2 warm-up launches, alone on the device
_ref = {}


def blaslt_tf(M, N, Kd, fl):
    key = (M, N, Kd, fl)
    if key not in _ref:
        try:
            if fl == "TT":
                a, b = torch.randn(M, Kd, device=dev, dtype=torch.bfloat16), torch.randn(N, Kd, device=dev, dtype=torch.bfloat16)
                f = lambda: torch.matmul(a, b.t())
            elif fl == "NN":
                a, b = torch.randn(Kd, M, device=dev, dtype=torch.bfloat16), torch.randn(Kd, N, device=dev, dtype=torch.bfloat16)
                f = lambda: torch.matmul(a.t(), b)
            elif fl == "TN":
                a, b = torch.randn(M, Kd, device=dev, dtype=torch.bfloat16), torch.randn(Kd, N, device=dev, dtype=torch.bfloat16)
                f = lambda: torch.matmul(a, b)
            else:
                a, b = torch.randn(Kd, M, device=dev, dtype=torch.bfloat16), torch.randn(N, Kd, device=dev, dtype=torch.bfloat16)
                f = lambda: torch.matmul(a.t(), b.t())
            for _ in range(2):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record()
            torch.cuda.synchronize()
            _ref[key] = 2.0 * M * N * Kd * 5 / e0.elapsed_time(e1) / 1e9
            del a, b
        except Exception:
            _ref[key] = float("nan")
    return _ref[key]


K.gemm = orig
print(f"{'M':>8} {'N':>6} {'K':>8} lay {'n':>4} {'ms':>8} {'avg_us':>8} {'TF':>7} {'blasLt':>7} {'ours/lt':>7}  epilogue   (blasLt = torch.matmul, plain bf16 GEMM of the shape, no epilogue)")
rows = sorted(stats.items(), key=lambda kv: -kv[1][1])
behind = []
for (M, N, Kd, fl, epi), (n, ms) in rows:
    tf = 2.0 * M * N * Kd * n / ms / 1e9
    lt = blaslt_tf(M, N, Kd, fl) if ms / tot > 0.002 else float("nan")       # (shapes below 0.2 % of the GEMM time: not timed)
    print(f"{M:8d} {N:6d} {Kd:8d} {fl}  {n:4d} {ms:8.2f} {ms / n * 1e3:8.1f} {tf:7.1f} {lt:7.1f} {tf / lt if lt == lt else float('nan'):7.2f}  {epi}")
    if lt == lt and epi in ("", "res", "bias") and tf < 0.9 * lt:
        behind.append((M, N, Kd, fl, epi, round(tf), round(lt)))
print(f"plain / bias / residual-class shapes more than 10 % behind the vendor GEMM: {len(behind)}")
for b_ in behind:
    print("   ", b_)
