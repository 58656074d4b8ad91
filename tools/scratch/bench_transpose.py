"""Time the per-step W^T refresh (vmvm_transpose_batched_bf16 over every Linear weight of the arena)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from pytorch_empirical_mvm_amd import config as CFG, kernels as K  # noqa: E402
from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain  # noqa: E402

args = CFG.get_args(vis_backbone_size="base", size_frame=8, max_size_frame=8, mvm_target=["pixel"])
model = VIOLET_Pretrain(args, None, device="cuda")
S = model.engine.store
S.refresh_transposed()
torch.cuda.synchronize()
nt = S.ttable.shape[0]
byt = sum(n * k for (_, n, k) in S.tmap.values()) * 2
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(2):
    e0.record()
    for _ in range(10):
        K.transpose_batched(S.shadow, S.shadowT, S.ttable)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"transpose_batched: {nt} tiles, {byt / 1e6:.0f} MB in + out each: {ms:.3f} ms = {2 * byt / ms / 1e6:.0f} GB/s")
# correctness spot check
for n, (o, N_, K_) in list(S.tmap.items())[:6] + list(S.tmap.items())[-6:]:
    w = S.shadow[o:o + N_ * K_].view(N_, K_)
    wt = S.shadowT[o:o + N_ * K_].view(K_, N_)
    assert torch.equal(w.t().contiguous(), wt), n
print("ok")
