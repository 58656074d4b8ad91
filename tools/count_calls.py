#!/usr/bin/env python3
"""Foreign (ctypes) calls into libvmvm per training step of the default bench workload (C2, B = 32): total and by entry point.
usage: python tools/count_calls.py [batch]     (GPU)"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from pytorch_empirical_mvm_amd import config as CFG, lib as L  # noqa: E402
from pytorch_empirical_mvm_amd.agent import Agent_Pretrain  # noqa: E402
from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda:0")
    args = CFG.get_args(vis_backbone_size="base", size_img=224, size_frame=8, max_size_frame=8, mvm_target=["pixel"], max_iter=1000)
    torch.manual_seed(88)
    model = VIOLET_Pretrain(args, None, device=dev)
    agent = Agent_Pretrain(args, model)
    img, txt, mask = bench.synth_batch(args, B, dev, 88)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    for _ in range(2):
        agent.step(agent.masking_device(img, txt, mask, generator=gen), is_train=True, sync=False)
    torch.cuda.synchronize()
    so = L.load()
    counts = collections.Counter()
    for name in L.exported_symbols():
        fn = getattr(so, name)

        def wrap(*a, _fn=fn, _n=name):
            counts[_n] += 1
            return _fn(*a)
        setattr(so, name, wrap)
    steps = 3
    for _ in range(steps):
        agent.step(agent.masking_device(img, txt, mask, generator=gen), is_train=True, sync=False)
    torch.cuda.synchronize()
    tot = sum(counts.values())
    print(f"foreign calls per step (B = {B}, block_abi = {model.engine.sw.block_abi}): {tot / steps:.0f}")
    for n, c in counts.most_common(40):
        print(f"  {c / steps:7.1f}  {n}")


if __name__ == "__main__":
    main()
