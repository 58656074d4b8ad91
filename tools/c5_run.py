"""BASELINE config 5 on one GPU at bf16: Swin-L-384 (window (8,12,12)), 16 x 384^2 frames, 32 text tokens, pixel target --
one full optimizer step (masking excluded), timed.  The fp8 GEMM path of config 5 is not built; this runs the bf16 kernels
(streaming attention for the 1152-token windows and the 2352-token fusion sequences).
usage: python tools/c5_run.py [B] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from pytorch_empirical_mvm_amd import config as CFG  # noqa: E402
from pytorch_empirical_mvm_amd.agent import Agent_Pretrain  # noqa: E402
from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda:0")
    args = CFG.get_args(vis_backbone_size="large", size_img=384, size_frame=16, max_size_frame=16, mvm_target=["pixel"], max_iter=1000)
    torch.manual_seed(88)
    model = VIOLET_Pretrain(args, None, device=dev)
    agent = Agent_Pretrain(args, model)
    img, txt, mask = bench.synth_batch(args, B, dev, 88)
    batches = [agent.prepare_batch(agent.masking(img, txt, mask, None)) for _ in range(2)]
    for i in range(2):
        r = agent.step(batches[i % 2], is_train=True)
    torch.cuda.synchronize()
    t = time.time()
    for i in range(steps):
        r = agent.step(batches[i % 2], is_train=True)
    torch.cuda.synchronize()
    dt = (time.time() - t) / steps
    print(f"C5 bf16 B={B}: {dt * 1e3:.1f} ms/step  {B / dt:.2f} clips/s  {B / dt * 21.74:.0f} TFLOP/s algorithmic  losses {r}  "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
