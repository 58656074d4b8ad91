"""standalone rate of vmvm_adamw / vmvm_sumsq_f32 / the gradient zero-fill on arenas of the step's sizes (Swin-B part 88 M, whole model 197 M f32)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pytorch_empirical_mvm_amd import kernels as K
dev = torch.device("cuda:0")
for n in (88_000_000, 197_000_000):
    p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
    v.abs_()
    pb = torch.empty(n, device=dev, dtype=torch.bfloat16)
    ss = torch.ones(1, device=dev)
    K.set_workspace(torch.empty(64 << 20, device=dev, dtype=torch.uint8))
    def run(fn, nbytes, name):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"n = {n / 1e6:.0f} M  {name}: {ms:.3f} ms  {nbytes / ms / 1e9:.2f} TB/s")
    run(lambda: K.adamw(p, g, m, v, pb, lr=1e-5, weight_decay=1e-3, beta1=0.9, beta2=0.98, eps=1e-8, step=3, sumsq_t=ss, max_grad_norm=1.0), n * 30, "adamw (16 B read + 14 B written per element)")
    run(lambda: K.sumsq(g, ss), n * 4, "sumsq")
    run(lambda: g.zero_(), n * 4, "zero fill (torch)")
