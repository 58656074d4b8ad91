"""the roofline GEMM (fusion FFN fc1 + bias + GELU + 8-bit GELU' code, 69120 x 3072 x 768) and the Swin stage-3 fc1 (47040 x 2048 x 512) on the
128x128 persistent kernel (variant 6, what the dispatch picks for this class) against the 256x256 ping-pong kernel (variant 7), after the
round-4 epilogue diet; + the same shapes with bias only, for scale"""
import torch
from pytorch_empirical_mvm_amd import kernels as K
dev = torch.device("cuda:0")
BF = torch.bfloat16
for (M, N, Kd) in [(69120, 3072, 768), (69120, 3072, 768), (47040, 2048, 512)]:
    A = (torch.randn(M, Kd, device=dev) * 0.5).to(BF)
    W = (torch.randn(N, Kd, device=dev) * 0.05).to(BF)
    b = torch.randn(N, device=dev)
    u = torch.empty((M, N), device=dev, dtype=torch.uint8)
    for name, kw in (("bias+GELU+code8", dict(bias=b, act=1, out_preact=u, code8=True)), ("bias", dict(bias=b))):
        for v in (6, 7, 6, 7, 0, 6, 7):
            try:
                fn = lambda: K.gemm(A, W, variant=v, **kw)
                fn(); torch.cuda.synchronize()
            except RuntimeError as e:
                print(f"{M}x{N}x{Kd} {name} variant {v}: {e}")
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f"{M}x{N}x{Kd} {name:16s} variant {v} ({'128x128 persistent' if v == 6 else '256x256 ping-pong' if v == 7 else 'dispatch'}): {ms * 1e3:.1f} us  {2.0 * M * N * Kd / ms / 1e9:.1f} TF")
