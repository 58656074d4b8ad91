"""Does splitting M at a whole number of rounds of the 256x256 ping-pong grid (remainder rows on the 128x128 kernel) beat one launch?"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from pytorch_empirical_mvm_amd import kernels as K
BF = torch.bfloat16
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, Kd) in [(69120, 768, 3072), (69120, 768, 2304), (69120, 768, 768), (50176, 512, 2048), (50176, 512, 1536), (50176, 512, 512)]:
    A = torch.randn(M, Kd, device="cuda").to(BF); W = torch.randn(N, Kd, device="cuda").to(BF)
    R = torch.randn(M, N, device="cuda").to(BF); bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=BF)
    nbn = N // 256
    tiles = ((M + 255) // 256) * nbn
    rounds = tiles // 256
    tm = (rounds * 256) // nbn
    Ms = tm * 256
    full = lambda: K.gemm(A, W, bias=bias, resid=R, out=out)
    def split(v2):
        K.gemm(A[:Ms], W, bias=bias, resid=R[:Ms], out=out[:Ms])
        K.gemm(A[Ms:], W, bias=bias, resid=R[Ms:], out=out[Ms:], variant=v2)
    a = t(full); b = t(lambda: split(6)); c = t(lambda: split(0)); d = t(lambda: K.gemm(A[:Ms], W, bias=bias, resid=R[:Ms], out=out[:Ms]))
    fl = 2.0 * M * N * Kd
    print(f"M={M} N={N} K={Kd}: tiles {tiles} ({tiles/256:.2f} rounds) one launch {a:7.1f} us ({fl/a/1e6:6.0f} TF) | first {Ms} rows {d:7.1f} | split + 128^2 rest {b:7.1f} us ({fl/b/1e6:6.0f} TF) | split + auto rest {c:7.1f}")
