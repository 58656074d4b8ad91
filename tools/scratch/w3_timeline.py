"""per-phase cycle stamps of one wave pair of attn_bwd_dkv_win3_kernel (probe build loaded through VMVM_LIB:
hipcc ... -I tools/probe/hooks -I pytorch_empirical_mvm_amd/csrc -DW3_TIMELINE: the stamps live in tools/probe/hooks/vmvm_probe_hooks.h)"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_empirical_mvm_amd import kernels as K, swin_index as SI, lib as L
dev = "cuda"
B_, heads, N = 32, 16, 392
nW = 4; nseq = B_ * nW; C_ = heads * 32
qkv = torch.randn(nseq * N, 3 * C_, device=dev).to(torch.bfloat16)
rc, rc0 = SI.rc_codes(N, (8, 7, 7)); pm = SI.win3_perm()
rc_t = torch.from_numpy(np.ascontiguousarray(rc[pm])).to(dev)
table = torch.randn(2535, heads, device=dev) * 0.1
kw = dict(q_off=0, k_off=C_, v_off=2 * C_, bias_table=table, rc=rc_t, rc0=rc0, region=None, n_win=nW, win_layout=1)
out, lse = K.attention_fwd(qkv, nseq, N, heads, 32, 0, 32 ** -0.5, **kw)
dout = torch.randn(nseq * N, C_, device=dev).to(torch.bfloat16)
for _ in range(3):
    K.attention_bwd(dout, qkv, out, lse, nseq, N, heads, 32, 0, 32 ** -0.5, dbias_table=None, **kw)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 4096)()
so = L.load()
so.vmvm_w3_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", so.vmvm_w3_debug_read(buf, 4096))
names = {1: "walk start", 2: "prologue done (reads0+mma1(0))", 10: "tr reads + dma issued", 11: "chain done", 12: "next reads issued", 13: "mma2 done(issued)", 14: "mma1 issued", 20: "walk end"}
for w, off in ((0, 0), (5, 2048)):
    n = int(buf[off]); prev = None; t0 = None
    print(f"--- wave {w}: {n - 1} stamps")
    rows = []
    for i in range(1, n):
        v = int(buf[off + i]); slot = v >> 48; t = v & ((1 << 48) - 1)
        if t0 is None: t0 = t
        rows.append((slot, t - t0, 0 if prev is None else t - prev)); prev = t
    # aggregate deltas by slot
    agg = {}
    for slot, t, d in rows: agg.setdefault(slot, []).append(d)
    for slot in sorted(agg): print(f"   -> {names.get(slot, slot):34s} n={len(agg[slot]):3d} mean delta {np.mean(agg[slot]):8.1f} ticks  (min {min(agg[slot])}, max {max(agg[slot])})")
    print("   total walk ticks", rows[-1][1])
    print("   first 12 stamps:", [(s_, d) for s_, _, d in rows[:12]])
