"""Per-wave s_memtime timeline of attn_fwd_win4_kernel (probe build: libvmvm_tl.so compiled with -DW4_TIMELINE), stage-3 shape."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_empirical_mvm_amd import kernels as K, swin_index as SI
dev = "cuda"; torch.manual_seed(0)
shifted = os.environ.get("SHIFTED", "0") == "1"
B_, heads, N, nW = 32, 16, 392, 4
nseq = B_ * nW; C_ = heads * 32
qkv = torch.randn(nseq * N, 3 * C_, device=dev).to(torch.bfloat16)
rc, rc0 = SI.rc_codes(N, (8, 7, 7)); pm = SI.win3_perm()
rc_t = torch.from_numpy(np.ascontiguousarray(rc[pm])).to(dev)
table = torch.randn(2535, heads, device=dev) * 0.1
reg = None
if shifted:
    reg = torch.from_numpy(np.ascontiguousarray(SI.region_ids(8, 14, 14, (8, 7, 7), (0, 3, 3))[:, pm])).to(dev)
which = os.environ.get("WHICH", "fwd")
dout = torch.randn(nseq * N, C_, device=dev).to(torch.bfloat16)
buf = torch.zeros(4 * 13 * 32 * 2, dtype=torch.int32, device=dev)
for _ in range(3):
    out, lse = K.attention_fwd(qkv, nseq, N, heads, 32, 0, 32 ** -0.5, q_off=0, k_off=C_, v_off=2 * C_, bias_table=table, rc=rc_t, rc0=rc0, region=reg, n_win=nW, win_layout=1, drop_mask=(buf if which == "fwd" else None))
    if which != "fwd":
        K.attention_bwd(dout, qkv, out, lse, nseq, N, heads, 32, 0, 32 ** -0.5, q_off=0, k_off=C_, v_off=2 * C_, bias_table=table, rc=rc_t, rc0=rc0, region=reg, n_win=nW, dbias_table=None, win_layout=1, drop_mask=buf)
torch.cuda.synchronize()
t = buf.cpu().numpy().view(np.uint64).reshape(4, 13, 32).astype(np.int64)
t0 = t[0, :, 0].min()
names = {0: "top", 1: "barrier", 2: "setup", 20: "stores", 21: "fetch"}
for b in range(4):
    print(f"--- sequence {b + 2} (cycles since the first wave reached the top of sequence 2)")
    for w in range(13):
        r = t[b, w] - t0
        blocks = " ".join(f"{r[3 + i] - (r[2] if i == 0 else r[3 + i - 1]):5d}" for i in range(13) if t[b, w, 3 + i])
        print(f"wave {w:2d}: top {r[0]:7d} barrier +{r[1] - r[0]:5d} setup +{r[2] - r[1]:4d} | blocks {blocks} | stores +{r[20] - max(r[3:16]):5d} fetch +{r[21] - r[20]:5d} | total {r[21] - r[0]:6d}")
