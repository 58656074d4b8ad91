"""Per-wave s_memtime timeline of attn_bwd_dq_fus4_kernel (probe build: libvmvm_f4tl.so = attention_fus4.hip with -DF4_TIMELINE), C2 fusion shape.
DROP=1: stored dropout decisions; 0: no dropout."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_empirical_mvm_amd import kernels as K
dev = "cuda"; torch.manual_seed(0)
drop = os.environ.get("DROP", "0") == "1"
nseq, Lq, heads, Hd = int(os.environ.get("NSEQ", "160")), 432, 12, 768
qkv = torch.randn(nseq * Lq, 3 * Hd, device=dev).to(torch.bfloat16)
km = torch.ones(nseq, Lq, dtype=torch.uint8, device=dev)
kw = dict(q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km)
if drop:
    kw.update(dropout_p=0.1, seed=1, drop_mask=K.attention_drop_mask(nseq, Lq, heads, 64, 1, 0.1, dev))
out, lse = K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, **kw)
dout = torch.randn(nseq * Lq, Hd, device=dev).to(torch.bfloat16)
buf = torch.zeros(12 * 64 * 2, dtype=torch.float32, device=dev)
for _ in range(3):
    K.attention_bwd(dout, qkv, out, lse, nseq, Lq, heads, 64, 1, 0.125, dbias_table=buf, **kw)
torch.cuda.synchronize()
t = buf.view(torch.int32).cpu().numpy().view(np.uint64).reshape(12, 64).astype(np.int64)
t0 = t[:, 0].min()
for w in range(8):
    r = t[w] - t0
    tsA = [i for i in range(14) if t[w, 3 + i]]
    tsB = [i for i in range(14, 27) if t[w, 3 + i]]
    mA = " ".join(f"{r[3 + i] - (r[2] if i == 0 else r[3 + i - 1]):4d}" for i in tsA)
    mB = " ".join(f"{r[3 + i] - (r[32] if i == 14 else r[3 + i - 1]):4d}" for i in tsB)
    pa = [i for i in range(14) if t[w, 33 + i]]
    pb = [i for i in range(14, 27) if t[w, 33 + i]]
    tA = " ".join(f"{r[33 + i] - (r[1] if i == pa[0] else r[33 + i - 1]):4d}" for i in pa)
    tB = " ".join(f"{r[33 + i] - (r[31] if i == pb[0] else r[33 + i - 1]):4d}" for i in pb)
    print(f"wave {w:2d}: top {r[0]:6d} P1 +{r[1] - r[0]:5d} | tail A {tA} | main A {mA} | loads + X +{r[31] - r[30]:5d} | tail B {tB} | main B {mB} | store +{r[62] - r[3 + 26]:4d} Y + reduce +{r[63] - r[62]:5d} | unit {r[63] - r[0]:6d}")
