import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_empirical_mvm_amd import kernels as K
dev = "cuda"; torch.manual_seed(0)
nseq, Lq, heads, Hd = 8, 432, 12, 768
qkv = (torch.randn(nseq * Lq, 3 * Hd, device=dev)).to(torch.bfloat16)
km = torch.ones(nseq, Lq, dtype=torch.uint8, device=dev)
km[0, 37] = 0; km[1, 200:216] = 0; km[2, 0:3] = 0; km[3, 120:330] = 0; km[4, 431] = 0; km[5, 400:] = 0; km[6, 16:32] = 0
kw = dict(q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km)
out, lse = K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, **kw)
qf = qkv.float().requires_grad_(True)
x = qf.view(nseq, Lq, 3, heads, 64).permute(2, 0, 3, 1, 4)
sc = (x[0] * 0.125) @ x[1].transpose(-1, -2) + torch.where(km.bool(), 0.0, float("-inf"))[:, None, None, :]
ref = (torch.softmax(sc, -1) @ x[2]).transpose(1, 2).reshape(nseq * Lq, Hd)
dout = torch.randn(nseq * Lq, Hd, device=dev).to(torch.bfloat16)
ref.backward(dout.float())
g = K.attention_bwd(dout, qkv, out, lse, nseq, Lq, heads, 64, 1, 0.125, **kw)
dq, rq = g[:, :Hd].float().view(nseq, Lq, heads, 64), qf.grad[:, :Hd].view(nseq, Lq, heads, 64)
for s in range(nseq):
    e = (dq[s] - rq[s]).abs()
    bad_rows = (e.amax(dim=(1, 2)) > 0.05 * rq[s].abs().max()).nonzero().flatten().tolist()
    print(f"seq {s}: max err {e.max().item():.4f} (ref max {rq[s].abs().max().item():.3f}); bad query tiles {sorted(set(r // 16 for r in bad_rows))[:30]}; nan {torch.isnan(dq[s]).any().item()}")
