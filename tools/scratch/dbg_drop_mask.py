"""debug: the stored dropout record of the fusion attention forward against the mask recovered from the kernel's own output (V = one-hot)"""
import torch
from pytorch_empirical_mvm_amd import kernels as K
dev = torch.device("cuda:0")
nseq, Lq, heads, Hd = 1, 432, 1, 64
BF = torch.bfloat16
true = torch.zeros(Lq, Lq, device=dev)
dm = K.attention_drop_mask(nseq, Lq, heads, 64, 1, 0.1, dev)
for k0 in range(0, Lq, 64):
    qkv = torch.zeros(nseq * Lq, 3 * Hd, device=dev, dtype=BF)
    n = min(64, Lq - k0)
    qkv[k0:k0 + n, 2 * Hd:2 * Hd + n] = torch.eye(n, device=dev, dtype=BF)
    dm.fill_(-1)
    out, lse = K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, dropout_p=0.1, seed=23, offset=77, drop_mask=dm)
    true[:, k0:k0 + n] = (out.float()[:, :n] == 0).float()
bits = ((dm.view(nseq * heads, 27, 27, 4, 2, 1).to(torch.int64) >> torch.arange(32, device=dev)) & 1)
bits = bits.reshape(nseq * heads, 27, 27, 4, 4, 16)
dec = bits.permute(0, 1, 5, 2, 4, 3).reshape(Lq, Lq).float()
print("true frac", true.mean().item(), "decoded frac", dec.mean().item(), "mismatch", (true != dec).float().mean().item())
print("untouched dwords", (dm == -1).float().mean().item())
d = dm.view(27, 27, 8)
print("record (0,0):", [hex(x & 0xffffffff) for x in d[0, 0].tolist()])
t00 = true[:16, :16]
for j in range(4):
    w = 0
    for g in range(4):
        for r in range(16):
            if t00[r, 4 * g + j] > 0:
                w |= 1 << (16 * g + r)
    print("expected word", j, hex(w & 0xffffffff), hex(w >> 32))
