"""practical HBM rates of plain torch kernels on this box: copy (1R + 1W), add (2R + 1W), read-reduce (1R), fill (1W)"""
import torch
dev = "cuda"
n = 1 << 29      # 512 Mi bf16 elements = 1 GiB per tensor
x = torch.randn(n, device=dev, dtype=torch.bfloat16); y = torch.empty_like(x); z = torch.randn_like(x)
def t(fn, byt, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:28s} {ms:7.3f} ms  {byt / ms / 1e9:7.2f} TB/s")
B = n * 2
t(lambda: y.copy_(x), 2 * B, "copy (1R+1W)")
t(lambda: torch.add(x, z, out=y), 3 * B, "add (2R+1W)")
t(lambda: x.float().sum() if False else torch.sum(x, dtype=torch.float32), B, "sum (1R)")
t(lambda: y.fill_(1.0), B, "fill (1W)")
xf = x.float()[: n // 2]; yf = torch.empty_like(xf)
t(lambda: yf.copy_(xf), 2 * xf.numel() * 4, "copy f32 (1R+1W)")
