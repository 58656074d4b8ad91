"""Does torch's index_add_ stall the host?  Host time of the call with ~20 ms of GPU work queued in front of it."""
import time, torch
dev = "cuda"
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
dx = torch.zeros(69120, 768, device=dev, dtype=torch.bfloat16)
rows = (torch.arange(128, device=dev, dtype=torch.int32) * 432 + 400)
dxc = torch.randn(128, 768, device=dev, dtype=torch.bfloat16)
rl = rows.long()
for name, fn in (("index_add_(rows.long())", lambda: dx.index_add_(0, rows.long(), dxc)),
                 ("index_add_(cached long)", lambda: dx.index_add_(0, rl, dxc)),
                 ("dx[rl] += dxc", lambda: dx.index_put_((rl,), dxc, accumulate=True)),
                 ("dx[rl] = dx[rl] + dxc", lambda: dx.__setitem__(rl, dx[rl] + dxc))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        for _ in range(12):
            b = a @ a                                   # ~20 ms of queued GPU work
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
    print(f"{name:28s} host ms per call: " + " ".join(f"{t:.3f}" for t in ts))
