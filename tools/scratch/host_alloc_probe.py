"""Where does the host stall inside a step?  Wraps every `_h2d` (pinned upload) of the package with a timer, runs bench.main() and
prints calls / total / max per upload site, plus torch's pinned-host-allocator statistics (a hipHostMalloc in steady state = a device-wide
stall).  Usage: python tools/scratch/host_alloc_probe.py [bench args]"""
import os, sys, time, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pytorch_empirical_mvm_amd.store as ST
import pytorch_empirical_mvm_amd.engine as E, pytorch_empirical_mvm_amd.engine_swin as ES, pytorch_empirical_mvm_amd.engine_heads as EH
stats = collections.defaultdict(lambda: [0, 0.0, 0.0])
orig = ST._h2d
def timed(t, device):
    t0 = time.perf_counter()
    r = orig(t, device)
    dt = time.perf_counter() - t0
    fr = traceback.extract_stack(limit=3)[0]
    k = f"{os.path.basename(fr.filename)}:{fr.lineno} {tuple(t.shape)}"
    s = stats[k]; s[0] += 1; s[1] += dt; s[2] = max(s[2], dt)
    return r
for m in (ST, E, ES, EH):
    if hasattr(m, "_h2d"): m._h2d = timed
    if hasattr(m, "_dev_i32"):
        m._dev_i32 = lambda a, device: timed(torch.from_numpy(__import__("numpy").ascontiguousarray(a, dtype="int32")), device)
import bench
sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[1:]
hs0 = torch.cuda.host_memory_stats() if hasattr(torch.cuda, "host_memory_stats") else None
bench.main()
print("--- pinned uploads (calls, total ms, max ms)")
for k, (n, tot, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{n:6d} {tot * 1e3:9.2f} {mx * 1e3:8.3f}  {k}")
if hs0 is not None:
    hs = torch.cuda.host_memory_stats()
    for k in sorted(hs):
        if ("alloc" in k or "segment" in k) and ("count" in k or "time" in k or k.endswith(".allocated")):
            print(k, hs[k])
