"""Is the host ahead of the GPU inside a step?  Wraps a few engine / agent entry points; at each entry it asks the main stream whether
everything enqueued so far has already run (stream.query() == True: the GPU is idle, the host is behind) and how long the host took
since the previous probe.  Usage: python tools/scratch/lead_probe.py [bench args]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pytorch_empirical_mvm_amd import engine as E, agent as A, engine_heads as EH
log = collections.defaultdict(lambda: [0, 0, 0.0])
last = [time.perf_counter()]
def probe(name):
    idle = torch.cuda.current_stream().query()
    now = time.perf_counter()
    s = log[name]; s[0] += 1; s[1] += int(idle); s[2] += now - last[0]
    last[0] = now
def wrap(cls, meth, name=None):
    f = getattr(cls, meth)
    def g(self, *a, **k):
        probe((name or meth) + " enter")
        r = f(self, *a, **k)
        probe((name or meth) + " exit")
        return r
    setattr(cls, meth, g)
Eng = E.VioletEngine
for m in ("sample_drop_path", "swin_forward", "encode", "go_cross", "forward_backward"):
    if hasattr(Eng, m): wrap(Eng, m)
for m in ("masking_device", "backward_step", "step"):
    wrap(A.Agent_Pretrain if hasattr(A, "Agent_Pretrain") else A.Agent, m)
import bench
sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[1:]
bench.main()
print("--- probe point: calls, GPU-idle count, host ms since previous probe (sum)")
for k, (n, idle, t) in log.items():
    print(f"{k:28s} {n:4d} {idle:4d} {t * 1e3:9.1f}")
