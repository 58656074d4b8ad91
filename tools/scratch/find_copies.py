"""which Python call sites issue the small eager ops (aten::copy_, fill_, cat, index, softmax ...) of one training step"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pytorch_empirical_mvm_amd import config as CFG
from pytorch_empirical_mvm_amd.agent import Agent_Pretrain
from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain
args = CFG.get_args(vis_backbone_size="base", size_frame=8, max_size_frame=8, mvm_target=["pixel"], max_iter=1000, seed=88)
model = VIOLET_Pretrain(args, None, device="cuda")
agent = Agent_Pretrain(args, model)
B = 32
img, txt, mask = bench.synth_batch(args, B, "cuda", 88)
def one():
    mb = agent.masking_device(img, txt, mask)
    agent.step(mb, is_train=True, sync=False)
for _ in range(2): one()
torch.cuda.synchronize()
import traceback
counts = collections.Counter()
orig = {}
def wrap(name):
    f = getattr(torch.Tensor, name)
    def g(self, *a, **k):
        st = traceback.extract_stack(limit=6)[:-1]
        key = name + " <- " + " <- ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in reversed(st[-3:]))
        counts[key] += 1
        return f(self, *a, **k)
    orig[name] = f
    setattr(torch.Tensor, name, g)
for n in ("copy_", "fill_", "zero_", "to", "contiguous", "clone", "index_add_", "float", "long", "__getitem__", "__setitem__"):
    wrap(n)
for fn in ("cat", "softmax", "zeros", "ones", "empty", "arange", "stack", "from_numpy", "tensor"):
    f = getattr(torch, fn)
    def mk(fn, f):
        def g(*a, **k):
            st = traceback.extract_stack(limit=6)[:-1]
            key = "torch." + fn + " <- " + " <- ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in reversed(st[-3:]))
            counts[key] += 1
            return f(*a, **k)
        return g
    setattr(torch, fn, mk(fn, f))
one()
torch.cuda.synchronize()
print("distinct sites", len(counts), "total", sum(counts.values()), flush=True)
tot = collections.Counter()
for k, v in counts.items():
    tot[k.split(" <- ")[0]] += v
print(dict(tot), flush=True)
for k, v in counts.most_common():
    if not k.startswith(("__getitem__", "torch.empty")):
        print(f"{v:5d}  {k}", flush=True)
