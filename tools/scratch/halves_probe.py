"""Probe: does running the Video-Swin forward + backward of a B = 32 batch as TWO half batches on two streams (kernels of the two chains
co-resident, the way the weight-gradient stream already overlaps the main one) beat one chain?  Timing only -- the two chains share the
scratch workspace here, so the gradients of this probe are not valid.
usage: python tools/scratch/halves_probe.py [B]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from pytorch_empirical_mvm_amd import config as CFG, lib as L
from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
args = CFG.get_args(vis_backbone_size="base", size_img=224, size_frame=8, max_size_frame=8, mvm_target=["pixel"], max_iter=1000)
torch.manual_seed(88)
model = VIOLET_Pretrain(args, None, device=dev)
eng = model.engine
from pytorch_empirical_mvm_amd.agent import Agent_Pretrain
agent = Agent_Pretrain(args, model)
img, txt, mask = bench.synth_batch(args, B, dev, 88)
mb = agent.prepare_batch(agent.masking(img, txt, mask, None))
for _ in range(2):
    agent.step(mb, is_train=True, sync=False)
torch.cuda.synchronize()
img = mb["unmask_img"].to(dev, torch.float32).contiguous()
cov = mb["cov"].to(dev).contiguous()
print("img", tuple(img.shape), img.dtype, "cov", tuple(cov.shape), cov.dtype, flush=True)


def fwd(x, c):
    eng.tape.clear()
    out, dims, C = eng.swin_forward(x, c, None)
    tape = list(eng.tape)
    eng.tape.clear()
    return out, tape


def run_one(x, c):
    out, tape = fwd(x, c)
    out.g = torch.ones_like(out.t) * 1e-3
    for f in reversed(tape):
        f()
    eng._wgrad_join()


def run_two(xa, ca, xb, cb, s1, s2):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with L.on_stream(s1):
        oa, ta = fwd(xa, ca)
    with L.on_stream(s2):
        ob, tb = fwd(xb, cb)
    with L.on_stream(s1):
        oa.g = torch.ones_like(oa.t) * 1e-3
    with L.on_stream(s2):
        ob.g = torch.ones_like(ob.t) * 1e-3
    n = len(ta)
    for i in range(n):
        with L.on_stream(s1):
            ta[n - 1 - i]()
        with L.on_stream(s2):
            tb[n - 1 - i]()
    cur.wait_stream(s1); cur.wait_stream(s2)
    eng._wgrad_join()
    return None


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with L.pin_current():
    t_full = timed(lambda: run_one(img, cov))
    print("one chain", t_full, flush=True)
    h = B // 2
    t_half_seq = timed(lambda: (run_one(img[:h].contiguous(), cov[:h].contiguous()), run_one(img[h:].contiguous(), cov[h:].contiguous())))
    print("halves in sequence", t_half_seq, flush=True)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    xa, xb, ca, cb = img[:h].contiguous(), img[h:].contiguous(), cov[:h].contiguous(), cov[h:].contiguous()
    t_two = timed(lambda: run_two(xa, ca, xb, cb, s1, s2))
print(f"Video-Swin forward + backward (eval, B = {B}): one chain {t_full:.2f} ms; two half batches one after the other {t_half_seq:.2f} ms; two half batches on two streams {t_two:.2f} ms")
