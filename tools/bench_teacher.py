#!/usr/bin/env python3
"""Time the dVAE tokenizer first pass (PyTorch conv2d) on 256 frames of 224^2 (= B 32 x T 8)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_empirical_mvm_amd.dvae import DalleTeacher
dev = "cuda"
img = torch.randn(256, 3, 224, 224, device=dev)
for name, setup in (("default", lambda t: None), ("benchmark", lambda t: setattr(torch.backends.cudnn, "benchmark", True)),
                    ("benchmark+channels_last", lambda t: setattr(t, "channels_last", True))):
    t = DalleTeacher(256, 8192, device=dev)
    setup(t)
    for _ in range(2): t.extract_vq_token(img)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3): t.extract_vq_token(img)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    print(f"{name:26s}: {dt * 1e3:.1f} ms  {256 * 208.5e9 / dt / 1e12:.0f} TFLOP/s")
