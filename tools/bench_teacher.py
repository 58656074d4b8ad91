#!/usr/bin/env python3
"""Time the dVAE tokenizer first pass (PyTorch conv2d) on 256 frames of 224^2 (= B 32 x T 8)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_empirical_mvm_amd.dvae import DalleTeacher
dev = "cuda"
img = torch.randn(256, 3, 224, 224, device=dev)
def _torch_path(t): t.native = False
ONLY = os.environ.get("TEACHER_NATIVE_ONLY") == "1"
for name, setup in ((("native implicit-GEMM", lambda t: None),) if ONLY else (("native implicit-GEMM", lambda t: None), ("torch conv2d NHWC", _torch_path))):
    t = DalleTeacher(256, 8192, device=dev)
    setup(t)
    for _ in range(2): t.extract_vq_token(img)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3): t.extract_vq_token(img)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    print(f"{name:26s}: {dt * 1e3:.1f} ms  {256 * 208.5e9 / dt / 1e12:.0f} TFLOP/s")

if ONLY:
    sys.exit(0)
# agreement of the native fp16 path with the PyTorch fp32 path on 8 frames (same weights)
t = DalleTeacher(256, 8192, device=dev)
tok_n = t.extract_vq_token(img[:8])
t32 = DalleTeacher(256, 8192, device=dev, dtype=torch.float32); t32.w = t.w; t32._refresh(); t32.native = False
tok_r = t32.extract_vq_token(img[:8])
t16 = DalleTeacher(256, 8192, device=dev); t16.w = t.w; t16._refresh(); t16.native = False
tok_h = t16.extract_vq_token(img[:8])
print(f"token agreement with the fp32 PyTorch path: native fp16 {float((tok_n == tok_r).float().mean()):.4f}   PyTorch fp16 {float((tok_h == tok_r).float().mean()):.4f}")
