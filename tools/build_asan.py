#!/usr/bin/env python3
"""ASAN + UBSAN build of libvmvm's HOST side (SURVEY 5 "race detection / sanitizers"): every translation unit of
pytorch_empirical_mvm_amd/csrc compiled with `-fsanitize=address,undefined -fno-gpu-sanitize` -- the device code objects are the
ordinary gfx950 ones, the host dispatch / validation / plan arithmetic is instrumented -- into build/libvmvm_asan.so (git-ignored,
gpurun-ignored: GPU AddressSanitizer / XNACK runs are not available on the pool and the sanitizer library never travels there).

    python tools/build_asan.py            # -> build/libvmvm_asan.so (content-hashed like the production library)
    VMVM_LIB=build/libvmvm_asan.so LD_PRELOAD=$(python tools/build_asan.py --runtime) ASAN_OPTIONS=detect_leaks=0 python tools/cabi_validation.py

tests/test_sanitizers_cpu.py does exactly that in the build container (CPU only).
"""
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_empirical_mvm_amd import build as B      # noqa: E402  (SOURCES, CSRC: one source list for both builds)

OUT_DIR = os.path.join(ROOT, "build")
LIB = os.path.join(OUT_DIR, "libvmvm_asan.so")
INFO = os.path.join(OUT_DIR, "libvmvm_asan.build.json")
SAN = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
FLAGS = ["--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-Wno-pass-failed"] + SAN


def runtime():
    """the shared ASAN runtime of hipcc's clang (to be pre-loaded into the un-instrumented python)"""
    c = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return c[-1] if c else None


def build(verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    h = hashlib.sha256((B._source_hash() + " ".join(FLAGS)).encode()).hexdigest()
    if os.path.exists(LIB) and os.path.exists(INFO) and json.load(open(INFO)).get("hash") == h:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    t0 = time.time()
    procs, objs = [], []
    for s in B.SOURCES:
        o = os.path.join(OUT_DIR, s.replace(".hip", ".asan.o"))
        objs.append(o)
        procs.append((s, subprocess.Popen([hipcc] + FLAGS + ["-I", os.path.join(B.CSRC, "hooks"), "-c", os.path.join(B.CSRC, s), "-o", o])))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc (sanitizer build) failed on {s}")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-shared-libsan"] + SAN + ["-o", LIB] + objs)
    json.dump(dict(hash=h, flags=FLAGS, seconds=round(time.time() - t0, 1)), open(INFO, "w"))
    if verbose:
        print(f"libvmvm_asan: built in {time.time() - t0:.0f} s", flush=True)
    return LIB


if __name__ == "__main__":
    if "--runtime" in sys.argv:
        print(runtime() or "")
    else:
        print(build())
