"""Build libvmvm.so (hand-written HIP kernels, gfx950 only) in-tree with hipcc.

    python -m pytorch_empirical_mvm_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvmvm.so")
SOURCES = ["gemm.hip", "gemm_pp.hip", "layernorm.hip", "attention.hip", "attention_win3.hip", "attention_win4.hip", "misc.hip", "dvae.hip", "patch_embed.hip", "blocks.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed"]


INFO = os.path.join(HERE, "libvmvm.build.json")       # next to the .so (git-ignored like it): what the library was built from


def _source_hash():
    """sha256 over every file the library is compiled from (csrc/*.hip, csrc/*.h, include/vmvm.h) + the flags"""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) 
    for f in files:
        h.update(f.encode()); h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(CSRC, "hooks", "vmvm_probe_hooks.h"), "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "vmvm.h"), "rb").read())
    return h.hexdigest()


def _stale():
    """content-based: the recorded source hash must match (mtimes say nothing after a checkout / snapshot copy)"""
    import json
    if not os.path.exists(LIB) or not os.path.exists(INFO):
        return True
    try:
        return json.load(open(INFO)).get("source_sha256") != _source_hash()
    except Exception:
        return True


def build(force=False, verbose=True):
    import json
    import time
    force = force or os.environ.get("VMVM_FORCE_BUILD") == "1"
    if not force and not _stale():
        if verbose:
            print(f"libvmvm: up to date (source hash {_source_hash()[:16]} matches {os.path.basename(INFO)}); VMVM_FORCE_BUILD=1 rebuilds", flush=True)
        return LIB
    t0 = time.time()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc] + FLAGS + ["-I", os.path.join(CSRC, "hooks"), "-c", os.path.join(CSRC, s), "-o", o]      # hooks/: the no-op measurement hooks (tools/probe/hooks holds the instrumented twin)
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    try:
        ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.strip().splitlines()[0]
    except Exception:
        ver = "?"
    json.dump(dict(source_sha256=_source_hash(), sources=SOURCES, flags=FLAGS, hipcc=ver, seconds=round(time.time() - t0, 1),
                   built_at=time.strftime("%Y-%m-%dT%H:%M:%S")), open(INFO, "w"), indent=1)
    if verbose:
        print(f"libvmvm: rebuilt {len(SOURCES)} sources in {time.time() - t0:.0f} s ({ver})", flush=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
