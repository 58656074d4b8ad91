"""SURVEY 5 "sanitizers" / VERDICT r5 item 9: the C ABI's host side under AddressSanitizer + UndefinedBehaviorSanitizer, in the build
container (never on the GPU box: the pool has no GPU ASAN / XNACK).  `tools/build_asan.py` compiles every translation unit with
`-fsanitize=address,undefined -fno-gpu-sanitize`; a child python with the sanitizer runtime pre-loaded then drives
`tools/cabi_validation.py` -- every entry point with null / inconsistent / overflowing descriptors and every workspace-size query --
through it.  A heap / stack / global out-of-bounds read of a descriptor or plan table, a signed overflow, a division by zero or a
misaligned load in the dispatch code aborts the child (`-fno-sanitize-recover`).  Round 6: the first run of this sweep found a real
one -- `vmvm_gemm_workspace_size` divided by a 32-bit tile count that had wrapped to zero (gemm.hip `gemm_tiles_ok`)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.join(os.path.dirname(__file__), "..")


@pytest.mark.timeout(900)
def test_c_abi_dispatch_under_asan_and_ubsan():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build_asan
    rt = build_asan.runtime()
    if rt is None:
        pytest.skip("no shared ASAN runtime in this toolchain")
    lib = build_asan.build(verbose=False)
    env = dict(os.environ, VMVM_LIB=lib, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tools", "cabi_validation.py")], capture_output=True, text=True, env=env, timeout=800)
    tail = p.stdout[-2500:] + p.stderr[-2500:]
    assert p.returncode == 0, tail
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, tail
    assert "all refused / answered on the host" in p.stdout, tail
    # the sanitizer library really was the one loaded (not the production .so)
    assert "libvmvm_asan" in lib and os.path.getsize(lib) > 1 << 20
