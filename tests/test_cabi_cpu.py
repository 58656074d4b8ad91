"""The C-ABI library loads and exports every symbol include/vmvm.h declares; ctypes struct layouts match the header's
field order; the product path fails loudly (no CPU fallback) when the library is missing.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    return open(os.path.join(ROOT, "include", "vmvm.h")).read()


def test_library_exports_every_declared_symbol():
    from pytorch_empirical_mvm_amd import build, lib
    build.build(force=False, verbose=False)
    declared = sorted(set(re.findall(r"^(?:int|int64_t) (vmvm_\w+)\(", _header(), flags=re.M)))
    assert len(declared) >= 25
    so = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(so, name), f"{name} declared in include/vmvm.h but not exported"
    assert sorted(lib.exported_symbols()) == declared, set(declared) ^ set(lib.exported_symbols())
    assert lib.load().vmvm_version() >= 1


@pytest.mark.parametrize("cname,pyname", [("vmvm_gemm_desc", "GemmDesc"), ("vmvm_ln_fwd_desc", "LnFwdDesc"), ("vmvm_ln_bwd_desc", "LnBwdDesc"),
                                          ("vmvm_attn_fwd_desc", "AttnFwdDesc"), ("vmvm_attn_bwd_desc", "AttnBwdDesc"), ("vmvm_adamw_desc", "AdamWDesc"), ("vmvm_bert_layer", "BertLayer"), ("vmvm_swin_block", "SwinBlock")])
def test_ctypes_structs_follow_header_field_order(cname, pyname):
    from pytorch_empirical_mvm_amd import lib
    m = re.search(r"typedef struct \{([^}]*)\} " + cname + ";", _header())
    assert m, cname
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        parts = decl.split(",")
        first = parts[0].split()
        names.append(first[-1].lstrip("*"))
        names += [p.strip().lstrip("*") for p in parts[1:]]
    py = [f[0] for f in getattr(lib, pyname)._fields_]
    assert py == names, (py, names)


def test_invalid_arguments_are_rejected_without_a_gpu():
    from pytorch_empirical_mvm_amd import lib
    l = lib.load()
    d = lib.GemmDesc()
    assert l.vmvm_gemm_bf16(ctypes.byref(d), None) == -1          # VMVM_EINVAL: null operands
    assert l.vmvm_sumsq_f32(None, 10, None, None, 0, None) == -1
    a = lib.AttnFwdDesc()
    assert l.vmvm_attention_fwd(ctypes.byref(a), None) == -1


def test_every_entry_point_validates_on_the_host():
    """tools/cabi_validation.py: null / inconsistent / overflowing arguments into EVERY entry point are refused (or, for the size
    queries, answered) before any HIP call -- the same sweep tests/test_sanitizers_cpu.py runs under ASAN + UBSAN"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("cabi_validation", os.path.join(ROOT, "tools", "cabi_validation.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    bad = m.main(verbose=False)
    assert not bad, bad


def test_workspace_size_queries_without_a_gpu():
    """every op that takes scratch reports how much (pure host arithmetic)."""
    from pytorch_empirical_mvm_amd import lib
    l = lib.load()
    d = lib.GemmDesc()
    d.M, d.N, d.K, d.out_fp32, d.accumulate = 3072, 768, 55296, 1, 1          # a weight gradient of the fusion FFN: splits
    need = l.vmvm_gemm_workspace_size(ctypes.byref(d))
    assert need > 0 and need % (3072 * 768 * 4) == 0 and need // (3072 * 768 * 4) >= 2
    d.accumulate = 0
    assert l.vmvm_gemm_workspace_size(ctypes.byref(d)) == 0                      # not a plain accumulator: never splits
    assert l.vmvm_gemm_workspace_size(None) < 0
    ln = lib.LnBwdDesc()
    ln.M, ln.C = 69120, 768
    assert l.vmvm_layernorm_bwd_workspace_size(ctypes.byref(ln)) == 256 * 4 * 2 * 768 * 4      # one partial row per resident workgroup (4 per CU at C <= 1024)
    a = lib.AttnBwdDesc()
    a.f.nseq, a.f.heads, a.f.L = 160, 12, 432
    assert l.vmvm_attention_bwd_workspace_size(ctypes.byref(a)) == 160 * 12 * 432 * 4
    assert l.vmvm_sumsq_workspace_size(225_000_000) == 2048 * 4


def test_missing_library_fails_loudly(monkeypatch):
    from pytorch_empirical_mvm_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libvmvm.so")
    with pytest.raises(RuntimeError, match="NO CPU"):
        lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pytorch_empirical_mvm_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle|violet_ref", src, flags=re.M), f"{f} imports the oracle"


@pytest.mark.timeout(600)
def test_attention_isa_lint():
    """tools/isa_lint.py on a fresh gfx950 compile of attention.hip (needs hipcc; cross-compiles without a GPU): no DMA-queue drains or
    waterfall loops in the hot loops of the batch-persistent window-attention kernels, no inline-asm VALU reading an MFMA result
    the compiler would have had to wait for, no back-to-back 16x16x16 -> 16x16x32 SrcC chain (hipcc 7.2 hazard)."""
    import shutil
    import subprocess
    import sys
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    root = os.path.join(os.path.dirname(__file__), "..")
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "isa_lint.py")], capture_output=True, text=True, timeout=580)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]


@pytest.mark.timeout(600)
def test_production_sources_carry_no_probe_switches_and_the_probe_hooks_still_build():
    """Source hygiene (round-3 review #14 / round-4 #8): no production translation unit mentions a probe switch -- the kernels call the
    no-op hooks of csrc/hooks/vmvm_probe_hooks.h -- and the instrumented twin under tools/probe/hooks offers the same hooks: the two
    translation units that call them compile against it with every switch on (device-only syntax pass, no GPU needed)."""
    import re
    import shutil
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    csrc = os.path.join(root, "pytorch_empirical_mvm_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            src = open(os.path.join(csrc, f)).read()
            assert not re.search(r"VMVM_PROBE_|VMVM_SCALAR_EPI|W3_ABL|W3_TIMELINE|W3_PINGPONG", src), f"{f} carries a probe switch"
            # round 6 (a name list goes stale): NO conditional compilation at all in a production translation unit --
            # every header is `#pragma once`, gfx950 is the only target, so any #if / #ifdef / #ifndef / #elif is a probe or dual-path switch
            cond = [ln for ln in src.splitlines() if re.match(r"\s*#\s*(if|ifdef|ifndef|elif)\b", ln)]
            assert not cond, f"{f} carries conditional compilation: {cond[:3]}"
    prod, probe = (open(os.path.join(d, "vmvm_probe_hooks.h")).read() for d in (os.path.join(csrc, "hooks"), os.path.join(root, "tools", "probe", "hooks")))
    names = set(re.findall(r"\b(?:struct|void|int)\s+(\w+)", prod)) | set(re.findall(r"constexpr \w+ (\w+)", prod))
    names -= {"init", "stamp", "next_tile", "flush"}
    assert names and all(n in probe for n in names), sorted(n for n in names if n not in probe)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    for unit, defs in (("attention_win3.hip", ["-DW3_TIMELINE"]), ("layernorm.hip", ["-DVMVM_PROBE_BUILD"]), ("attention_win4.hip", ["-DW4_TIMELINE", "-DW4_NO_ODD"])):
        p = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-Wno-pass-failed", "--offload-device-only", "-fsyntax-only", *defs,
                            "-I", os.path.join(root, "tools", "probe", "hooks"), "-I", csrc, os.path.join(csrc, unit)], capture_output=True, text=True, timeout=280)
        assert p.returncode == 0, p.stderr[-3000:]
