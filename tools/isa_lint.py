#!/usr/bin/env python3
"""ISA lint of attention.hip, attention_win3.hip and attention_win4.hip (compiles them to gfx950 assembly with hipcc --save-temps and inspects the text): properties the
compiler does not guarantee and that were each lost once without any test noticing until a flaky run --

1. the hot loops of the batch-persistent window-attention kernels hold no `s_waitcnt vmcnt(0)` (= the wave drains the next
   sequence's DMA prefetch it has just issued) and no waterfall loop around a buffer instruction;
2. every inline-asm `v_max3_f32` reads registers whose producing MFMA was issued >= MIN_DIST instructions earlier (hipcc does not
   insert MFMA-result wait states in front of inline asm);
3. no `v_mfma_f32_16x16x16_bf16` result is the SrcC of the next-issued MFMA of the other shape (hipcc 7.2 emits no wait states for
   that pair and the hardware does not forward between them: tools/probe/bias_mfma_probe.hip);
4. (round 5) the key-block walks of the win4 kernels hold no scratch traffic: a spill inside a walk is a scratch reload + s_waitcnt
   vmcnt(0) per key block (the masked dK / dV build, opt-in and known to spill, is exempt).

usage: python tools/isa_lint.py [path/to/attention.s]      (exit code 0 = clean)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIN_DIST = 16


def assembly():
    if len(sys.argv) > 1:
        return open(sys.argv[1]).read()
    d = tempfile.mkdtemp()
    csrc = os.path.join(ROOT, "pytorch_empirical_mvm_amd", "csrc")
    out = []
    for name in ("attention", "attention_win3", "attention_win4"):          # the attention translation units (the win_layout = 1 kernels use the same idioms)
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-I", os.path.join(csrc, "hooks"),
                        "-c", os.path.join(csrc, name + ".hip"), "-o", os.path.join(d, name + ".o"), "-save-temps"], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out.append(open(os.path.join(d, name + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read())
    return "\n".join(out)


def kernels(s):
    for m in re.finditer(r"^(_Z\S+):\n", s, re.M):
        j = s.find(".Lfunc_end", m.end())
        ins = [l.strip() for l in s[m.end():j].splitlines()]
        yield m.group(1), [l for l in ins if l and not l.startswith((".", ";")) and not l.endswith(":")]


def main():
    bad = []
    for name, ins in kernels(assembly()):
        if "kernel" not in name:                          # (data symbols, e.g. the W3_PA table)
            continue
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:64]
        mf = [k for k, l in enumerate(ins) if l.startswith("v_mfma")]
        ex = [k for k, l in enumerate(ins) if l.startswith("v_exp_f32")]
        if ("win2_kernel" in name or "win3_kernel" in name) and mf and ex:
            loop = ins[min(mf[0], ex[0]):max(mf[-1], ex[-1]) + 1]
            n_vm = sum(1 for l in loop if l.startswith("s_waitcnt") and "vmcnt(0)" in l)
            n_wf = sum(1 for k, l in enumerate(loop) if l.startswith("v_cmp_eq_u64") and any(x.startswith("s_and_saveexec") for x in loop[k:k + 5]))
            if n_vm or n_wf:
                bad.append(f"{short}: {n_vm} vmcnt(0) drains and {n_wf} waterfall loops inside the hot loop")
        if "win4_kernel" in name and ex and "attn_bwd_dkv_win4_kernelILb1" not in name:
            exs = set(ex)
            for k, l in enumerate(ins):
                if l.startswith("scratch_"):
                    before = any((k - d) in exs for d in range(1, 120))
                    after = any((k + d) in exs for d in range(1, 120))
                    if before and after:
                        bad.append(f"{short}: {l.split()[0]} inside a key-block walk (instruction {k})")
        last = {}
        prev_mfma = None
        for k, l in enumerate(ins):
            m = re.match(r"(v_mfma_\S+) v\[(\d+):(\d+)\], (?:v|a)\[\d+:\d+\], (?:v|a)\[\d+:\d+\], (\S+)", l)
            if m:
                if prev_mfma is not None and prev_mfma[0] != m.group(1) and "16x16x16" in (prev_mfma[0] + m.group(1)):
                    c = re.match(r"v\[(\d+):(\d+)\]", m.group(4))
                    if c and (int(c.group(1)), int(c.group(2))) == prev_mfma[1]:
                        bad.append(f"{short}: {m.group(1)} reads the result of the preceding {prev_mfma[0]} as SrcC (instruction {k})")
                prev_mfma = (m.group(1), (int(m.group(2)), int(m.group(3))))
                for r in range(int(m.group(2)), int(m.group(3)) + 1):
                    last[r] = k
            if l.startswith("v_max3_f32") and "win4_kernel" not in name:      # (the win4 kernels have no inline-asm maxima: theirs come from the compiler, which pads them)
                for r in [int(x) for x in re.findall(r"v(\d+)", l)][1:]:
                    if r in last and k - last[r] < MIN_DIST:
                        bad.append(f"{short}: inline-asm v_max3_f32 reads v{r} {k - last[r]} instructions after the MFMA that writes it")
    for b in bad:
        print("LINT:", b)
    print(f"isa_lint: {len(bad)} finding(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
