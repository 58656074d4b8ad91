#!/bin/bash
# experiment: non-temporal 16-byte OUTPUT stores in the GEMM epilogues (bf16 C / C2) -> tools/scratch/abl/gemm_nt.so
set -e
root=$(git rev-parse --show-toplevel)
mkdir -p "$root/tools/scratch/abl"
tmp=$(mktemp -d)
mkdir -p "$tmp/pytorch_empirical_mvm_amd"; cp -r "$root/include" "$tmp/include"
cp -r "$root/pytorch_empirical_mvm_amd/csrc" "$tmp/pytorch_empirical_mvm_amd/csrc"
cd "$tmp/pytorch_empirical_mvm_amd/csrc"
python3 - <<'PY'
import re
n = 0
for f in ("gemm_pp.h", "gemm.hip", "gemm_epi.h"):
    s = open(f).read()
    s, k = re.subn(r'\*reinterpret_cast<uint4\*>\((reinterpret_cast<(?:u16|unsigned char)\*>\(p\.C2?\) \+ [^;=]*)\) = ([^;]*);', r'st_nt16(\1, \2);', s)
    n += k
    open(f, "w").write(s)
print("store sites:", n)
PY
for f in gemm gemm_pp dvae; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -I hooks -c $f.hip -o $f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/scratch/abl/gemm_nt.so" *.o
ls -la "$root/tools/scratch/abl/gemm_nt.so"
