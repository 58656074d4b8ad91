#!/bin/bash
# experiment: non-temporal loads of the GEMM epilogues' residual / saved-activation operands (read once) -> tools/scratch/abl/epi_nt.so
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
    s, k = re.subn(r'\*reinterpret_cast<const uint4\*>\((reinterpret_cast<const (?:u16|unsigned char)\*>\(p\.(?:resid|aux)\) \+ [^;]*)\);', r'ld_nt16(\1);', s)
    n += k
    open(f, "w").write(s)
print("load sites:", n)
PY
for f in gemm gemm_pp dvae; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -I hooks -c $f.hip -o $f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/scratch/abl/epi_nt.so" *.o
ls -la "$root/tools/scratch/abl/epi_nt.so"
