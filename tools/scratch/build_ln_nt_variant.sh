#!/bin/bash
# experiment: non-temporal loads (and, with "st", stores) in the LayerNorm kernels -> tools/scratch/abl/ln_nt[_st].so
set -e
root=$(git rev-parse --show-toplevel)
mkdir -p "$root/tools/scratch/abl"
for v in ld st; do
  tmp=$(mktemp -d)
  mkdir -p "$tmp/pytorch_empirical_mvm_amd"; cp -r "$root/include" "$tmp/include"
  cp -r "$root/pytorch_empirical_mvm_amd/csrc" "$tmp/pytorch_empirical_mvm_amd/csrc"
  f="$tmp/pytorch_empirical_mvm_amd/csrc/layernorm.hip"
  python3 - "$f" "$v" <<'PY'
import sys, re
f, v = sys.argv[1], sys.argv[2]
s = open(f).read()
helper = '''
typedef unsigned v4u_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_nt16(const void* p) { const v4u_nt v = __builtin_nontemporal_load(reinterpret_cast<const v4u_nt*>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st_nt16(void* p, const uint4 x) { const v4u_nt v = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(v, reinterpret_cast<v4u_nt*>(p)); }
'''
i = s.index('// 8 consecutive input elements of a row')
s = s[:i] + helper + s[i:]
n0 = s.count('*reinterpret_cast<const uint4*>(')
s = s.replace('*reinterpret_cast<const uint4*>(', 'ld_nt16(')
if v == 'st':
    s = re.sub(r'\*reinterpret_cast<uint4\*>\((.*)\) = pack_bf8\(o\);', r'st_nt16(\1, pack_bf8(o));', s)
print("replaced load sites:", n0 - s.count('*reinterpret_cast<const uint4*>('))
open(f, 'w').write(s)
PY
  ( cd "$tmp/pytorch_empirical_mvm_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -I hooks -c layernorm.hip -o layernorm.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/scratch/abl/ln_nt_$v.so" *.o ) &
done
wait
ls -la "$root/tools/scratch/abl/" | grep ln_nt
