#!/bin/bash
# experiment: non-temporal stores of the split-K f32 slabs (gemm_epi.h epi_store8, EF_SPLIT with a workspace) -> tools/scratch/abl/slab_nt.so
set -e
root=$(git rev-parse --show-toplevel)
mkdir -p "$root/tools/scratch/abl"
tmp=$(mktemp -d)
mkdir -p "$tmp/pytorch_empirical_mvm_amd"; cp -r "$root/include" "$tmp/include"
cp -r "$root/pytorch_empirical_mvm_amd/csrc" "$tmp/pytorch_empirical_mvm_amd/csrc"
cd "$tmp/pytorch_empirical_mvm_amd/csrc"
python3 - <<'PY'
s = open("gemm_epi.h").read()
old = '''      float* c = reinterpret_cast<float*>(p.workspace) + ((size_t)e_.slice * e_.M + dst) * e_.N + n;
      *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);'''
new = '''      float* c = reinterpret_cast<float*>(p.workspace) + ((size_t)e_.slice * e_.M + dst) * e_.N + n;
      st_nt16(c, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
      st_nt16(c + 4, make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])));'''
assert old in s
open("gemm_epi.h", "w").write(s.replace(old, new, 1))
PY
for f in gemm gemm_pp dvae; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -I hooks -c $f.hip -o $f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/scratch/abl/slab_nt.so" *.o
ls -la "$root/tools/scratch/abl/slab_nt.so"
