#!/bin/bash
# libvmvm variants that differ in the workgroup block of attn_bwd_dbias_stream_kernel (NWB waves x KCB keys): tools/scratch/abl/dbias_<NWB>_<KCB>.so
set -e
root=$(git rev-parse --show-toplevel)
mkdir -p "$root/tools/scratch/abl"
for v in "$@"; do
  nw=${v%_*}; kc=${v#*_}
  tmp=$(mktemp -d)
  mkdir -p "$tmp/pytorch_empirical_mvm_amd"; cp -r "$root/include" "$tmp/include"
  cp -r "$root/pytorch_empirical_mvm_amd/csrc" "$tmp/pytorch_empirical_mvm_amd/csrc"; ln -s "$tmp/pytorch_empirical_mvm_amd/csrc" "$tmp/csrc"
  sed -i "s/constexpr int NWB = 4, KCB = 128;/constexpr int NWB = $nw, KCB = $kc;/" "$tmp/csrc/attention.hip"
  grep -q "NWB = $nw, KCB = $kc" "$tmp/csrc/attention.hip"
  ( cd "$tmp/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -I hooks -c attention.hip -o attention.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/scratch/abl/dbias_$v.so" *.o ) &
done
wait
ls -la "$root/tools/scratch/abl/" | grep dbias
