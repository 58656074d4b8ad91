#!/bin/bash
# build libvmvm from the csrc of a git revision into tools/scratch/abl/<name>.so (for same-box A/B runs through VMVM_LIB)
# usage: tools/scratch/build_ref_lib.sh <rev> <name>
set -e
rev="$1"; name="$2"
root=$(git rev-parse --show-toplevel)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" pytorch_empirical_mvm_amd/csrc include | tar -x -C "$tmp"
mkdir -p "$root/tools/scratch/abl"
cd "$tmp/pytorch_empirical_mvm_amd/csrc"
for f in gemm gemm_pp layernorm attention attention_win3 attention_win4 misc dvae patch_embed blocks; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -I hooks -c $f.hip -o $f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/scratch/abl/$name.so" *.o
rm -rf "$tmp"
echo "$root/tools/scratch/abl/$name.so"
