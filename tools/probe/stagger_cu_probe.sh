#!/bin/bash
# 128x128 persistent kernel: the second workgroup to ARRIVE on a CU (HW_ID, per-CU arrival counter) starts P % of an estimated tile time late,
# so that one workgroup's epilogue runs under the other's main loop (probe build -DVMVM_PROBE_STAGGER_CU=P).   (run on the GPU box)
cd "$(dirname "$0")/../.."
for P in 0 25 50 75; do
  D=""; [ $P -gt 0 ] && D="-DVMVM_PROBE_STAGGER_CU=$P"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed $D -I tools/probe/hooks -I pytorch_empirical_mvm_amd/csrc tools/probe/gemm_probe.hip -o /tmp/gemm_probe_sc$P &
done
wait
for rd in 1 2; do
  for P in 0 25 50 75; do
    echo "== 128x128 kernel, second workgroup of a CU late by $P % of a tile (round $rd)"
    PROBE_CODE8=1 /tmp/gemm_probe_sc$P roof 10 old128 3 2>&1 | grep -A1 "epi=" | grep -v "^--"
  done
done
