#!/bin/bash
# M panels per raster group (probe builds -DVMVM_PROBE_GM=g / -DVMVM_PROBE_GM_PP=g): does a group size whose tile count is not a multiple of the
# XCD's 64 (32) resident workgroups spread the cold-panel switch?   (run on the GPU box)
cd "$(dirname "$0")/../.."
for g in 8 4 6 7 12 16; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -DVMVM_PROBE_GM=$g -DVMVM_PROBE_GM_PP=$g -I tools/probe/hooks -I pytorch_empirical_mvm_amd/csrc tools/probe/gemm_probe.hip -o /tmp/gemm_probe_gm$g &
done
wait
for rd in 1 2; do
  for g in 8 4 6 7 12 16; do
    echo "== group of $g M panels (round $rd)"
    PROBE_CODE8=1 /tmp/gemm_probe_gm$g ${SET:-stag} 10 ${V:-old128} 3 2>&1 | grep -A1 "epi=plain\|epi=bias+gelu" | grep -v "^--"
    PROBE_CODE8=1 /tmp/gemm_probe_gm$g ${SET:-stag} 10 pp2_64x2 3 2>&1 | grep -A1 "epi=plain" | grep -v "^--" | grep pp2
  done
done
