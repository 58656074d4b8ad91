#!/bin/bash
# 128x128 persistent kernel (two workgroups per CU): start phases spread over a tile time (probe build -DVMVM_PROBE_STAGGER=P) vs lockstep
cd "$(dirname "$0")/../.."
for P in 0 2 4 16; do
  D=""; [ $P -gt 0 ] && D="-DVMVM_PROBE_STAGGER=$P"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed $D -I tools/probe/hooks -I pytorch_empirical_mvm_amd/csrc tools/probe/gemm_probe.hip -o /tmp/gemm_probe_s$P &
done
wait
for rd in 1 2; do
  for P in 0 2 4 16; do
    echo "== 128x128 kernel, start phases $P (round $rd)"
    PROBE_CODE8=1 /tmp/gemm_probe_s$P roof 10 old128 3 2>&1 | grep -A1 "epi=" | grep -v "^--"
  done
done
