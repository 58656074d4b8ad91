#!/bin/bash
# How much of the persistent kernels' time is L2-miss latency of the operand panels?  Same launches with the A and / or B operand made L2-resident
# (row stride 8 elements: the rows overlap, the operand is ~1 MB; results are garbage, timing only).   (run on the GPU box)
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -I tools/probe/hooks -I pytorch_empirical_mvm_amd/csrc tools/probe/gemm_probe.hip -o /tmp/gemm_probe_hot
for v in old128 pp2_64x2; do
  for mode in "" "PROBE_HOT_A=1" "PROBE_HOT_B=1" "PROBE_HOT_A=1 PROBE_HOT_B=1"; do
    echo "== $v  ${mode:-operands from HBM / MALL as in the step}"
    env PROBE_CODE8=1 $mode /tmp/gemm_probe_hot ${SET:-fc1} 10 $v 3 2>&1 | grep -A1 "epi=plain\|epi=bias+gelu" | grep -v "^--"
  done
done
