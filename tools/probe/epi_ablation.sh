#!/bin/bash
# Epilogue ablation of the 128x128 persistent kernel at the roofline shapes (fc1 + bias + GELU + 8-bit GELU' code):
# full / no global stores / no epilogue math / neither -- what the epilogue costs, and which half.   (run on the GPU box)
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for e in 0 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -DVMVM_PROBE_EPI=$e -I tools/probe/hooks -I pytorch_empirical_mvm_amd/csrc tools/probe/gemm_probe.hip -o /tmp/gemm_probe_e$e &
done
wait
for rd in 1 2; do
  for e in 0 1 2 3; do
    echo "== VMVM_PROBE_EPI=$e (1 = no stores, 2 = no math) round $rd"
    PROBE_CODE8=1 /tmp/gemm_probe_e$e ${SET:-roof} 10 old128 3 2>&1 | grep -A1 "epi=plain\|epi=bias+gelu\|epi=act3"
  done
done
