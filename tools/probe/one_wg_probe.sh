#!/bin/bash
# Is the 128x128 persistent kernel's rate the sum of two independent workgroups (latency-bound each) or one shared MFMA pipe?
# Same kernel with ONE workgroup per CU (VMVM_PROBE_ONE_WG) against the shipped two, with and without the epilogue.   (run on the GPU box)
set -e
cd "$(dirname "$0")/../.."
for e in 0 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -DVMVM_PROBE_EPI=$e -I tools/probe/hooks -I pytorch_empirical_mvm_amd/csrc tools/probe/gemm_probe.hip -o /tmp/gemm_probe_e$e &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -DVMVM_PROBE_EPI=$e -DVMVM_PROBE_ONE_WG -I tools/probe/hooks -I pytorch_empirical_mvm_amd/csrc tools/probe/gemm_probe.hip -o /tmp/gemm_probe_one_e$e &
done
wait
for rd in 1 2; do
  for e in 0 3; do
    echo "== two workgroups per CU, VMVM_PROBE_EPI=$e (3 = no epilogue math, no stores) round $rd"
    PROBE_CODE8=1 /tmp/gemm_probe_e$e ${SET:-roof} 10 old128 3 2>&1 | grep -A1 "epi=plain\|epi=bias+gelu"
    echo "== ONE workgroup per CU, VMVM_PROBE_EPI=$e round $rd"
    PROBE_CODE8=1 /tmp/gemm_probe_one_e$e ${SET:-roof} 10 old128 3 2>&1 | grep -A1 "epi=plain\|epi=bias+gelu"
  done
done
