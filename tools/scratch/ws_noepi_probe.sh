#!/bin/bash
# wave-specialised fc1 kernel with the epilogue waves working and idle (-DVMVM_PROBE_WS_NOEPI: the multiplying waves' rate alone)
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -DVMVM_PROBE_WS_NOEPI -I tools/probe/hooks -I pytorch_empirical_mvm_amd/csrc tools/probe/gemm_probe.hip -o /tmp/gemm_probe_wsne || exit 1
for set in fc1 small; do
  PROBE_CODE8=1 timeout 120 /tmp/gemm_probe_wsne $set 10 128 3 2>&1 | grep -A2 "epi=bias+gelu" | grep -v "^--"
done
