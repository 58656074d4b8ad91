#!/bin/bash
# fc1 class (bias + GELU + 8-bit GELU' code): specialised multiply / epilogue waves (ws128 = gemm_fc1_ws_kernel, one 8-wave workgroup per CU) against the
# 4-wave 128x128 persistent kernel with its own epilogue (old128, two workgroups per CU), interleaved rounds in one process.   (run on the GPU box)
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -I tools/probe/hooks -I pytorch_empirical_mvm_amd/csrc tools/probe/gemm_probe.hip -o /tmp/gemm_probe_ws || exit 1
for set in ${SETS:-fc1 roof swin1 small}; do
  PROBE_CODE8=1 timeout 120 /tmp/gemm_probe_ws $set 10 128 3 2>&1 | grep -A2 "epi=bias+gelu" | grep -v "^--"
done
