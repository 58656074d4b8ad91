#!/bin/bash
# fc1 class (bias + GELU + 8-bit GELU' code): the epilogue inside the next tile's K loop (il128 = gemm_fc1_il_kernel) against the separate-epilogue
# 128x128 persistent kernel (old128) and the ping-pong kernel, interleaved rounds in one process.   (run on the GPU box)
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -I tools/probe/hooks -I pytorch_empirical_mvm_amd/csrc tools/probe/gemm_probe.hip -o /tmp/gemm_probe_il || exit 1
for set in ${SETS:-fc1 roof swin1}; do
  PROBE_CODE8=1 /tmp/gemm_probe_il $set 10 128 3 2>&1 | grep -A2 "epi=bias+gelu" | grep -v "^--"
done
