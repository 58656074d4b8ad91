#!/bin/bash
# Timeline of the two workgroups that share a CU in the 128x128 persistent kernel (probe build -DVMVM_PROBE_TIMELINE): are their epilogues in
# lockstep or do they alternate?   (run on the GPU box)
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -DVMVM_PROBE_TIMELINE -I tools/probe/hooks -I pytorch_empirical_mvm_amd/csrc tools/probe/gemm_probe.hip -o /tmp/gemm_probe_tl
PROBE_CODE8=1 /tmp/gemm_probe_tl ${SET:-fc1} 5 old128 1 2>&1 | grep -v "act3\|bias+resid" | head -${LINES_:-120}
