#!/bin/bash
cd "$(dirname "$0")"
cd $GRAFT_REPO_ROOT
for v in "" "-DVMVM_PROBE_WS_HALF" "-DVMVM_PROBE_WS_NOEPI"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed $v -I tools/probe/hooks -I pytorch_empirical_mvm_amd/csrc tools/probe/gemm_probe.hip -o /tmp/gp_ws_$( [ -z "$v" ] && echo full || echo ${v#-DVMVM_PROBE_WS_} ) &
done
wait
for b in full HALF NOEPI; do echo "== $b"; PROBE_CODE8=1 /tmp/gp_ws_$b fc1 10 128 3 2>&1 | grep -A2 "epi=bias+gelu" | grep "ws128\|old128"; done
