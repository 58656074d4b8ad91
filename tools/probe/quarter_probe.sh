#!/bin/bash
# ping-pong kernel on a quarter of the CUs (grid 64) vs the full grid: per-tile time with and without epilogue -- is the store tail a
# per-CU limit (same per-tile time) or a chip-wide one (shorter per-tile time on 64 CUs)?
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -I tools/probe/hooks -I pytorch_empirical_mvm_amd/csrc tools/probe/gemm_probe.hip -o /tmp/gemm_probe || exit 1
PROBE_CODE8=1 /tmp/gemm_probe stag 5 pp2_ 3 2>&1 | grep -v MISMATCHxx
