#!/bin/bash
# rocprofv3 kernel trace of the window-attention micro-benchmark (tools/gpu_check.py benchattn): per-kernel durations of the
# win2 / win3 kernels at the stage shapes.  usage: bash tools/prof_attn.sh [cases, e.g. "stage-3,stage-3 unshifted"]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_attn
export VMVM_BENCH_ONLY="${1:-stage-3,stage-3 unshifted}"
rocprofv3 --kernel-trace -d /tmp/prof_attn -- python3 $GRAFT_REPO_ROOT/tools/gpu_check.py benchattn > /tmp/prof_attn.log 2>&1
grep "win " /tmp/prof_attn.log
DB=$(find /tmp/prof_attn -name "*.db" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 $GRAFT_REPO_ROOT/tools/prof_shapes.py $DB > $GRAFT_REPO_ROOT/gpurun_out/attn_shapes.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $DB 40 | grep -i "attn\|total" | cut -c1-200
