#!/bin/bash
# rocprofv3 kernel trace of bench.py with another MVM target ($1 = vq | 2d_feature | 3d_feature); summary to gpurun_out/
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_t
rocprofv3 --kernel-trace -d /tmp/prof_t -- python3 $GRAFT_REPO_ROOT/bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --mvm-target $1 > /tmp/prof_t.log 2>&1
tail -1 /tmp/prof_t.log | cut -c1-200
DB=$(find /tmp/prof_t -name "*.db" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $DB 40 > $GRAFT_REPO_ROOT/gpurun_out/step_trace_$1.txt 2>&1
head -44 $GRAFT_REPO_ROOT/gpurun_out/step_trace_$1.txt | cut -c1-170
