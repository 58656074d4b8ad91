#!/bin/bash
# rocprofv3 kernel trace of the config-5 (Swin-L-384, 16 x 384^2, bf16) step; per-kernel summary to gpurun_out/
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c5
rocprofv3 --kernel-trace -d /tmp/prof_c5 -- python3 $GRAFT_REPO_ROOT/tools/c5_run.py ${1:-4} 2 > /tmp/prof_c5.log 2>&1
tail -1 /tmp/prof_c5.log | cut -c1-300
DB=$(find /tmp/prof_c5 -name "*.db" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $DB > $GRAFT_REPO_ROOT/gpurun_out/c5_trace.txt 2>&1
head -40 $GRAFT_REPO_ROOT/gpurun_out/c5_trace.txt | cut -c1-170
