#!/bin/bash
# rocprofv3 kernel trace of the config-5 geometry step (Swin-L, 16 x 384^2, B = 8): per-kernel summary -> gpurun_out/c5_trace.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c5
rocprofv3 --kernel-trace -d /tmp/prof_c5 -- python3 $GRAFT_REPO_ROOT/bench.py --size large --img 384 --frames 16 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/prof_c5.log 2>&1
tail -1 /tmp/prof_c5.log | cut -c1-200
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $(find /tmp/prof_c5 -name "*.db" | head -1) 60 > $GRAFT_REPO_ROOT/gpurun_out/c5_trace.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_shapes.py $(find /tmp/prof_c5 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/c5_shapes.txt 2>&1
head -34 $GRAFT_REPO_ROOT/gpurun_out/c5_trace.txt | cut -c1-170
