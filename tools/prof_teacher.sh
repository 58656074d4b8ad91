#!/bin/bash
# rocprofv3 kernel trace of the dVAE tokenizer bench (native path only): per-kernel summary -> gpurun_out/teacher_trace.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt; TEACHER_NATIVE_ONLY=1 rocprofv3 --kernel-trace -d /tmp/pt -- python3 $GRAFT_REPO_ROOT/tools/bench_teacher.py > /tmp/pt.log 2>&1
tail -3 /tmp/pt.log
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $(find /tmp/pt -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/teacher_trace.txt
head -40 $GRAFT_REPO_ROOT/gpurun_out/teacher_trace.txt | cut -c1-200
