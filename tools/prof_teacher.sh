#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt; rocprofv3 --kernel-trace -d /tmp/pt -- python3 $GRAFT_REPO_ROOT/tools/bench_teacher.py > /tmp/pt.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $(find /tmp/pt -name "*.db" | head -1) | head -24 | cut -c1-170
