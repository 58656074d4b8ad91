#!/bin/bash
# rocprofv3 kernel trace of the default bench step; writes the per-kernel summary to gpurun_out/
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_step
rocprofv3 --kernel-trace -d /tmp/prof_step -- python3 $GRAFT_REPO_ROOT/bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/prof_step.log 2>&1
tail -1 /tmp/prof_step.log | cut -c1-200
tail -1 /tmp/prof_step.log | python3 -c "import json,sys; o=json.loads(sys.stdin.read()); r=o['roofline']; print('roofline by HIP events:', r['achieved'], r['unit'], 'frac', r['frac'], '=', round(2*69120*3072*768/r['achieved']/1e6,1), 'us')"
DB=$(find /tmp/prof_step -name "*.db" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $DB 80 > $GRAFT_REPO_ROOT/gpurun_out/step_trace.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_shapes.py $DB > $GRAFT_REPO_ROOT/gpurun_out/step_shapes.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_gaps.py $DB > $GRAFT_REPO_ROOT/gpurun_out/step_gaps.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_neighbors.py $DB copyBuffer > $GRAFT_REPO_ROOT/gpurun_out/step_copies.txt 2>&1
# the roofline kernel's instantiation also serves the Swin fc1 shapes: one duration cluster per shape (the fusion FFN fc1 of the VTM pass = the 12 per step around 350-390 us)
python3 $GRAFT_REPO_ROOT/tools/prof_hist.py $DB gemm_pers_kernelILb1ELb1ELi37E > $GRAFT_REPO_ROOT/gpurun_out/step_roofline_kernel_clusters.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_hist.py $DB gemm_pp_kernelILb1ELb1ELi16432E >> $GRAFT_REPO_ROOT/gpurun_out/step_roofline_kernel_clusters.txt 2>&1
head -40 $GRAFT_REPO_ROOT/gpurun_out/step_trace.txt | cut -c1-170
python3 $GRAFT_REPO_ROOT/tools/prof_streams.py $DB 400 > $GRAFT_REPO_ROOT/gpurun_out/step_streams.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_window.py $DB patch_embed_fwd 40 6 > $GRAFT_REPO_ROOT/gpurun_out/step_boundary.txt 2>&1
