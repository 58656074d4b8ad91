#!/bin/bash
# end-of-round evidence pass (round 3): default bench line, step kernel trace (+ shapes, gaps, roofline-kernel clusters), counter passes of the roofline kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python3 $R/bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/r03_bench_n1.json
bash $R/tools/prof_step.sh > /dev/null 2>&1
cp $O/step_trace.txt $O/r03_step_kernel_trace_b32.txt; cp $O/step_shapes.txt $O/r03_step_kernel_shapes_b32.txt; cp $O/step_gaps.txt $O/r03_step_idle_gaps.txt
cp $O/step_roofline_kernel_clusters.txt $O/r03_step_roofline_kernel_clusters.txt
bash $R/tools/pmc_roofline.sh > /dev/null 2>&1
cut -c1-1200 $O/r03_bench_n1.json; head -12 $O/r03_step_kernel_trace_b32.txt | cut -c1-150; cat $O/r03_step_roofline_kernel_clusters.txt | head -20
# what bounds the 128x128 persistent kernel: one workgroup per CU vs two, the two workgroups' timeline on a CU, operands forced L2-resident
bash $R/tools/probe/one_wg_probe.sh > $O/r03_gemm_probe_one_wg_per_cu.txt 2>&1
LINES_=64 bash $R/tools/probe/timeline_probe.sh > $O/r03_gemm_probe_cu_timeline.txt 2>&1
bash $R/tools/probe/hot_operand_probe.sh > $O/r03_gemm_probe_hot_operands.txt 2>&1
bash $R/tools/probe/stagger_cu_probe.sh > $O/r03_gemm_probe_stagger_per_cu.txt 2>&1
