#!/bin/bash
# end-of-round-4 measurement bundle -> gpurun_out/r04/ (copied into profiles/ by hand)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
./tools/probe/valu_probe > $O/probe_valu_issue.txt 2>&1
./tools/probe/lds_atomic_probe > $O/probe_lds.txt 2>&1
python tools/gpu_check.py benchattn 2>&1 | grep "win \|bert" > $O/window_attention_bench.txt
bash tools/prof_attn.sh "stage-1,stage-2,stage-3,stage-3 unshifted,stage-4" > $O/window_attention_kernel_times.txt 2>&1
VMVM_PMC_SHIFTED=0 bash tools/pmc_attn.sh > /dev/null 2>&1; cp gpurun_out/pmc_attn.txt $O/pmc_window_attention_unshifted.txt
VMVM_PMC_SHIFTED=1 bash tools/pmc_attn.sh > /dev/null 2>&1; cp gpurun_out/pmc_attn.txt $O/pmc_window_attention_shifted.txt
bash tools/prof_step.sh > $O/prof_step.log 2>&1
for f in step_trace step_shapes step_gaps step_streams step_roofline_kernel_clusters; do cp gpurun_out/$f.txt $O/$f.txt; done
bash tools/scratch/ab_win3.sh > $O/ab_win_layout.txt 2>&1
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
tail -1 $O/bench_n1.json | cut -c1-300
