#!/bin/bash
# end-of-round-5 measurement bundle -> gpurun_out/r05/ (copied into profiles/ by hand)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
./tools/probe/valu4_probe > $O/probe_valu_saturated.txt 2>&1
VMVM_BENCH_LAYOUTS=1 python tools/gpu_check.py benchattn 2>&1 | grep "win \|bert" > $O/attention_microbench.txt
VMVM_BENCH_LAYOUTS=1 VMVM_BENCH_WINDOW_ONLY=1 VMVM_NO_WIN4=1 python tools/gpu_check.py benchattn 2>&1 | grep "win " > $O/attention_microbench_win3_kernels.txt
VMVM_BENCH_LAYOUTS=1 bash tools/prof_attn.sh "stage-1,stage-2,stage-3,stage-3 unshifted,stage-4" > $O/window_attention_kernel_times.txt 2>&1
VMVM_PMC_SHIFTED=0 bash tools/pmc_attn.sh > /dev/null 2>&1; cp gpurun_out/pmc_attn.txt $O/pmc_window_attention_unshifted.txt
VMVM_PMC_SHIFTED=1 bash tools/pmc_attn.sh > /dev/null 2>&1; cp gpurun_out/pmc_attn.txt $O/pmc_window_attention_shifted.txt
bash tools/prof_step.sh > $O/prof_step.log 2>&1
for f in step_trace step_shapes step_gaps step_streams step_roofline_kernel_clusters step_copies; do cp gpurun_out/$f.txt $O/$f.txt; done
bash tools/pmc_roofline.sh > /dev/null 2>&1; cp gpurun_out/r05_pmc_roofline_gemm.txt $O/pmc_roofline_gemm.txt
bash tools/scratch/ab_env.sh "VMVM_NO_WIN4=1" "VMVM_X=0" --steps 10 --warmup 3 > $O/ab_win4_bench.txt 2>&1
python tools/gemm_shapes.py 32 > $O/gemm_shapes_single_stream.txt 2>&1
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
tail -1 $O/bench_n1.json | cut -c1-300
