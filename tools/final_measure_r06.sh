#!/bin/bash
# end-of-round-6 measurement bundle -> gpurun_out/r06/ (copied into profiles/r06_* by hand)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
bash tools/prof_step.sh > $O/prof_step.log 2>&1
for f in step_trace step_shapes step_gaps step_streams step_roofline_kernel_clusters step_copies step_boundary; do cp gpurun_out/$f.txt $O/$f.txt; done
bash tools/pmc_roofline.sh > /dev/null 2>&1; cp gpurun_out/r06_pmc_roofline_gemm.txt $O/pmc_roofline_gemm.txt
python tools/gemm_shapes.py 32 > $O/gemm_shapes_single_stream.txt 2>&1
python tools/count_calls.py 2>&1 | grep -v amdgpu.ids > $O/foreign_calls_per_step.txt
VMVM_BLOCK_ABI=0 python tools/count_calls.py 2>&1 | grep -v amdgpu.ids | head -12 >> $O/foreign_calls_per_step.txt
bash tools/scratch/ab_env.sh "VMVM_BLOCK_ABI=1" "VMVM_BLOCK_ABI=0" --steps 20 --warmup 5 > $O/ab_block_abi.txt 2>&1
python bench.py --mvm-target vq --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_vq_n1.json 2>/dev/null
python bench.py --mvm-target 2d_feature --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_2d_feature_n1.json 2>/dev/null
python bench.py --mvm-target 3d_feature --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_3d_feature_n1.json 2>/dev/null
for i in 1 2; do
  python bench.py --size large --img 384 --frames 16 --batch 8 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c5_bf16_n1.json 2>/dev/null
  python bench.py --size large --img 384 --frames 16 --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --fp8 > $O/bench_c5_fp8_n1.json 2>/dev/null
  for f in bf16 fp8; do tail -1 $O/bench_c5_${f}_n1.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config-5 geometry $f:', d['value'], 'clips/s', d['ms_per_step'], 'ms/step')"; done
done > $O/c5_bf16_vs_fp8.txt 2>&1
python tools/bench_teacher.py > $O/teacher_bench.txt 2>&1
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
tail -1 $O/bench_n1.json | cut -c1-400
