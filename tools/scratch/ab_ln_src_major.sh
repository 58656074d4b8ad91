#!/bin/bash
# LayerNorm backward of the window-gathered norm1 (Video-Swin stage 1-2): output-major walk (VMVM_LN_SRC_MAJOR=0) vs source-major
mkdir -p gpurun_out
{
python tools/gpu_check.py lng 2>&1 | tail -12
for v in 0 1; do
  echo "=== VMVM_LN_SRC_MAJOR=$v"
  export VMVM_LN_SRC_MAJOR=$v
  bash tools/prof_step.sh > /dev/null 2>&1
  grep "ln_bwd_pk\|copy_batches\|invert_map" gpurun_out/step_trace.txt | cut -c1-150
  tail -1 /tmp/prof_step.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step (under the profiler)', d['ms_per_step'])"
done
unset VMVM_LN_SRC_MAJOR
bash tools/scratch/ab_env.sh "VMVM_LN_SRC_MAJOR=0" "VMVM_LN_SRC_MAJOR=1" --steps 12 --warmup 4
} 2>&1 | tee gpurun_out/ab_ln_src_major.txt
