#!/bin/bash
# forward window attention on the region-major layout: attn_fwd_win3_kernel vs the win2 kernel on the same layout (VMVM_NO_WIN3_FWD=1)
mkdir -p gpurun_out
{
python tools/gpu_check.py attnw 2>&1 | tail -25
for v in 1 0; do
  echo "=== VMVM_NO_WIN3_FWD=$v"
  if [ $v = 1 ]; then export VMVM_NO_WIN3_FWD=1; else unset VMVM_NO_WIN3_FWD; fi
  VMVM_BENCH_LAYOUTS=1 python tools/gpu_check.py benchattn 2>&1 | grep "win fwd\|fwd:" | head -12
done
} 2>&1 | tee gpurun_out/ab_win3_fwd.txt
