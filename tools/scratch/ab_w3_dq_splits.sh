#!/bin/bash
# dQ kernel of the win_layout = 1 window attention: 2 key parts (8 waves, 2 per SIMD) vs 3 key parts (12 waves, 3 per SIMD, bf16 partials)
mkdir -p gpurun_out
{
for v in 2 3 2 3; do
  echo "=== VMVM_W3_DQ_SPLITS=$v"
  [ $v = 3 ] && VMVM_W3_DQ_SPLITS=$v python tools/gpu_check.py attnw 2>&1 | grep "layout=1\|FAILED" | grep "dq\|dtable\|FAILED"
  VMVM_W3_DQ_SPLITS=$v VMVM_BENCH_LAYOUTS=1 python tools/gpu_check.py benchattn 2>&1 | grep "win bwd" | head -10
done
} 2>&1 | tee gpurun_out/ab_w3_dq_splits.txt
