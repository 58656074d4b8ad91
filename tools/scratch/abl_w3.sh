#!/bin/bash
# ablation of attn_bwd_dkv_win3_kernel (W3_ABL bitmask builds): which phase content the per-sequence time is made of
# (historical: the W3_ABL / W3_PINGPONG switches were taken out of the kernel after the measurement -- results in
#  profiles/r04_window_attention_dkv_win3_anatomy.txt; the instrumented kernel is in the round-4 history before the hooks commit)
export VMVM_BENCH_ONLY="stage-3 unshifted" VMVM_BENCH_LAYOUTS=1
for v in ${W3_VARIANTS:-0 1 2 4 8 16 32 7 63}; do
  echo "== W3_ABL=$v"; VMVM_LIB=$PWD/tools/scratch/abl/libvmvm_w3_$v.so python tools/gpu_check.py benchattn 2>&1 | grep "no table"
done
