#!/bin/bash
# ablation builds of attn_bwd_fused_kernel (-DF_ABL bits: 1 no dQ phase, 2 no barrier, 4 no dS write, 8 no dV/dK products); timing only
for v in ${F_VARIANTS:-0 1 2 3 8 11}; do
  echo "== F_ABL=$v"; VMVM_FUSED_BWD=1 VMVM_LIB=$PWD/tools/scratch/abl/libvmvm_f$v.so VMVM_BENCH_ONLY=none python tools/gpu_check.py benchattn 2>&1 | grep "bert bwd" | head -3
done
