#!/bin/bash
# A/B of the fusion-encoder attention kernels (L = 432 = 27 tiles): waves per workgroup -- 8 waves walk the 27 owner tiles in 4 passes
# (the last one with 3 of 8 waves busy), 9 waves in 3 full passes (VMVM_BERT_BWD_NW / VMVM_BERT_FWD_NW = 8 | 9); round-4 also tried two
# key tiles per wave in dK/dV (VMVM_BERT_DKV_KT = 2 | 3) and two query tiles per wave in dQ (VMVM_BERT_DQ_QT = 2): no gain.
mkdir -p gpurun_out
for cfg in "8 8" "9 8" "8 9" "9 9" "8 8" "9 9"; do
  set -- $cfg
  echo "=== BWD_NW=$1 FWD_NW=$2"
  VMVM_BERT_BWD_NW=$1 VMVM_BERT_FWD_NW=$2 python tools/gpu_check.py attnb 2>&1 | tail -1
  VMVM_BERT_BWD_NW=$1 VMVM_BERT_FWD_NW=$2 VMVM_BENCH_ONLY=none python tools/gpu_check.py benchattn 2>&1 | grep "bert [fb]wd" | head -4
done 2>&1 | tee gpurun_out/ab_bert_attn.txt
