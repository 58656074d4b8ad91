#!/bin/bash
# ablation of the round-3 window attention kernels (libraries built with -DABL_* from a scratch copy of attention.hip) + LDS probe
mkdir -p gpurun_out
./tools/probe/lds_atomic_probe > gpurun_out/lds_atomic_probe.txt 2>&1
for r in 1 2; do
for v in base NODMA NOFETCH NOEXP ALL; do
  echo "== $v (round $r)"
  VMVM_LIB=$PWD/tools/scratch/abl/libvmvm_$v.so python tools/gpu_check.py benchattn 2>&1 | grep "win " 
done
done > gpurun_out/abl_attn.txt 2>&1
