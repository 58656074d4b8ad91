#!/bin/bash
mkdir -p gpurun_out
VMVM_LIB=$PWD/tools/scratch/abl/libvmvm_pipe.so python tools/gpu_check.py attnw 2>&1 | tail -30 > gpurun_out/pipe_check.txt
for r in 1 2; do
for v in base pipe; do
  echo "== $v (round $r)"
  VMVM_LIB=$PWD/tools/scratch/abl/libvmvm_$v.so python tools/gpu_check.py benchattn 2>&1 | grep "win "
done
done > gpurun_out/abl_pipe.txt 2>&1
