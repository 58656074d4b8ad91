#!/bin/bash
for v in "" tools/scratch/abl/dbias_8_128.so tools/scratch/abl/dbias_8_192.so tools/scratch/abl/dbias_4_192.so; do
  if [ -n "$v" ]; then export VMVM_LIB=$PWD/$v; fi
  python tools/scratch/dbias_bench.py 2>&1 | grep -v amdgpu.ids
done
