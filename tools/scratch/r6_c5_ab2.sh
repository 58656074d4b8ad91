#!/bin/bash
# config-5 geometry step with libvmvm variants (VMVM_LIB), interleaved; VMVM_TABLE_SIDE=0
export VMVM_TABLE_SIDE=0
for i in 1 2; do
  for v in "" tools/scratch/abl/dbias_8_128.so tools/scratch/abl/dbias_8_192.so; do
    if [ -n "$v" ]; then export VMVM_LIB=$PWD/$v; else unset VMVM_LIB; fi
    python bench.py --size large --img 384 --frames 16 --batch 8 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib=${v:-tree}', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done
