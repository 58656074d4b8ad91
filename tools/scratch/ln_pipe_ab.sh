#!/bin/bash
python -m pytest tests/test_kernels_gpu.py -x -q -k "check_ln" 2>&1 | tail -2
for v in tools/scratch/abl/ref_head.so ""; do
  if [ -n "$v" ]; then export VMVM_LIB=$PWD/$v; else unset VMVM_LIB; fi
  echo "== lib ${v:-tree}"
  python tools/scratch/ln_cold.py 2>&1 | grep -v amdgpu | grep "bwd"
done
for i in 1 2 3; do
  for v in tools/scratch/abl/ref_head.so ""; do
    if [ -n "$v" ]; then export VMVM_LIB=$PWD/$v; else unset VMVM_LIB; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib=${v:-tree}', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done
