#!/bin/bash
for i in 1 2 3; do
  for v in "" tools/scratch/abl/epi_nt.so; do
    if [ -n "$v" ]; then export VMVM_LIB=$PWD/$v; else unset VMVM_LIB; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib=${v:-tree}', d['value'], 'clips/s', d['ms_per_step'], 'ms', d['roofline']['achieved'])"
  done
done
