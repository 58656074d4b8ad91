#!/bin/bash
# A/B on one box: win_layout off / on (win3 attention kernels), default bench step
for r in 1 2; do
  for v in 0 1; do
    echo "== VMVM_WIN_LAYOUT=$v (round $r)"
    VMVM_WIN_LAYOUT=$v python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | python3 -c "import json,sys; o=json.loads(sys.stdin.read()); print(o['ms_per_step'], o['value'])"
  done
done
