#!/bin/bash
# config-5 geometry: table gradient on the second stream (VMVM_TABLE_SIDE) A/B, interleaved
mkdir -p gpurun_out/r06
python -m pytest tests/test_round6_gpu.py -x -q -s -k "streaming_window_table" 2>&1 | tail -8
for i in 1 2; do
  for v in 1 0; do
    VMVM_TABLE_SIDE=$v python bench.py --size large --img 384 --frames 16 --batch 8 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('table_side=$v', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done
