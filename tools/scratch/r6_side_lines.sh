#!/bin/bash
# round 6: re-measure the side lines the round-5 review called stale (C4 vq, feature targets, config-5 geometry, dVAE tokenizer)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --mvm-target vq 2>/dev/null | tail -1 > $O/bench_vq_n1.json
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --mvm-target 2d_feature 2>/dev/null | tail -1 > $O/bench_2d_feature_n1.json
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --mvm-target 3d_feature 2>/dev/null | tail -1 > $O/bench_3d_feature_n1.json
python bench.py --no-cpu-baseline --steps 5 --warmup 2 --size large --img 384 --frames 16 --batch 8 2>/dev/null | tail -1 > $O/bench_c5_bf16_n1.json
python tools/bench_teacher.py > $O/teacher_bench.txt 2>&1
for f in bench_vq_n1 bench_2d_feature_n1 bench_3d_feature_n1 bench_c5_bf16_n1; do python -c "
import json,sys
d=json.load(open('$O/$f.json')); print('$f', d['value'], d['unit'], d['ms_per_step'], 'ms/step; peak mem', d.get('peak_mem_gib'))"; done
tail -5 $O/teacher_bench.txt
