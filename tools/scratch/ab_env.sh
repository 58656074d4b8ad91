#!/bin/bash
# A/B of environment switches on one box with the default bench step.  Usage: ab_env.sh "VAR1=a VAR2=b" "VAR1=c" [bench args]
# (each quoted group is one configuration; every configuration runs twice, interleaved)
mkdir -p gpurun_out
A="$1"; B="$2"; shift 2
for r in 1 2; do
  for cfg in "$A" "$B"; do
    env $cfg python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg :', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; host issue', d.get('host_issue_ms_per_step'), 'ms; roofline', d['roofline']['achieved'], 'TF; peak mem', d.get('peak_mem_gib'))"
  done
done 2>&1 | tee -a gpurun_out/ab_env.txt
