#!/bin/bash
# A/B of two builds of libvmvm.so on one box with the default bench step: tools/probe/libvmvm_old.so vs the in-tree library
cp pytorch_empirical_mvm_amd/libvmvm.so /tmp/new.so
for r in 1 2; do
  for which in old new; do
    if [ $which = old ]; then cp tools/probe/libvmvm_old.so pytorch_empirical_mvm_amd/libvmvm.so; else cp /tmp/new.so pytorch_empirical_mvm_amd/libvmvm.so; fi
    VMVM_NO_BUILD=1 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  done
done
cp /tmp/new.so pytorch_empirical_mvm_amd/libvmvm.so
