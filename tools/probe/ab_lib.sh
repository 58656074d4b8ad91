#!/bin/bash
# A/B of two builds of libvmvm.so on one box with the default bench step: tools/probe/libvmvm_old.so (VMVM_LIB) vs the in-tree library
cd "$(dirname "$0")/../.."
for r in 1 2 3; do
  for w in old new; do
    if [ $w = old ]; then export VMVM_LIB=$PWD/tools/probe/libvmvm_old.so; else unset VMVM_LIB; fi
    echo $w $(python bench.py --no-cpu-baseline "$@" 2>&1 | grep -o "\"ms_per_step\": [0-9.]*\|achieved\": [0-9.]*" | head -2 | tr "\n" " ")
  done
done
