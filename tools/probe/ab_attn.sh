#!/bin/bash
# A/B of two builds of the attention kernels on one box: tools/probe/libvmvm_old.so (previous commit) vs the in-tree library
cp pytorch_empirical_mvm_amd/libvmvm.so /tmp/new.so
for r in 1 2; do
  cp tools/probe/libvmvm_old.so pytorch_empirical_mvm_amd/libvmvm.so; echo "== old (round $r)"; python tools/gpu_check.py benchattn 2>&1 | grep "win\|bert"
  cp /tmp/new.so pytorch_empirical_mvm_amd/libvmvm.so; echo "== new (round $r)"; python tools/gpu_check.py benchattn 2>&1 | grep "win\|bert"
done
