#!/bin/bash
cp pytorch_empirical_mvm_amd/libvmvm.so /tmp/new.so
for r in 1 2; do
  cp tools/probe/libvmvm_old.so pytorch_empirical_mvm_amd/libvmvm.so; echo "== old (round $r)"; python tools/gpu_check.py benchln 2>&1 | grep -i "ln \|layernorm\|bwd\|fwd"
  cp /tmp/new.so pytorch_empirical_mvm_amd/libvmvm.so; echo "== new (round $r)"; python tools/gpu_check.py benchln 2>&1 | grep -i "ln \|layernorm\|bwd\|fwd"
done
