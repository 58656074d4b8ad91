#!/bin/bash
# counter passes over the fusion-encoder attention kernels in three forms: every kernel evaluates Philox / the forward's stored dropout
# decisions / the opt-in one-pass backward  -> gpurun_out/pmc_bert.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_bert.txt; : > $O
for cfg in "0 0" "1 0" "1 1"; do
  set -- $cfg
  export VMVM_PMC_DROPMASK=$1
  if [ $2 = 1 ]; then export VMVM_FUSED_BWD=1; else unset VMVM_FUSED_BWD; fi
  echo "##### stored decisions = $1, one-pass backward = $2" >> $O
  PMC_WHICH=bert bash $R/tools/pmc_attn.sh > /dev/null 2>&1
  grep -v "^$" $R/gpurun_out/pmc_attn.txt | cut -c1-330 >> $O
done
cat $O | cut -c1-200 | head -80
