#!/bin/bash
# counter passes over the fusion-encoder attention kernels in two forms: every kernel evaluates Philox / the forward's stored dropout
# decisions  -> gpurun_out/pmc_bert.txt   (round 4 also profiled the one-pass backward, now tools/scratch/attention_fused_experiment.hip)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_bert.txt; : > $O
for cfg in "0 0" "1 0"; do
  set -- $cfg
  export VMVM_PMC_DROPMASK=$1
  echo "##### stored decisions = $1" >> $O
  PMC_WHICH=bert bash $R/tools/pmc_attn.sh > /dev/null 2>&1
  grep -v "^$" $R/gpurun_out/pmc_attn.txt | cut -c1-330 >> $O
done
cat $O | cut -c1-200 | head -80
