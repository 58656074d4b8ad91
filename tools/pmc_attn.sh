#!/bin/bash
# rocprofv3 counter passes over the window-attention kernels (tools/pmc_kernels.py attn; separate --pmc passes, only --kernel-trace beside them)
# -> gpurun_out/pmc_attn.txt   (PMC_WHICH=bert: the fusion-encoder attention kernels instead)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_attn.txt
: > $OUT
run() {
  local n=$1; shift
  rm -rf /tmp/pa_$n
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pa_$n -- python3 $R/tools/pmc_kernels.py ${PMC_WHICH:-attn} > /tmp/pa_$n.log 2>&1
  echo "--- pass $n: $*" >> $OUT
  python3 $R/tools/pmc_summary.py $(find /tmp/pa_$n -name "*.db" | head -1) >> $OUT 2>&1
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE
run c SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
cat $OUT | cut -c1-260
