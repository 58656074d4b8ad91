#!/bin/bash
# rocprofv3 counter passes over the GEMM probe (separate --pmc passes, only --kernel-trace beside them)
# usage: tools/probe/pmc_probe.sh <shape-set> <variant-filter> <tag>      -> gpurun_out/pmc_<tag>.txt
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$3.txt
P="$R/tools/probe/gemm_probe $1 2 $2 1"
: > $OUT
run() {   # name, counters...
  local n=$1; shift
  rm -rf /tmp/pp_$n
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pp_$n -- $P > /tmp/pp_$n.log 2>&1
  echo "--- pass $n: $*" >> $OUT
  python3 $R/tools/pmc_summary.py $(find /tmp/pp_$n -name "*.db" | head -1) >> $OUT 2>&1
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE
run c FETCH_SIZE
run d WRITE_SIZE
run e TCC_HIT_sum TCC_MISS_sum
rm -rf /tmp/pp_t; rocprofv3 --kernel-trace -d /tmp/pp_t -- $P > /tmp/pp_t.log 2>&1
echo "--- kernel trace" >> $OUT
python3 $R/tools/prof_summary.py $(find /tmp/pp_t -name "*.db" | head -1) >> $OUT 2>&1
cat $OUT
