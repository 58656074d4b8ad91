#!/bin/bash
# counter passes over the fusion attention kernels (stored dropout decisions): HBM bytes + SQ activity -> gpurun_out/pmc_f4.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_f4.txt; : > $OUT
export VMVM_PMC_DROPMASK=1
run() { local n=$1; shift; rm -rf /tmp/pf_$n; rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pf_$n -- python3 $R/tools/pmc_kernels.py bert > /tmp/pf_$n.log 2>&1; echo "--- pass $n: $*" >> $OUT; python3 $R/tools/pmc_summary.py $(find /tmp/pf_$n -name "*.db" | head -1) >> $OUT 2>&1; }
run f FETCH_SIZE
run w WRITE_SIZE
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE
run c SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_IFETCH SQ_IFETCH_LEVEL GRBM_GUI_ACTIVE
cat $OUT | cut -c1-300
