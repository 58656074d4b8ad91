#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06/pmc_wgrad.txt; mkdir -p $R/gpurun_out/r06; : > $O
run() { local n=$1; shift; rm -rf /tmp/pw_$n; rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pw_$n -- python3 $R/tools/scratch/pmc_wgrad.py > /tmp/pw_$n.log 2>&1
        echo "--- pass $n: $*" >> $O; python3 $R/tools/pmc_summary.py $(find /tmp/pw_$n -name "*.db" | head -1) >> $O 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
rm -rf /tmp/pw_t; rocprofv3 --kernel-trace -d /tmp/pw_t -- python3 $R/tools/scratch/pmc_wgrad.py > /tmp/pw_t.log 2>&1
python3 $R/tools/prof_summary.py $(find /tmp/pw_t -name "*.db" | head -1) | grep -i "gemm\|splitk\|total" >> $O 2>&1
cat $O
