#!/bin/bash
# Counter passes (separate --pmc runs, only --kernel-trace beside them) + kernel trace of the roofline kernels of bench.py:
# the fusion FFN fc1 GEMM (bias + GELU + saved 8-bit GELU' code, M = 69120, N = 3072, K = 768) and AdamW over a 225 M parameter arena.
#   gpurun -- tools/pmc_roofline.sh      -> gpurun_out/r06_pmc_roofline_gemm.txt  (copy to profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_pmc_roofline_gemm.txt; mkdir -p $R/gpurun_out
: > $O
run() { local n=$1; shift; rm -rf /tmp/pr_$n; rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pr_$n -- python3 $R/tools/pmc_roofline.py > /tmp/pr_$n.log 2>&1
        echo "--- pass $n: $*" >> $O; python3 $R/tools/pmc_summary.py $(find /tmp/pr_$n -name "*.db" | head -1) >> $O 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
rm -rf /tmp/pr_t; rocprofv3 --kernel-trace -d /tmp/pr_t -- python3 $R/tools/pmc_roofline.py > /tmp/pr_t.log 2>&1
echo "--- kernel trace (rocprofv3 --kernel-trace, same command)" >> $O
python3 $R/tools/prof_summary.py $(find /tmp/pr_t -name "*.db" | head -1) | grep -i "gemm\|adamw\|total_ms" >> $O 2>&1
cat $O
