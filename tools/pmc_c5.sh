#!/bin/bash
# Counter passes (separate --pmc runs, only --kernel-trace beside them) over one config-5 geometry step (Swin-L, 16 x 384^2, B = 8):
# the streaming window-attention kernels -> gpurun_out/r06/pmc_c5_stream.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06/pmc_c5_stream.txt; mkdir -p $R/gpurun_out/r06
: > $O
run() { local n=$1; shift; rm -rf /tmp/pc_$n; rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pc_$n -- python3 $R/bench.py --size large --img 384 --frames 16 --batch 8 --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pc_$n.log 2>&1
        echo "--- pass $n: $*" >> $O; python3 $R/tools/pmc_summary.py $(find /tmp/pc_$n -name "*.db" | head -1) | grep -A1 "stream_kernel" >> $O 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
rm -rf /tmp/pc_t; rocprofv3 --kernel-trace -d /tmp/pc_t -- python3 $R/bench.py --size large --img 384 --frames 16 --batch 8 --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pc_t.log 2>&1
echo "--- kernel trace (same command): per (kernel, grid)" >> $O
python3 $R/tools/prof_shapes.py $(find /tmp/pc_t -name "*.db" | head -1) 200 2>&1 | grep "stream_kernel" >> $O
cat $O | cut -c1-260
