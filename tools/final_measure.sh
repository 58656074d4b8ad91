#!/bin/bash
# final evidence pass: default bench line, step kernel trace, PMC traffic of the roofline kernel (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python3 $R/bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_n1.json
rm -rf /tmp/p1; rocprofv3 --kernel-trace -d /tmp/p1 -- python3 $R/bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/p1.log 2>&1
python3 $R/tools/prof_summary.py $(find /tmp/p1 -name "*.db" | head -1) > $O/step_trace.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c; rocprofv3 --pmc $c --kernel-trace -d /tmp/p_$c -- python3 $R/tools/pmc_roofline.py > /tmp/p_$c.log 2>&1
  python3 $R/tools/pmc_summary.py $(find /tmp/p_$c -name "*.db" | head -1) >> $O/pmc_roofline.txt 2>&1
done
rm -rf /tmp/p_sq; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d /tmp/p_sq -- python3 $R/tools/pmc_roofline.py > /tmp/p_sq.log 2>&1
echo "--- SQ counters (separate pass)" >> $O/pmc_roofline.txt
python3 $R/tools/pmc_summary.py $(find /tmp/p_sq -name "*.db" | head -1) >> $O/pmc_roofline.txt 2>&1
# rocprofv3 --stats style duration of the roofline kernel under the same command
rm -rf /tmp/p_rk; rocprofv3 --kernel-trace -d /tmp/p_rk -- python3 $R/tools/pmc_roofline.py > /tmp/p_rk.log 2>&1
python3 $R/tools/prof_summary.py $(find /tmp/p_rk -name "*.db" | head -1) | grep -i gemm > $O/roofline_kernel_trace.txt 2>&1
cat $O/bench_n1.json | cut -c1-1500; cat $O/pmc_roofline.txt; cat $O/roofline_kernel_trace.txt
