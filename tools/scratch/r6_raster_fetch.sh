#!/bin/bash
# Round 6: which panels make FETCH_SIZE of the fc1 GEMM several times its operand bytes?  The 128 x 128 persistent kernel's
# raster walks GM consecutive M panels per N tile (gemm_epi.h raster, GM = 8 in the library).  Same kernel, library builds with GM = 4 / 8 / 16 / 24
# (probe hook VMVM_PROBE_GM), FETCH_SIZE per launch (x 2: gfx950 reports 64-byte units as 32) and the kernel's duration in the same pass.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06/raster_fetch.txt; mkdir -p $R/gpurun_out/r06
: > $O
for g in 4 8 16 24; do
  lib=$R/tools/scratch/abl/libvmvm_gm$g.so; [ $g = 8 ] && lib=$R/pytorch_empirical_mvm_amd/libvmvm.so
  rm -rf /tmp/rf_$g; VMVM_LIB=$lib rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/rf_$g -- python3 $R/tools/pmc_roofline.py > /tmp/rf_$g.log 2>&1
  echo "--- GM = $g M panels per raster group" >> $O
  python3 $R/tools/pmc_summary.py $(find /tmp/rf_$g -name "*.db" | head -1) 2>&1 | grep -A1 "gemm_pers" >> $O
  python3 $R/tools/prof_summary.py $(find /tmp/rf_$g -name "*.db" | head -1) 2>&1 | grep "gemm_pers" | cut -c1-120 >> $O
done
cat $O
