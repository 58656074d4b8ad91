#!/bin/bash
# kernel trace of the RCCL world-1 smoke: which RCCL kernels run, and how much of their time other kernels overlap
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 VMVM_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0
rm -rf /tmp/p_rccl; rocprofv3 --kernel-trace -d /tmp/p_rccl -- python3 $R/tools/rccl_smoke.py --steps 2 > /tmp/p_rccl.log 2>&1
tail -2 /tmp/p_rccl.log
python3 $R/tools/prof_overlap.py $(find /tmp/p_rccl -name "*.db" | head -1) nccl rccl > $O/rccl_world1_overlap.txt 2>&1
cat $O/rccl_world1_overlap.txt
