cd /tmp && export TMPDIR=/tmp
for cfg in "16 256" "8 512" "4 1024" "2 2048" "4 512"; do
  set -- $cfg
  rm -rf /tmp/cs_$1_$2; VMVM_COLSUM_IT=$1 VMVM_COLSUM_CAP=$2 rocprofv3 --kernel-trace -d /tmp/cs_$1_$2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/cs.log 2>&1
  echo "it $1 cap $2: $(python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $(find /tmp/cs_$1_$2 -name '*.db' | head -1) 80 | grep -E 'colsum_kernel|ln_colreduce' | cut -c1-64 | tr '\n' '|')"
done
