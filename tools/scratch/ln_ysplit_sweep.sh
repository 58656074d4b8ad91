cd /tmp && export TMPDIR=/tmp
for y in 8 16 32 64; do
  rm -rf /tmp/ys_$y; VMVM_LN_YSPLIT=$y rocprofv3 --kernel-trace -d /tmp/ys_$y -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/ys_$y.log 2>&1
  echo "ysplit $y: $(python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $(find /tmp/ys_$y -name '*.db' | head -1) 80 | grep ln_colreduce | cut -c1-70)"
done
