mkdir -p gpurun_out/r6a
python -m pytest tests/test_kernels_gpu.py -q -x -k "spike or mask_boundary or nonfinite" -s > gpurun_out/r6a/kern.log 2>&1; echo "kern rc=$?"
python -m pytest tests/test_round6_gpu.py -q -s > gpurun_out/r6a/round6.log 2>&1; echo "round6 rc=$?"
python -m pytest tests/test_round3_gpu.py -q -x -k droppath_dead -s > gpurun_out/r6a/dce.log 2>&1; echo "dce rc=$?"
python tools/scratch/diag_c1_grads.py > gpurun_out/r6a/diag_c1.log 2>&1; echo "diag rc=$?"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r6a/bench.log 2>&1; echo "bench rc=$?"
tail -3 gpurun_out/r6a/kern.log; tail -30 gpurun_out/r6a/round6.log; tail -3 gpurun_out/r6a/dce.log; tail -1 gpurun_out/r6a/bench.log | cut -c1-600
