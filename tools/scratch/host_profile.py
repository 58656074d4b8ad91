"""cProfile of the host side of the bench step (where do the ~60 ms of Python per step go?).  Usage: python tools/scratch/host_profile.py"""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "6", "--warmup", "2"]
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(70)
print(s.getvalue()[:14000])
