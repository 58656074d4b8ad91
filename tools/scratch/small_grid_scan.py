#!/usr/bin/env python3
"""Long launches on small grids in a prof_shapes.py listing (serial tails): small_grid_scan.py shapes.txt [min_us]"""
import re, sys
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
out = []
for l in open(sys.argv[1]):
    m = re.match(r"\s*([\d.]+) ms\s+[\d.]+% n=\s*(\d+) avg=\s*([\d.]+) us grid=\((\d+),(\d+)\) (\S+)", l)
    if not m:
        continue
    ms, n, avg, gx, gy, name = float(m.group(1)), int(m.group(2)), float(m.group(3)), int(m.group(4)), int(m.group(5)), m.group(6)
    if gx * gy < 256 * 256 * 2 and avg > thr and "gemm_p" not in name and "win2" not in name:
        out.append((ms, n, avg, gx * gy, name[:90]))
out.sort(reverse=True)
for r in out[:20]:
    print("%7.2f ms n=%4d avg=%8.1f us threads=%8d %s" % r)
