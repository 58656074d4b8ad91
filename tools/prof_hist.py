#!/usr/bin/env python3
"""Launch-duration clusters of one kernel in a rocprofv3 --kernel-trace sqlite database: an instantiation that serves several shapes
shows one cluster per shape (count, mean, min, max).  Two launches belong to one cluster when their durations are within 12 %.
  prof_hist.py trace.db gemm_pers_kernelILb1ELb1ELi37E"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
d = sorted((e - s) / 1e3 for s, e, n in cur.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id") if pat in n)
print(f"{len(d)} launches of *{pat}*")
clusters = []
for x in d:
    if clusters and x <= clusters[-1][0] * 1.12:
        clusters[-1].append(x)
    else:
        clusters.append([x])
print(f"{'n':>5} {'mean_us':>9} {'min_us':>9} {'max_us':>9}")
for c in clusters:
    print(f"{len(c):5d} {sum(c) / len(c):9.1f} {c[0]:9.1f} {c[-1]:9.1f}")
