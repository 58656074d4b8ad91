#!/usr/bin/env python3
"""For the dispatches whose kernel name contains PATTERN in a rocprofv3 --kernel-trace sqlite database: what ran just before and
just after each (histogram of neighbour names, steady-state window = the last `win` ms), with the dispatch's duration.
  prof_neighbors.py trace.db copyBuffer [win_ms]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
win = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 400e6
lo = max(r[1] for r in rows) - win
short = lambda n: n.replace("_ZN12_GLOBAL__N_1", "")[:60]
h = collections.defaultdict(lambda: [0, 0.0])
for i, (s, e, n) in enumerate(rows):
    if pat in n and s >= lo:
        k = (short(rows[i - 1][2]) if i else "-", short(rows[i + 1][2]) if i + 1 < len(rows) else "-")
        h[k][0] += 1
        h[k][1] += (e - s) / 1e3
print(f"{'n':>5} {'us':>9}  prev -> next   (dispatches matching '{pat}' in the last {win / 1e6:.0f} ms)")
for k, (c, us) in sorted(h.items(), key=lambda kv: -kv[1][1]):
    print(f"{c:5d} {us:9.1f}  {k[0]}  ->  {k[1]}")
