#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace sqlite database: per-kernel total/avg time (the `--stats` view)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
tot = list(cur.execute(f"select sum(end-start)/1e6 from {kd}"))[0][0]
print(f"total kernel time {tot:.2f} ms over {list(cur.execute(f'select count(*) from {kd}'))[0][0]} dispatches")
print(f"{'total_ms':>10s} {'pct':>6s} {'calls':>7s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s}  kernel")
q = f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc limit {int(sys.argv[2]) if len(sys.argv) > 2 else 30}"
for r in cur.execute(q):
    print(f"{r[2]:10.2f} {100*r[2]/tot:5.1f}% {r[1]:7d} {r[3]:10.1f} {r[4]:9.1f} {r[5]:10.1f}  {r[0][:120]}")
