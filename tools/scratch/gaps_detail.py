#!/usr/bin/env python3
"""Every idle gap > 100 us in the last `win` ms of a rocprofv3 kernel trace, with the kernels before / after it (steady-state steps)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
t1 = max(r[1] for r in rows); lo = t1 - (float(sys.argv[2]) if len(sys.argv) > 2 else 400) * 1e6
busy_end, last = None, None
for s, e, n in rows:
    if e < lo: continue
    if busy_end is not None and s - busy_end > 100_000:
        print(f"gap {(s - busy_end) / 1e3:7.1f} us at t-{(t1 - s) / 1e6:6.1f} ms   after {last[:70]}   before {n[:70]}")
    if busy_end is None or e > busy_end: busy_end, last = e, n
