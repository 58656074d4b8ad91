#!/usr/bin/env python3
"""Dispatches around the LAST launch of a kernel (name substring) in a rocprofv3 --kernel-trace database: start offset, duration, gap to the previous
dispatch's end, stream, name.   python3 tools/prof_window.py <db> <substring> [before=25] [after=8]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2]; nb = int(sys.argv[3]) if len(sys.argv) > 3 else 25; na = int(sys.argv[4]) if len(sys.argv) > 4 else 8
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
key = "stream_id" if "stream_id" in cols else "queue_id"
rows = list(cur.execute(f"select d.start, d.end, d.{key}, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
idx = [i for i, r in enumerate(rows) if pat in r[3]]
if not idx: sys.exit("no match")
i0 = idx[-1]
t0 = rows[i0][0]
prev_end = {}
for i in range(max(0, i0 - nb), min(len(rows), i0 + na + 1)):
    s, e, k, n = rows[i]
    gap = (s - max(r[1] for r in rows[max(0, i - 40):i])) / 1e3 if i else 0.0
    print(f"{'>>' if i == i0 else '  '} t={(s - t0) / 1e3:10.1f} us  dur={(e - s) / 1e3:8.1f}  idle-before={max(gap, 0.0):7.1f}  {key}={k}  {n[:90]}")
