#!/usr/bin/env python3
"""Per-stream view of a rocprofv3 --kernel-trace database (round 3: the step runs on two streams): busy time of each stream / queue, the
time both are busy, and -- for the last `window_ms` -- the union busy time (what the wall clock sees) against the sum of kernel durations.
    python3 tools/prof_streams.py <db> [window_ms]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
key = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
print("dispatch columns:", cols)
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 400e6
rows = list(cur.execute(f"select start, end, {key or '0'} from {kd} order by start"))
t1 = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t1 - win]
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
by = {}
for s, e, k in rows: by.setdefault(k, []).append((s, e))
span = max(r[1] for r in rows) - min(r[0] for r in rows)
print(f"window {span/1e6:.1f} ms, {len(rows)} dispatches, sum of durations {sum(e-s for s,e,_ in rows)/1e6:.1f} ms, union busy {union([(s,e) for s,e,_ in rows])/1e6:.1f} ms")
for k, iv in sorted(by.items(), key=lambda x: -len(x[1])):
    print(f"  {key}={k}: {len(iv):6d} dispatches, busy (union) {union(iv)/1e6:8.1f} ms, sum {sum(e-s for s,e in iv)/1e6:8.1f} ms")
