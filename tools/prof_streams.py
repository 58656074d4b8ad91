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

# per stream: the kernels by total time, and for each how much of its run time another stream was busy too (a main-stream kernel that shares the
# chip with a side-stream kernel runs longer than alone: "shared" is the part of its duration to look at)
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows2 = list(cur.execute(f"select d.start, d.end, d.{key or 'queue_id'}, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id where d.start >= {t1 - win} order by d.start"))
import bisect
def merged(iv):
    iv = sorted(iv); out = []; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: out.append((cs, ce)); cs, ce = s, e
        else: ce = max(ce, e)
    out.append((cs, ce)); return out
def overlap(m, s, e):
    i = bisect.bisect_left(m, (s, s)) - 1; tot = 0
    for a, b in m[max(i, 0):]:
        if a >= e: break
        tot += max(0, min(b, e) - max(a, s))
    return tot
for k in by:
    others = [(s, e) for s, e, kk, _ in rows2 if kk != k]
    m = merged(others) if others else []
    agg = {}
    for s, e, kk, name in rows2:
        if kk != k: continue
        a = agg.setdefault(name, [0, 0, 0]); a[0] += 1; a[1] += e - s; a[2] += overlap(m, s, e) if m else 0
    tot = sum(a[1] for a in agg.values())
    print(f"\n{key}={k}: {tot/1e6:.1f} ms of kernels, {sum(a[2] for a in agg.values())/1e6:.1f} ms of it with another stream busy")
    print(f"{'total_ms':>10s} {'shared_ms':>10s} {'calls':>6s} {'avg_us':>9s}  kernel")
    for name, a in sorted(agg.items(), key=lambda x: -x[1][1])[:28]:
        print(f"{a[1]/1e6:10.2f} {a[2]/1e6:10.2f} {a[0]:6d} {a[1]/a[0]/1e3:9.1f}  {name[:110]}")
