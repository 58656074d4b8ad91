#!/usr/bin/env python3
"""GPU idle time between kernels in a rocprofv3 --kernel-trace sqlite database: per step-sized window, the time during which no
kernel of the process was executing (launch gaps, host syncs)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
named = list(cur.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
rows = [(a, b) for a, b, _ in named]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy_end, idle, big = rows[0][1], 0, []
for s, e in rows[1:]:
    if s > busy_end:
        idle += s - busy_end
        if s - busy_end > 200_000:
            big.append((s - t0, s - busy_end))
    busy_end = max(busy_end, e)
print(f"span {(t1 - t0) / 1e6:.1f} ms, idle {idle / 1e6:.2f} ms ({100 * idle / (t1 - t0):.1f} %), {len(rows)} dispatches")
# steady state: the last `win` ms of the trace (default 400 = ~3 steps of the C2 bench)
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 400e6
lo = t1 - win
busy_end, idle, hist = None, 0, {}
for s, e in rows:
    if e < lo:
        continue
    if busy_end is None:
        busy_end = e
        continue
    if s > busy_end:
        g = s - busy_end
        idle += g
        b = 10 if g < 10_000 else 20 if g < 20_000 else 50 if g < 50_000 else 200 if g < 200_000 else 1000
        hist[b] = hist.get(b, (0, 0)); hist[b] = (hist[b][0] + 1, hist[b][1] + g)
    busy_end = max(busy_end, e)
print(f"last {win / 1e6:.0f} ms: idle {idle / 1e6:.2f} ms ({100 * idle / win:.1f} %)")
for b in sorted(hist):
    print(f"   gaps < {b:5d} us: {hist[b][0]:6d}  total {hist[b][1] / 1e6:.2f} ms")

# what runs on either side of the large steady-state gaps
busy_end, prev = None, None
for s_, e_, n_ in named:
    if e_ < lo:
        continue
    if busy_end is not None and s_ - busy_end > 500_000:
        print(f"   gap {((s_ - busy_end) / 1e6):.2f} ms  after {prev[:70]}  before {n_[:70]}")
    if busy_end is None or e_ > busy_end:
        busy_end, prev = e_, n_
