#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace sqlite database: the kernels whose name contains one of the given substrings (e.g. nccl), and
for each of them how much of its [start, end] interval is covered by OTHER kernels running at the same time (side-stream overlap)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pats = [p.lower() for p in sys.argv[2:]] or ["nccl"]
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
sel = [r for r in rows if any(p in r[0].lower() for p in pats)]
print(f"{len(rows)} dispatches, {len(sel)} match {pats}")
if not sel:
    print("no matching kernel was launched (at world size 1 RCCL's all_reduce / broadcast return without a device kernel for in-place buffers)")
    names = sorted({r[0][:80] for r in rows if "gemm" not in r[0] and "attn" not in r[0] and "ln_" not in r[0]})
    print("other kernels seen:", "; ".join(names[:40]))
    sys.exit(0)
tot = cov = 0
for name, a, e in sel:
    c = 0
    for n2, a2, e2 in rows:
        if n2 is name and a2 == a:
            continue
        lo, hi = max(a, a2), min(e, e2)
        if hi > lo and not any(p in n2.lower() for p in pats):
            c += hi - lo
    tot += e - a; cov += min(c, e - a)
    print(f"{(e - a) / 1e3:10.1f} us  overlapped {100.0 * min(c, e - a) / max(e - a, 1):5.1f} %  {name[:100]}")
print(f"total {tot / 1e3:.1f} us, overlapped by other kernels {100.0 * cov / max(tot, 1):.1f} %")
