#!/usr/bin/env python3
"""Per-(kernel, grid) time breakdown of a rocprofv3 kernel-trace database (separates GEMM shapes)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
tot = list(cur.execute(f"select sum(end-start)/1e6 from {kd}"))[0][0]
q = f"select s.kernel_name, d.grid_size_x, d.grid_size_y, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name, d.grid_size_x, d.grid_size_y order by 5 desc limit {int(sys.argv[2]) if len(sys.argv)>2 else 40}"
print(f"total {tot:.1f} ms")
for r in cur.execute(q):
    name = r[0].replace("_ZN12_GLOBAL__N_1", "")[:60]
    print(f"{r[4]:9.2f} ms {100*r[4]/tot:5.1f}% n={r[3]:5d} avg={r[5]:9.1f} us grid=({r[1]},{r[2]}) {name}")
