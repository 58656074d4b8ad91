#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc results (sqlite): mean counter value per kernel name."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
def t(prefix): return [x for x in tabs if x.startswith(prefix)][0]
pmc, pinfo, kd, ks = t('rocpd_pmc_event'), t('rocpd_info_pmc'), t('rocpd_kernel_dispatch'), t('rocpd_info_kernel_symbol')
q = f"""select s.kernel_name, d.grid_size_x, i.name, avg(e.value), count(*) from {pmc} e join {pinfo} i on e.pmc_id=i.id
        join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by s.kernel_name, d.grid_size_x, i.name"""
res = collections.defaultdict(dict)
for name, grid, cname, val, n in cur.execute(q):
    res[(name.replace("_ZN12_GLOBAL__N_1", "")[:58], grid)][cname] = val
for k, v in res.items():
    if not any(x in k[0] for x in ("gemm", "attn", "ln_", "adamw", "sumsq")): continue
    print(k[0], "grid", k[1])
    print("    " + "  ".join(f"{c}={val:.3g}" for c, val in sorted(v.items())))
