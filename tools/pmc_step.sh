#!/bin/bash
# one --pmc pass (SQ wait / busy / LDS counters) over a short run of the default bench step: per-kernel totals -> gpurun_out/pmc_step.txt
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ps
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/ps -- python3 $R/bench.py --batch 32 --steps 1 --warmup 1 --no-cpu-baseline > /tmp/ps.log 2>&1
python3 - <<PY > $R/gpurun_out/pmc_step.txt
import sqlite3, collections, glob, re
db = sqlite3.connect(glob.glob('/tmp/ps/**/*.db', recursive=True)[0]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
def t(p): return [x for x in tabs if x.startswith(p)][0]
pmc, pinfo, kd, ks = t('rocpd_pmc_event'), t('rocpd_info_pmc'), t('rocpd_kernel_dispatch'), t('rocpd_info_kernel_symbol')
q = f"select s.kernel_name, i.name, sum(e.value), count(*) from {pmc} e join {pinfo} i on e.pmc_id=i.id join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by s.kernel_name, i.name"
res = collections.defaultdict(dict)
for name, cname, val, n in cur.execute(q):
    k = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', name)[:70]
    res[k][cname] = val; res[k]['n'] = n
rows = sorted(res.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))
print("%-72s %6s %12s %7s %7s %7s %7s %7s" % ("kernel", "calls", "wave_cycles", "wait%", "winst%", "valu%", "lds%", "ldsconf%"))
for k, v in rows[:45]:
    w = v.get('SQ_WAVE_CYCLES', 0) or 1
    li = v.get('SQ_LDS_IDX_ACTIVE', 0) or 1
    print("%-72s %6d %12.3g %7.1f %7.1f %7.1f %7.1f %7.1f" % (k, v['n'], w, 100 * v.get('SQ_WAIT_ANY', 0) / w, 100 * v.get('SQ_WAIT_INST_ANY', 0) / w,
          100 * v.get('SQ_ACTIVE_INST_VALU', 0) / w, 100 * v.get('SQ_ACTIVE_INST_LDS', 0) / w, 100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / li))
PY
head -50 $R/gpurun_out/pmc_step.txt | cut -c1-150
