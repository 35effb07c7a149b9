#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 --pmc results database (rocpd sqlite). usage: pmc_db.py <pmc_results.db> [top]"""
import sqlite3, collections, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda n: [t for t in tabs if t.startswith(n)][0]
pmc = {r[0]: r[1] for r in cur.execute("select id,name from %s" % T('rocpd_info_pmc'))}
ks = {r[0]: r[1] for r in cur.execute("select id,kernel_name from %s" % T('rocpd_info_kernel_symbol'))}
import subprocess
def short(n):
    n = n.replace('.kd', '')
    try:
        n = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], capture_output=True, text=True).stdout.strip() or n
    except Exception:
        pass
    return n.split('(')[0].replace('void ', '').replace('bpg::', '')
names = {k: short(v) for k, v in ks.items()}
disp = {r[3]: (names[r[0]], r[2] - r[1]) for r in cur.execute("select kernel_id,start,end,event_id from %s" % T('rocpd_kernel_dispatch'))}
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float)
for ev, pid, val in cur.execute("select event_id,pmc_id,value from %s" % T('rocpd_pmc_event')):
    if ev in disp: agg[disp[ev][0]][pmc[pid]] += val
for ev, (k, d) in disp.items(): cnt[k] += 1; dur[k] += d
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for k in sorted(agg, key=lambda k: -dur[k])[:top]:
    n = cnt[k]
    print("%-26s n=%-4d avg %.3f ms  " % (k, n, dur[k] / n / 1e6) + "  ".join("%s=%.4g" % (c, v / n) for c, v in sorted(agg[k].items())))
