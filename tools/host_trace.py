#!/usr/bin/env python3
"""host-side view of the last `n` ms of a rocprofv3 --kernel-trace --hip-runtime-trace database: per thread, the HIP API calls longer than `min_us`
(and every kernel launch), next to the kernel timeline -- where the enqueueing threads wait"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); span_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0; min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
def tab(prefix):
    c = [t for t in tabs if t.startswith(prefix)]
    return c[0] if c else None
if "--schema" in sys.argv:
    for t in tabs:
        print(t, [r[1] for r in db.execute("pragma table_info(%s)" % t)])
    sys.exit(0)
kd, ks = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol")
kcols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
st, en = ("start", "end") if "start" in kcols else ("start_timestamp", "end_timestamp")
kern = list(db.execute("select s.kernel_name, d.%s, d.%s, d.stream_id from %s d join %s s on d.kernel_id = s.id order by d.%s" % (st, en, kd, ks, st)))
t_end = max(r[2] for r in kern); t0 = t_end - span_ms * 1e6
reg, strs = tab("rocpd_region"), tab("rocpd_string")
rcols = [r[1] for r in db.execute("pragma table_info(%s)" % reg)]
rows = list(db.execute("select s.string, r.start, r.end, r.tid from %s r join %s s on r.name_id = s.id where r.end >= ? and r.start <= ? order by r.start" % (reg, strs), (t0, t_end)))
ev = []
for n, s, e, q in kern:
    if s >= t0:
        ev.append((s, "K", "stream %s" % q, n.split("(")[0].replace("_ZN12_GLOBAL__N_1", "")[:30], (e - s) / 1e3))
for n, s, e, tid in rows:
    if (e - s) / 1e3 >= min_us or "Launch" in n:
        ev.append((s, "H", "tid %s" % tid, n, (e - s) / 1e3))
ev.sort()
base = ev[0][0]
for s, kind, who, name, dur in ev:
    print("%9.1f us  %s %-14s +%8.1f  %s" % ((s - base) / 1e3, kind, who, dur, name))
