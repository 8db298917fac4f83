#!/usr/bin/env python3
"""print the kernel timeline (start offset, duration, queue/stream, name) of the last `n` ms of a rocprofv3 rocpd database"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); span_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
st, en = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)  # HIP stream when recorded, else the hardware queue
sel = "s.kernel_name, d.%s, d.%s, %s" % (st, en, ("d." + qcol) if qcol else "0")
rows = list(db.execute("select %s from %s d join %s s on d.kernel_id = s.id order by d.%s" % (sel, kd, ks, st)))
t_end = max(r[2] for r in rows); t0 = t_end - span_ms * 1e6
rows = [r for r in rows if r[1] >= t0]
base = rows[0][1]
qs = {}
for n, s, e, q in rows:
    qi = qs.setdefault(q, len(qs))
    short = n.split("(")[0].replace("_ZN12_GLOBAL__N_1", "")[:28]
    print("%9.1f us  +%8.1f  q%d %s%s" % ((s - base) / 1e3, (e - s) / 1e3, qi, "    " * qi, short))
