#!/usr/bin/env python3
"""list individual dispatch durations of kernels matching a substring from a rocprofv3 rocpd database"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
st, en = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
rows = list(db.execute("select s.kernel_name, d.%s, d.%s from %s d join %s s on d.kernel_id = s.id order by d.%s" % (st, en, kd, ks, st)))
out = [(n, (e - s) / 1e3) for n, s, e in rows if pat in n]
print(len(out), "calls;", " ".join("%.0f" % d for _, d in out[-int(sys.argv[3]) if len(sys.argv) > 3 else 0:]))
