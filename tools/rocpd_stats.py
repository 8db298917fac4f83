#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel stats table
(calls, total / avg / min / max duration, % of GPU kernel time) -- the same content as `--stats` CSV output."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
    start, end = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    q = ("select s.kernel_name, count(*), sum(d.%s - d.%s), min(d.%s - d.%s), max(d.%s - d.%s) from %s d join %s s on d.kernel_id = s.id "
         "group by s.kernel_name order by 3 desc" % (end, start, end, start, end, start, kd, ks))
    rows = list(db.execute(q))
    total = sum(r[2] for r in rows) or 1
    lines = ["%-72s %8s %12s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
    for name, n, tot, mn, mx in rows:
        short = name if len(name) <= 72 else name[:69] + "..."
        lines.append("%-72s %8d %12.1f %12.2f %12.2f %12.2f %6.2f%%" % (short, n, tot / 1e3, tot / 1e3 / n, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    lines.append("total kernel time: %.3f ms" % (total / 1e6))
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
