#!/usr/bin/env python3
"""per kernel: FETCH_SIZE (TCC, KB) summed over the instances of a dispatch, the three largest dispatches, with their durations.
usage: tools/fetch_per_kernel.py <rocprofv3 --pmc FETCH_SIZE --kernel-trace database>"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
pe, pi, kd, ks = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
st, en = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
q = ("select s.kernel_name, d.id, sum(e.value), d.%s - d.%s from %s e join %s i on e.pmc_id = i.id join %s d on e.event_id = d.event_id "
     "join %s s on d.kernel_id = s.id where i.name = 'FETCH_SIZE' group by d.id order by d.id" % (en, st, pe, pi, kd, ks))
agg = collections.defaultdict(list)
for name, did, kb, dur in db.execute(q):
    agg[name.split("(")[0].replace("_ZN12_GLOBAL__N_1", "")[:44]].append((kb, dur / 1e3))
print("# FETCH_SIZE per dispatch (raw counter, MB; MI355X_MICROARCH.md: wide coalesced streaming reads are under-reported by 2x on gfx950 -- the x2 column),")
print("# the three largest dispatches of every kernel; bench.py --no-extras, scales serialised (BCD_HIP_SERIAL_SCALES=1)")
for k, v in sorted(agg.items(), key=lambda kv: -max(x[0] for x in kv[1])):
    top = sorted(v, key=lambda x: -x[0])[:3]
    print("%-46s n=%3d  %s" % (k, len(v), " | ".join("%6.0f MB (x2 %6.0f) in %6.0f us" % (kb / 1024, 2 * kb / 1024, d) for kb, d in top)))
