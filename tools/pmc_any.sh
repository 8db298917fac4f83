#!/bin/bash
# usage (GPU box): tools/pmc_any.sh <tag> "<counters>" <kernel-name-substring> <command...>
#   -> gpurun_out/pmc_<tag>.txt: per-kernel averages of the counters (rocprofv3 --pmc with --kernel-trace only)
TAG=$1; CTR=$2; PAT=$3; shift; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTR --kernel-trace -d $R/gpurun_out/pmc_$TAG -o $TAG -- "$@" > $R/gpurun_out/pmc_${TAG}.log 2>&1
DB=$(ls $R/gpurun_out/pmc_$TAG/*.db | head -1)
python - "$DB" "$PAT" <<'PY' | tee $R/gpurun_out/pmc_${TAG}.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2].split(",")
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
pe, pi, kd, ks = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
q = ("select s.kernel_name, i.name, count(*), avg(e.value) from %s e join %s i on e.pmc_id = i.id join %s d on e.event_id = d.event_id "
     "join %s s on d.kernel_id = s.id group by s.kernel_name, i.name order by s.kernel_name" % (pe, pi, kd, ks))
for name, ctr, n, v in db.execute(q):
    if any(k in name for k in pat):
        print("%-60s %-28s n=%-4d avg=%.5g" % (name[:60], ctr, n, v))
PY
rm -rf $R/gpurun_out/pmc_$TAG
