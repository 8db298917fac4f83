#!/bin/bash
# HBM traffic of the pair-distance kernel from the TCC counters (GPU box).  Two separate passes (FETCH_SIZE uses 3 of the
# 4 TCC slots, WRITE_SIZE 2), kernel-trace only, as MI355X_MICROARCH.md prescribes.  Output: gpurun_out/pmc_traffic.json
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_$C -o t -- python $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 "$@" > $R/gpurun_out/pmc_$C.log 2>&1
done
python - $R "$@" <<'PY'
import sqlite3, sys, json, glob, os, hashlib
R = sys.argv[1]
SRC_HASH = hashlib.sha256(open(os.path.join(R, "bcd_amd", "csrc", "k_similarity_fast.hip"), "rb").read()).hexdigest()[:16]
W, H = 1920, 1080
for i, a in enumerate(sys.argv):
    if a == "--width": W = int(sys.argv[i + 1])
    if a == "--height": H = int(sys.argv[i + 1])
res = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sqlite3.connect(glob.glob(os.path.join(R, "gpurun_out", "pmc_" + C, "*.db"))[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    pe, pi, kd, ks = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
    q = ("select s.kernel_name, sum(e.value), count(distinct d.id) from %s e join %s i on e.pmc_id = i.id join %s d on e.event_id = d.event_id "
         "join %s s on d.kernel_id = s.id where i.name = '%s' group by s.kernel_name" % (pe, pi, kd, ks, C))
    for name, tot, nd in db.execute(q):
        if "pairdist" in name:
            res[C] = {"sum_kb": tot, "dispatches": nd}
            print(C, name[:50], "sum", tot, "dispatches", nd)
fetch_kb = res["FETCH_SIZE"]["sum_kb"] / res["FETCH_SIZE"]["dispatches"]
write_kb = res["WRITE_SIZE"]["sum_kb"] / res["WRITE_SIZE"]["dispatches"]
# MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide
# (16 B/lane) coalesced streaming read -> doubled.  WRITE_SIZE is uncalibrated, reported as is.
path = os.path.join(R, "gpurun_out", "pmc_traffic.json")
out = json.load(open(path)) if os.path.exists(path) else {}
out["%dx%d_s3" % (W, H)] = {"kernel": "k_pairdist_rw<60>", "fetch_size_kb_per_launch_raw": fetch_kb, "write_size_kb_per_launch_raw": write_kb,
                       "hbm_bytes_per_launch_avg": int((2 * fetch_kb + write_kb) * 1024), "kernel_source_sha256_16": SRC_HASH,
                       "note": "avg over the 3 scale launches of a 3-scale step; read side = 2 x FETCH_SIZE (gfx950 correction), write side WRITE_SIZE as reported"}
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out))
PY
