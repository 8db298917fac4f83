#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof.sh <tag> [bench args...]  -> gpurun_out/<tag>.db + <tag>_stats.txt
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_bench.log 2>&1 || true
DB=$(ls $R/gpurun_out/prof_$TAG/*.db | head -1)
python $R/tools/rocpd_stats.py $DB $R/gpurun_out/${TAG}_stats.txt
grep '^{' $R/gpurun_out/${TAG}_bench.log | tail -1 | cut -c1-300
