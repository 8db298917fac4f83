#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export BCD_HIP_GRAB=${BCD_HIP_GRAB:-2}
tools/prof.sh r3_tl --no-extras --steps 10 --warmup 3 > /dev/null 2>&1
DB=$(ls gpurun_out/prof_r3_tl/*.db | head -1)
python tools/timeline.py $DB 7.0 > gpurun_out/r3_tl_timeline.txt
grep '^{' gpurun_out/r3_tl_bench.log | cut -c1-140
cat gpurun_out/r3_tl_timeline.txt | cut -c1-100
