#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export BCD_HIP_SERIAL_SCALES=1
tools/prof.sh r3_tex --no-extras --steps 4 --warmup 2 --pattern 1 > /dev/null 2>&1
echo "== textured (serial scales)"; head -14 gpurun_out/r3_tex_stats.txt | cut -c1-60,76-
tools/prof.sh r3_texl --no-extras --steps 4 --warmup 2 --pattern 1 --sigma 0.1 --spikes 0 > /dev/null 2>&1
echo "== textured low noise (serial scales)"; head -14 gpurun_out/r3_texl_stats.txt | cut -c1-60,76-
unset BCD_HIP_SERIAL_SCALES
tools/prof.sh r3_texc --no-extras --steps 6 --warmup 2 --pattern 1 > /dev/null 2>&1
DB=$(ls gpurun_out/prof_r3_texc/*.db | head -1)
python tools/timeline.py $DB 8.2 | cut -c1-90
