#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export BCD_HIP_SERIAL_SCALES=1
tools/prof.sh r3_tex --no-extras --steps 4 --warmup 2 --pattern 1 > /dev/null 2>&1
echo "== textured (serial scales)"; head -16 gpurun_out/r3_tex_stats.txt | cut -c1-60,76-
tools/prof.sh r3_b12 --no-extras --steps 3 --warmup 1 --width 3840 --height 2160 --search-radius 12 > /dev/null 2>&1
echo "== 4K b12 (serial scales)"; head -16 gpurun_out/r3_b12_stats.txt | cut -c1-60,76-
tools/prof.sh r3_m0 --no-extras --steps 2 --warmup 1 --skip-prob 0 > /dev/null 2>&1
echo "== m0 (serial scales)"; head -12 gpurun_out/r3_m0_stats.txt | cut -c1-60,76-
