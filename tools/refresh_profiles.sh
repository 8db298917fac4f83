#!/bin/bash
# GPU box: regenerate the round's evidence under gpurun_out/ (copied into profiles/ afterwards)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tools/prof.sh final_noextras --no-extras --steps 10 --warmup 2 > /dev/null
tools/prof.sh final_default --steps 10 --warmup 2 > /dev/null
BCD_HIP_SERIAL_SCALES=1 tools/prof.sh final_serial --no-extras --steps 10 --warmup 2 > /dev/null
DB=$(ls gpurun_out/prof_final_noextras/*.db | head -1)
python tools/timeline.py $DB 6.75 > gpurun_out/final_timeline.txt
for t in final_noextras final_default final_serial; do grep '^{' gpurun_out/${t}_bench.log | tail -1 > gpurun_out/${t}_line.json; done
python bench.py --steps 30 --warmup 5 > gpurun_out/final_bench_unprofiled.log 2>&1
grep '^{' gpurun_out/final_bench_unprofiled.log | tail -1 > gpurun_out/final_unprofiled_line.json
python tools/exp_host.py 1280 720 2>&1 | tail -1; python tools/exp_host.py 1920 1080 2>&1 | tail -1
for cfg in "--width 1280 --height 720" "--width 3840 --height 2160 --steps 6" "--width 3840 --height 2160 --search-radius 12 --steps 4"; do
  python bench.py --no-extras --no-cpu-baseline $cfg 2>&1 | tail -1 | cut -c1-260
done
