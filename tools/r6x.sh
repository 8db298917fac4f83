#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
K="processed_set or mono_parity or strip_visiting or multiscale_parity or greedy or band_path_exact or quarter_hd or multi_rank_driver_equals or large_window or config4_chain"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$K" > gpurun_out/r6x_tests.log 2>&1
echo "rc=$? $(tail -1 gpurun_out/r6x_tests.log)"
B12="--no-extras --width 3840 --height 2160 --spp 8 --sigma 0.25 --spikes 0.01 --search-radius 12 --steps 4 --warmup 2"
BCD_HIP_SERIAL_SCALES=1 tools/prof.sh r6x_b12 $B12 > /dev/null
BCD_HIP_SERIAL_SCALES=1 tools/prof.sh r6x_hd --no-extras --steps 10 --warmup 3 > /dev/null
for t in b12 hd; do echo "== $t"; grep -E "mark_round|mark_deps|total kernel" gpurun_out/r6x_${t}_stats.txt | cut -c1-42,76-; grep '^{' gpurun_out/r6x_${t}_bench.log | tail -1 | cut -c60-120; done
python tools/kernel_calls.py $(ls gpurun_out/prof_r6x_b12/*.db | head -1) mark_round 30
python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 10 2>&1 | tail -1 | cut -c60-130
rm -rf gpurun_out/prof_r6x_*
