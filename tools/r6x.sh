#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pyramid or fixtures or pixel_cov or multiscale_parity or mono_parity or band_path_exact or multi_rank_driver or finalize or merge or config4_chain or host_entry" > gpurun_out/r6x_tests.log 2>&1
echo "rc=$? $(tail -1 gpurun_out/r6x_tests.log)"
tools/prof.sh r6x_4k --no-extras --width 3840 --height 2160 --spp 8 --sigma 0.15 --spikes 0 --steps 6 --warmup 2 | cut -c60-130
grep -E "downscale|pixel_cov|interpolate|merge_px|finalize|total kernel" gpurun_out/r6x_4k_stats.txt | cut -c1-50,76-
python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 10 2>&1 | tail -1 | cut -c60-130
python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 10 2>&1 | tail -1 | cut -c60-130
rm -rf gpurun_out/prof_r6x_*
