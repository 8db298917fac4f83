#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python tools/exp_m0_leg.py --head-only 2>&1 | tail -2
BCD_HIP_TWO_LANES=0 python tools/exp_m0_leg.py --head-only 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "1080p or low_sample or budget" > gpurun_out/r6x_tests.log 2>&1
echo "rc=$? $(tail -1 gpurun_out/r6x_tests.log)"
