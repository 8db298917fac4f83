#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6z_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6z_tests.log
tail -4 gpurun_out/r6z_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6z_bench.log 2>&1
grep '^{' gpurun_out/r6z_bench.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline', d['value'], d['ms_per_step'], d['roofline']['frac']); print(d['legs'])"
