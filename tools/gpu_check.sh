#!/bin/bash
# GPU box: tests + eigensolver timing + bench for one build.  usage: tools/gpu_check.sh <tag> [quick]
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
if [ "$2" != "quick" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1
  echo "tests rc=$?" >> gpurun_out/${TAG}_tests.log
  tail -4 gpurun_out/${TAG}_tests.log
fi
python tools/exp_eig.py 32768 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/${TAG}_bench.log') if x.startswith('{')]
if not l:
    print(open('gpurun_out/${TAG}_bench.log').read()[-3000:])
else:
    d=json.loads(l[-1])
    print('headline', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'iso', d['roofline'].get('isolated_avg_launch_ms'))
    for k in ('end_to_end','low_noise','textured','textured_low_noise','m0','frame_4k','frame_4k_b12_prefilter'):
        if k in d: print(k, d[k]['value'], d[k].get('ms_per_step', d[k].get('ms_per_frame')))
PY
BCD_HIP_SERIAL_SCALES=1 tools/prof.sh ${TAG}_serial --no-extras --steps 6 --warmup 2 > /dev/null 2>&1
head -12 gpurun_out/${TAG}_serial_stats.txt | cut -c1-60,76-
