#!/bin/bash
# compact multi-config bench (GPU box): tools/bench_sweep.sh [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "--sigma 0.35 --spikes 0.01" "--sigma 0.15 --spikes 0.0" "--sigma 0.08 --spikes 0.0"; do
  python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline $cfg "$@" 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s0 = d['config']['per_scale'][0]
print('$cfg', '| ms/step', d['ms_per_step'], '| Mpix/s', d['value'], '| pairdist avg ms', d['roofline']['avg_launch_ms'], '| s0 proc', s0['processed_frac'], 'fb', s0['fallback_frac'], 'meanS', s0['mean_similar'], 'rounds', [x['rounds'] for x in d['config']['per_scale']])
"
done
