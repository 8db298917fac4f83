#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6z_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6z_tests.log
tail -3 gpurun_out/r6z_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
tail -4 gpurun_out/refresh.log | cut -c1-200
