#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for x in 0 8192 33400; do echo "extra LDS $x: $(BCD_HIP_RW_EXTRA_LDS=$x python tools/exp_similarity.py --quick 1920x1080 2>&1 | grep pairdist | tr '\n' ' ')"; done
