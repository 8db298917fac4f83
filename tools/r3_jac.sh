#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in 0 1 2 3 0 1 2 3; do echo "variant $v: $(BCD_JAC_VARIANT=$v python tools/exp_eig.py 65536 2>&1 | grep matrices)"; done
S2="SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_WAVES"
for v in 0 1 2 3; do
  BCD_JAC_VARIANT=$v timeout 300 tools/pmc_any.sh r3_jv$v "$S2" jacobi27 python $R/tools/exp_eig.py 32768 > /dev/null 2>&1
  echo "== variant $v"; cut -c60- gpurun_out/pmc_r3_jv$v.txt
done
