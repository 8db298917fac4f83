#!/bin/bash
# GPU box, round 3: counter refresh of the kernels that ship (rocprofv3 --pmc passes with --kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
S1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
S2="SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
S3="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
timeout 300 tools/pmc_any.sh r3_pd_s1 "$S1" pairdist_rw python $R/tools/exp_similarity.py --quick > /dev/null 2>&1
timeout 300 tools/pmc_any.sh r3_pd_s2 "$S2" pairdist_rw python $R/tools/exp_similarity.py --quick > /dev/null 2>&1
timeout 300 tools/pmc_any.sh r3_eig_s1 "$S1" jacobi27 python $R/tools/exp_eig.py 32768 > /dev/null 2>&1
timeout 300 tools/pmc_any.sh r3_eig_s2 "$S2" jacobi27 python $R/tools/exp_eig.py 32768 > /dev/null 2>&1
timeout 300 tools/pmc_any.sh r3_eig_s3 "$S3" jacobi27 python $R/tools/exp_eig.py 32768 > /dev/null 2>&1
export BCD_HIP_SERIAL_SCALES=1
timeout 300 tools/pmc_any.sh r3_bay_s1 "$S1" bayes python $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
timeout 300 tools/pmc_any.sh r3_bay_s2 "$S2" bayes python $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
timeout 300 tools/pmc_any.sh r3_bay_s3 "$S3" bayes python $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
unset BCD_HIP_SERIAL_SCALES
for f in gpurun_out/pmc_r3_*.txt; do echo "== $f"; cat $f; done
tail -3 gpurun_out/pmc_r3_eig_s3.log
