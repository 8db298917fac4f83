#!/bin/bash
# GPU box: regenerate the round's evidence under gpurun_out/ (copied into profiles/ afterwards by tools/collect_profiles.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tools/prof.sh final_noextras --no-extras --steps 10 --warmup 2 > /dev/null
tools/prof.sh final_default --steps 10 --warmup 2 > /dev/null
BCD_HIP_SERIAL_SCALES=1 tools/prof.sh final_serial --no-extras --steps 10 --warmup 2 > /dev/null
DB=$(ls gpurun_out/prof_final_noextras/*.db | head -1)
python tools/timeline.py $DB 6.2 > gpurun_out/final_timeline.txt
for t in final_noextras final_default final_serial; do grep '^{' gpurun_out/${t}_bench.log | tail -1 > gpurun_out/${t}_line.json; done
python bench.py --steps 30 --warmup 5 > gpurun_out/final_bench_unprofiled.log 2>&1
grep '^{' gpurun_out/final_bench_unprofiled.log | tail -1 > gpurun_out/final_unprofiled_line.json
# BASELINE configs[3] / [4] on one GPU (round 4): per-kernel tables of the 4K frame and of the 4K b = 12 frame
tools/prof.sh final_4k --no-extras --width 3840 --height 2160 --spp 8 --sigma 0.15 --spikes 0 --steps 6 --warmup 2 > /dev/null
tools/prof.sh final_4k_b12 --no-extras --width 3840 --height 2160 --spp 8 --sigma 0.25 --spikes 0.01 --search-radius 12 --steps 4 --warmup 1 > /dev/null
BCD_HIP_SERIAL_SCALES=1 tools/prof.sh final_4k_b12_serial --no-extras --width 3840 --height 2160 --spp 8 --sigma 0.25 --spikes 0.01 --search-radius 12 --steps 4 --warmup 1 > /dev/null
# general sample counts: the headline frame at a uniform 24 spp -- the RATIO form of the distance kernel (round 6; round 5: own-list kernel + pixel-major masks)
tools/prof.sh final_nonuni --no-extras --spp 24 --steps 10 --warmup 2 > /dev/null
BCD_HIP_SERIAL_SCALES=1 tools/prof.sh final_nonuni_serial --no-extras --spp 24 --steps 10 --warmup 2 > /dev/null
# one rank's band of the 4K frame through the band driver with RCCL in loopback (bench.py predicted_8gpu): kernel timeline of one step
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/prof_final_band -o band -- python $R/bench.py --predict-band-child > $R/gpurun_out/final_band.log 2>&1)
python tools/timeline.py $(ls gpurun_out/prof_final_band/*.db | head -1) 3.0 > gpurun_out/final_band_timeline.txt
# ... and where its host threads wait (HIP runtime trace beside the kernel trace; no counters in this run)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --hip-runtime-trace -d $R/gpurun_out/prof_final_band_host -o band -- python $R/bench.py --predict-band-child > /dev/null 2>&1)
python tools/host_trace.py $(ls gpurun_out/prof_final_band_host/*.db | head -1) 3.2 15 | head -400 > gpurun_out/final_band_host_trace.txt
# FETCH_SIZE of every kernel of a step (scales serialised; one counter pass, kernel trace only)
(cd /tmp && export TMPDIR=/tmp && BCD_HIP_SERIAL_SCALES=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch_all -o x -- python $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1)
python tools/fetch_per_kernel.py $(ls gpurun_out/pmc_fetch_all/*.db | head -1) > gpurun_out/final_fetch_per_kernel.txt
# HBM traffic of the pair-distance kernel (TCC counters, separate passes): the headline size and the 4K frame
tools/pmc_traffic.sh > gpurun_out/final_pmc_traffic.log 2>&1
tools/pmc_traffic.sh --width 3840 --height 2160 --spp 8 --sigma 0.15 --spikes 0 >> gpurun_out/final_pmc_traffic.log 2>&1
# SQ counters of the kernels that ship
S1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
S2="SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
S3="SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT"
i=1
for S in "$S1" "$S2" "$S3"; do
  timeout 300 tools/pmc_any.sh final_pd_s$i "$S" pairdist_rw python $R/tools/exp_similarity.py --quick > /dev/null 2>&1
  timeout 300 tools/pmc_any.sh final_eig_s$i "$S" jacobi27 python $R/tools/exp_eig.py 32768 > /dev/null 2>&1
  BCD_HIP_SERIAL_SCALES=1 timeout 300 tools/pmc_any.sh final_est_s$i "$S" bayes,finish27 python $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
  i=$((i+1))
done
python tools/exp_eig.py 32768 > gpurun_out/final_eig.log 2>&1
python tools/exp_eig.py 65536 >> gpurun_out/final_eig.log 2>&1
for cfg in "1280 720" "1920 1080" "3840 2160"; do python tools/exp_host.py $cfg 2>&1 | tail -1; BCD_HIP_STREAM_UPLOADS=0 python tools/exp_host.py $cfg 2>&1 | tail -1; done > gpurun_out/final_host.log
for cfg in "--width 1280 --height 720" "--width 3840 --height 2160 --steps 6" "--width 3840 --height 2160 --search-radius 12 --steps 4"; do
  python bench.py --no-extras --no-cpu-baseline $cfg 2>&1 | tail -1 | cut -c1-260
done > gpurun_out/final_sizes.log
cat gpurun_out/final_host.log gpurun_out/final_sizes.log; tail -3 gpurun_out/final_eig.log
