#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
run() { # tag, size b spp sigma spikes, env...
  TAG=$1; ARGS=$2; shift; shift
  (cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/tools/exp_masks.py $ARGS 2 > $R/gpurun_out/${TAG}.log 2>&1)
  python tools/rocpd_stats.py $(ls gpurun_out/prof_$TAG/*.db | head -1) gpurun_out/${TAG}_stats.txt > /dev/null
  echo "== $TAG $(grep checksum gpurun_out/${TAG}.log)"; grep -E "verify" gpurun_out/${TAG}_stats.txt | cut -c1-42,76-
  rm -rf gpurun_out/prof_$TAG
}
for g in 2048 1536 1024 768 512; do run v_$g "3840x2160 12 8 0.25 0.01" BCD_HIP_VERIFY_GRID=$g; done
