#!/bin/bash
# usage: gpu_retry.sh <timeout> '<command>'  -- retries while the pod's slots are busy (exit 3 / transient)
T=$1; shift
for i in $(seq 1 12); do
  out=$(timeout $((T+1500)) gpurun --timeout $T -- "$@" 2>&1)
  echo "$out" | tail -60
  if echo "$out" | grep -q "status=transient"; then sleep 120; continue; fi
  break
done
