#!/bin/bash
# development: batches per window of the pipeline (TPNET_DEV_WINDOW, dev build) on the C2 epoch (cold / replayed) and a 2 048-batch stream
R=${GRAFT_REPO_ROOT:-/root/repo}
export TPNET_DEV_LIB=$R/tpnet_amd/libtpnet_hip_dev.so HIP_FORCE_DEV_KERNARG=1
for w in ${WINS:-24 27 32 40 53 64}; do
  e=$(TPNET_DEV_WINDOW=$w timeout -k 10 120 python3 $R/tools/profile_stream.py --config C2 --edges -1 --reps 6 2>/dev/null | grep rep | tail -3 | awk '{print $6}' | tr '\n' ' ')
  r=$(TPNET_DEV_WINDOW=$w timeout -k 10 120 python3 $R/tools/profile_stream.py --config C2 --edges -1 --reps 6 --replay 2>/dev/null | grep rep | tail -3 | awk '{print $6}' | tr '\n' ' ')
  l=$(TPNET_DEV_WINDOW=$w timeout -k 10 120 python3 $R/tools/profile_stream.py --config C2 --batches 2048 --reps 4 2>/dev/null | grep rep | tail -2 | awk '{print $6}' | tr '\n' ' ')
  echo "window $w batches: epoch cold M edges/s [$e] replayed [$r] long stream [$l]"
done
