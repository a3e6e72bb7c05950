#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export TPNET_DEV_LIB=$R/tpnet_amd/libtpnet_hip_dev.so HIP_FORCE_DEV_KERNARG=1
for H in 96 128 192 256; do echo "C5 B=1000 H=$H: $(TPNET_DEV_WIN_HEAVY=$H python tools/profile_stream.py --config C5 --batch 1000 --batches 600 --reps 3 2>/dev/null | tail -1)"; done
for H in 96 128 192; do echo "C3 B=1000 (d=256) H=$H: $(TPNET_DEV_WIN_HEAVY=$H python tools/profile_stream.py --config C3 --batch 1000 --batches 600 --reps 3 2>/dev/null | tail -1)"; done
for H in 96 128 192; do echo "C2 long H=$H: $(TPNET_DEV_WIN_HEAVY=$H python tools/profile_stream.py --config C2 --batches 1280 --reps 3 2>/dev/null | tail -1)"; done
for H in 24 32 40; do echo "C1 short(160 batches) H=$H: $(TPNET_DEV_WIN_HEAVY=$H python tools/profile_stream.py --config C1 --batches 160 --reps 4 2>/dev/null | tail -1)"; done
