#!/bin/bash
# C4: the per-batch kernel's launch geometry knobs (developer build)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export TPNET_DEV_LIB=$R/tpnet_amd/libtpnet_hip_dev.so HIP_FORCE_DEV_KERNARG=1
O=$R/gpurun_out/r05k; mkdir -p $O
run() { echo "== $*"; env "$@" timeout -k 10 280 python tools/profile_stream.py --config C4 --batches 40 --reps 3 2>&1 | grep "rep [12]" ; }
run A=1 || exit 1
run TPNET_DEV_GRID_CAP=768 || exit 1
run TPNET_DEV_GRID_CAP=1536 || exit 1
run TPNET_DEV_GRID_CAP=3072 || exit 1
run TPNET_DEV_GRID_CAP=4096 || exit 1
run TPNET_DEV_ITEMS_FIRST=0 || exit 1
run TPNET_DEV_HEAVY_BLOCKS=64 || exit 1
run TPNET_DEV_HEAVY_BLOCKS=512 || exit 1
