#!/bin/bash
# development: k_step on the C4 table with MORE workgroups than the default cap of 2 048 (dev build: make VARIANT=dev VARFLAGS=-DTPNET_DEV)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for envs in "X=0" "TPNET_DEV_GRID_CAP=3072" "TPNET_DEV_GRID_CAP=4096" "TPNET_DEV_GRID_CAP=6144" "TPNET_DEV_GRID_CAP=4096 TPNET_DEV_HEAVY_BLOCKS=32"; do
  rm -rf $R/gpurun_out/c4g
  ( export TPNET_DEV_LIB=$R/tpnet_amd/libtpnet_hip_dev.so HIP_FORCE_DEV_KERNARG=1 $envs; timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c4g -- python3 $R/tools/profile_stream.py --config ${CFG:-C4} --batches 40 --reps 2 > $R/gpurun_out/c4g.log 2>&1 )
  python3 - $R/gpurun_out/c4g "$envs" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_step" in r["Name"]:
            print("%-60s k_step calls=%s avg=%.0f min=%s max=%s" % (sys.argv[2], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
done
