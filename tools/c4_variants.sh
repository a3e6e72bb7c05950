#!/bin/bash
# development: k_step variants on the C4 table (10 M rows, 72 GB): per-batch kernel time from a rocprofv3 kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for v in base nolds lds32 dev:TPNET_DEV_HEAVY_BLOCKS=64 dev:TPNET_DEV_GRID_CAP=4096 dev:TPNET_DEV_NT_STATE=1 dev:TPNET_DEV_NT_STATE=0; do
  lib=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs=""
  so=$R/tpnet_amd/libtpnet_hip_$lib.so; [ "$lib" = base ] && so=$R/tpnet_amd/libtpnet_hip.so
  rm -rf $R/gpurun_out/c4v
  ( export TPNET_DEV_LIB=$so HIP_FORCE_DEV_KERNARG=1; [ -n "$envs" ] && export $envs; timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c4v -- python3 $R/tools/profile_stream.py --config C4 --batches 40 --reps 2 > $R/gpurun_out/c4v.log 2>&1 )
  echo "== $v: $(grep -h 'us/batch' $R/gpurun_out/c4v.log | tail -1)"
  python3 - $R/gpurun_out/c4v <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_step" in r["Name"]:
            print("   k_step calls=%s avg=%.0f min=%s max=%s" % (r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
done
