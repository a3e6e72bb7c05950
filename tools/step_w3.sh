#!/bin/bash
# development: the 256-thread step kernel on rows of 256 floats at three waves per SIMD (168 VGPRs, the default since round 4) against
# two (173 VGPRs: make VARIANT=w2 VARFLAGS=-DTPNET_MINW_WIDE256=2), C4 and C3: per-batch kernel time from a rocprofv3 kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for cfg in C4 C3; do
for v in base w2; do
  so=$R/tpnet_amd/libtpnet_hip_$v.so; [ "$v" = base ] && so=$R/tpnet_amd/libtpnet_hip.so
  rm -rf $R/gpurun_out/w3v
  ( export TPNET_DEV_LIB=$so HIP_FORCE_DEV_KERNARG=1; timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/w3v -- python3 $R/tools/profile_stream.py --config $cfg --batches 40 --reps 2 > $R/gpurun_out/w3v.log 2>&1 )
  echo "== $cfg $v: $(grep -h 'us/batch' $R/gpurun_out/w3v.log | tail -1)"
  python3 - $R/gpurun_out/w3v <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_step" in r["Name"]:
            print("   k_step calls=%s avg=%.0f min=%s max=%s" % (r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
done; done
