#!/bin/bash
# the fused step's variant: A/B of a kernel change against the previous build (tpnet_amd/libtpnet_hip_old.so), same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r05m; mkdir -p $O
for lib in new old; do
  if [ $lib = old ]; then export TPNET_DEV_LIB=$R/tpnet_amd/libtpnet_hip_old.so; else unset TPNET_DEV_LIB; fi
  for c in C4 C3 C5; do
    nb=40; [ $c = C4 ] || nb=60
    timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${lib}_$c -- python3 $R/tools/profile_stream.py --config $c --batches $nb --reps 2 > $O/${lib}_$c.log 2>&1 || exit 1
    f=$(find $O/${lib}_$c -name '*kernel_stats.csv' | head -1); [ -n "$f" ] || exit 1
    python3 - "$f" $lib $c <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_step" in r["Name"]:
        print(sys.argv[2], sys.argv[3], "k_step calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 2), "min", round(float(r["MinNs"]) / 1e3, 2))
PY
  done
done
