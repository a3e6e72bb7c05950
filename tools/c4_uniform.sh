#!/bin/bash
# development: k_step on the C4 table with the config's power-law stream and with uniform ids (no hubs, no repeated rows)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for law in "2.0 3.0" "1.0 1.0" "1.5 2.0"; do
  set -- $law
  rm -rf $R/gpurun_out/c4u
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c4u -- python3 $R/tools/profile_stream.py --config C4 --batches 40 --reps 2 --pu $1 --pi $2 > $R/gpurun_out/c4u.log 2>&1
  python3 - $R/gpurun_out/c4u "pu=$1 pi=$2" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_step" in r["Name"]:
            print("%-20s k_step calls=%s avg=%.0f min=%s max=%s" % (sys.argv[2], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
done
