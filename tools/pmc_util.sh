#!/bin/bash
# rocprofv3 derived-metric passes (one metric per pass) for the step kernel of one config.  usage: tools/pmc_util.sh C3 [batches]
CFG=${1:-C3}; NB=${2:-60}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for m in VALUBusy MemUnitStalled VmemLatency MeanOccupancyPerCU SALUBusy; do
  O=$R/gpurun_out/util_${CFG}_$m
  timeout -k 10 200 rocprofv3 --pmc $m --kernel-trace --output-format csv -d $O -- python3 $R/tools/profile_stream.py --config $CFG --batches $NB --reps 1 > $O.log 2>&1 || { echo "$m failed"; tail -3 $O.log; continue; }
  python3 - "$O" "$m" <<'PY'
import csv, glob, sys, statistics as st
f = sorted(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"))
if not f:
    print(sys.argv[2], "no output"); sys.exit(0)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[-1])) if "k_step" in r["Kernel_Name"]]
print(f"{sys.argv[2]:22s} k_step avg {st.mean(v):12.2f}  (n={len(v)})" if v else f"{sys.argv[2]} no k_step rows")
PY
done
