#!/bin/bash
# rocprofv3 kernel-trace of developer variants: tools/variants.sh "TAG ENV=.. ENV=.." ...   (PSARGS = profile_stream args)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for spec in "$@"; do
  set -- $spec; tag=$1; shift
  ( for kv in "$@"; do export "$kv"; done; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/var_$tag -- python3 $R/tools/profile_stream.py ${PSARGS} > $R/gpurun_out/var_$tag.log 2>&1 )
  echo "== $tag ($*)"; grep -h "us/batch" $R/gpurun_out/var_$tag.log | tail -1
  python3 - $R/gpurun_out/var_$tag <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_step" in r["Name"] or "k_finish" in r["Name"]:
            print("   %-10s calls=%s avg=%.0f min=%s max=%s pct=%s" % (r["Name"].split("(")[0][-12:], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
PY
done
