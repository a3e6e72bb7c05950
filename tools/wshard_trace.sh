#!/bin/bash
# development: kernel timeline of the windowed shard's plan + run (tools/wshard_rates.py), per kernel name
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ws_trace
WS_ONLY_G=${1:-8} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ws_trace -- python3 $R/tools/wshard_rates.py ${2:-20} > $R/gpurun_out/ws_trace.log 2>&1
python3 - <<PY
import csv,glob,os
f=max(glob.glob("$R/gpurun_out/ws_trace/*/*_kernel_stats.csv"), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:28]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1000:8.1f} us  total {float(r['TotalDurationNs'])/1000:9.0f} us")
PY
