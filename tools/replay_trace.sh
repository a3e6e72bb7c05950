#!/bin/bash
# development: kernel timeline of one REPLAYED epoch of C2 (the plan of the update replayed, only the negatives resolved again)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
rm -rf $R/gpurun_out/replay_trace
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/replay_trace -- python3 $R/tools/profile_stream.py --config C2 --edges -1 --reps 4 --replay > $R/gpurun_out/replay_trace.log 2>&1
tail -4 $R/gpurun_out/replay_trace.log
python3 - <<PY
import csv, glob, os
f = sorted(glob.glob("$R/gpurun_out/replay_trace/*/*_kernel_trace.csv"), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last epoch: from the last k_state_init on
last = max(i for i, r in enumerate(rows) if "k_state_init" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
prev = None
for r in rows[last:]:
    s = int(r["Start_Timestamp"]) - t0; e = int(r["End_Timestamp"]) - t0
    nm = r["Kernel_Name"].replace("void ", "").replace("tpnet::", "").split("(")[0][:40]
    print(f"{s / 1e3:8.1f} {e / 1e3:8.1f} {(e - s) / 1e3:6.1f}  gap {((s - prev) / 1e3) if prev is not None else 0:5.1f}  {nm}")
    prev = e
PY
