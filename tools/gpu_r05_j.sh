#!/bin/bash
# the first timed region of a process against the later ones: kernel timeline of bench.py
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r05j; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin > $O/bench.json 2> $O/bench.err || exit 1
f=$(find $O/trace -name '*kernel_trace.csv' | head -1); [ -n "$f" ] || exit 1
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# regions: each timed call = k_dense_group ... k_wwriteback_dense with 5 k_wpipe in between where grid matches 20 batches
calls = []
cur = None
for r in rows:
    n = r["Kernel_Name"]
    short = n.split("(")[0].split("::")[-1][:24]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "k_dense_group" in n or "k_dense_sort" in n:
        cur = {"k": [], "grid": r.get("Grid_Size_X") or r.get("Grid_Size")}
        calls.append(cur)
    if cur is not None:
        cur["k"].append((short, s, e))
        if "k_wwriteback" in n:
            cur = None
for i, c in enumerate(calls):
    k = c["k"]
    if len([x for x in k if "k_wpipe" in x[0]]) != 5:
        continue
    t0 = k[0][1]
    print(f"call {i} grid {c['grid']}: total {(k[-1][2]-t0)/1e3:.1f} us: " + " ".join(f"{x[0][:10]}@{(x[1]-t0)/1e3:.1f}+{(x[2]-x[1])/1e3:.1f}" for x in k))
PY
