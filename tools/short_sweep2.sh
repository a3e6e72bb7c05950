#!/bin/bash
# development: heavy-threshold x window-length sweep of short streams + a kernel timeline
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for K in 4 5 8 12; do for H in 12 16 24 32 48; do TPNET_DEV_WIN_HEAVY=$H TPNET_DEV_WINDOW=$K python tools/short_sweep.py 20,40,158 windowed 2>/dev/null | sed "s/^/H=$H /"; done; done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ss_trace
TPNET_DEV_WIN_HEAVY=16 TPNET_DEV_WINDOW=5 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ss_trace -- python3 $R/tools/short_sweep.py 20 windowed > $R/gpurun_out/ss_trace.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/ss_trace/*/*_kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_wwriteback' in r['Kernel_Name']]
a,b=idx[-2]+1,idx[-1]+1
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    n=r['Kernel_Name'].replace('void ','').replace('tpnet::','')
    if 'rocprim' in n: continue
    print(f"{s/1000:9.1f} {e/1000:9.1f} {(e-s)/1000:7.1f}  {n[:40]}  grid {r['Grid_Size_X']}")
PY
