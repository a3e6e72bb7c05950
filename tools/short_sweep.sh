#!/bin/bash
# development: window-length sweep of short streams + a kernel timeline.  usage (GPU box): tools/short_sweep.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python tools/short_sweep.py 20,40,158 batch
for K in 2 3 4 5 8 12 24; do TPNET_DEV_WINDOW=$K python tools/short_sweep.py 20,40,158 windowed; done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ss_trace
TPNET_DEV_WINDOW=5 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ss_trace -- python3 $R/tools/short_sweep.py 20 windowed > $R/gpurun_out/ss_trace.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/ss_trace/*/*_kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_wwriteback' in r['Kernel_Name']]
# the last timed call: kernels between the second-to-last and the last write-back
a,b=idx[-2]+1,idx[-1]+1
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    n=r['Kernel_Name'].replace('void ','').replace('tpnet::','')
    n=('rocprim:'+('onesweep_iter' if 'onesweep_iteration' in n else ('hist' if 'global_offsets' in n else n[60:100]))) if 'rocprim' in n else n[:40]
    print(f"{s/1000:9.1f} {e/1000:9.1f} {(e-s)/1000:7.1f}  {n}  grid {r['Grid_Size_X']}")
PY
