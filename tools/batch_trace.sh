#!/bin/bash
# development: kernel timeline of one 20-batch run_stream call on the per-batch schedule
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/bt_trace
HIP_FORCE_DEV_KERNARG=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/bt_trace -- python3 $R/tools/short_sweep.py ${1:-20} batch > $R/gpurun_out/bt_trace.log 2>&1
grep nb= $R/gpurun_out/bt_trace.log
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/bt_trace/*/*_kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_plan_one' in r['Kernel_Name']]
a=idx[-1]
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:a+22]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    print(f"{s/1000:8.1f} {e/1000:8.1f} {(e-s)/1000:6.1f} {r['Kernel_Name'].replace('void ','').replace('tpnet::','')[:30]} grid {r['Grid_Size_X']} wg {r['Workgroup_Size_X']} vgpr {r.get('VGPR_Count','')} lds {r.get('LDS_Block_Size','')}")
PY
HIP_FORCE_DEV_KERNARG=1 python3 $R/tools/short_sweep.py 20,40 batch 2>/dev/null
