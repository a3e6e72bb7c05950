#!/bin/bash
# development: kernel timeline of the decoder-level per-batch loop
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && python3 tools/dropin_trace.py 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/dropin_trace
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/dropin_trace -- python3 $R/tools/dropin_trace.py > $R/gpurun_out/dropin_trace.log 2>&1
grep rep $R/gpurun_out/dropin_trace.log
python3 - <<PY
import csv,glob,os
f=sorted(glob.glob('$R/gpurun_out/dropin_trace/*/*_kernel_trace.csv'), key=os.path.getmtime)[-1]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_plan_one' in r['Kernel_Name']]
a=idx[-3]
t0=int(rows[a]['Start_Timestamp'])
tail=[r for r in rows if 'k_pair_gram' in r['Kernel_Name']][-6:]+rows[-10:]
for r in rows[a:a+8]+tail:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    print(f"{s/1000:8.1f} {e/1000:8.1f} {(e-s)/1000:6.1f} {r['Kernel_Name'].replace('void ','').replace('tpnet::','')[:40]} grid {r['Grid_Size_X']} wg {r['Workgroup_Size_X']} vgpr {r.get('VGPR_Count','')} agpr {r.get('Accum_VGPR_Count','')} scratch {r.get('Private_Segment_Size', r.get('Scratch_Size',''))} lds {r.get('LDS_Block_Size','')}")
PY
