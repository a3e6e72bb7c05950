#!/bin/bash
# development: kernel timeline of one batch of the device-resident encoder-level loop of bench.py
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/enc_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/enc_trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/enc_trace.json 2> $R/gpurun_out/enc_trace.err
python3 - <<PY
import csv,glob,os
f=sorted(glob.glob('$R/gpurun_out/enc_trace/*/*_kernel_trace.csv'), key=os.path.getmtime)[-1]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_encoder_sample' in r['Kernel_Name']]
# fp32-class loop comes first, the bf16 loop after: take a batch in the first third
a=idx[len(idx)//4]
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:a+13]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    print(f"{s/1000:8.1f} {e/1000:8.1f} {(e-s)/1000:6.1f} {r['Kernel_Name'].replace('void ','').replace('tpnet::','')[:44]} grid {r['Grid_Size_X']} wg {r['Workgroup_Size_X']}")
PY
