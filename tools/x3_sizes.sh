#!/bin/bash
# development: kernel durations of k_mlp64_x3 by list length (rocprofv3 kernel trace of tools/mlp_f32_rate.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/x3_sizes
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/x3_sizes -- python3 $R/tools/mlp_f32_rate.py > $R/gpurun_out/x3_sizes.log 2>&1
python3 - <<PY
import csv,glob,os,collections,statistics as st
f=sorted(glob.glob('$R/gpurun_out/x3_sizes/*/*_kernel_trace.csv'), key=os.path.getmtime)[-1]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_mlp64_x3' in r['Kernel_Name'] or 'k_pair_feature_bf16' in r['Kernel_Name']:
        acc[(r['Kernel_Name'][:40], r['Grid_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
for k,v in acc.items(): print(k, 'n=%d' % len(v), 'median %.2f us' % st.median(v))
PY
