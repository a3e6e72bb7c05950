#!/bin/bash
# development: median duration of the fused pair-feature kernel in the decoder-level loop for diagnostic-library settings
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export TPNET_DEV_LIB=$R/tpnet_amd/libtpnet_hip_dev.so
for v in "${@}"; do
  rm -rf $R/gpurun_out/fv_trace
  env $v timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/fv_trace -- python3 $R/tools/dropin_trace.py > $R/gpurun_out/fv_trace.log 2>&1 || { echo "$v: failed"; tail -3 $R/gpurun_out/fv_trace.log; continue; }
  python3 - "$v" <<PY
import csv,glob,os,sys,statistics as st
f=sorted(glob.glob('$R/gpurun_out/fv_trace/*/*_kernel_trace.csv'), key=os.path.getmtime)[-1]
rows=list(csv.DictReader(open(f)))
out=[]
for name in ('k_pair_feature<','k_pair_feature_bf16','k_plan_one','k_step'):
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000 for r in rows if name in r['Kernel_Name']]
    r0=[r for r in rows if name in r['Kernel_Name']]
    if d: out.append(f"{name} n={len(d)} median {st.median(d):.2f} us grid {r0[0]['Grid_Size_X']} vgpr {r0[0].get('VGPR_Count','')}")
print(sys.argv[1], '|', ' | '.join(out))
PY
done
