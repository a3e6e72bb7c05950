#!/bin/bash
# development: kernel timeline of one short run_stream call per setting (dev build: libtpnet_hip_dev.so).
# usage: tools/short_trace.sh "K:H:nb ..."   (K = 0: the library's own window choice, H = 0: its own threshold)
R=${GRAFT_REPO_ROOT:-/root/repo}
export TPNET_DEV_LIB=$R/tpnet_amd/libtpnet_hip_dev.so
cd /tmp && export TMPDIR=/tmp
for s in $1; do
  K=${s%%:*}; r=${s#*:}; H=${r%%:*}; NB=${r#*:}
  rm -rf $R/gpurun_out/st_trace
  TPNET_DEV_WIN_HEAVY=$H TPNET_DEV_WINDOW_FIXED=$K timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/st_trace -- python3 $R/tools/short_sweep.py $NB ${SCHED:-windowed} > $R/gpurun_out/st_trace.log 2>&1
  echo "== K=$K H=$H nb=$NB: $(grep 'nb=' $R/gpurun_out/st_trace.log)"
  python3 - <<PY
import csv,glob
import os; f=max(glob.glob("$R/gpurun_out/st_trace/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
starts=[i for i,r in enumerate(rows) if 'k_wsort' in r['Kernel_Name'] or 'k_make_keys_w' in r['Kernel_Name'] or 'k_dense_sort<' in r['Kernel_Name']]
a=starts[-1]-1 if ('fillBuffer' in rows[starts[-1]-1]['Kernel_Name'] and 'k_dense_sort<' not in rows[starts[-1]]['Kernel_Name']) else starts[-1]
b=a
while b < len(rows) and not ('k_wpipe' in rows[b]['Kernel_Name']): b+=1
while b < len(rows) and ('k_wpipe' in rows[b]['Kernel_Name'] or 'k_wwriteback' in rows[b]['Kernel_Name']): b+=1
t0=int(rows[a]['Start_Timestamp'])
wp=[]; first=None; plan=[]
for r in rows[a:b]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    n=r['Kernel_Name']
    if 'k_wpipe' in n:
        wp.append((e-s)/1000)
        if first is None: first=s/1000
    elif first is None: plan.append('%s %.1f-%.1f' % (n.replace('void ','').replace('tpnet::','')[:9], s/1000, e/1000))
    last=e/1000
print("   plan: " + '; '.join(plan))
print("   pipeline from %.1f us; k_wpipe x%d: %s  sum %.1f; call ends %.1f" % (first, len(wp), ' '.join('%.0f'%x for x in wp[:12]) + (' ...' if len(wp)>12 else ''), sum(wp), last))
PY
done
