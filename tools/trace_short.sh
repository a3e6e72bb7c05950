# development: kernel timeline of the driver's short bench run (bench.py --steps 20 --warmup 5)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/r2e_trace
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r2e_trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r2e.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/r2e_trace/*/*_kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
# the timed region: find the last k_batch_desc before the roofline pass... print the window plan + steps sequences
idx=[i for i,r in enumerate(rows) if 'k_batch_desc' in r['Kernel_Name']]
print('k_batch_desc launches:', len(idx))
for start in idx[:3]:
    t0=int(rows[start]['Start_Timestamp'])
    print('--- call starting at row', start)
    for r in rows[start:start+34]:
        s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
        n=r['Kernel_Name'].replace('void ','').replace('tpnet::','')[:46]
        print(f"{s/1000:9.1f} {e/1000:9.1f} {(e-s)/1000:7.1f}  {n}  grid {r['Grid_Size_X']}")
        if "wwriteback" in r["Kernel_Name"]: break
PY
