# development: kernel times of the windowed schedule under a few knob settings (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=${CFG:-C2}
for v in "$@"; do
  ( export $v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2d_v -- python3 $R/tools/profile_stream.py --config $CFG --batches 320 --reps 2 > $R/gpurun_out/r2d.log 2>&1
  echo "$v: $(grep 'rep 1' $R/gpurun_out/r2d.log | cut -d: -f2)"
  python3 - <<PY
import csv,glob,os
f=sorted(glob.glob('$R/gpurun_out/r2d_v/*/*_kernel_stats.csv'),key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)):
    if 'k_w' in r['Name'] or 'finish_w' in r['Name'] or 'edge_refs' in r['Name'] or 'onesweep' in r['Name']:
        print('   ', r['Name'][13:40].split('(')[0], r['Calls'], round(float(r['AverageNs'])/1000,1), int(r['MinNs'])/1000, int(r['MaxNs'])/1000)
PY
  rm -rf $R/gpurun_out/r2d_v )
done
