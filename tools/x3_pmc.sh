#!/bin/bash
# development: SQ counters of the dense-layer kernel k_mlp64_x3 on 800 000 rows (one --pmc pass per counter group)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
cat > /tmp/x3_run.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from tpnet_amd import fused_feature as ff
torch.manual_seed(0)
mlp = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 64)).cuda()
x = torch.rand(800000, 64, device="cuda") * 12
with torch.no_grad():
    for _ in range(6): ff.mlp_f32(mlp, x)
torch.cuda.synchronize()
PY
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  O=$R/gpurun_out/x3_pmc; rm -rf $O
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O -- python3 /tmp/x3_run.py > $R/gpurun_out/x3_pmc.log 2>&1 || { echo "$grp: failed"; tail -3 $R/gpurun_out/x3_pmc.log; continue; }
  python3 - <<PY
import csv,glob,os,collections
f=sorted(glob.glob('$R/gpurun_out/x3_pmc/*/*_counter_collection.csv'), key=os.path.getmtime)[-1]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_mlp64_x3' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v[1:])/max(1,len(v)-1)) for k,v in acc.items()})
PY
done
