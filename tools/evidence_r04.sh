#!/bin/bash
# round-4 evidence in one GPU call: bench lines (driver flags, default, C1 / C3 / C5), rocprofv3 of the driver's command, kernel
# traces + PMC passes per config, the driver's line ten times.  Outputs under gpurun_out/ (summarised into profiles/ by
# tools/summarize_r04.py and by hand).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/profile_round_r04.sh 2>&1 | tail -8
MODES="epoch long" bash tools/prof_r04.sh C1 2>&1 | grep -E "exit [^0]" 
MODES="batch b1000" NB=60 bash tools/prof_r04.sh C3 2>&1 | grep -E "exit [^0]"
MODES="batch b1000" NB=60 bash tools/prof_r04.sh C5 2>&1 | grep -E "exit [^0]"
MODES="batch" NB=40 bash tools/prof_r04.sh C4 2>&1 | grep -E "exit [^0]"
for i in 1 2 3 4 5 6 7 8; do timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('driver line run', round(j['value']/1e6,1), 'M edges/s; regions us', [round(x,1) for x in j['timed_regions']['wall_us']])"; done
echo done
