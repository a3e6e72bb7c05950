#!/bin/bash
# round-4 bench lines (after the per-config PMC files exist: bench.py reads roofline.traffic from them) + the driver's line eight times
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/profile_round_r04.sh 2>&1 | tail -8
for i in 1 2 3 4 5 6 7 8; do timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('driver line run', round(j['value']/1e6,1), 'M edges/s; regions us', [round(x,1) for x in j['timed_regions']['wall_us']], 'kernel', j['roofline']['kernel_short'], round(j['roofline']['frac'],3))"; done
echo done
