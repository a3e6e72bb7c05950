#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export HIP_FORCE_DEV_KERNARG=1
O=$R/gpurun_out/r05e; mkdir -p $O
TPNET_BENCH_FORCE_DIST=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_force.json 2> $O/bench_force.err; echo "force rc $?"; python - <<'PY'
import json
l=json.loads([x for x in open("gpurun_out/r05e/bench_force.json").read().splitlines() if x.startswith("{")][-1])
print("forced one-rank:", l["value"], l["ms_per_step"]*1e3*20, "us for 20 steps", l["roofline"]["kernel_short"], l["roofline"]["frac"], l["timed_regions"])
PY
TPNET_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_n2.json 2> $O/bench_n2.err; echo "n2 rc $?"; wc -l $O/bench_n2.json; tail -3 $O/bench_n2.err; python - <<'PY'
import json
l=json.loads([x for x in open("gpurun_out/r05e/bench_n2.json").read().splitlines() if x.startswith("{")][-1])
print(l["value"], l["n_gpus"], l["ranks_seen"], l["config"].get("schedule"), l["roofline"]["kernel_short"], l["roofline"]["frac"], l["roofline"]["avg_launch_period_us"], l["roofline"]["exchange"])
c=l["c4_rows"]; print("c4", c.get("value"), c.get("error"), c.get("roofline",{}).get("avg_launch_period_us"), c.get("roofline",{}).get("kernel_short"))
print(l["cpu_baseline"]["value"], l["cpu_baseline"]["cores"], l["timed_regions"])
PY
