#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export HIP_FORCE_DEV_KERNARG=1
O=$R/gpurun_out/r05f; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
TPNET_BENCH_FORCE_DIST=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_force.json 2> $O/bench_force.err; echo "force rc $?"; python - <<'PY'
import json
l=json.loads([x for x in open("gpurun_out/r05f/bench_force.json").read().splitlines() if x.startswith("{")][-1])
print("forced one-rank:", l["value"], l["ms_per_step"]*1e3*20, "us for 20 steps", l["roofline"]["kernel_short"], l["roofline"]["frac"], l["timed_regions"]["wall_us"])
PY
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver rc $?"
python - <<'PY'
import json
l=json.load(open("gpurun_out/r05f/bench_driver.json"))
print("value", l["value"], "frac", l["roofline"]["frac"], "first", l["timed_regions"]["first_region"]["value"])
print("cpu", l["cpu_baseline"]["value"], l["cpu_baseline"]["cores"], l["cpu_baseline"]["value_all_physical_cores"], l["cpu_baseline"]["value_3_threads"])
print("dropin", l["dropin"]["us_per_batch"], l["dropin"]["encoder_level_device"].get("us_per_batch"))
PY
