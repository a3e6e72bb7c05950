#!/bin/bash
# round 5, first GPU cycle: the suite, the driver's line, the forced one-rank run of the N > 1 path, two gloo ranks on the one GPU
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export HIP_FORCE_DEV_KERNARG=1
O=$R/gpurun_out/r05a; mkdir -p $O
timeout -k 10 420 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver rc $?"
python - <<'PY'
import json
l=json.load(open("gpurun_out/r05a/bench_driver.json"))
print("value", l["value"], "frac", l["roofline"]["frac"], "traffic", l["roofline"]["traffic"], l["roofline"]["traffic_source"])
print("first", l["timed_regions"]["first_region"]["value"], "cpu", l["cpu_baseline"]["value"], l["cpu_baseline"]["cores"], l["cpu_baseline"].get("value_16_threads"), l["cpu_baseline"].get("value_3_threads"))
print("dropin", l["dropin"]["us_per_batch"], l["dropin"]["encoder_level_device"].get("us_per_batch"))
PY
TPNET_BENCH_FORCE_DIST=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/bench_force.json 2> $O/bench_force.err; echo "force rc $?"; cut -c1-600 $O/bench_force.json
TPNET_BENCH_BACKEND=gloo TPNET_BENCH_C4_BATCH=2000 timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_n2.json 2> $O/bench_n2.err; echo "n2 rc $?"; cut -c1-900 $O/bench_n2.json; tail -5 $O/bench_n2.err
