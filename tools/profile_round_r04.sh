#!/bin/bash
# round-4 bench lines + rocprofv3 of bench.py itself (the command the roofline object comes from).  usage (GPU box): tools/profile_round.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_C2_driver.json 2> gpurun_out/r04_bench_C2_driver.err
timeout -k 10 400 python bench.py > gpurun_out/r04_bench_C2.json 2> gpurun_out/r04_bench_C2.err
timeout -k 10 300 python bench.py --config C1 --steps 8000 --warmup 200 --no-cpu-baseline > gpurun_out/r04_bench_C1.json 2> gpurun_out/r04_bench_C1.err
timeout -k 10 300 python bench.py --config C3 --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/r04_bench_C3.json 2> gpurun_out/r04_bench_C3.err
timeout -k 10 300 python bench.py --config C5 --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/r04_bench_C5.json 2> gpurun_out/r04_bench_C5.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r04_bench_prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_bench_prof -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dropin > $R/gpurun_out/r04_bench_C2_driver_under_rocprof.json 2> $R/gpurun_out/r04_bench_prof.err
cd $R; for f in gpurun_out/r04_bench_C2_driver.json gpurun_out/r04_bench_C2.json gpurun_out/r04_bench_C1.json gpurun_out/r04_bench_C3.json gpurun_out/r04_bench_C5.json gpurun_out/r04_bench_C2_driver_under_rocprof.json; do python - $f <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); r = j["roofline"]
print(sys.argv[1].split("/")[-1], round(j["value"] / 1e6, 1), "M/s", r["kernel_short"], "frac", round(r["frac"], 3), "mem", r.get("memory_side_frac"), "period us", round(r["avg_launch_period_us"], 2),
      "epoch", (j.get("epoch") or {}).get("cold", {}).get("value"), (j.get("epoch") or {}).get("replay", {}).get("value"), "long", (j.get("long_stream") or {}).get("value"))
PY
done
