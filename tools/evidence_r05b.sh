#!/bin/bash
# round-5 evidence, phase 2 (GPU box; after profiles/r05_*_pmc.json exist: bench.py reads roofline.traffic from them): the bench lines
# (driver flags, default, C1 / C3 / C5, the forced one-rank run of the N > 1 path, two gloo ranks on the one GPU), rocprofv3 of the
# driver's command, the driver's line eight times
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_C2_driver.json 2> $O/r05_bench_C2_driver.err
timeout -k 10 400 python bench.py > $O/r05_bench_C2.json 2> $O/r05_bench_C2.err
timeout -k 10 300 python bench.py --config C1 --steps 8000 --warmup 200 --no-cpu-baseline > $O/r05_bench_C1.json 2> $O/r05_bench_C1.err
timeout -k 10 300 python bench.py --config C3 --steps 300 --warmup 20 --no-cpu-baseline > $O/r05_bench_C3.json 2> $O/r05_bench_C3.err
timeout -k 10 300 python bench.py --config C5 --steps 300 --warmup 20 --no-cpu-baseline > $O/r05_bench_C5.json 2> $O/r05_bench_C5.err
TPNET_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/r05_bench_force.json 2> $O/r05_bench_force.err
TPNET_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/r05_bench_gloo2.json 2> $O/r05_bench_gloo2.err
(cd /tmp && export TMPDIR=/tmp && rm -rf $R/$O/r05_bench_prof && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r05_bench_prof -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dropin > $R/$O/r05_bench_C2_driver_under_rocprof.json 2> $R/$O/r05_bench_prof.err)
for f in $O/r05_bench_C2_driver.json $O/r05_bench_C2.json $O/r05_bench_C1.json $O/r05_bench_C3.json $O/r05_bench_C5.json $O/r05_bench_force.json $O/r05_bench_gloo2.json $O/r05_bench_C2_driver_under_rocprof.json; do python - $f <<'PY'
import json, sys
try:
    j = json.loads([x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1]); r = j["roofline"]
    print(sys.argv[1].split("/")[-1], round(j["value"] / 1e6, 2), "M/s n_gpus", j["n_gpus"], r["kernel_short"], "frac", round(r["frac"], 3), "traffic", r.get("traffic"), "stale", (r.get("traffic_source") or {}).get("stale"),
          "period us", round(r["avg_launch_period_us"], 2), "epoch", (j.get("epoch") or {}).get("cold", {}).get("value"), (j.get("epoch") or {}).get("replay", {}).get("value"),
          "long", (j.get("long_stream") or {}).get("value"), "dropin", (j.get("dropin") or {}).get("us_per_batch"), ((j.get("dropin") or {}).get("encoder_level_device") or {}).get("us_per_batch"))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
done
for i in 1 2 3 4 5 6 7 8; do timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('driver line run', round(j['value']/1e6,1), 'M edges/s; regions us', [round(x,1) for x in j['timed_regions']['wall_us']], 'kernel', j['roofline']['kernel_short'], round(j['roofline']['frac'],3), 'first', round(j['timed_regions']['first_region']['value']/1e6,1))"; done
echo done
