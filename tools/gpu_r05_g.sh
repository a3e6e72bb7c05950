#!/bin/bash
# round 5: phase A of the dense planner without a sort (k_dense_group) against the sorting one (TPNET_DEV_DENSE_SORT=1, dev build)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export HIP_FORCE_DEV_KERNARG=1
O=$R/gpurun_out/r05g; mkdir -p $O
[ -n "$SKIP_TESTS" ] || timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py tests/test_callers.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for i in $BENCH_RUNS; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin > $O/bench_$i.json 2> $O/bench_$i.err || exit 1
done
[ -z "$BENCH_RUNS" ] || python - <<'PY'
import json
for i in (1,2,3,4):
    l=json.load(open(f"gpurun_out/r05g/bench_{i}.json"))
    print("driver", round(l["value"]/1e6,1), "frac", round(l["roofline"]["frac"],3), [round(x,1) for x in l["timed_regions"]["wall_us"]])
PY
cd /tmp && export TMPDIR=/tmp
for mode in group sort; do
  for what in "timed20:--batches 20 --reps 6 --schedule auto" "epoch:--edges -1 --reps 3" "long:--batches 2048 --reps 2"; do
    name=${what%%:*}; args=${what#*:}
    if [ $mode = sort ]; then export TPNET_DEV_LIB=$R/tpnet_amd/libtpnet_hip_dev.so TPNET_DEV_DENSE_SORT=1; else unset TPNET_DEV_LIB TPNET_DEV_DENSE_SORT; fi
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${mode}_$name -- python3 $R/tools/profile_stream.py --config C2 $args > $O/prof_${mode}_$name.log 2>&1 || exit 1
    f=$(find $O/prof_${mode}_$name -name '*kernel_stats.csv' | head -1)
    echo "== $mode $name"; [ -n "$f" ] || exit 1; grep -E "k_dense|k_wpipe|k_wwrite" $f | cut -d, -f1-5 | cut -c1-40,80-200
  done
done
