#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r05l; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_callers.py tests/test_sharded.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for i in 1 2 3 4 5; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin > $O/bench_$i.json 2> $O/bench_$i.err || exit 1
  python - $O/bench_$i.json <<'PY'
import json, sys
l = json.load(open(sys.argv[1])); tr = l["timed_regions"]
print("driver", round(l["value"] / 1e6, 1), "frac", round(l["roofline"]["frac"], 3), [round(x, 1) for x in tr["wall_us"]], "plain", round(tr["plain_call"]["value"] / 1e6, 1), [round(x, 1) for x in tr["plain_call"]["wall_us"]])
PY
done
