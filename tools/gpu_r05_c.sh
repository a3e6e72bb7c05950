#!/bin/bash
# round 5: the forced one-rank run again + the 20-batch call per window length / heavy threshold (developer build), with its kernel timeline
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export HIP_FORCE_DEV_KERNARG=1
O=$R/gpurun_out/r05c; mkdir -p $O
TPNET_BENCH_FORCE_DIST=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_force.json 2> $O/bench_force.err; echo "force rc $?"; python - <<'PY'
import json
l=json.loads([x for x in open("gpurun_out/r05c/bench_force.json").read().splitlines() if x.startswith("{")][-1])
print("forced one-rank:", l["value"], l["ms_per_step"]*1e3*20, "us for 20 steps", l["roofline"]["kernel_short"], l["roofline"]["frac"])
PY
python tools/short_sweep.py 20 auto 2>/dev/null
export TPNET_DEV_LIB=$R/tpnet_amd/libtpnet_hip_dev.so
for K in 5 7 10 20; do for H in 0 64 96; do TPNET_DEV_WIN_HEAVY=$H TPNET_DEV_WINDOW_FIXED=$K python tools/short_sweep.py 20 windowed 2>/dev/null | sed "s/^/K=$K H=$H /"; done; done
unset TPNET_DEV_LIB
bash tools/short_trace.sh "0:0:20 7:0:20"
