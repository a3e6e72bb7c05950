#!/bin/bash
# one GPU cycle: parity tests, bench line, rocprofv3 kernel trace of a shorter bench run.  usage: tools/gpu_cycle.sh TAG
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/test_$TAG.log 2>&1; echo "pytest exit $?" >> gpurun_out/test_$TAG.log; tail -4 gpurun_out/test_$TAG.log
timeout -k 10 300 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench exit $?"; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --no-cpu-baseline --steps 300 --warmup 20 > $R/gpurun_out/prof_$TAG.log 2>&1; echo "prof exit $?"
head -6 $R/gpurun_out/prof_$TAG/*/*_kernel_stats.csv | cut -c1-150
