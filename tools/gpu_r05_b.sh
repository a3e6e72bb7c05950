#!/bin/bash
# round 5, second GPU cycle: the sharded tests, the forced one-rank run of the N > 1 path, two gloo ranks on the one GPU (C4 leg at full batch)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export HIP_FORCE_DEV_KERNARG=1
O=$R/gpurun_out/r05b; mkdir -p $O
timeout -k 10 420 python -m pytest tests/test_sharded.py tests/test_abi.py tests/test_bench_launcher.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
TPNET_BENCH_FORCE_DIST=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/bench_force.json 2> $O/bench_force.err; echo "force rc $?"; cut -c1-300 $O/bench_force.json; tail -3 $O/bench_force.err
TPNET_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_n2.json 2> $O/bench_n2.err; echo "n2 rc $?"; cut -c1-300 $O/bench_n2.json; tail -5 $O/bench_n2.err
