#!/bin/bash
# round 5: plan replay on the per-batch schedule (one-chunk streams) + the exact mode's readout launch: parity, then the epoch of C3 / C5 cold against replayed
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export HIP_FORCE_DEV_KERNARG=1
O=$R/gpurun_out/r05h; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for c in C3 C5; do
  for mode in cold replay; do
    extra=""; [ $mode = replay ] && extra="--replay"
    timeout -k 10 200 python tools/profile_stream.py --config $c --edges -1 --reps 4 $extra > $O/${c}_$mode.log 2>&1 || exit 1
    echo "== $c $mode"; grep rep $O/${c}_$mode.log
  done
done
