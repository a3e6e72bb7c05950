#!/bin/bash
# development: the windowed / planner parity tests, then timelines of short calls
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "three_launch or plan_replay or windowed or stream_matches or c2_full or internal_chunking or odd_shapes or packed" > gpurun_out/t2.log 2>&1; tail -5 gpurun_out/t2.log
python tools/short_sweep.py 12,20,40,158,400 windowed 2>/dev/null
bash tools/short_trace.sh "0:0:20 0:0:158"
