#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python tools/short_sweep.py 20,40,158 batch 2>/dev/null
python tools/short_sweep.py 12,20,40,158,400 windowed 2>/dev/null
python tools/short_sweep.py 12,20,40,158,400 auto 2>/dev/null
export TPNET_DEV_LIB=$R/tpnet_amd/libtpnet_hip_dev.so
for K in 5 7 10 20; do for H in 24 32 48; do TPNET_DEV_WIN_HEAVY=$H TPNET_DEV_WINDOW_FIXED=$K python tools/short_sweep.py 20 windowed 2>/dev/null | sed "s/^/K=$K H=$H /"; done; done
for K in 10 14 20; do for H in 32 48 64; do TPNET_DEV_WIN_HEAVY=$H TPNET_DEV_WINDOW_FIXED=$K python tools/short_sweep.py 40 windowed 2>/dev/null | sed "s/^/K=$K H=$H /"; done; done
for K in 12 16 20 24; do for H in 48 64 96; do TPNET_DEV_WIN_HEAVY=$H TPNET_DEV_WINDOW_FIXED=$K python tools/short_sweep.py 158 windowed 2>/dev/null | sed "s/^/K=$K H=$H /"; done; done
bash tools/short_trace.sh "0:0:20 0:0:158"
