#!/bin/bash
# development: one epoch-length stream (158 batches) and a long one (1280) of C2 per setting of the windowed schedule's developer
# knobs on the diagnostic library (wall clock of run_stream: tools/short_sweep.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
export TPNET_DEV_LIB=$R/tpnet_amd/libtpnet_hip_dev.so
cd /tmp
for v in "$@"; do
  a=$(env $v python3 $R/tools/short_sweep.py 158 windowed 2>/dev/null | tail -1 | sed 's/.*total //')
  b=$(env $v python3 $R/tools/short_sweep.py 1280 windowed 2>/dev/null | tail -1 | sed 's/.*total //')
  echo "$v | 158: $a | 1280: $b"
done
