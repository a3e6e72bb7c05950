#!/bin/bash
# A/B of library variants: tools/ab.sh name1 name2 ...   (name "" = product lib)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  lib=$R/tpnet_amd/libtpnet_hip_$v.so; [ "$v" = base ] && lib=$R/tpnet_amd/libtpnet_hip.so
  ( export TPNET_DEV_LIB=$lib; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$v -- python3 $R/tools/profile_stream.py $PSARGS > $R/gpurun_out/ab_$v.log 2>&1 )
  echo "== $v"; grep -h "us/batch" $R/gpurun_out/ab_$v.log | tail -1
  grep -h "k_step" $R/gpurun_out/ab_$v/*/*_kernel_stats.csv | awk -F'",' '{print "   "$2}' | head -2
  grep -h "k_step" $R/gpurun_out/ab_$v/*/*_kernel_stats.csv | sed 's/.*)",//' | head -1
done
