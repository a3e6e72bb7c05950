#!/bin/bash
# rocprofv3 kernel-trace of a few developer variants; prints the k_step line of each stats file
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" true; ( export "$@"; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/var_$tag -- python3 $R/tools/profile_stream.py ${PSARGS} > $R/gpurun_out/var_$tag.log 2>&1 ); echo "== $tag ($*)"; grep -h "us/batch" $R/gpurun_out/var_$tag.log | tail -1; grep -h "k_step" $R/gpurun_out/var_$tag/*/*_kernel_stats.csv | awk -F, '{print "   k_step calls="$(NF-6)" avg_ns="$(NF-4)" min="$(NF-2)" max="$(NF-1)}'; }
run base X=1
run readout TPNET_DEV_ROLE_MASK=1
run update TPNET_DEV_ROLE_MASK=2
run upd_noheavy TPNET_DEV_ROLE_MASK=2 TPNET_DEV_HEAVY_THRESHOLD=100000
run thr4 TPNET_DEV_HEAVY_THRESHOLD=4
run thr16 TPNET_DEV_HEAVY_THRESHOLD=16
run thr32 TPNET_DEV_HEAVY_THRESHOLD=32
