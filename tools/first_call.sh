#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2 3; do
for m in 0 1 2; do echo "mode $m: $(TPNET_WARM_MODE=$m python tools/first_call.py 2>/dev/null)"; done
echo "mode 0 + explicit: $(TPNET_WARM_MODE=0 WARM=64:100 python tools/first_call.py 2>/dev/null)"
done
