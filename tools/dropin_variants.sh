#!/bin/bash
# development: the decoder-level loop with the product library, then the diagnostic one at several pair-feature kernel choices
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
true
for from in 512 992; do
  echo "== dev lib, FEATURE_MFMA_FROM=$from"
  TPNET_DEV_LIB=$R/tpnet_amd/libtpnet_hip_dev.so TPNET_DEV_FEATURE_MFMA_FROM=$from bash tools/dropin_trace.sh 2>&1 | grep -v "^[WE]2026"
done
