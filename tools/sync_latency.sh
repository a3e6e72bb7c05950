#!/bin/bash
# the 20-step region under runtime settings that change how the host waits for the GPU (one process per setting)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 120 python tools/sync_latency.py
HSA_ENABLE_INTERRUPT=0 timeout -k 10 120 python tools/sync_latency.py
ROC_ACTIVE_WAIT_TIMEOUT=2000 timeout -k 10 120 python tools/sync_latency.py
timeout -k 10 120 python tools/sync_latency.py --spin
HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=2000 timeout -k 10 120 python tools/sync_latency.py
