#!/bin/bash
# round-5 evidence, phase 1 (GPU box): kernel traces + separate PMC passes per config (tools/prof_round.sh) -> gpurun_out/r05_*;
# summarised locally by `ROUND=r05 python tools/summarize_round.py C1 C2 C3 C4 C5` into profiles/r05_<CFG>.md / _pmc.json.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export ROUND=r05
MODES="timed20 epoch long" bash tools/prof_round.sh C2 2>&1 | grep -E "exit [^0]|killed"
MODES="epoch long" bash tools/prof_round.sh C1 2>&1 | grep -E "exit [^0]|killed"
MODES="batch b1000" NB=60 bash tools/prof_round.sh C3 2>&1 | grep -E "exit [^0]|killed"
MODES="batch b1000" NB=60 bash tools/prof_round.sh C5 2>&1 | grep -E "exit [^0]|killed"
MODES="batch" NB=40 bash tools/prof_round.sh C4 2>&1 | grep -E "exit [^0]|killed"
ls gpurun_out | grep r05_ | head -40
echo done
