#!/bin/bash
# rocprofv3 passes for one config: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc runs
# (MI355X_MICROARCH.md: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2).  usage: tools/pmc.sh C2 [batches]
CFG=${1:-C2}; NB=${2:-640}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/pmc_$CFG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/profile_stream.py --config $CFG --batches $NB --reps 2 > $O.trace.log 2>&1; echo "trace exit $?"; tail -1 $O.trace.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/tools/profile_stream.py --config $CFG --batches $NB --reps 1 > $O.fetch.log 2>&1; echo "fetch exit $?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/tools/profile_stream.py --config $CFG --batches $NB --reps 1 > $O.write.log 2>&1; echo "write exit $?"
ls $O/*/*/ | head -30
