#!/bin/bash
# One round's evidence for one config (ROUND=r05 by default): per mode (timed20 = the driver's 20 timed batches, epoch = the config's own stream, long =
# 2048 batches) a rocprofv3 kernel trace + stats and SEPARATE --pmc passes for FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md:
# FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2).  usage (GPU box): tools/prof_round.sh C2 ; then locally: python tools/summarize_round.py C2
CFG=${1:-C2}
ROUND=${ROUND:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=1
declare -A ARGS
ARGS[timed20]="--config $CFG --batches 20 --reps 6 --schedule auto"
ARGS[epoch]="--config $CFG --edges -1 --reps 3"
ARGS[long]="--config $CFG --batches 2048 --reps 2"
ARGS[batch]="--config $CFG --batches ${NB:-60} --reps 2"
ARGS[b1000]="--config $CFG --batch 1000 --batches 600 --reps 2"
# (a step that was killed at its limit ends the script: no further GPU step behind a hung one)
step() { "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed (rc $rc): stopping"; exit $rc; fi; return $rc; }
for mode in ${MODES:-timed20 epoch long}; do
  O=$R/gpurun_out/${ROUND}_${CFG}_$mode
  rm -rf $O
  step timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/profile_stream.py ${ARGS[$mode]} > $O.trace.log 2>&1; echo "$mode trace exit $?"; tail -1 $O.trace.log
  step timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/tools/profile_stream.py ${ARGS[$mode]} > $O.fetch.log 2>&1; echo "$mode fetch exit $?"
  step timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/tools/profile_stream.py ${ARGS[$mode]} > $O.write.log 2>&1; echo "$mode write exit $?"
  echo "${ARGS[$mode]}" > $O/args.txt
  (cd $R && python3 -c "import bench; print(bench.csrc_fingerprint())") > $O/sha.txt 2>/dev/null
done
