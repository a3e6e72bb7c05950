# development: throughput of the windowed schedule over knob settings.  usage: CFG=C2 tools/win_sweep.sh "A=1 B=2" "..."
R=$GRAFT_REPO_ROOT
CFG=${CFG:-C2}
for v in "$@"; do
  ( export $v; echo "$v: $(python3 $R/tools/profile_stream.py --config $CFG --batches ${NBATCH:-640} --reps 3 2>&1 | grep 'rep 2' | cut -d: -f2)" )
done
