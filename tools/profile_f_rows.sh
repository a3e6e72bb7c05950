#!/bin/bash
# rocprofv3 evidence for the rows either side of the path (f-1, f-2): kernel trace + stats of the rate tools, and FETCH_SIZE
# of the encoder readout kernels (generic / shared-first-node / anchored) in a separate --pmc pass.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/prof_f
for t in encoder_readout feature_rate mlp_rate decoder_rate; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$t -- python3 $R/tools/$t.py > $O.$t.log 2>&1; echo "$t exit $?"
done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/encoder_fetch -- python3 $R/tools/encoder_readout.py > $O.encoder_fetch.log 2>&1; echo "fetch exit $?"
