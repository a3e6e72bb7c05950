#!/bin/bash
# one round's committed evidence: rocprofv3 kernel trace + separate PMC passes per config (tools/pmc.sh) and a bench line per
# config.  usage (on the GPU box): tools/profile_round.sh ; then locally: PROF_BATCHES=.. python tools/summarize_prof.py rNN Cx
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/pmc.sh C2 1280 > gpurun_out/pmc_C2.log 2>&1
bash tools/pmc.sh C1 2560 > gpurun_out/pmc_C1.log 2>&1
bash tools/pmc.sh C3 60 > gpurun_out/pmc_C3.log 2>&1
bash tools/pmc.sh C5 60 > gpurun_out/pmc_C5.log 2>&1
cd $R
timeout -k 10 300 python bench.py --config C2 > gpurun_out/r02_bench_C2.json 2> gpurun_out/r02_bench_C2.err
timeout -k 10 300 python bench.py --config C1 --steps 8000 --warmup 200 --no-cpu-baseline > gpurun_out/r02_bench_C1.json 2> gpurun_out/r02_bench_C1.err
timeout -k 10 300 python bench.py --config C3 --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/r02_bench_C3.json 2> gpurun_out/r02_bench_C3.err
timeout -k 10 300 python bench.py --config C5 --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/r02_bench_C5.json 2> gpurun_out/r02_bench_C5.err
tail -c 600 gpurun_out/r02_bench_C*.json
