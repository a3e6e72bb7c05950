#!/bin/bash
# the C4 access-mix probe (tools/probes/c4_mix_probe.hip) over launch sizes and variants; compile first: hipcc -O3 --offload-arch=gfx950
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/tools/probes/c4_mix_probe
[ -x $P ] || /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $R/tools/probes/c4_mix_probe.hip -o $P
for args in "10000 40 256 12 0 0" "10000 40 256 12 1 0" "10000 40 256 6 0 0" "10000 40 256 4 0 0" "10000 40 256 12 0 1" "10000 40 512 12 0 0" \
            "40000 20 256 12 0 0" "40000 20 256 12 1 0" "40000 20 256 4 0 0" "160000 10 256 12 0 0" "160000 10 256 12 1 0" "160000 10 256 12 0 0 2048" "160000 10 256 12 0 0 4096" "10000 40 256 12 0 0 0 10000000 1" "10000 40 256 4 0 0 0 10000000 1" "40000 20 256 12 0 0 0 10000000 1"; do
  timeout -k 10 120 $P $args || exit 1
done
