#!/bin/bash
# Round-4 evidence for the encoder-side kernels at C2 (d=128, K=20, 80 000 pairs per call): per-kernel duration and SQ / TCC
# counters, one --pmc pass per counter group.  usage (GPU box): tools/encoder_pmc.sh [TAG] ; writes gpurun_out/encoder_pmc_TAG.json
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=1
cat > /tmp/enc_run.py <<PY
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, "$R")
import tpnet_amd
from tpnet_amd import _lib, fused_feature as ff
from tpnet_amd.stream import CONFIGS, synthetic_stream
from tpnet_amd.callers import RecentNeighborSampler
lib = _lib.load()
c = CONFIGS["C2"]; B = c["B"]; K = 20; nb = 12; E = nb * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
dev = torch.device("cuda:0")
torch.manual_seed(0)
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
rp.run_stream(D(src[:-B]), D(dst[:-B]), None, D(t[:-B]), B, want_neg=False, want_pos=False)
s = slice(E - B, E)
sampler = RecentNeighborSampler(src, dst, t)
nodes = np.concatenate([src[s], dst[s]])
neigh, _, _ = sampler.get_historical_neighbors(nodes, np.tile(t[s], 2), K)
w = neigh.reshape(-1); a = np.repeat(np.tile(src[s], 2), K); b_ = np.repeat(np.tile(dst[s], 2), K)
n = len(w)
du, dv = D(np.tile(w, 2)), D(np.concatenate([a, b_]))
dn, d1, d2 = D(neigh), D(np.tile(src[s], 2)), D(np.tile(dst[s], 2))
out = torch.empty((2 * n, 64), device=dev)
st = rp._state(); stream = rp._stream(); now = rp._now_host; lam = float(c["lam"])
REPS = 6
for _ in range(REPS):
    lib.tpnet_pair_gram(C.byref(st), du.data_ptr(), dv.data_ptr(), 2 * n, now, lam, 0, out.data_ptr(), stream)
for _ in range(REPS):
    lib.tpnet_pair_gram_anchored(C.byref(st), dn.data_ptr(), d1.data_ptr(), d2.data_ptr(), neigh.shape[0], K, now, lam, 0,
                                 out.data_ptr(), out[n:].data_ptr(), stream)
with torch.no_grad():
    for _ in range(REPS): rp.get_pair_wise_feature_anchored(dn, d1, d2)          # readout + dense layers in one launch
    for _ in range(REPS): ff.mlp_f32(rp.mlp, out)
    for _ in range(REPS): rp.get_pair_wise_feature(src[s], dst[s])
torch.cuda.synchronize()
PY
O=$R/gpurun_out/encoder_pmc_$TAG; rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 /tmp/enc_run.py > $O/trace.log 2>&1; echo "trace exit $?"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SALU SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc$i -- python3 /tmp/enc_run.py > $O/pmc$i.log 2>&1 || { echo "$grp: failed"; tail -3 $O/pmc$i.log; continue; }
  echo "$grp" > $O/pmc$i/group.txt
done
python3 $R/tools/encoder_pmc_summary.py $O > $R/gpurun_out/encoder_pmc_$TAG.json; cat $R/gpurun_out/encoder_pmc_$TAG.json | cut -c1-3000
