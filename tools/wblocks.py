#!/usr/bin/env python3
"""Developer tool: start / end of every block of ONE pipeline step of the windowed schedule (diagnostic build, STAMPS=1).
   python tools/wblocks.py [C2] [nb] [step]   -- stamps pipeline step `step` (default 10) of a warm stream of nb batches (default 160)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TPNET_DEV_LIB", os.path.join(ROOT, "tpnet_amd", "libtpnet_hip_stampsdev.so"))
os.environ["TPNET_DEV_STAMP_LAYER"] = "-1"
os.environ.setdefault("TPNET_DEV_STAMP_STEP", sys.argv[3] if len(sys.argv) > 3 else "10")
import tpnet_amd
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
c = dict(CONFIGS[cfgname])
B = c["B"]; nb = int(sys.argv[2]) if len(sys.argv) > 2 else 160; E = nb * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
neg = synthetic_negatives(c["U"], N, E, B, 1)
dev = torch.device("cuda:0")
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
D = lambda x: torch.from_numpy(x).to(dev)
ds, dd, dn, dt = D(src), D(dst), D(neg), D(t)
rp._workspace(E, B, stream=True)
for rep in range(2):
    rp.reset_random_projections()
    rp._eng["ws"][: 1 << 20].zero_()
    rp.run_stream(ds, dd, dn, dt, B, t_end=float(t[-1]), schedule="windowed", replay=False)
torch.cuda.synchronize()
# every step overwrote the block stamps: what is left is the last step that had each block index; take blocks whose
# stamps fall in the last 'full' step by looking at the latest start times
raw = rp._eng["ws"][(1 << 19): (1 << 20)].cpu().numpy().view(np.uint64).reshape(-1, 4)[:16000].astype(np.int64)
ok = raw[:, 0] > 0
st, en, kind = raw[ok, 0], raw[ok, 1], raw[ok, 2]
# cluster by step: steps are separated in time; keep the cluster with the most blocks
t0 = st.min()
print(f"{cfgname}: step with {len(st)} stamped blocks, span {(en.max() - t0) * 10} ns")
names = {0: "hub L1", 1: "hub L2", 2: "hub L3", 3: "chains L1", 4: "chains L2", 5: "chains L3", 6: "readout", 7: "medium L1", 8: "medium L2", 9: "medium L3"}
for k in sorted(set(kind.tolist())):
    m = kind == k
    d = (en[m] - st[m]) * 10
    busy = d > 1500
    print(f"  {names.get(k, k):10s} blocks {m.sum():5d} (busy {busy.sum():5d}): start {int((st[m].min() - t0) * 10):6d}..{int((st[m].max() - t0) * 10):6d} ns, "
          f"end max {int((en[m].max() - t0) * 10):6d}; busy duration med {int(np.median(d[busy])) if busy.any() else 0} p95 {int(np.percentile(d[busy], 95)) if busy.any() else 0} max {d.max()}")
# concurrency profile
T = np.arange(0, (en.max() - t0) + 1, 100)
conc = [(int(((st - t0 <= x) & (en - t0 > x) & ((en - st) * 10 > 1500)).sum())) for x in T]
print("  busy blocks in flight every 1 us:", conc)
