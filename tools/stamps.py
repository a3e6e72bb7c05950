#!/usr/bin/env python3
"""Developer tool: in-kernel phase stamps of k_step (diagnostic build: make -C tpnet_amd/csrc STAMPS=1).
TPNET_DEV_LIB=tpnet_amd/libtpnet_hip_stamps.so python tools/stamps.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TPNET_DEV_LIB", os.path.join(ROOT, "tpnet_amd", "libtpnet_hip_stamps.so"))
import tpnet_amd
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
c = dict(CONFIGS[cfgname])
if len(sys.argv) > 3: c["d"] = int(sys.argv[2]); c["B"] = int(sys.argv[3])      # overrides: stamps.py C2 16 8000
B = c["B"]; nbw = 40; E = (nbw + 1) * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
neg = synthetic_negatives(c["U"], N, E, B, 1)
dev = torch.device("cuda:0")
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
D = lambda x: torch.from_numpy(x).to(dev)
ds, dd, dn, dt = D(src), D(dst), D(neg), D(t)
rp._workspace(E, B)
for rep in range(3):
    rp.run_stream(ds[:nbw * B], dd[:nbw * B], dn[:nbw * B], dt[:nbw * B], B, t_end=float(t[nbw * B - 1]))   # warm
rp._eng["ws"][: 1 << 20].zero_()
rp.run_stream(ds[nbw * B:], dd[nbw * B:], dn[nbw * B:], dt[nbw * B:], B, t_end=float(t[-1]))           # ONE launch
torch.cuda.synchronize()
raw = rp._eng["ws"][: 1 << 20].cpu().numpy().view(np.uint64).reshape(-1, 8, 2)[:8000].astype(np.int64)
clk, wall = raw[:, :, 0], raw[:, :, 1]
used = clk[:, 0] > 0
t0 = wall[used, 0].min()
print(f"{cfgname}: waves stamped {used.sum()}; kernel span by wall clock (100 MHz ticks*10ns): "
      f"{(wall[used].max() - t0) * 10} ns")
names = {0: "start", 1: "ids loaded/meta issue", 2: "meta+ptrs ready", 3: "rows+FMA done", 4: "reduce done", 5: "pair stored",
         6: "light item done", 7: "heavy item done"}
for s in range(8):
    m = used & (clk[:, s] > 0)
    if m.sum() == 0:
        continue
    rel = (wall[m, s] - t0) * 10
    print(f"  slot {s} {names[s]:24s} n={m.sum():5d}  wall-rel ns: min {rel.min():6d} med {int(np.median(rel)):6d} max {rel.max():6d}")
ro = used & (clk[:, 5] > 0)
for a, b in ((0, 1), (1, 2), (2, 3), (3, 4), (4, 5)):
    dc = clk[ro, b] - clk[ro, a]; dw = (wall[ro, b] - wall[ro, a]) * 10
    print(f"  readout waves phase {a}->{b}: cycles med {int(np.median(dc))} p95 {int(np.percentile(dc, 95))}; ns med {int(np.median(dw))}")
tot_c = clk[ro, 5] - clk[ro, 0]; tot_w = (wall[ro, 5] - wall[ro, 0]) * 10
print(f"  readout wave total: cycles med {int(np.median(tot_c))}, ns med {int(np.median(tot_w))} -> clock ~{np.median(tot_c) / np.median(tot_w):.2f} GHz")
up = used & (clk[:, 6] > 0)
if up.sum():
    dw = (wall[up, 6] - wall[up, 0]) * 10
    print(f"  light-update waves total ns: med {int(np.median(dw))} max {dw.max()}")
hv = used & (clk[:, 7] > 0)
if hv.sum():
    dw = (wall[hv, 7] - wall[hv, 0]) * 10
    print(f"  heavy waves (last item done) ns: med {int(np.median(dw))} max {dw.max()}  n={hv.sum()}")
