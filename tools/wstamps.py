#!/usr/bin/env python3
"""Developer tool (written for the first version of the windowed schedule; the slot layout it decodes has not been re-checked
against k_wpipe): in-kernel phase stamps of ONE launch of the windowed schedule (diagnostic build:
make -C tpnet_amd/csrc STAMPS=1).   TPNET_DEV_STAMP_LAYER=2 python tools/wstamps.py [C2]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TPNET_DEV_LIB", os.path.join(ROOT, "tpnet_amd", "libtpnet_hip_stamps.so"))
os.environ.setdefault("TPNET_DEV_STAMP_LAYER", "2")
import tpnet_amd
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
c = dict(CONFIGS[cfgname])
B = c["B"]; K = min(64, 16384 // B); nbw = 5 * K; E = (nbw + K) * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
neg = synthetic_negatives(c["U"], N, E, B, 1)
dev = torch.device("cuda:0")
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
D = lambda x: torch.from_numpy(x).to(dev)
ds, dd, dn, dt = D(src), D(dst), D(neg), D(t)
rp._workspace(E, B, stream=True)
for rep in range(2):
    rp.run_stream(ds[:nbw * B], dd[:nbw * B], dn[:nbw * B], dt[:nbw * B], B, t_end=float(t[nbw * B - 1]))   # warm
rp._eng["ws"][: 1 << 20].zero_()
rp.run_stream(ds[nbw * B:], dd[nbw * B:], dn[nbw * B:], dt[nbw * B:], B, t_end=float(t[-1]))           # ONE window
torch.cuda.synchronize()
raw = rp._eng["ws"][: 1 << 20].cpu().numpy().view(np.uint64).reshape(-1, 8, 2)[:4000].astype(np.int64)
clk, wall = raw[:, :, 0], raw[:, :, 1]
hv = clk[:, 0] > 0
lt = clk[:, 5] > 0
t0 = min(wall[hv, 0].min() if hv.any() else 1 << 62, wall[lt, 5].min() if lt.any() else 1 << 62)
print(f"{cfgname} layer {os.environ['TPNET_DEV_STAMP_LAYER']}: heavy waves {hv.sum()}, light waves {lt.sum()}")
if hv.any():
    end = (wall[hv, 4] - t0) * 10
    print(f"  heavy units (last per block): start rel ns med {int(np.median((wall[hv,0]-t0)*10))} max {((wall[hv,0]-t0)*10).max()}; end med {int(np.median(end))} max {end.max()}")
    names = ["setup(tables)", "block sums", "run sums", "chain walk"]
    cnt = clk[hv, 7]
    big = np.argsort(-cnt)[:8]
    for a in range(4):
        dw = (wall[hv, a + 1] - wall[hv, a]) * 10
        print(f"    {names[a]:14s} ns: med {int(np.median(dw)):6d} p95 {int(np.percentile(dw, 95)):6d} max {dw.max():6d}   | biggest chains: {dw[big].tolist()}")
    print(f"    chain sizes of the biggest: {cnt[big].tolist()}; total unit ns of those: {((wall[hv,4]-wall[hv,0])*10)[big].tolist()}")
if lt.any():
    dw = (wall[lt, 6] - wall[lt, 5]) * 10
    end = (wall[lt, 6] - t0) * 10
    print(f"  light waves: start rel ns med {int(np.median((wall[lt,5]-t0)*10))} max {((wall[lt,5]-t0)*10).max()}; duration med {int(np.median(dw))} p95 {int(np.percentile(dw,95))} max {dw.max()}; end max {end.max()}")
if hv.any():
    dc = (clk[hv, 4] - clk[hv, 0]).astype(float); dwall = (wall[hv, 4] - wall[hv, 0]).astype(float) * 10
    print(f"  shader clock during heavy units: {np.median(dc / dwall):.2f} GHz (median over waves)")
