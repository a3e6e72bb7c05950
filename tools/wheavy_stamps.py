#!/usr/bin/env python3
"""Developer tool: phase stamps of the workgroup-walked (hub) chains of ONE pipeline step of a short windowed call
(diagnostic build with -DTPNET_STAMPS -DTPNET_DEV).   python tools/wheavy_stamps.py [nb] [step] [layer]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TPNET_DEV_LIB", os.path.join(ROOT, "tpnet_amd", "libtpnet_hip_stampsdev.so"))
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
os.environ["TPNET_DEV_STAMP_STEP"] = sys.argv[2] if len(sys.argv) > 2 else "0"
os.environ["TPNET_DEV_STAMP_LAYER"] = sys.argv[3] if len(sys.argv) > 3 else "1"
import tpnet_amd
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
c = dict(CONFIGS["C2"]); B = c["B"]; E = nb * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
neg = synthetic_negatives(c["U"], N, E, B, 1)
dev = torch.device("cuda:0")
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
D = lambda x: torch.from_numpy(x).to(dev)
ds, dd, dn, dt = D(src), D(dst), D(neg), D(t)
rp._workspace(E, B, stream=True)
for rep in range(3):
    rp.reset_random_projections()
    rp._eng["ws"][: 1 << 20].zero_()
    rp.run_stream(ds, dd, dn, dt, B, t_end=float(t[-1]), schedule="windowed", replay=False)
torch.cuda.synchronize()
raw = rp._eng["ws"][: 1 << 19].cpu().numpy().view(np.uint64).reshape(-1, 8, 2)[:4000].astype(np.int64)
clk, wall = raw[:, :, 0], raw[:, :, 1]
hv = (clk[:, 0] > 0) & (clk[:, 4] > 0)
t0 = wall[hv, 0].min()
print(f"nb={nb} step {os.environ['TPNET_DEV_STAMP_STEP']} layer {os.environ['TPNET_DEV_STAMP_LAYER']}: stamped hub waves {hv.sum()}")
names = ["setup(tables)", "-> block sums (last segment)", "run sums", "chain walk + tail"]
cnt = clk[hv, 7]
big = np.argsort(-cnt)[:8]
print(f"  unit start rel ns: med {int(np.median((wall[hv,0]-t0)*10))} max {((wall[hv,0]-t0)*10).max()}; end med {int(np.median((wall[hv,4]-t0)*10))} max {((wall[hv,4]-t0)*10).max()}")
for a in range(4):
    dw = (wall[hv, a + 1] - wall[hv, a]) * 10
    print(f"    {names[a]:30s} ns: med {int(np.median(dw)):6d} p95 {int(np.percentile(dw, 95)):6d} max {dw.max():6d}   | biggest chains: {dw[big].tolist()}")
print(f"    chain sizes of the biggest: {cnt[big].tolist()}; total unit ns of those: {((wall[hv,4]-wall[hv,0])*10)[big].tolist()}")
sizes = np.unique(cnt)
for sz in sizes[-12:]:
    m = cnt == sz
    print(f"    chain of {sz:5d}: waves {m.sum():4d}, unit ns med {int(np.median((wall[hv,4]-wall[hv,0])[m]*10))}, setup {int(np.median((wall[hv,1]-wall[hv,0])[m]*10))}")
