#!/usr/bin/env python3
"""Developer tool: phase stamps of k_wsort's middle workgroup (diagnostic build: make -C tpnet_amd/csrc STAMPS=1), one epoch of C2."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TPNET_DEV_LIB", os.path.join(ROOT, "tpnet_amd", "libtpnet_hip_stamps.so"))
import tpnet_amd, bench
from tpnet_amd.stream import CONFIGS
c = CONFIGS["C2"]; B = c["B"]; nb = (c["E"] + B - 1) // B
src, dst, neg, t, N = bench.make_workload(c, nb, 0)
E = c["E"]
dev = torch.device("cuda:0")
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=E, dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
D = [torch.from_numpy(x[:E]).to(dev) for x in (src, dst, neg, t)]
names = ["start", "edges loaded", "sorted", "keys in LDS, arrays stored, scan", "run heads ranked", "hash inserts done", "descriptor"]
acc = []
for rep in range(6):
    rp.reset_random_projections()
    rp.run_stream(D[0], D[1], D[2], D[3], B, t_end=float(t[E - 1]), schedule="windowed")
    torch.cuda.synchronize()
    raw = rp._eng["ws"][:7 * 16].cpu().numpy().view(np.uint64).reshape(7, 2).astype(np.int64)
    if rep >= 2:
        acc.append((raw[:, 1] - raw[0, 1]) * 10)
a = np.median(np.array(acc), axis=0)
for i, nm in enumerate(names):
    print(f"{nm:36s} {a[i]:8.0f} ns  (+{a[i] - (a[i - 1] if i else 0):.0f})")
