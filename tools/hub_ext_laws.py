#!/usr/bin/env python3
"""Developer tool: one epoch-length stream (158 batches of 1000, C2's shape) on four degree laws, windowed schedule, wall clock
(best of 4) -- run per setting of TPNET_DEV_WIN_HUB_EXT on the diagnostic library."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpnet_amd
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
c = CONFIGS["C2"]; B = c["B"]; nb = 158; E = nb * B
dev = torch.device("cuda:0")
out = []
for pu, pi in ((2.0, 3.0), (1.5, 4.0), (3.0, 5.0), (1.0, 1.0)):
    src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0, pu, pi)
    neg = synthetic_negatives(c["U"], N, E, B, 1)
    rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
            device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
    D = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, t)]
    op = torch.empty((E, 64), device=dev); on = torch.empty((E, 64), device=dev)
    best = 1e9
    for r in range(5):
        rp.reset_random_projections()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rp.run_stream(D[0], D[1], D[2], D[3], B, out_pos=op, out_neg=on, t_end=float(t[-1]), schedule="windowed")
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        if r: best = min(best, el * 1e6)
    out.append(f"({pu},{pi}) {best:.0f} us")
print(os.environ.get("TPNET_DEV_WIN_HUB_EXT", "default"), " | ".join(out))
