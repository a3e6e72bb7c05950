#!/usr/bin/env python3
"""Developer tool: host-side cost of a short run_stream call (enqueue time vs GPU time)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpnet_amd
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
c = CONFIGS["C2"]; B = c["B"]; nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
E = (nb + 5) * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
neg = synthetic_negatives(c["U"], N, E, B, 1)
dev = torch.device("cuda:0")
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
D = lambda x: torch.from_numpy(x).to(dev)
ds, dd, dn, dt = D(src), D(dst), D(neg), D(t)
op = torch.empty((nb * B, 64), device=dev); on = torch.empty((nb * B, 64), device=dev)
sched = sys.argv[2] if len(sys.argv) > 2 else "auto"
for rep in range(6):
    rp.reset_random_projections()
    rp.run_stream(ds[:5 * B], dd[:5 * B], dn[:5 * B], dt[:5 * B], B, out_pos=op[:5 * B], out_neg=on[:5 * B], t_end=float(t[5 * B - 1]), schedule=sched)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rp.run_stream(ds[5 * B:], dd[5 * B:], dn[5 * B:], dt[5 * B:], B, out_pos=op, out_neg=on, t_end=float(t[-1]), schedule=sched)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rep {rep}: enqueue {1e6 * (t1 - t0):.0f} us, until done {1e6 * (t2 - t0):.0f} us  ({nb * B / (t2 - t0) / 1e6:.1f} M edges/s)")
