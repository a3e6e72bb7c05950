#!/usr/bin/env python3
"""Developer tool: the C2 stream rate for several seeds of the synthetic stream (different hub statistics): the
kernel-variant and grid choices of launch_step are tuned on seed 0 and must not depend on it."""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpnet_amd
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
c = CONFIGS["C2"]; B = c["B"]; nb = 600; E = nb * B
dev = torch.device("cuda:0")
for seed in (0, 1, 2, 3):
    src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], seed)
    neg = synthetic_negatives(c["U"], N, E, B, 1 + seed)
    rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
            device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
    D = lambda x: torch.from_numpy(x).to(dev)
    ds, dd, dn, dt = D(src), D(dst), D(neg), D(t)
    op = torch.empty((E, 64), device=dev); on = torch.empty((E, 64), device=dev)
    for r in range(3):
        rp.reset_random_projections()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rp.run_stream(ds, dd, dn, dt, B, out_pos=op, out_neg=on, t_end=float(t[-1]))
        torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"seed {seed}: {E / el / 1e6:.1f} M edges/s, {el / nb * 1e6:.2f} us/batch", flush=True)
