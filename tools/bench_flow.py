#!/usr/bin/env python3
"""Developer tool: bench.py's timed region (W=5 warm-up batches, then K=20 timed) repeated, to separate first-call costs."""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpnet_amd
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
c = CONFIGS["C2"]; B = c["B"]; W, K = 5, 20
E = (W + K) * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
neg = synthetic_negatives(c["U"], N, E, B, 1)
dev = torch.device("cuda:0")
D = lambda x: torch.from_numpy(x).to(dev)
ds, dd, dn, dt = D(src), D(dst), D(neg), D(t)
pre = os.environ.get("PRE_SPIN") == "1"
pre_n = int(os.environ.get("PRE_CALL", "0"))
if pre_n:
    rq = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
            device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
    rq.run_stream(ds[:pre_n * B], dd[:pre_n * B], dn[:pre_n * B], dt[:pre_n * B], B, t_end=float(t[pre_n * B - 1]))
    torch.cuda.synchronize()
for trial in range(3):
    rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
            device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
    op = torch.empty((K * B, 64), device=dev); on = torch.empty((K * B, 64), device=dev)
    rp._workspace(K * B, B, stream=True)
    def run(a, b):
        sl = slice(a * B, b * B)
        rp.run_stream(ds[sl], dd[sl], dn[sl], dt[sl], B, out_pos=op[:(b - a) * B], out_neg=on[:(b - a) * B], t_end=float(t[b * B - 1]))
    if pre:
        x = torch.empty(1 << 26, device=dev); y = torch.empty_like(x)
        for _ in range(20): y.copy_(x)
    run(0, W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(W, W + K)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"trial {trial}: timed {el * 1e6:.0f} us -> {K * B / el / 1e6:.1f} M edges/s")
