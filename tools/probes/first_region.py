#!/usr/bin/env python3
"""The first timed 20-batch call of a process against the later ones: host time until run_stream returns (all launches issued) and
until the synchronise behind it returns.  Same order of calls as bench.py's main leg (reserve, 5 warm-up batches on the windowed
schedule, then the timed calls, each from a reset table)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
import tpnet_amd
from tpnet_amd.stream import CONFIGS
import bench
cfg = CONFIGS["C2"]; B, d, K, W = cfg["B"], cfg["d"], 20, 5
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
src, dst, neg, t, N = bench.make_workload(cfg, K + W, 0)
D = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, t)]
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=cfg["E"], dim_factor=10, num_layer=3, time_decay_weight=cfg["lam"],
                                      device=str(dev), use_matrix=False, beginning_time=np.float64(0.0), not_scale=False,
                                      enforce_dim=d).to(dev)
rp.reserve_stream(K * B, B)
op = torch.empty(K * B, 64, device=dev); on = torch.empty(K * B, 64, device=dev)
extra = int(sys.argv[1]) if len(sys.argv) > 1 else 0          # extra warm-up calls of the timed shape before the first timed one
rows = []
for reg in range(5 + extra):
    rp.reset_random_projections()
    sl = slice(0, W * B)
    rp.run_stream(D[0][sl], D[1][sl], D[2][sl], D[3][sl], B, out_pos=op[:W * B], out_neg=on[:W * B], t_end=float(t[W * B - 1]),
                  schedule="windowed")
    sl = slice(W * B, (W + K) * B)
    a = (D[0][sl], D[1][sl], D[2][sl], D[3][sl]); te = float(t[(W + K) * B - 1])
    torch.cuda.synchronize()
    mode = os.environ.get("FR_MODE", "")
    if mode == "sleep" and reg >= 3:
        time.sleep(0.05)                                  # an idle host in front of the call
    if mode == "spin":
        tb = time.perf_counter()
        while time.perf_counter() - tb < 0.003:           # a busy host in front of the call
            pass
    t0 = time.perf_counter()
    rp.run_stream(*a, B, out_pos=op, out_neg=on, t_end=te)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6, rp.last_stream_replayed))
print("region: issue us / total us / replayed:", [(round(a, 1), round(b, 1), r) for a, b, r in rows], flush=True)
