#!/usr/bin/env python3
"""Developer tool: wall clock of short run_stream calls (C2 shape) per schedule; window length via TPNET_DEV_WINDOW.
usage: python tools/short_sweep.py "20,40,158" [windowed|batch|auto]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpnet_amd
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
c = CONFIGS[os.environ.get("CFG", "C2")]; B = c["B"]
nbs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20").split(",")]
sched = sys.argv[2] if len(sys.argv) > 2 else "windowed"
nbmax = max(nbs)
E = (nbmax + 5) * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
neg = synthetic_negatives(c["U"], N, E, B, 1)
dev = torch.device("cuda:0")
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
D = lambda x: torch.from_numpy(x).to(dev)
ds, dd, dn, dt = D(src), D(dst), D(neg), D(t)
op = torch.empty((nbmax * B, 64), device=dev); on = torch.empty((nbmax * B, 64), device=dev)
rp._workspace(nbmax * B, B, stream=True).zero_(); op.zero_(); on.zero_()
for nb in nbs:
    res = []
    for rep in range(7):
        rp.reset_random_projections()
        rp.run_stream(ds[:5 * B], dd[:5 * B], dn[:5 * B], dt[:5 * B], B, out_pos=op[:5 * B], out_neg=on[:5 * B], t_end=float(t[5 * B - 1]), schedule=sched)
        torch.cuda.synchronize()
        e = (5 + nb) * B
        t0 = time.perf_counter()
        rp.run_stream(ds[5 * B:e], dd[5 * B:e], dn[5 * B:e], dt[5 * B:e], B, out_pos=op[:nb * B], out_neg=on[:nb * B], t_end=float(t[e - 1]), schedule=sched)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append((t1 - t0, t2 - t0))
    enq = np.median([r[0] for r in res[2:]]) * 1e6; tot = np.median([r[1] for r in res[2:]]) * 1e6
    print(f"W={os.environ.get('TPNET_DEV_WINDOW','-')} {sched} nb={nb}: enqueue {enq:.0f} us, total {tot:.0f} us, {nb * B / tot:.1f} M edges/s", flush=True)
