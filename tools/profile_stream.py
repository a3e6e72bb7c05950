#!/usr/bin/env python3
"""Developer tool: run the device-resident stream of one config a few times (for rocprofv3 / timing experiments).
usage: python tools/profile_stream.py [--config C2] [--batches 300] [--reps 3]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpnet_amd
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2"); ap.add_argument("--batches", type=int, default=300)
ap.add_argument("--reps", type=int, default=3); ap.add_argument("--batch", type=int, default=0); ap.add_argument("--dim", type=int, default=0)
ap.add_argument("--edges", type=int, default=0, help="exact stream length (overrides --batches; -1 = the config's own E)")
ap.add_argument("--schedule", default=None)
ap.add_argument("--replay", action="store_true", help="let every repetition after the first replay the stream's plan (a later epoch)")
ap.add_argument("--pu", type=float, default=2.0, help="degree-law exponent of the users (1.0 = uniform)")
ap.add_argument("--pi", type=float, default=3.0, help="degree-law exponent of the items (1.0 = uniform)")
a = ap.parse_args()
c = dict(CONFIGS[a.config]); c["d"] = a.dim or c["d"]; B = a.batch or c["B"]; E = a.batches * B
if a.edges:
    E = c["E"] if a.edges < 0 else a.edges
    a.batches = (E + B - 1) // B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0, a.pu, a.pi)
print(f"{a.config}: N={N} E={E} d={c['d']} state {(2 * N * 3 + N) * c['d'] * 4 / 1e9:.1f} GB", flush=True)
neg = synthetic_negatives(c["U"], N, E, B, 1)
dev = torch.device("cuda:0")
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
D = lambda x: torch.from_numpy(x).to(dev)
ds, dd, dn, dt = D(src), D(dst), D(neg), D(t)
op = torch.empty((E, 64), device=dev); on = torch.empty((E, 64), device=dev)
for r in range(a.reps):
    rp.reset_random_projections()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rp.run_stream(ds, dd, dn, dt, B, out_pos=op, out_neg=on, t_end=float(t[-1]), schedule=a.schedule, replay=(None if a.replay else False))
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"{a.config} d={c['d']} B={B} rep {r}: {E / el / 1e6:.1f} M edges/s, {el / a.batches * 1e6:.2f} us/batch", flush=True)
