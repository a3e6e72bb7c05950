#!/usr/bin/env python3
"""Developer tool: the targeted exchange's plan for one run_stream call of a multi-GPU bench shape (C2's law, G ranks, global batch
G * 1000, 20 batches), rank 0: torch plan + relabelling against tpnet_xplan_targeted (wall clock incl. the read-back)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tpnet_amd import _lib
from tpnet_amd.sharded import ShardedStreamRunner
from tpnet_amd.stream import CONFIGS
class _Rp:
    @staticmethod
    def _stream():
        return None
for G in (2, 8):
    cfg = dict(CONFIGS["C2"], B=CONFIGS["C2"]["B"] * G)
    B = cfg["B"]; nb = 20
    src, dst, neg, t, N = bench.make_workload(cfg, nb, 0)
    dev = torch.device("cuda:0")
    D = [torch.from_numpy(x[:nb * B]).to(dev) for x in (src, dst, neg)]
    r = object.__new__(ShardedStreamRunner)
    r.G, r.me, r.N, r.n_cap, r.H, r.rp, r.group = G, 0, N, (N + G - 1) // G, 3 * B, _Rp(), None
    for name, fn in (("torch", r.relabel_targeted), ("device", r.relabel_targeted_device)):
        ts = []
        for i in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            R = fn(D[0], D[1], D[2], B)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
        print(f"G={G} global batch {B}, {nb} batches: {name} plan {np.median(ts[2:]):.0f} us per call (rows received per batch {int(R['rtot'].mean())})")
