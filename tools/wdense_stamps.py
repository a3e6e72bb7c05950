#!/usr/bin/env python3
"""Developer tool: phase stamps of the dense planner's middle workgroup (diagnostic build: make -C tpnet_amd/csrc STAMPS=1).
usage: python tools/wdense_stamps.py [nb]   (C2 shape, nb batches; default 20)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TPNET_DEV_LIB", os.path.join(ROOT, "tpnet_amd", "libtpnet_hip_stamps.so"))
import tpnet_amd, bench
from tpnet_amd.stream import CONFIGS
c = CONFIGS["C2"]; B = c["B"]; nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
src, dst, neg, t, N = bench.make_workload(c, nb, 0)
E = nb * B
dev = torch.device("cuda:0")
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
D = [torch.from_numpy(np.ascontiguousarray(x[:E])).to(dev) for x in (src, dst, neg, t)]
names = {0: "A start", 1: "A edges loaded", 2: "A sorted", 3: "A arrays stored, scan", 4: "A run heads ranked", 5: "A rows written", 6: "A descriptor",
         8: "kernel start", 9: "A done", 10: "barrier 1", 11: "B1a done", 12: "barrier 2", 13: "B1b done", 14: "barrier 3", 15: "B2 + C prologue", 16: "C done", 17: "last workgroup done", 20: "block 0: B1b start", 21: "block 0: m stored, chains ranked", 22: "block 0: B1b done"}
rp._workspace(E, B, stream=True)
acc = []
for rep in range(6):
    rp.reset_random_projections()
    rp._eng["ws"][:23 * 16].zero_()
    rp.run_stream(D[0], D[1], D[2], D[3], B, t_end=float(t[E - 1]), schedule="windowed", replay=False)
    torch.cuda.synchronize()
    raw = rp._eng["ws"][:23 * 16].cpu().numpy().view(np.uint64).reshape(23, 2).astype(np.int64)
    if rep >= 2:
        acc.append((raw[:, 1] - raw[8, 1]) * 10)
a = np.median(np.array(acc), axis=0)
order = [8, 0, 1, 2, 3, 4, 5, 6, 9, 10, 11, 12, 20, 21, 22, 13, 14, 15, 16, 17]
prev = 0
for i in order:
    print(f"{names[i]:28s} {a[i]:8.0f} ns  (+{a[i] - prev:.0f})")
    prev = a[i]
