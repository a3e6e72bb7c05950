#!/usr/bin/env python3
"""Developer tool: where the HOST time of the encoder's call from host index arrays goes (the reference's own call pattern,
models/TPNet.py:311-316: get_pair_wise_feature(tile(neigh, 2), concat(repeat(src, K), repeat(dst, K))) on numpy arrays):
cProfile of 200 calls at C2's shape (80 000 pairs) + the wall clock per call with and without the GPU work awaited."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpnet_amd
from tpnet_amd.stream import CONFIGS, synthetic_stream
c = CONFIGS["C2"]; B = c["B"]; K = 20; E = 6 * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
dev = torch.device("cuda:0"); torch.manual_seed(0)
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
rp.run_stream(D(src), D(dst), None, D(t), B, want_neg=False, want_pos=False)
rng = np.random.RandomState(0)
s = slice(0, B)
nodes = np.concatenate([src[s], dst[s]])
neigh = rng.randint(1, N, (len(nodes), K)).astype(np.int64)
u = np.tile(neigh.reshape(-1), 2)
v = np.concatenate([np.repeat(np.tile(src[s], 2), K), np.repeat(np.tile(dst[s], 2), K)])
with torch.no_grad():
    for _ in range(10): rp.get_pair_wise_feature(u, v)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): rp.get_pair_wise_feature(u, v)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host issue time per call {(t1 - t0) / 200 * 1e6:.1f} us; with the GPU awaited {(t2 - t0) / 200 * 1e6:.1f} us")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): rp.get_pair_wise_feature(u, v)
    pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
