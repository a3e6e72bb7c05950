#!/usr/bin/env python3
"""Developer tool: rate of the drop-in Python path (host numpy in, per-call H2D, 3 calls per batch) vs run_stream."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpnet_amd
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
c = CONFIGS[cfg]; B = c["B"]; nb = 200; E = nb * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
neg = synthetic_negatives(c["U"], N, E, B, 1)
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to("cuda:0")
for rep in range(2):
    rp.reset_random_projections()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        for b in range(nb):
            s = slice(b * B, (b + 1) * B)
            rp.get_pair_wise_feature(src[s], dst[s]); rp.get_pair_wise_feature(src[s], neg[s])
            rp.update(src[s], dst[s], t[s])
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"{cfg} drop-in API (numpy in, mlp included): {E / el / 1e6:.2f} M edges/s, {el / nb * 1e6:.1f} us/batch")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for b in range(50):
    s = slice(b * B, (b + 1) * B)
    rp.pair_gram(src[s], dst[s]); rp.update(src[s], dst[s], t[s])
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
