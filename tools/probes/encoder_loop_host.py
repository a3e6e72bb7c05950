#!/usr/bin/env python3
"""Probe: host issue time against wall time of the encoder-level loop with device-resident ids (bench.py's encoder_level_device)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tpnet_amd
from tpnet_amd.stream import CONFIGS, synthetic_stream
from tpnet_amd.sampler import GpuRecentNeighborSampler
c = CONFIGS["C2"]; B = c["B"]; K = 20; nbe = 24; E = (nbe + 4) * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
neg = np.random.RandomState(1).randint(1, N, E).astype(np.int64)
dev = torch.device("cuda:0"); torch.manual_seed(0)
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
smp = GpuRecentNeighborSampler(src, dst, t, device="cuda:0", num_nodes=N)
for rep in range(4):
    rp.reset_random_projections()
    with torch.no_grad():
        for b in range(nbe + 4):
            if b == 4:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            s = slice(b * B, (b + 1) * B)
            for other in (dst[s], neg[s]):
                rp.encoder_pair_features(smp, src[s], other, t[s], K)
            rp.get_pair_wise_feature(src[s], dst[s])
            rp.get_pair_wise_feature(src[s], neg[s])
            rp.update(src[s], dst[s], t[s])
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"pass {rep}: host issue {(t1 - t0) / nbe * 1e6:.1f} us per batch, wall {(t2 - t0) / nbe * 1e6:.1f} us per batch")
