#!/usr/bin/env python3
"""development: cProfile of the decoder-level per-batch loop (host side of 2 x get_pair_wise_feature + update)."""
import cProfile, pstats, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tpnet_amd, bench
from tpnet_amd.stream import CONFIGS
cfg = CONFIGS["C2"]; B, d = cfg["B"], cfg["d"]
nb = 150
src, dst, neg, t, N = bench.make_workload(cfg, nb + 8, 0)
dev = torch.device("cuda", 0)
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=cfg["E"], dim_factor=10, num_layer=3, time_decay_weight=cfg["lam"],
                                      device=str(dev), use_matrix=False, beginning_time=np.float64(0.0), not_scale=False,
                                      enforce_dim=d).to(dev)
def loop(n0, n1):
    for b in range(n0, n1):
        s = slice(b * B, (b + 1) * B)
        rp.get_pair_wise_feature(src[s], dst[s])
        rp.get_pair_wise_feature(src[s], neg[s])
        rp.update(src[s], dst[s], t[s])
with torch.no_grad():
    loop(0, 4); torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable(); loop(4, nb); pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
