#!/usr/bin/env python3
"""development: the decoder-level per-batch loop (2 x get_pair_wise_feature + update from host arrays), 60 batches; prints the
host time per batch with and without the GPU in the loop (issue-only = the loop's time when the final synchronize is excluded)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tpnet_amd, bench
from tpnet_amd.stream import CONFIGS
cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"]; B, d = cfg["B"], cfg["d"]
nb = 60
src, dst, neg, t, N = bench.make_workload(cfg, nb + 8, 0)
dev = torch.device("cuda", 0)
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=cfg["E"], dim_factor=10, num_layer=3, time_decay_weight=cfg["lam"],
                                      device=str(dev), use_matrix=False, beginning_time=np.float64(0.0), not_scale=False,
                                      enforce_dim=d).to(dev)
with torch.no_grad():
    for rep in range(3):
        rp.reset_random_projections()
        per = []
        for b in range(nb + 4):
            if b == 4:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            s = slice(b * B, (b + 1) * B)
            a0 = time.perf_counter()
            rp.get_pair_wise_feature(src[s], dst[s])
            a1 = time.perf_counter()
            rp.get_pair_wise_feature(src[s], neg[s])
            a2 = time.perf_counter()
            rp.update(src[s], dst[s], t[s])
            a3 = time.perf_counter()
            if b >= 4: per.append((a1 - a0, a2 - a1, a3 - a2))
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        p = np.array(per) * 1e6
        print(f"rep {rep}: issue {1e6 * (t1 - t0) / nb:.1f} us/batch, with final sync {1e6 * (t2 - t0) / nb:.1f} us/batch; "
              f"host per call median: pair {np.median(p[:, 0]):.1f} pair {np.median(p[:, 1]):.1f} update {np.median(p[:, 2]):.1f}")

    # the plain readout (no dense layers) from the same host ids, and from device-resident ids: kernels to compare in the trace
    for b in range(8):
        s = slice(b * B, (b + 1) * B)
        rp.pair_gram(src[s], dst[s])
    torch.cuda.synchronize()
    du, dv = torch.from_numpy(src[:B]).to(dev), torch.from_numpy(dst[:B]).to(dev)
    for b in range(8):
        rp.get_pair_wise_feature(du, dv)
    torch.cuda.synchronize()
