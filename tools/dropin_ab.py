#!/usr/bin/env python3
"""The decoder-level per-batch loop (2 x get_pair_wise_feature + update from host arrays, C2) with the staging ring in pinned host
memory and in device memory behind the BAR (tpnet_stage_create_ex): us per batch, the faster of three passes each."""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tpnet_amd
from tpnet_amd import random_projection as RPM
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"]; B = cfg["B"]; nb = 60
src, dst, t, N = synthetic_stream(cfg["U"], cfg["I"], (nb + 3) * B, cfg["span"] * (nb + 3) * B / cfg["E"], 0)
neg = synthetic_negatives(cfg["U"], N, len(src), B, 1)
for mode in (0, -1, 0, -1):
    RPM.STAGE_MODE = mode
    rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=cfg["E"], dim_factor=10, num_layer=3, time_decay_weight=cfg["lam"], device="cuda:0",
                                          use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=cfg["d"]).to("cuda:0")
    def one():
        rp.reset_random_projections()
        with torch.no_grad():
            for b in range(nb + 3):
                if b == 3:
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                s = slice(b * B, (b + 1) * B)
                rp.get_pair_wise_feature(src[s], dst[s]); rp.get_pair_wise_feature(src[s], neg[s]); rp.update(src[s], dst[s], t[s])
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / nb * 1e6
    r = [one() for _ in range(4)]
    from tpnet_amd import _lib
    print(f"stage mode {mode} (in device memory: {_lib.load().tpnet_stage_in_device_memory(rp._eng['stage'].handle)}): {min(r):.1f} us per batch ({' '.join('%.1f' % x for x in r)})", flush=True)
