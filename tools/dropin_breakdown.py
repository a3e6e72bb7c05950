#!/usr/bin/env python3
"""Developer tool: where a drop-in batch (2 x get_pair_wise_feature + update from host arrays) spends its time:
host enqueue time per call (no sync inside the loop) and wall time per batch with a sync at the end."""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
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
def loop(what):
    rp.reset_random_projections()
    acc = {"feat": 0.0, "upd": 0.0}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        for b in range(nb):
            s = slice(b * B, (b + 1) * B)
            a = time.perf_counter()
            if what & 1:
                rp.get_pair_wise_feature(src[s], dst[s]); rp.get_pair_wise_feature(src[s], neg[s])
            m = time.perf_counter()
            if what & 2:
                rp.update(src[s], dst[s], t[s])
            z = time.perf_counter()
            acc["feat"] += m - a; acc["upd"] += z - m
    t1 = time.perf_counter()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    return el / nb * 1e6, (t1 - t0) / nb * 1e6, acc["feat"] / nb * 1e6, acc["upd"] / nb * 1e6
for what, name in ((3, "both"), (1, "readouts only"), (2, "update only"), (3, "both")):
    loop(what)
    w, h, f, u = loop(what)
    print(f"{cfg} {name}: wall {w:.1f} us/batch, host enqueue {h:.1f} (2 readouts {f:.1f}, update {u:.1f})")
# pure host cost per call: bursts of 2 batches (6 calls < 8 ring slots, so no call waits for the GPU), GPU idle at the start
rp.reset_random_projections()
acc = [0.0, 0.0]; reps = 60
with torch.no_grad():
    for r in range(reps):
        torch.cuda.synchronize()
        for b in (2 * r, 2 * r + 1):
            s = slice(b * B, (b + 1) * B)
            a = time.perf_counter()
            rp.get_pair_wise_feature(src[s], dst[s]); rp.get_pair_wise_feature(src[s], neg[s])
            m = time.perf_counter()
            rp.update(src[s], dst[s], t[s])
            z = time.perf_counter()
            acc[0] += m - a; acc[1] += z - m
print(f"{cfg} host-only enqueue cost: get_pair_wise_feature {acc[0] / (4 * reps) * 1e6:.1f} us per call, update {acc[1] / (2 * reps) * 1e6:.1f} us per call")
