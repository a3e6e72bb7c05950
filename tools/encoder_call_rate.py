#!/usr/bin/env python3
"""Developer tool: the encoder's call (anchored readout + self.mlp) on C3's 800 000 pairs and C2's 80 000: readout alone,
readout then dense layers, and the one-call version that runs a chunk's dense layers beside the next chunk's readout."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpnet_amd
from tpnet_amd import _lib
from tpnet_amd.stream import CONFIGS, synthetic_stream
lib = _lib.load()
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for cfg, K in (("C2", 20), ("C3", 20), ("C5", 20)):
    c = CONFIGS[cfg]; B = c["B"]; nb = 6; E = nb * B
    src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
    dev = torch.device("cuda:0")
    rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
            device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
    D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    rp.run_stream(D(src), D(dst), None, D(t), B, want_neg=False, want_pos=False)
    rng = np.random.RandomState(1)
    n = 2 * B
    neigh = D(rng.randint(1, N, (n, K)).astype(np.int64)); a1 = D(np.tile(src[-B:], 2)); a2 = D(np.tile(dst[-B:], 2))
    with torch.no_grad():
        tr = timeit(lambda: rp.pair_gram_anchored(neigh, a1, a2))
        rp.fused_mlp = True      # (switches the one-call path off: readout, then the dense layers)
        g = rp.pair_gram_anchored(neigh, a1, a2).view(-1, 64)
        from tpnet_amd import fused_feature as ff
        tm = timeit(lambda: ff.mlp_f32(rp.mlp, g))
        rp.fused_mlp = False
        to = timeit(lambda: rp.get_pair_wise_feature_anchored(neigh, a1, a2))
        y1 = rp.get_pair_wise_feature_anchored(neigh, a1, a2)
        y2 = ff.mlp_f32(rp.mlp, g)
    print(f"{cfg} d={c['d']} {2 * n * K} pairs: anchored readout {tr:.1f} us; dense layers alone {tm:.1f} us; one call, {to:.1f} us "
          f"= {to / tr:.2f} x the readout (sequential: {(tr + tm) / tr:.2f} x); same bits: {bool(torch.equal(y1, y2))}", flush=True)
