#!/usr/bin/env python3
"""Developer tool: encoder-style readout (4*B*K pairs, models/TPNet.py:311-316): generic pair kernel on the full pair
list vs the shared-first-node kernel, ids resident on the device, HIP-event timing."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpnet_amd
from tpnet_amd import _lib
from tpnet_amd.stream import CONFIGS, synthetic_stream
from tpnet_amd.callers import RecentNeighborSampler
lib = _lib.load()
for cfg, K in (("C1", 20), ("C2", 20), ("C3", 20)):
    c = CONFIGS[cfg]; B = c["B"]; nb = 12; E = nb * B
    src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
    dev = torch.device("cuda:0")
    rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
            device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
    D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    rp.run_stream(D(src[:-B]), D(dst[:-B]), None, D(t[:-B]), B, want_neg=False, want_pos=False)
    s = slice(E - B, E)
    sampler = RecentNeighborSampler(src, dst, t)
    nodes = np.concatenate([src[s], dst[s]])
    neigh, _, _ = sampler.get_historical_neighbors(nodes, np.tile(t[s], 2), K)
    w = neigh.reshape(-1); a = np.repeat(np.tile(src[s], 2), K); b_ = np.repeat(np.tile(dst[s], 2), K)
    n = len(w)
    dw, da, db = D(w), D(a), D(b_)
    du, dv = D(np.tile(w, 2)), D(np.concatenate([a, b_]))
    out = torch.empty((2 * n, 64), device=dev)
    st = rp._state(); stream = rp._stream(); now = rp._now_host; lam = float(c["lam"])
    def timeit(fn, reps=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    tg = timeit(lambda: lib.tpnet_pair_gram(C.byref(st), du.data_ptr(), dv.data_ptr(), 2 * n, now, lam, 0, out.data_ptr(), stream))
    ts = timeit(lambda: lib.tpnet_pair_gram_shared(C.byref(st), dw.data_ptr(), da.data_ptr(), db.data_ptr(), n, now, lam, 0,
                                                    out.data_ptr(), out[n:].data_ptr(), stream))
    dn, d1, d2 = D(neigh), D(np.tile(src[s], 2)), D(np.tile(dst[s], 2))
    ta = timeit(lambda: lib.tpnet_pair_gram_anchored(C.byref(st), dn.data_ptr(), d1.data_ptr(), d2.data_ptr(), neigh.shape[0], K,
                                                      now, lam, 0, out.data_ptr(), out[n:].data_ptr(), stream))
    tv = timeit(lambda: lib.tpnet_pair_gram_anchored(C.byref(st), dn.data_ptr(), d1.data_ptr(), d2.data_ptr(), neigh.shape[0], K,
                                                      now, lam, 256, out.data_ptr(), out[n:].data_ptr(), stream))
    byt = 2 * n * (8 * c["d"] * 4 + 256)
    print(f"{cfg} B={B} K={K}: {2 * n} pairs, d={c['d']}: generic {tg:.1f} us ({byt / tg / 1e3:.0f} GB/s algorithmic), "
          f"shared-first {ts:.1f} us -> {tg / ts:.2f}x, anchored {ta:.1f} us -> {tg / ta:.2f}x (matrix cores where they apply; "
          f"vector-ALU walk {tv:.1f} us)", flush=True)
