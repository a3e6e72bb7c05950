#!/usr/bin/env python3
"""Probe: the pre-mlp features of the same pairs three ways, many times over -- host ids through the staging ring (k_pair_gram), the
fused kernel's out_gram (host ids), device-resident ids (k_pair_gram, no ring) -- looking for entries that differ by more than
summation order allows."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tpnet_amd
from tpnet_amd import _lib, fused_feature as ff
d = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.RandomState(d + 1)
N = 300
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=4 * N, dim_factor=10, num_layer=3, time_decay_weight=1e-6, device="cuda:0",
                                      use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=d).to("cuda:0")
t0 = 0.0
for _ in range(3):
    src = rng.randint(1, N, 150); dst = rng.randint(1, N, 150); src[rng.rand(150) < 0.3] = 3; dst[rng.rand(150) < 0.1] = 5
    t = np.sort(rng.uniform(t0, t0 + 2e5, 150)); t0 = t[-1]
    rp.update(src.astype(np.int64), dst.astype(np.int64), t)
lib = _lib.load()
prep = ff.prepared(rp.mlp, 64)
bad = 0
for it in range(300):
    n = int(rng.choice([2048, 2500, 3001, 4099]))
    u, v = rng.randint(0, N, n).astype(np.int64), rng.randint(0, N, n).astype(np.int64)
    with torch.no_grad():
        g_dev = rp.pair_gram(torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda())
        g_host = rp.pair_gram(u, v)
        gram = torch.empty((n, 64), device="cuda:0"); out = torch.empty((n, 64), device="cuda:0")
        rc = lib.tpnet_host_pair_feature(rp._st_ref(), rp._eng["stage"].handle, u.ctypes.data, v.ctypes.data, n, rp._now_host,
                                         1e-6, 0, prep[2], gram.data_ptr(), out.data_ptr(), None)
        assert rc == 0
        want = rp.mlp(g_dev)
    for name, g in (("host ring k_pair_gram", g_host), ("fused out_gram", gram)):
        diff = (g - g_dev).abs()
        big = diff > 1e-3 * (1 + g_dev.abs())
        if bool(big.any()):
            idx = big.nonzero().cpu().numpy()
            bad += 1
            print(f"it {it} n {n} {name}: {len(idx)} entries differ; rows {sorted(set(idx[:, 0].tolist()))[:8]} cols {sorted(set(idx[:, 1].tolist()))[:16]} max {float(diff.max()):.3g}")
    e = (out - want).abs().max().item()
    if e > 1e-3 * max(1.0, want.abs().max().item()):
        bad += 1
        print(f"it {it} n {n} fused OUT differs from mlp(g_dev): {e:.3g}")
print("d", d, "iterations with a difference:", bad)
