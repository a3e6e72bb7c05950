#!/usr/bin/env python3
"""Probe: do the matrix-core readout (VALU-heavy) and the dense-layer kernel (MFMA / LDS-heavy) overlap when they share the chip?
The two kernels of one 80 000-pair encoder call at C2 on two streams (independent buffers) against one after the other."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tpnet_amd
from tpnet_amd import _lib, fused_feature as ff
from tpnet_amd.stream import CONFIGS, synthetic_stream
lib = _lib.load()
c = CONFIGS["C2"]; B = c["B"]; K = 20; E = 6 * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
dev = torch.device("cuda:0")
torch.manual_seed(0)
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
rp.run_stream(D(src), D(dst), None, D(t), B, want_neg=False, want_pos=False)
rng = np.random.RandomState(0)
n = 2 * B
neigh = D(rng.randint(1, N, (n, K)).astype(np.int64)); a1 = D(rng.randint(1, N, n).astype(np.int64)); a2 = D(rng.randint(1, N, n).astype(np.int64))
out = torch.empty((2 * n * K, 64), device=dev)
x = torch.rand((2 * n * K, 64), device=dev) * 8
y = torch.empty_like(x)
st = rp._state(); now = rp._now_host; lam = float(c["lam"])
prep = ff.prepared(rp.mlp, 64)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def gram(s):
    lib.tpnet_pair_gram_anchored(C.byref(st), neigh.data_ptr(), a1.data_ptr(), a2.data_ptr(), n, K, now, lam, 0, out.data_ptr(),
                                 out[n * K:].data_ptr(), C.c_void_p(s.cuda_stream))
def mlp(s):
    lib.tpnet_mlp64_f32(x.data_ptr(), 2 * n * K, prep[2], y.data_ptr(), C.c_void_p(s.cuda_stream))
def timeit(fn, reps=30):
    fn(); torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e3
cur = torch.cuda.current_stream()
print("readout alone        %.1f us" % timeit(lambda: gram(cur)))
print("dense layers alone   %.1f us" % timeit(lambda: mlp(cur)))
print("one after the other  %.1f us" % timeit(lambda: (gram(cur), mlp(cur))))
def both():
    s1.wait_stream(cur); s2.wait_stream(cur)
    gram(s1); mlp(s2)
    cur.wait_stream(s1); cur.wait_stream(s2)
print("side by side (2 streams, events included) %.1f us" % timeit(both))
