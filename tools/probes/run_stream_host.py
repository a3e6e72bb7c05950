#!/usr/bin/env python3
"""Developer probe: host time of RandomProjectionModule.run_stream OUTSIDE the C call (the C entry point replaced by a stub), and
of the C call's own host side (20 batches of C2), against the wall clock of the whole call with a synchronise behind it."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tpnet_amd, bench
from tpnet_amd import _lib
from tpnet_amd.stream import CONFIGS
c = CONFIGS["C2"]; B = c["B"]; K = 20
src, dst, neg, t, N = bench.make_workload(c, K + 5, 0)
dev = torch.device("cuda:0")
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
d = [torch.from_numpy(np.ascontiguousarray(x[:K * B])).to(dev) for x in (src, dst, neg, t)]
NG = rp.pair_wise_feature_dim
o_p = torch.empty((K * B, NG), dtype=torch.float32, device=dev); o_n = torch.empty_like(o_p)
t_end = float(t[K * B - 1])
def call():
    rp.run_stream(d[0], d[1], d[2], d[3], B, out_pos=o_p, out_neg=o_n, t_end=t_end, replay=False)
for _ in range(5):
    call(); torch.cuda.synchronize()
def med(f, n=200):
    xs = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); xs.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(xs))
whole = med(lambda: (call(), torch.cuda.synchronize()))
issue = med(call)
fast = _lib.fast()
real = fast.run_stream
class Stub:
    def __getattr__(self, k): return getattr(fast, k)
    def run_stream(self, *a): return 0, 0.0
_lib_fast = _lib.fast
_lib.fast = lambda: Stub()
py_only = med(call)
_lib.fast = _lib_fast
marks = []
class Wrap:
    def __getattr__(self, k): return getattr(fast, k)
    def run_stream(self, *a):
        t1 = time.perf_counter(); r = real(*a); marks.append((t1, time.perf_counter())); return r
w = Wrap()
_lib.fast = lambda: w
pre, inc, post = [], [], []
for _ in range(200):
    torch.cuda.synchronize(); t0 = time.perf_counter(); call(); t3 = time.perf_counter()
    t1, t2 = marks[-1]
    pre.append((t1 - t0) * 1e6); inc.append((t2 - t1) * 1e6); post.append((t3 - t2) * 1e6)
_lib.fast = _lib_fast
print(f"in place: python before the C call {np.median(pre):.1f} us, the C call (21 launches) {np.median(inc):.1f} us, python behind it {np.median(post):.1f} us")
print(f"whole call + synchronise {whole:.1f} us; call returns after {issue:.1f} us; python outside the C call {py_only:.1f} us")
