#!/usr/bin/env python3
"""Developer probe: what the FIRST 20-batch call of a fresh process pays over the steady state (the driver's own run: 5 warm-up
steps as four short calls, then the timed call).  One process per setting:
    python tools/probes/fresh_process_call.py [spin_us]     spin_us > 0: the GPU kept busy that long right before the warm-up steps
Prints the wall clock of the timed call, then of five more calls of the same shape (the steady state)."""
import os, sys, time, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tpnet_amd, bench
from tpnet_amd import _lib
from tpnet_amd.stream import CONFIGS
spin = int(sys.argv[1]) if len(sys.argv) > 1 else 0
c = CONFIGS["C2"]; B = c["B"]; K = 20; W = 5
src, dst, neg, t, N = bench.make_workload(c, 8 * K + W, 0)
dev = torch.device("cuda:0")
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (src, dst, neg, t)]
NG = rp.pair_wise_feature_dim
o_p = torch.empty((K * B, NG), dtype=torch.float32, device=dev); o_n = torch.empty_like(o_p)
rp.reserve_stream(K * B, B); rp._eng["ws"].zero_(); o_p.zero_(); o_n.zero_()
import gc; gc.collect(); gc.disable()
def call(a, b):
    sl = slice(a * B, b * B)
    return (d[0][sl], d[1][sl], d[2][sl], d[3][sl], o_p[:(b - a) * B], o_n[:(b - a) * B], float(t[b * B - 1]))
def run(p):
    rp.run_stream(p[0], p[1], p[2], p[3], B, out_pos=p[4], out_neg=p[5], t_end=p[6])
time.sleep(0.5)                                  # (the GPU idles, as it does while the driver's run builds its workload)
if spin:
    _lib.check(_lib.load().tpnet_runtime_warmup(0, spin, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "spin")
for a, b in ((0, 1), (1, 2), (2, 3), (3, 5)):
    run(call(a, b))
out = []
for r in range(6):
    p = call(W + r * K, W + (r + 1) * K)
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(p); torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 1e6)
print(f"spin {spin} us: timed call {out[0]:.1f} us; the next five {' '.join('%.1f' % x for x in out[1:])}")
