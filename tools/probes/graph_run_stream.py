#!/usr/bin/env python3
"""Developer probe: the per-batch schedule's 21 launches of a 20-batch C2 call, issued on a stream against captured ONCE and replayed as a
hipGraph (torch.cuda.CUDAGraph around the same C call).  The replay repeats launch ids, so its RESULTS are not the stream's -- only
its timing is looked at: what a graph would save on the chain of dependent launches."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tpnet_amd, bench
from tpnet_amd.stream import CONFIGS
c = CONFIGS["C2"]; B = c["B"]; K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
src, dst, neg, t, N = bench.make_workload(c, K + 5, 0)
dev = torch.device("cuda:0")
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
d = [torch.from_numpy(np.ascontiguousarray(x[:K * B])).to(dev) for x in (src, dst, neg, t)]
NG = rp.pair_wise_feature_dim
o_p = torch.empty((K * B, NG), dtype=torch.float32, device=dev); o_n = torch.empty_like(o_p)
t_end = float(t[K * B - 1])
def call():
    rp.run_stream(d[0], d[1], d[2], d[3], B, out_pos=o_p, out_neg=o_n, t_end=t_end, replay=False, schedule="batch")
for _ in range(5):
    call(); torch.cuda.synchronize()
def med(f, n=100):
    xs = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); xs.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(xs)), float(np.min(xs))
print("stream: median %.1f us, min %.1f us" % med(call))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    call(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        call()
torch.cuda.synchronize()
for _ in range(5):
    g.replay(); torch.cuda.synchronize()
print("graph : median %.1f us, min %.1f us" % med(g.replay))
