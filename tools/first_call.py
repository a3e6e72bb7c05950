#!/usr/bin/env python3
"""Developer tool: the first-call cost of a fresh process.  bench.py's flow (module, W=5 warm-up batches as two calls, K=20
timed) three times in ONE process; WARM="launches:spin_us" calls tpnet_runtime_warmup first (0:0 = off)."""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpnet_amd
from tpnet_amd import _lib
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
c = CONFIGS["C2"]; B = c["B"]; W, K = 5, int(os.environ.get("K", "20"))
E = (W + K) * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
neg = synthetic_negatives(c["U"], N, E, B, 1)
dev = torch.device("cuda:0")
D = lambda x: torch.from_numpy(x).to(dev)
ds, dd, dn, dt = D(src), D(dst), D(neg), D(t)
nl, spin = [int(x) for x in os.environ.get("WARM", "0:0").split(":")]
if nl or spin:
    _lib.check(_lib.load().tpnet_runtime_warmup(nl, spin, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "warm")
    torch.cuda.synchronize()
out = []
for trial in range(3):
    rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
            device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
    op = torch.empty((K * B, 64), device=dev); on = torch.empty((K * B, 64), device=dev)
    rp._workspace(K * B, B, stream=True).zero_(); op.zero_(); on.zero_()
    def run(a, b):
        sl = slice(a * B, b * B)
        rp.run_stream(ds[sl], dd[sl], dn[sl], dt[sl], B, out_pos=op[:(b - a) * B], out_neg=on[:(b - a) * B], t_end=float(t[b * B - 1]))
    run(0, W // 2); run(W // 2, W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(W, W + K)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) * 1e6)
print(f"WARM={nl}:{spin} K={K}: timed call of trial 0/1/2: " + " / ".join(f"{x:.0f}" for x in out) + " us", flush=True)
