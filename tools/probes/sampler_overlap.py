#!/usr/bin/env python3
"""Probe: the encoder call's sampler kernel (7.5 us, latency-bound, reads only the neighbour store) on a SIDE stream under the
previous call's fused readout + dense-layer kernel (38 us, every CU), handed over by an event, against both on one stream.
80 000-pair calls at C2, ids resident on the device."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tpnet_amd
from tpnet_amd import _lib, fused_feature as ff
from tpnet_amd.stream import CONFIGS, synthetic_stream
from tpnet_amd.sampler import GpuRecentNeighborSampler
lib = _lib.load()
c = CONFIGS["C2"]; B = c["B"]; K = 20; E = 30 * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
dev = torch.device("cuda:0"); torch.manual_seed(0)
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
rp.run_stream(D(src), D(dst), None, D(t), B, want_neg=False, want_pos=False)
smp = GpuRecentNeighborSampler(src, dst, t, device="cuda:0", num_nodes=N)
s_ = slice(20 * B, 21 * B)
ds, do, dt = D(src[s_]), D(dst[s_]), D(t[s_])
nbytes = lib.tpnet_encoder_scratch_bytes(B, K)
scr = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
out = [torch.empty((4 * B * K, 64), device=dev) for _ in range(2)]
st = rp._state(); now = rp._now_host; lam = float(c["lam"])
prep = ff.prepared(rp.mlp, 64)
cur = torch.cuda.current_stream(); side = torch.cuda.Stream()
P = lambda s: C.c_void_p(s.cuda_stream)

def rows(i, s):
    _lib.check(lib.tpnet_encoder_rows(C.byref(st), smp._buf.data_ptr(), smp.E, smp.num_nodes, ds.data_ptr(), do.data_ptr(), dt.data_ptr(),
                                      B, K, scr[i].data_ptr(), nbytes, P(s)), "rows")
def fused(i, s):
    base = (scr[i].data_ptr() + 255) // 256 * 256
    a1 = base + 4 * B * 8; a2 = a1 + 2 * B * 8; ng = a2 + 2 * B * 8
    _lib.check(lib.tpnet_anchored_features(C.byref(st), ng, a1, a2, 2 * B, K, now, lam, 0, prep[2], None, out[i].data_ptr(), P(s)), "fused")

def wall(fn, reps=60):
    for i in range(6): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps): fn(i)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t2 - t0) / reps * 1e6, (t1 - t0) / reps * 1e6

def one_stream(i):
    rows(i & 1, cur); fused(i & 1, cur)
ready = [torch.cuda.Event() for _ in range(2)]
done = [torch.cuda.Event() for _ in range(2)]
for e in done: e.record(cur)
def two_streams(i):
    j = i & 1
    side.wait_event(done[j])          # the fused kernel that read this scratch buffer two calls ago
    rows(j, side)
    ready[j].record(side)
    cur.wait_event(ready[j])
    fused(j, cur)
    done[j].record(cur)
def two_streams_no_done(i):           # (what the waits on `done` cost: unsafe, timing only)
    j = i & 1
    rows(j, side)
    ready[j].record(side)
    cur.wait_event(ready[j])
    fused(j, cur)
def fused_only(i):
    fused(i & 1, cur)
rows(0, cur); rows(1, cur); torch.cuda.synchronize()
for name, fn in (("fused kernel alone", fused_only), ("sampler + fused on one stream", one_stream), ("sampler on a side stream", two_streams),
                 ("the same without the reuse events", two_streams_no_done), ("sampler + fused on one stream", one_stream)):
    w, h = wall(fn)
    print(f"{name:40s} {w:6.1f} us per call (host issue {h:.1f})")
