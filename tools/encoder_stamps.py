#!/usr/bin/env python3
"""Developer tool: wall-clock stamps (100 MHz) of the fused encoder kernel's waves (diagnostic build: make -C tpnet_amd/csrc STAMPS=1):
per producer wave when its tiles' rows arrived / were split / multiplied / published, per consumer wave when it got a tile and
when its dense-layer passes started and ended.  C2 shape, 80 000 pairs."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TPNET_DEV_LIB", os.path.join(ROOT, "tpnet_amd", "libtpnet_hip_stamps.so"))
import tpnet_amd
from tpnet_amd import _lib
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
with torch.no_grad():
    for _ in range(4):
        rp.get_pair_wise_feature_anchored(neigh, a1, a2)
torch.cuda.synchronize()
raw = np.zeros(4096 * 64, dtype=np.uint64)
cdll = C.CDLL(os.environ["TPNET_DEV_LIB"])
assert cdll.tpnet_dev_encoder_stamps(raw.ctypes.data_as(C.c_void_p), C.c_size_t(raw.nbytes)) == 0
s = raw.reshape(4096, 64).astype(np.int64)
nw = 250 * 8
s = s[:nw]
t0 = s[:, 0].min()
rel = lambda x: (x - t0) * 10            # ns
prod = s.reshape(-1, 8, 64)[:, :4].reshape(-1, 64); cons = s.reshape(-1, 8, 64)[:, 4:].reshape(-1, 64)
med = lambda x: float(np.median(x))
print("workgroups start (ns after the first): median %.0f max %.0f" % (med(rel(s[:, 0])), rel(s[:, 0]).max()))
print("weights in LDS:   median %.0f  (+%.0f after the wave's start)" % (med(rel(s[:, 1])), med(rel(s[:, 1]) - rel(s[:, 0]))))
print("producer: ids + meta of the group arrived: median %.0f" % med(rel(prod[:, 2])))
for k in range(10):
    a = [med(rel(prod[:, 4 + 4 * k + i])) for i in range(4)]
    print("producer tile %d: rows there %.0f | split %.0f | products + free buffer %.0f | published %.0f   (tile %.0f ns)" % (k, *a, a[3] - (med(rel(prod[:, 4 * k + 3])) if k else med(rel(prod[:, 2])))))
for k in range(10):
    a = [med(rel(cons[:, 4 + 4 * k + i])) for i in range(3)]
    print("consumer tile %d: got it %.0f | dense start %.0f | dense end %.0f" % (k, *a))
print("last consumer done: %.0f ns" % rel(cons[:, 4:44]).max())
