#!/usr/bin/env python3
"""Developer tool: phase stamps of the fused pair-feature kernel (diagnostic build: make -C tpnet_amd/csrc STAMPS=1; in that build
out_gram is the stamp buffer), host-array calls of 1000 pairs at C2."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TPNET_DEV_LIB", os.path.join(ROOT, "tpnet_amd", "libtpnet_hip_stamps.so"))
import tpnet_amd, bench
from tpnet_amd import _lib, fused_feature as ff
from tpnet_amd.stream import CONFIGS
c = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"]; B = c["B"]
src, dst, neg, t, N = bench.make_workload(c, 40, 0)
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to("cuda:0")
for b in range(10):
    s = slice(b * B, (b + 1) * B); rp.update(src[s], dst[s], t[s])
lib = _lib.load()
prep = ff.prepared(rp.mlp, 64)
dbg = torch.zeros(4000 * 8 * 2, dtype=torch.int64, device="cuda:0")
out = torch.empty((B, 64), device="cuda:0")
names = ["start", "readout done (features in LDS)", "weights issued + barrier", "layer 1 + barrier", "layer 2 + barrier", "outputs stored + barrier"]
res = []
for b in range(10, 30):
    s = slice(b * B, (b + 1) * B)
    dbg.zero_(); torch.cuda.synchronize()
    rc = lib.tpnet_host_pair_feature(rp._st_ref(), rp._eng["stage"].handle, src[s].ctypes.data, dst[s].ctypes.data, B, rp._now_host,
                                     float(rp.time_decay_weight), 0, prep[2], dbg.data_ptr(), out.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    raw = dbg.cpu().numpy().reshape(-1, 8, 2)
    wall = raw[:, :, 1]
    used = wall[:, 0] > 0
    t0 = wall[used, 0].min()
    res.append([(np.median(wall[used, k]) - t0) * 10 for k in range(6)] + [(wall[used, :6].max() - t0) * 10, used.sum()])
a = np.median(np.array(res), axis=0)
print(f"waves stamped {a[7]:.0f}; first wave start -> last stamp {a[6]:.0f} ns")
for i, nm in enumerate(names):
    print(f"{nm:36s} median over waves {a[i]:8.0f} ns  (+{a[i] - (a[i - 1] if i else 0):.0f})")
