#!/usr/bin/env python3
"""Developer tool: tpnet_update (device-resident ids: k_plan_one reads HBM) next to rp.update (host arrays: k_plan_one reads
the pinned staging slot), run under `rocprofv3 --kernel-trace --stats` to compare the plan kernel's duration."""
import os, sys, time, ctypes as C
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpnet_amd
from tpnet_amd import _lib
from tpnet_amd.stream import CONFIGS, synthetic_stream
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
mode = sys.argv[2] if len(sys.argv) > 2 else "dev"
c = CONFIGS[cfg]; B = c["B"]; nb = 200; E = nb * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to("cuda:0")
rp.update(src[:B], dst[:B], t[:B])
lib = _lib.load()
ds, dd, dt = (torch.from_numpy(x).cuda() for x in (src, dst, t))
ws = rp._workspace(B, B)
torch.cuda.synchronize(); t0 = time.perf_counter()
for b in range(1, nb):
    s = slice(b * B, (b + 1) * B)
    if mode == "dev":
        lid = rp._next_launch_ids(1)
        _lib.check(lib.tpnet_update(rp._st_ref(), ds[s].data_ptr(), dd[s].data_ptr(), dt[s].data_ptr(), B, rp._now_host,
                                    c["lam"], lid, 0, ws.data_ptr(), ws.numel(), None))
        rp._now_host = float(t[(b + 1) * B - 1])
    else:
        rp.update(src[s], dst[s], t[s])
torch.cuda.synchronize()
print(f"{cfg} {mode}: {(time.perf_counter() - t0) / (nb - 1) * 1e6:.1f} us per update")
