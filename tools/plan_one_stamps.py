#!/usr/bin/env python3
"""Developer tool: phase stamps of k_plan_one (diagnostic build: make -C tpnet_amd/csrc STAMPS=1), host-array update calls."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TPNET_DEV_LIB", os.path.join(ROOT, "tpnet_amd", "libtpnet_hip_stamps.so"))
import tpnet_amd, bench
from tpnet_amd.stream import CONFIGS
c = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"]; B = c["B"]
src, dst, neg, t, N = bench.make_workload(c, 40, 0)
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to("cuda:0")
names = (["start", "edges loaded, weights in LDS", "sorted", "sorted keys in LDS", "items written", "descriptor written"]
         if os.environ.get("TPNET_DEV_PLAN_ONE_SORT") else
         ["start", "edges loaded, weights in LDS", "grouped (hash table)", "spans scanned, members listed", "ranked, sorted arrays written",
          "items written"])
acc = []
for b in range(30):
    s = slice(b * B, (b + 1) * B)
    rp.update(src[s], dst[s], t[s])
    torch.cuda.synchronize()
    raw = rp._eng["ws"][:96].cpu().numpy().view(np.uint64).reshape(6, 2).astype(np.int64)
    if b >= 5:
        acc.append((raw[:, 1] - raw[0, 1]) * 10)       # 100 MHz wall clock -> ns
a = np.median(np.array(acc), axis=0)
for i, nm in enumerate(names):
    print(f"{nm:32s} {a[i]:8.0f} ns  (+{a[i] - (a[i - 1] if i else 0):.0f})")
