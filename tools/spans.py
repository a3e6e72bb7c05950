#!/usr/bin/env python3
"""Developer tool: GPU-side span of every k_step launch and the gaps between launches (diagnostic STAMPS build)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TPNET_DEV_LIB", os.path.join(ROOT, "tpnet_amd", "libtpnet_hip_stamps.so"))
import tpnet_amd
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
c = CONFIGS[cfgname]; B = c["B"]; nb = 200; E = nb * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
neg = synthetic_negatives(c["U"], N, E, B, 1)
dev = torch.device("cuda:0")
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
D = lambda x: torch.from_numpy(x).to(dev)
ds, dd, dn, dt = D(src), D(dst), D(neg), D(t)
rp._workspace(E, B)
for rep in range(3):
    rp.reset_random_projections()
    ws = rp._eng["ws"][: 1 << 20].view(torch.int64)
    ws.zero_(); ws[4000 * 16: 4000 * 16 + 2 * 2048: 2] = (1 << 62)      # min slots
    rp.run_stream(ds, dd, dn, dt, B, t_end=float(t[-1]))
    torch.cuda.synchronize()
sp = rp._eng["ws"][: 1 << 20].cpu().numpy().view(np.int64)[4000 * 16: 4000 * 16 + 2 * nb].reshape(nb, 2)
span = (sp[:, 1] - sp[:, 0]) * 10
gap = (sp[1:, 0] - sp[:-1, 1]) * 10
per = (sp[1:, 0] - sp[:-1, 0]) * 10
q = lambda x: [int(np.percentile(x, p)) for p in (5, 50, 95)]
print(f"{cfgname} role_mask={os.environ.get('TPNET_DEV_ROLE_MASK', '3')}: in-kernel span ns p5/50/95 {q(span[20:])}; "
      f"gap (last wave end -> next first wave start) {q(gap[20:])}; launch period {q(per[20:])}")
