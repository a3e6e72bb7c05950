import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import test_gpu_parity as T
d, L, N, B, nb, per = 64, 2, 3000, 200, 150, 60
rng = np.random.RandomState(d + B + nb)
E = nb * B - B // 3
src, dst, neg, t = T._random_stream(rng, N, E, 4.0e5)
P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
dev = lambda x: torch.from_numpy(x).to("cuda:0")
ds, dd, dn, dt = dev(src), dev(dst), dev(neg), dev(t)
rp = T._module(N, d, L, 2e-6, t[0], P0=P0)
rp.stream_log_cap_bytes = per * B * 2 * L * d * 4
for i in range(3):
    rp.reset_random_projections()
    rp.random_projections[0].data.copy_(torch.from_numpy(P0))
    rp.run_stream(ds, dd, dn, dt, B, schedule="windowed")
    tag = rp._plan_tag
    print(i, rp.last_stream_replayed, "ws", rp._eng["ws"].numel(), "built", [int(x) for x in tag.built][:20], "sigs", tag.table_sig, tag.stream_sig)
