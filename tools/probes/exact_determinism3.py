"""Eager decay differs run to run at d=64, B=1000, N=500: which launch property?  (developer build: TPNET_DEV_* knobs from the env)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_parity as T
from tpnet_amd import _lib
DEV = T.DEV
mode = sys.argv[1]
exact = mode != "lazy"
if mode == "eager":
    _lib.FLAG_SEQUENTIAL = 0
d, L, N, B, nb = 64, 2, 500, 1000, 7
rng = np.random.RandomState(1)
E = nb * B
src, dst, neg, t = T._random_stream(rng, N, E, 4.0e5)
P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
dev = lambda x: torch.from_numpy(x).to(DEV)
ds, dd, dn, dt = dev(src), dev(dst), dev(neg), dev(t)
rp = T._module(N, d, L, 2e-6, t[0], P0=P0, exact=exact)
outs = []
for rep in range(6):
    rp.reset_random_projections()
    rp.random_projections[0].data.copy_(torch.from_numpy(P0))
    kw = dict(want_pos=False, want_neg=False) if os.environ.get("NO_READOUT") else {}
    rp.run_stream(ds, dd, (None if os.environ.get("NO_READOUT") else dn), dt, B, schedule="batch", replay=False, **kw)
    outs.append(T._layers(rp))
print(mode, {k: v for k, v in os.environ.items() if (k.startswith("TPNET_DEV_") and k != "TPNET_DEV_LIB") or k == "NO_READOUT"},
      "state equal to run 0:", [bool((o == outs[0]).all()) for o in outs[1:]],
      "rows differing:", [int((o != outs[0]).any(axis=(0, 2)).sum()) for o in outs[1:]], flush=True)
