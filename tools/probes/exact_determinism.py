"""Is a stream bit-reproducible run to run?  exact / lazy mode, several batch sizes, the per-batch schedule."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_parity as T
DEV = T.DEV
for exact in (True, False):
    for (d, L, N, B, nb) in ((64, 2, 500, 1000, 7), (64, 2, 500, 2048, 7), (64, 2, 500, 2049, 7), (64, 2, 500, 2500, 7), (128, 3, 3000, 5000, 5)):
        rng = np.random.RandomState(1)
        E = nb * B
        src, dst, neg, t = T._random_stream(rng, N, E, 4.0e5)
        P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
        dev = lambda x: torch.from_numpy(x).to(DEV)
        ds, dd, dn, dt = dev(src), dev(dst), dev(neg), dev(t)
        rp = T._module(N, d, L, 2e-6, t[0], P0=P0, exact=exact)
        outs = []
        for rep in range(3):
            rp.reset_random_projections()
            rp.random_projections[0].data.copy_(torch.from_numpy(P0))
            fp, fn = rp.run_stream(ds, dd, dn, dt, B, schedule="batch", replay=False)
            outs.append((fp.clone(), fn.clone(), T._layers(rp)))
        for k in (1, 2):
            bp = (outs[k][0] != outs[0][0]).any(dim=1).nonzero().flatten()
            print(f"exact={exact} B={B} N={N} d={d}: run {k} vs 0: pos rows differ {bp.numel()} first {bp[:3].tolist()} state equal {bool((outs[k][2] == outs[0][2]).all())}", flush=True)
