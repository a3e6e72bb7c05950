#!/usr/bin/env python3
"""Developer tool: random shapes of a windowed stream cut into several chunks by a log cap (one chunk's plan region each in front of
one version log, api.hip: window_chunk) -- cold, replayed and replayed again with new negatives -- against the same stream as ONE
chunk: features and final state bit for bit.  usage: python tools/soak_chunks.py [cases] [seed]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DEV = "cuda:0"
stats = {"replayed": 0, "not_replayed": 0, "cases": 0}
for case in range(cases):
    d = int(rng.choice([16, 32, 64, 128, 256, 120, 512]))
    L = int(rng.choice([1, 2, 3, 3, 3, 4]))
    N = int(rng.choice([90, 400, 2000, 9000]))
    B = int(rng.choice([20, 64, 100, 200, 500, 1000, 1170]))
    nb = int(rng.randint(8, 120))
    packed = bool(rng.rand() < 0.25)
    want_neg = bool(rng.rand() < 0.8)
    per = int(rng.randint(4, max(5, nb)))
    E = nb * B - int(rng.randint(0, B))
    if E * L * d * 8 > 3e9:
        continue
    src, dst, neg, t = T._random_stream(rng, N, E, 4.0e5)
    neg2 = rng.randint(0, N, E).astype(np.int64)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    ds, dd, dn, dn2, dt = dev(src), dev(dst), dev(neg), dev(neg2), dev(t)

    def epoch(rp, negs, **kw):
        rp.reset_random_projections()
        rp.random_projections[0].data.copy_(torch.from_numpy(P0))
        fp, fn = rp.run_stream(ds, dd, negs if want_neg else None, dt, B, schedule="windowed", packed=packed, **kw)
        return fp.clone(), (fn.clone() if fn is not None else None), T._layers(rp), rp.last_stream_replayed

    one = T._module(N, d, L, 2e-6, t[0], P0=P0)
    c1 = epoch(one, dn, replay=False); c2 = epoch(one, dn2, replay=False)
    rp = T._module(N, d, L, 2e-6, t[0], P0=P0)
    rp.stream_log_cap_bytes = per * B * 2 * L * d * 4
    # (a cap below one window's log leaves the per-batch schedule, whose sums differ from the pipeline's in the last bits: the capped
    # stream planned afresh is the reference then; the one-chunk run wherever the chunks' plans were replayed, i.e. the pipeline ran)
    k1 = epoch(rp, dn, replay=False); k2 = epoch(rp, dn2, replay=False)
    got = [epoch(rp, dn), epoch(rp, dn2), epoch(rp, dn)]
    for g, w, wk in zip(got, (c1, c2, c1), (k1, k2, k1)):
        ok = torch.equal(g[0], wk[0]) and (g[1] is None or torch.equal(g[1], wk[1])) and np.array_equal(g[2], wk[2])
        if g[3]:
            ok = ok and torch.equal(g[0], w[0]) and (g[1] is None or torch.equal(g[1], w[1])) and np.array_equal(g[2], w[2])
        if not ok:
            print(f"MISMATCH case {case}: d={d} L={L} N={N} B={B} nb={nb} per={per} packed={packed} neg={want_neg} replayed={g[3]}", flush=True)
            sys.exit(1)
    rp.check_device_errors()
    stats["cases"] += 1
    stats["replayed" if got[1][3] and got[2][3] else "not_replayed"] += 1
    print(f"case {case}: d={d} L={L} N={N} B={B} nb={nb} cap={per} batches packed={int(packed)} neg={int(want_neg)}: "
          f"ws {rp._eng['ws'].numel() / 1e6:.1f} MB (one chunk: {one._eng['ws'].numel() / 1e6:.1f}), replayed {got[1][3]}", flush=True)
    del one, rp
print(stats)
