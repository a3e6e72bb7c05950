#!/usr/bin/env python3
"""Developer tool: random shapes of the encoder's call from HOST index arrays (the reference's tile / repeat layout, and arrays that
just miss it) through get_pair_wise_feature -- the one-crossing path (tpnet_host_anchored_features), the staged path behind it, the
general path -- against the general kernel on device ids + torch's layers.  usage: python tools/soak_encoder.py [cases] [seed]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fused_feature as F
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(cases):
    d = int(rng.choice([32, 64, 128, 128, 256, 512]))
    L = int(rng.choice([3, 3, 3, 2, 4]))
    N = int(rng.choice([50, 400, 3000]))
    K = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 20, 20, 33]))
    m = int(rng.choice([1, 2, 7, 64, 333, 1000, 2000, 3200]))
    if rng.rand() < 0.1:
        m = int(rng.randint(120000 // K, 140000 // K))                  # around the 1 MB staging slot (n/2 + 2m ids)
    kind = str(rng.choice(["pattern", "pattern", "pattern", "equal anchors", "broken tile", "broken repeat", "padding ids"]))
    grad = bool(rng.rand() < 0.25) and m * K < 20000
    rp = F._module(N, d, L)
    for src, dst, t in F._stream(rng, N, min(120, N), 3):
        rp.update(src, dst, t)
    neigh = rng.randint(0 if kind == "padding ids" else 1, N, (m, K)).astype(np.int64)
    a1, a2 = rng.randint(1, N, m).astype(np.int64), rng.randint(1, N, m).astype(np.int64)
    if kind == "equal anchors" and m > 3:
        a1[1:3] = a1[0]; a2[:] = a2[0]
    u = np.tile(neigh.reshape(-1), 2)
    v = np.concatenate([np.repeat(a1, K), np.repeat(a2, K)])
    n = u.size
    if kind == "broken tile":
        u[rng.randint(n // 2, n)] ^= 1
        u = np.clip(u, 0, N - 1)
    if kind == "broken repeat" and K > 1:
        j = rng.randint(0, n); v[j] = (v[j] % (N - 1)) + 1 if v[j] != (v[j] % (N - 1)) + 1 else 1
    ud, vd = torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()
    with torch.no_grad():
        want = rp.mlp(rp.pair_gram(ud, vd))
    if grad:
        got = rp.get_pair_wise_feature(u, v)
        assert got.requires_grad
        gy = torch.randn_like(want)
        gram = rp.pair_gram(ud, vd)
        w2 = rp.mlp(gram)
        F._assert_mlp_grads_close(rp.mlp, gram, gy, torch.autograd.grad(got, list(rp.mlp.parameters()), gy),
                                  torch.autograd.grad(w2, list(rp.mlp.parameters()), gy))
        got = got.detach()
    else:
        with torch.no_grad():
            got = rp.get_pair_wise_feature(u, v)
    rp.check_device_errors()
    err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
    worst = max(worst, err)
    if got.shape != want.shape or not err <= 2e-5:
        print(f"MISMATCH case {case}: d={d} L={L} N={N} K={K} m={m} {kind} grad={grad}: {err:.3g}", flush=True)
        sys.exit(1)
    print(f"case {case}: d={d} L={L} N={N} K={K} m={m} {kind} grad={int(grad)}: {err:.2g}", flush=True)
print("worst relative error", worst)
