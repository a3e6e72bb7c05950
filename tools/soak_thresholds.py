#!/usr/bin/env python3
"""Developer tool: the per-batch calls from HOST arrays at sizes either side of the staging ring's thresholds (pairs per slot, batch
planned from the slot, batch staged and copied) against the same calls on device tensors: features and state bit for bit."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for d, L, N in ((128, 3, 3000), (64, 2, 500), (256, 3, 9000), (120, 3, 700)):
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    a = T._module(N, d, L, 2e-6, 1.0e6, P0=P0)
    b = T._module(N, d, L, 2e-6, 1.0e6, P0=P0)
    a._ensure_engine()
    st = a._eng["stage"]
    sizes = sorted(set([1, 2, 3, 999, 1000, 1001, st.max_batch - 1, st.max_batch, st.max_batch + 1, st.max_host_batch - 1, st.max_host_batch,
                        st.max_host_batch + 1, st.max_pairs - 1, st.max_pairs, st.max_pairs + 1, 2 * st.max_pairs + 5, 40000]))
    tnow = 1.0e6
    for n in sizes:
        u = rng.randint(0, N, n).astype(np.int64); v = rng.randint(0, N, n).astype(np.int64)
        with torch.no_grad():
            fa = a.get_pair_wise_feature(u, v)
            fb = a.get_pair_wise_feature(torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda())     # (the same self.mlp)
            ga = a.pair_gram(u, v); gb = b.pair_gram(torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda())
        assert torch.equal(ga, gb), ("pair_gram", d, n)
        ef = (fa - fb).abs().max().item() / max(1.0, fb.abs().max().item())
        assert ef <= 2e-5, ("get_pair_wise_feature", d, n, ef)           # (host arrays: readout + mlp in one launch; tensors: two)
        if n <= 40000:
            src = rng.randint(1, N, n).astype(np.int64); dst = rng.randint(1, N, n).astype(np.int64)
            t = np.sort(rng.uniform(tnow, tnow + 1e4, n)); tnow = t[-1]
            a.update(src, dst, t)
            b.run_stream(torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda(), None, torch.from_numpy(t).cuda(), n,
                         want_pos=False, want_neg=False, schedule="batch")
            la, lb = T._layers(a), T._layers(b)
            sc = max(1e-30, float(np.abs(lb).max()))
            e = float(np.abs(la - lb).max()) / sc
            assert e <= 1e-5 and float(a.now_time.item()) == float(b.now_time.item()), ("update", d, n, e)
        print(f"d={d} L={L} N={N} n={n}: features {ef:.1e} ok", flush=True)
    a.check_device_errors(); b.check_device_errors()
print("thresholds ok")
