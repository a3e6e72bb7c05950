"""Development aid: windowed run_stream against the oracle on a small random stream, printing where they differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tpnet_oracle as O
from tpnet_amd import RandomProjectionModule

d, L, N, B = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (64, 3, 300, 50))]
nb = int(sys.argv[5]) if len(sys.argv) > 5 else 5
rng = np.random.RandomState(1)
E = nb * B - B // 3
lam = 2e-6
src = rng.randint(1, N, E).astype(np.int64); dst = rng.randint(1, N, E).astype(np.int64)
src[rng.rand(E) < 0.2] = 1 + rng.randint(0, 3); dst[rng.rand(E) < 0.1] = 7
neg = rng.randint(0, N, E).astype(np.int64)
t = np.sort(rng.uniform(1.0e6, 1.4e6, E))
P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
rp = RandomProjectionModule(node_num=N, edge_num=1000, dim_factor=10, num_layer=L, time_decay_weight=lam, device="cuda:0",
                            use_matrix=False, beginning_time=np.float64(t[0]), not_scale=False, enforce_dim=d)
rp.random_projections[0].data = torch.from_numpy(P0)
rp = rp.to("cuda:0")
dev = lambda x: torch.from_numpy(x).to("cuda:0")
fp, fn = rp.run_stream(dev(src), dev(dst), dev(neg), dev(t), B)
fp = fp.cpu().numpy(); fn = fn.cpu().numpy()
st = O.OracleState(P0, L, lam, t[0])
for o in range(0, E, B):
    s = slice(o, min(o + B, E))
    for name, got, v in (("pos", fp[s], dst[s]), ("neg", fn[s], neg[s])):
        want = O.pair_gram(st, src[s], v)
        err = np.abs(got - want)
        bad = err > 1e-4 * np.abs(want) + 1e-4
        print(f"batch {o // B} {name}: max err {err.max():.3e}, bad {int(bad.sum())}/{bad.size}",
              ("first bad (pair, entry): %s got %.4f want %.4f" % (np.argwhere(bad)[0], got[bad][0], want[bad][0])) if bad.any() else "")
    O.update(st, src[s], dst[s], t[s])
for i in range(1, L + 1):
    got = rp.random_projections[i].detach().cpu().numpy()
    err = np.abs(got - st.P[i]); sc = np.abs(st.P[i]).max()
    print(f"layer {i}: max err {err.max():.3e} (scale {sc:.3e}), rows off: {int((err.max(axis=1) > 1e-4 * sc).sum())}")
rp.check_device_errors()
if os.environ.get("DBG_DUMP"):
    st0 = O.OracleState(P0, L, lam, t[0])
    np.set_printoptions(precision=3, linewidth=200, suppress=True)
    for pr in (0, 1):
        print("pair", pr, "src", src[pr], "dst", dst[pr], "neg", neg[pr])
        print("got pos\n", fp[pr].reshape(2 * L + 2, -1)); print("want\n", O.pair_gram(st0, src[pr:pr + 1], dst[pr:pr + 1])[0].reshape(2 * L + 2, -1))
        print("got neg\n", fn[pr].reshape(2 * L + 2, -1)); print("want\n", O.pair_gram(st0, src[pr:pr + 1], neg[pr:pr + 1])[0].reshape(2 * L + 2, -1))
