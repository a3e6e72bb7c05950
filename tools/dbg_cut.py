"""Development aid: one run_stream call vs two calls cut at a batch boundary (windowed schedule)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tpnet_amd import RandomProjectionModule
d, L, N, B, nb = [int(x) for x in sys.argv[1:6]]
rng = np.random.RandomState(d + B)
E = nb * B - B // 3
lam = 2e-6
def _random_stream(rng, N, E, span, hub_frac=0.2):
    src = rng.randint(1, N, E).astype(np.int64); dst = rng.randint(1, N, E).astype(np.int64)
    src[rng.rand(E) < hub_frac] = 1 + rng.randint(0, 3); dst[rng.rand(E) < hub_frac / 2] = 7
    dst[::17] = src[::17]
    t = np.sort(rng.uniform(1.0e6, 1.0e6 + span, E)); neg = rng.randint(0, N, E).astype(np.int64)
    return src, dst, neg, t
src, dst, neg, t = _random_stream(rng, N, E, 4.0e5)
P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
def mod():
    rp = RandomProjectionModule(node_num=N, edge_num=1000, dim_factor=10, num_layer=L, time_decay_weight=lam, device="cuda:0",
                                use_matrix=False, beginning_time=np.float64(t[0]), not_scale=False, enforce_dim=d)
    rp.random_projections[0].data = torch.from_numpy(P0)
    return rp.to("cuda:0")
dev = lambda x: torch.from_numpy(x).to("cuda:0")
ds, dd, dn, dt = dev(src), dev(dst), dev(neg), dev(t)
a = mod(); fa, na = a.run_stream(ds, dd, dn, dt, B)
c = mod(); cut = (nb // 2 + 1) * B
f1, n1 = c.run_stream(ds[:cut], dd[:cut], dn[:cut], dt[:cut], B)
lay1 = [c.random_projections[i].detach().clone() for i in range(1, L + 1)]
a1 = mod(); a1.run_stream(ds[:cut], dd[:cut], dn[:cut], dt[:cut], B)
print("first part features equal:", torch.equal(f1, fa[:cut]), torch.equal(n1, na[:cut]))
f2, n2 = c.run_stream(ds[cut:], dd[cut:], dn[cut:], dt[cut:], B)
for nm, x, y in (("pos", f2, fa[cut:]), ("neg", n2, na[cut:])):
    df = (x - y).abs()
    bad = (df > 0).nonzero()
    print(nm, "second part equal:", torch.equal(x, y), "n diff", len(bad), "max", float(df.max()), "first diffs (edge, entry):", bad[:5].tolist(),
          "batches with diffs:", sorted(set((bad[:, 0] // B).tolist()))[:10])
for i in range(L):
    x = c.random_projections[i + 1].detach(); y = a.random_projections[i + 1].detach()
    df = (x - y).abs().max(dim=1).values
    print("layer", i + 1, "rows differing:", int((df > 0).sum()), "nodes", (df > 0).nonzero().flatten()[:10].tolist())
