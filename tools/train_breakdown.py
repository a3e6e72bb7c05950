#!/usr/bin/env python3
"""Developer tool: where a decoder-level TRAINING step's time goes (C2): features with grad, update, loss, backward, Adam step --
each phase closed by a synchronise (so the sum exceeds the pipelined loop), and the loop as bench.py's dropin.train runs it."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tpnet_amd, bench
from tpnet_amd.stream import CONFIGS
c = CONFIGS["C2"]; B = c["B"]; nb = 40
src, dst, neg, t, N = bench.make_workload(c, nb + 4, 0)
dev = torch.device("cuda:0")
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
fused = len(sys.argv) > 1 and sys.argv[1] == "fused"
opt = torch.optim.Adam(rp.mlp.parameters(), lr=1e-4, fused=fused)
labels = torch.cat([torch.ones(B, device=dev), torch.zeros(B, device=dev)])
lossf = torch.nn.BCEWithLogitsLoss()
acc = {k: [] for k in ("features", "update", "loss", "zero_grad", "backward", "step")}
def tick():
    torch.cuda.synchronize(); return time.perf_counter()
for b in range(nb):
    s = slice(b * B, (b + 1) * B)
    t0 = tick()
    f1 = rp.get_pair_wise_feature(src[s], dst[s]); f2 = rp.get_pair_wise_feature(src[s], neg[s])
    t1 = tick()
    rp.update(src[s], dst[s], t[s])
    t2 = tick()
    loss = lossf(torch.cat([f1.sum(1), f2.sum(1)]), labels)
    t3 = tick()
    opt.zero_grad()
    t4 = tick()
    loss.backward()
    t5 = tick()
    opt.step()
    t6 = tick()
    if b >= 5:
        for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
            acc[k].append(v * 1e6)
print("Adam fused" if fused else "Adam (default)", {k: round(float(np.median(v)), 1) for k, v in acc.items()}, "sum", round(sum(float(np.median(v)) for v in acc.values()), 1))
# the loop without the per-phase synchronise
def loop(train):
    rp.reset_random_projections()
    for b in range(nb):
        if b == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        s = slice(b * B, (b + 1) * B)
        if train:
            f1 = rp.get_pair_wise_feature(src[s], dst[s]); f2 = rp.get_pair_wise_feature(src[s], neg[s])
            rp.update(src[s], dst[s], t[s])
            loss = lossf(torch.cat([f1.sum(1), f2.sum(1)]), labels)
            opt.zero_grad(); loss.backward(); opt.step()
        else:
            with torch.no_grad():
                rp.get_pair_wise_feature(src[s], dst[s]); rp.get_pair_wise_feature(src[s], neg[s])
                rp.update(src[s], dst[s], t[s])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (nb - 5) * 1e6
print("loop: no_grad %.1f us per batch, train %.1f us per batch" % (loop(False), loop(True)))
# the same training step on plain torch layers (what the reference's module would pay for loss + backward + step on this GPU)
x = torch.randn(B, 64, device=dev)
def torch_only():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        f1 = rp.mlp(x); f2 = rp.mlp(x)
        loss = lossf(torch.cat([f1.sum(1), f2.sum(1)]), labels)
        opt.zero_grad(); loss.backward(); opt.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e6
torch_only()
print("torch layers on resident features + the same loss / backward / step: %.1f us per step" % torch_only())
