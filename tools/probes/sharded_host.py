#!/usr/bin/env python3
"""Probe: where the time of a 20-batch row-sharded stream goes on ONE rank (world 1 on real RCCL: no wire, every fixed cost):
cProfile of ShardedStreamRunner.run_stream, wall clock of prepare / steps / finish."""
import cProfile, os, pstats, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import tpnet_amd, bench
from tpnet_amd.stream import CONFIGS
from tpnet_amd.sharded import ShardedStreamRunner
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
c = CONFIGS["C2"]; B = c["B"]; K = 20; W = 5
src, dst, neg, t, N = bench.make_workload(c, W + K, 0)
D = lambda x: torch.from_numpy(x).to(dev)
ds, dd, dn, dt = D(src), D(dst), D(neg), D(t)
r = ShardedStreamRunner.create(node_num=N, edge_num=c["E"], dim=c["d"], num_layer=3, time_decay_weight=c["lam"], device=dev,
                               beginning_time=np.float64(0.0), halo_rows=3 * B)
r.exchange = "targeted"
r.rp._workspace(K * B, B)
def run(a, b_):
    sl = slice(a * B, b_ * B)
    t_last = t[np.minimum(np.arange(a + 1, b_ + 1) * B, len(t)) - 1]
    r.run_stream(ds[sl], dd[sl], dn[sl], dt[sl], B, t_host_last=t_last, merge_outputs=False)
for rep in range(3):
    r.rp.reset_random_projections()
    run(0, W); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(W, W + K); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"20 batches: call returns after {(t1 - t0) * 1e6:.0f} us, done after {(t2 - t0) * 1e6:.0f} us = {(t2 - t0) / K * 1e6:.1f} us per batch")
r.rp.reset_random_projections(); run(0, W); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); run(W, W + K); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
dist.destroy_process_group()
