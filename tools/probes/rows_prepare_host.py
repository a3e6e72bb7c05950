#!/usr/bin/env python3
"""Where the host time of ONE cold sharded run_stream call goes with one rank (the forced one-rank run of bench.py's N > 1 path):
pieces of ShardedStreamRunner.prepare_targeted / steps_targeted timed one by one, then the whole call, wall clock."""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tpnet_amd.sharded import ShardedStreamRunner
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives

cfg = CONFIGS["C2"]; B = cfg["B"]; nbt = 20; W = 5
DIST = os.environ.get("PROBE_DIST") == "1"        # with a one-rank RCCL process group and a barrier in front of every call, as bench.py's forced run
if DIST:
    import torch.distributed as dist
    os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
src, dst, t, N = synthetic_stream(cfg["U"], cfg["I"], (W + nbt) * 40 * B, cfg["span"] * 0.3, seed=0)
neg = synthetic_negatives(cfg["U"], N, len(src), B)
dev = torch.device("cuda:0")
D = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, t)]
r = ShardedStreamRunner.create(node_num=N, edge_num=cfg["E"], dim=cfg["d"], num_layer=3, time_decay_weight=cfg["lam"], device=dev,
                               beginning_time=np.float64(0.0), halo_rows=3 * B, world=1, rank=0)
r.exchange = "targeted"
rp = r.rp
rp._ensure_engine(); rp.reserve_stream(nbt * B, B)
op = torch.zeros((nbt * B, 64), device=dev); on = torch.zeros_like(op)

def call(k):
    a = k * nbt * B
    sl = slice(a, a + nbt * B)
    tl = t[np.arange(1, nbt + 1) * B + a - 1]
    if DIST:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.run_stream(D[0][sl], D[1][sl], D[2][sl], D[3][sl], B, t_host_last=tl, merge_outputs=False, out_pos=op, out_neg=on)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) * 1e6, (t2 - t0) * 1e6

for k in range(12):
    rp.reset_random_projections()
    h, w = call(k)
    print(f"cold call {k}: host {h:.0f} us, wall {w:.0f} us")
# the same stream again: everything kept
for k in range(4):
    rp.reset_random_projections()
    h, w = call(11)
    print(f"same stream again: host {h:.0f} us, wall {w:.0f} us")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for k in range(12, 32):
    rp.reset_random_projections()
    call(k)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
