#!/usr/bin/env python3
"""Developer tool: where the row-sharded runner's time per step goes (single rank, collectives forced)."""
import os, sys, time
os.environ.setdefault("TPNET_DEV_FORCE_COLLECTIVES", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tpnet_amd.sharded import ShardedStreamRunner
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
c = CONFIGS["C2"]; B = c["B"]; nb = 400; E = nb * B
src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
neg = synthetic_negatives(c["U"], N, E, B, 1)
D = lambda x: torch.from_numpy(x).to(dev)
ds, dd, dn, dt = D(src), D(dst), D(neg), D(t)
run = ShardedStreamRunner.create(node_num=N, edge_num=c["E"], dim=c["d"], num_layer=3, time_decay_weight=c["lam"], device=dev,
                                 beginning_time=np.float64(0.0), halo_rows=3 * B)
t_last = t[np.arange(1, nb + 1) * B - 1]
for exchange in ("allgather", "targeted"):
    run.exchange = exchange
    for rep in range(3):
        run.rp.reset_random_projections()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        R = run.relabel(ds, dd, dn, B) if exchange == "allgather" else run.relabel_targeted(ds, dd, dn, B)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        run.run_stream(ds, dd, dn, dt, B, t_host_last=t_last, merge_outputs=False)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{exchange} rep {rep}: relabel alone {1e3 * (t1 - t0):.2f} ms per call ({1e6 * (t1 - t0) / nb:.1f} us/step); run_stream (incl. its own "
              f"relabel) {1e6 * (t2 - t1) / nb:.1f} us/step -> loop alone {1e6 * ((t2 - t1) - (t1 - t0)) / nb:.1f} us/step", flush=True)
# the plain per-batch stream on the same module shape, for the ratio
import tpnet_amd
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
for rep in range(3):
    rp.reset_random_projections()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rp.run_stream(ds, dd, dn, dt, B, t_end=float(t[-1]), schedule="batch")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"plain per-batch stream rep {rep}: {1e6 * (t1 - t0) / nb:.1f} us/step", flush=True)
print("C-side RCCL communicator:", "yes" if getattr(run, "_comm", None) else "no (torch.distributed collective)")
run.close()
dist.destroy_process_group()
