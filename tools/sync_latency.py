#!/usr/bin/env python3
"""Wall clock of the driver's 20-step timed region (barrier + synchronize, one run_stream call of 20 batches, synchronize)
beside the HIP-event time of the same call, to split the region into device time and host-side start / wake-up latency.
Run once per runtime setting (environment variables are read by the ROCm runtime at start-up):
    python tools/sync_latency.py [--config C2] [--steps 20] [--reps 60] [--spin]
--spin: hipSetDeviceFlags(hipDeviceScheduleSpin) before the first stream call."""
import argparse, ctypes, os, statistics as st, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tpnet_amd
from tpnet_amd.stream import CONFIGS
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2"); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--reps", type=int, default=60); ap.add_argument("--spin", action="store_true")
a = ap.parse_args()
cfg = CONFIGS[a.config]; B, d = cfg["B"], cfg["d"]; K = a.steps
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
if a.spin:
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(spin) ->", hip.hipSetDeviceFlags(1))
src, dst, neg, t, N = bench.make_workload(cfg, K * (a.reps + 2), 0)
D = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, t)]
torch.manual_seed(0)
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=cfg["E"], dim_factor=10, num_layer=3, time_decay_weight=cfg["lam"],
                                      device=str(dev), use_matrix=False, beginning_time=np.float64(0.0), not_scale=False,
                                      enforce_dim=d).to(dev)
NG = 4 * (3 + 1) ** 2
op = torch.empty(K * B, NG, device=dev); on = torch.empty(K * B, NG, device=dev)
def run(i):
    sl = slice(i * K * B, (i + 1) * K * B)
    rp.run_stream(D[0][sl], D[1][sl], D[2][sl], D[3][sl], B, out_pos=op, out_neg=on, t_end=float(t[(i + 1) * K * B - 1]))
run(0); run(1)
wall, evs = [], []
for i in range(2, a.reps + 2):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(i)
    torch.cuda.synchronize()
    wall.append((time.perf_counter() - t0) * 1e6)
for i in range(2, a.reps + 2):        # same calls again with events (the state moved on; timing only)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); run(i); e1.record(); torch.cuda.synchronize()
    evs.append(e0.elapsed_time(e1) * 1e3)
# an empty region: synchronize, one trivial kernel, synchronize
x = torch.zeros(8, device=dev); emp = []
for i in range(200):
    torch.cuda.synchronize(); t0 = time.perf_counter(); x.add_(1); torch.cuda.synchronize(); emp.append((time.perf_counter() - t0) * 1e6)
env = {k: os.environ[k] for k in ("HSA_ENABLE_INTERRUPT", "ROC_ACTIVE_WAIT_TIMEOUT", "GPU_MAX_HW_QUEUES") if k in os.environ}
print("first calls (us):", [round(w, 1) for w in wall[:10]])
print(f"{env} spin={a.spin} wall us median {st.median(wall):.1f} min {min(wall):.1f} | events median {st.median(evs):.1f} | "
      f"one-kernel region median {st.median(emp):.1f} min {min(emp):.1f} | edges/s at median {K * B / st.median(wall):.1f} M")
