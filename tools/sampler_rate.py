#!/usr/bin/env python3
"""Developer tool: recent-neighbour sampling, device kernel vs the host loop (per query batch of 2B nodes, K=20)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tpnet_amd.sampler import GpuRecentNeighborSampler
from tpnet_amd.callers import RecentNeighborSampler
from tpnet_amd.stream import CONFIGS, synthetic_stream
for cfg in ("C2", "C3"):
    c = CONFIGS[cfg]; B = c["B"]; E = c["E"]
    src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"], 0)
    t0 = time.perf_counter(); gpu = GpuRecentNeighborSampler(src, dst, t); torch.cuda.synchronize(); tb = time.perf_counter() - t0
    t0 = time.perf_counter(); host = RecentNeighborSampler(src, dst, t); th = time.perf_counter() - t0
    s = slice(E - B, E)
    nodes = np.concatenate([src[s], dst[s]]); times = np.tile(t[s], 2)
    dn, dt = torch.from_numpy(nodes).cuda(), torch.from_numpy(times).cuda()
    gpu.sample_device(dn, dt, 20); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): gpu.sample_device(dn, dt, 20)
    e1.record(); torch.cuda.synchronize()
    tg = e0.elapsed_time(e1) / 50 * 1e3
    t0 = time.perf_counter(); host.get_historical_neighbors(nodes, times, 20); thq = (time.perf_counter() - t0) * 1e6
    print(f"{cfg}: E={E} build device {tb * 1e3:.1f} ms / host {th * 1e3:.1f} ms; query 2B={2 * B} nodes x K=20: "
          f"device {tg:.1f} us, host loop {thq:.0f} us -> {thq / tg:.0f}x")
