#!/usr/bin/env python3
"""Compute side of the row-sharded stream at G = 1, 2, 4, 8 on ONE GPU (the pool has no multi-GPU box): the G shards of the C2 table
live in one process and run ONE AFTER THE OTHER, so every figure is what ONE rank's GPU does, uncontended -- the plan of a call
(tpnet_wshard_plan: wall clock incl. its two host synchronisations), the pipeline's launches, the pack / unpack launches, and the
rows a rank sends / receives per launch -- on the windowed shard (csrc/wshard.hip) and on the per-batch shard (csrc/rows_rccl.hip)
for the same stream.  Rows move by plain copies between the shards' buffers (no wire time: that is the part a one-GPU box cannot
measure).  usage: python tools/wshard_rates.py [batches] [config]"""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tpnet_amd import _lib
from tpnet_amd.sharded import ShardedStreamRunner
from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives, bytes_per_edge

nbt = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "C2"]
B1, d, L = cfg["B"], cfg["d"], 3
dev = torch.device("cuda:0")
lib = _lib.load()
PH_LAUNCH, PH_PACK, PH_UNPACK = 1, 2, 8
ev = lambda: torch.cuda.Event(enable_timing=True)


def run(G, reps=4):
    Bg = B1 * G
    E = nbt * Bg
    src, dst, t, N = synthetic_stream(cfg["U"], cfg["I"], E, cfg["span"] * E / cfg["E"], seed=0)
    neg = synthetic_negatives(cfg["U"], N, E, Bg, seed=1)
    D = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, t)]
    runs = [ShardedStreamRunner.create(node_num=N, edge_num=cfg["E"], dim=d, num_layer=L, time_decay_weight=cfg["lam"], device=dev,
                                       beginning_time=np.float64(0.0), halo_rows=max(3 * Bg, N), seed=r, world=G, rank=r) for r in range(G)]
    outs = [(torch.zeros((E, 64), device=dev), torch.zeros((E, 64), device=dev)) for _ in range(G)]
    res = []
    for rep in range(reps):
        for r in range(G):
            runs[r].rp.reset_random_projections()
        torch.cuda.synchronize()
        plan_us, plans = [], []
        for r in range(G):
            t0 = time.perf_counter()
            p = runs[r].plan_windowed(*D, Bg)
            torch.cuda.synchronize()
            plan_us.append((time.perf_counter() - t0) * 1e6)
            assert p is not None, "declined"
            plans.append(p)
        nst = plans[0]["nsteps"]
        k_us = np.zeros((G, nst)); x_us = np.zeros((G, nst))
        for r in range(G):
            _lib.check(lib.tpnet_wshard_begin(plans[r]["handle"], None, PH_PACK, runs[r].rp._stream()), "begin")
        for r in range(G):
            rp = runs[r].rp
            p0_t = rp._plist()[0].data
            q_t = rp._eng["q"].view(2, rp.node_num, L * d)
            for o in range(G):
                c = int(plans[r]["chunk_cnt"][o])
                if o != r and c:
                    a0 = runs[r].n_cap + int(plans[r]["hstart"][o])
                    p0_t[a0:a0 + c].copy_(plans[o]["bufs"]["send_p0"][:c])
                    q_t[0, a0:a0 + c].copy_(plans[o]["bufs"]["send_q"][:c])
        for j in range(nst):
            for r in range(G):
                e0, e1, e2 = ev(), ev(), ev()
                e0.record()
                _lib.check(lib.tpnet_wshard_step(plans[r]["handle"], None, j, PH_LAUNCH, outs[r][0].data_ptr(), outs[r][1].data_ptr(), runs[r].rp._stream()), "step")
                e1.record()
                _lib.check(lib.tpnet_wshard_step(plans[r]["handle"], None, j, PH_PACK, outs[r][0].data_ptr(), outs[r][1].data_ptr(), runs[r].rp._stream()), "step")
                e2.record()
                torch.cuda.synchronize()
                k_us[r, j] = e0.elapsed_time(e1) * 1e3
                x_us[r, j] = e1.elapsed_time(e2) * 1e3
            for r in range(G):
                ro = 0
                for o in range(G):
                    c = int(plans[r]["recv_cnt"][j][o])
                    if c:
                        a0 = int(plans[o]["send_cnt"][j][:r].sum())
                        plans[r]["bufs"]["recvbuf"][ro:ro + c].copy_(plans[o]["bufs"]["sendbuf"][a0:a0 + c])
                    ro += c
            for r in range(G):
                e0, e1 = ev(), ev()
                e0.record()
                _lib.check(lib.tpnet_wshard_step(plans[r]["handle"], None, j, PH_UNPACK, outs[r][0].data_ptr(), outs[r][1].data_ptr(), runs[r].rp._stream()), "step")
                e1.record()
                torch.cuda.synchronize()
                x_us[r, j] += e0.elapsed_time(e1) * 1e3
        for r in range(G):
            _lib.check(lib.tpnet_wshard_finish(plans[r]["handle"], runs[r].rp._next_launch_ids(1), runs[r].rp._stream()), "finish")
        torch.cuda.synchronize()
        sent = np.array([p["send_cnt"].sum() for p in plans], dtype=np.float64)
        halo = np.array([p["halo"] for p in plans], dtype=np.float64)
        for r in range(G):
            lib.tpnet_wshard_destroy(plans[r]["handle"])
            rp = runs[r].rp
            rp._now_host = float(t[-1]); rp._params_valid = False; rp._now_dirty = True; rp._table_written()
        if rep >= 1:
            res.append((np.max(plan_us), k_us.sum(axis=1).max(), x_us.sum(axis=1).max(), nst, sent.max(), halo.max(), k_us.max(axis=0)))
    med = lambda i: float(np.median([x[i] for x in res]))
    nst = res[0][3]
    per_launch = np.median(np.stack([x[6] for x in res]), axis=0)
    row = dict(G=G, Bg=Bg, launches=nst, plan_us=med(0), pipeline_us=med(1), pack_unpack_us=med(2), rows_sent=med(4), halo_rows=med(5),
               per_launch=" ".join(f"{x:.0f}" for x in per_launch))
    gpu = row["pipeline_us"] + row["pack_unpack_us"]
    row["edges_per_s_compute"] = nbt * Bg / ((gpu + row["plan_us"]) * 1e-6)
    row["edges_per_s_kernels"] = nbt * Bg / (gpu * 1e-6)
    # wire: what travels per rank per call, at the per-link rate of xGMI (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU; a rank's
    # G - 1 peers share its links evenly -> at most min(G - 1, 7) links busy)
    bytes_call = row["rows_sent"] * d * 4 + row["halo_rows"] * (L + 1) * d * 4
    links = max(1, min(G - 1, 7))
    row["wire_us_at_xgmi"] = bytes_call / (links * 153e9) * 1e6 if G > 1 else 0.0
    return row


def run_batch(G, reps=4):
    """The same stream on the PER-BATCH shard (tpnet_pack_split + tpnet_step_batch per batch, rows moved by plain copies): the plan
    of a call (exchange plan + per-batch plan of the kernels, wall clock) and the launches, per rank."""
    Bg = B1 * G
    E = nbt * Bg
    src, dst, t, N = synthetic_stream(cfg["U"], cfg["I"], E, cfg["span"] * E / cfg["E"], seed=0)
    neg = synthetic_negatives(cfg["U"], N, E, Bg, seed=1)
    D = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, t)]
    runs = [ShardedStreamRunner.create(node_num=N, edge_num=cfg["E"], dim=d, num_layer=L, time_decay_weight=cfg["lam"], device=dev,
                                       beginning_time=np.float64(0.0), halo_rows=3 * Bg, seed=r, world=G, rank=r) for r in range(G)]
    for r_ in runs:
        r_.reuse_plans = False
    res = []
    nb = nbt
    for rep in range(reps):
        for r in range(G):
            runs[r].rp.reset_random_projections()
        torch.cuda.synchronize()
        plan_us, cb = [], []
        for r in range(G):
            t0 = time.perf_counter()
            c = runs[r].prepare_targeted(*D, Bg, comm=None)
            torch.cuda.synchronize()
            plan_us.append((time.perf_counter() - t0) * 1e6)
            cb.append(c)
        k_us = np.zeros(G); x_us = np.zeros(G)
        for b in range(nb):
            now = cb[0]["now"] if b == 0 else float(cb[0]["t_last"][b - 1])
            for r in range(G):
                c = cb[r]
                e0, e1 = ev(), ev()
                e0.record()
                _lib.check(lib.tpnet_pack_split(C.byref(c["st"]), c["R"]["pack_ids"].data_ptr() + 8 * int(c["sstart"][b]), int(c["stot"][b]), now,
                                                c["lam"], c["send_p0"].data_ptr(), c["send_q"].data_ptr(), runs[r].n_cap, int(c["rtot"][b]),
                                                c["stream"]), "pack_split")
                e1.record()
                torch.cuda.synchronize()
                x_us[r] += e0.elapsed_time(e1) * 1e3
            for r in range(G):
                c = cb[r]
                rp = runs[r].rp
                n_cap = runs[r].n_cap
                ro = 0
                for o in range(G):
                    n = int(c["rcnt"][b][o])
                    if n:
                        a0 = int(cb[o]["scnt"][b][:r].sum())
                        rp._plist()[0].data[n_cap + ro:n_cap + ro + n].copy_(cb[o]["send_p0"][a0:a0 + n])
                        rp._eng["q"].view(2, rp.node_num, L * d)[0, n_cap + ro:n_cap + ro + n].copy_(cb[o]["send_q"][a0:a0 + n])
                    ro += n
            for r in range(G):
                c = cb[r]
                e0, e1 = ev(), ev()
                e0.record()
                _lib.check(lib.tpnet_step_batch(C.byref(c["st"]), c["ls"].data_ptr(), c["ld"].data_ptr(), c["ln"].data_ptr(), c["t"].data_ptr(),
                                                E, Bg, b, c["lam"], c["lid0"] + b, c["flags"], 0, runs[r].n_cap, c["out_pos"].data_ptr(),
                                                c["out_neg"].data_ptr(), c["ws"].data_ptr(), c["ws"].numel(), c["stream"]), "step_batch")
                e1.record()
                torch.cuda.synchronize()
                k_us[r] += e0.elapsed_time(e1) * 1e3
        for r in range(G):
            runs[r].finish_targeted(cb[r], merge_outputs=False)
        sent = max(float(c["stot"].sum()) for c in cb)
        if rep >= 1:
            res.append((max(plan_us), k_us.max(), x_us.max(), sent))
    med = lambda i: float(np.median([x[i] for x in res]))
    gpu = med(1) + med(2)
    bytes_call = med(3) * (L + 1) * d * 4
    links = max(1, min(G - 1, 7))
    return dict(G=G, Bg=Bg, plan_us=med(0), step_us=med(1), pack_us=med(2), rows_sent=med(3), kern=nbt * Bg / (gpu * 1e-6),
                comp=nbt * Bg / ((gpu + med(0)) * 1e-6), wire=bytes_call / (links * 153e9) * 1e6)


print(f"# {nbt} batches of {B1} edges per GPU, {cfg['desc']}; every figure the slowest of the G ranks, median of 3 runs")
print("| G | global batch | launches | plan (wall, us) | k_wpipe launches (us) | pack + unpack (us) | rows sent per rank | halo rows | per launch (us) | edges/s, kernels only | edges/s incl. plan | wire time at 153 GB/s per link (us) |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
ONLY = os.environ.get("WS_ONLY_G")
for G in (1, 2, 4, 8):
    if ONLY and int(ONLY) != G:
        continue
    if G == 1:
        # the single-GPU call through the same driver is not a shard: report run_stream's kernels instead
        continue
    r = run(G)
    print(f"| {r['G']} | {r['Bg']} | {r['launches']} | {r['plan_us']:.0f} | {r['pipeline_us']:.0f} | {r['pack_unpack_us']:.0f} | {r['rows_sent']:.0f} | "
          f"{r['halo_rows']:.0f} | {r['per_launch']} | {r['edges_per_s_kernels'] / 1e6:.1f} M | {r['edges_per_s_compute'] / 1e6:.1f} M | {r['wire_us_at_xgmi']:.0f} |", flush=True)

print()
print(f"# the same stream on the PER-BATCH shard (pack + k_step per batch; {nbt} exchanges per call instead of the pipeline's launches)")
print("| G | global batch | plan (wall, us) | k_step launches (us) | pack launches (us) | rows sent per rank (whole bundles) | edges/s, kernels only | edges/s incl. plan | wire time at 153 GB/s per link (us) |")
print("|---|---|---|---|---|---|---|---|---|")
for G in (2, 4, 8):
    if ONLY and int(ONLY) != G:
        continue
    r = run_batch(G)
    print(f"| {r['G']} | {r['Bg']} | {r['plan_us']:.0f} | {r['step_us']:.0f} | {r['pack_us']:.0f} | {r['rows_sent']:.0f} | {r['kern'] / 1e6:.1f} M | "
          f"{r['comp'] / 1e6:.1f} M | {r['wire']:.0f} |", flush=True)
