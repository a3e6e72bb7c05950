#!/usr/bin/env python3
"""TEST DRIVER (run in a process of its own by tests/test_sharded.py::test_rccl_branch_through_loopback): two row shards of one
table in ONE process on one GPU, each with its own host thread and stream, driven through tpnet_rows_stream_targeted with a
communicator of the in-process stand-in tests/loopback/librccl_loopback.so -- the `comm && G > 1` branch of
tpnet_rows_step_targeted (message offsets, halo placement, two messages per peer inside one group) executes, which no gloo test
reaches.  Checks: (1) bit for bit against the same two shards with the rows moved by plain copies (the gloo transport's data
path: tpnet_pack_split -> copies into the halo rows -> tpnet_step_batch); (2) against the single-GPU run with the tolerances of
test_sharded_stream_equals_single_gpu.  Prints one line 'LOOPBACK OK ...' or raises."""
import ctypes as C
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import tpnet_amd                                           # noqa: E402
from tpnet_amd import _lib                                 # noqa: E402
from tpnet_amd.sharded import ShardedStreamRunner          # noqa: E402


def stream_of(seed, N, E):
    rng = np.random.RandomState(seed)
    src = rng.randint(1, N, E).astype(np.int64)
    dst = rng.randint(1, N, E).astype(np.int64)
    src[rng.rand(E) < 0.15] = 3                            # hubs: rows that every batch sends both ways
    dst[rng.rand(E) < 0.10] = 4
    neg = rng.randint(1, N, E).astype(np.int64)
    t = np.sort(rng.uniform(0.0, 4.0e5, E))
    return src, dst, neg, t


def shards(N, d, L, E, B, lam, t0, P0, dev):
    out = []
    for r in range(2):
        run = ShardedStreamRunner.create(node_num=N, edge_num=E, dim=d, num_layer=L, time_decay_weight=lam, device=dev,
                                         beginning_time=np.float64(t0), halo_rows=3 * B, seed=r, world=2, rank=r)
        run.set_full_p0(P0)
        out.append(run)
    return out


def full_layers(runs, N, L, d):
    """[L + 1, N, d] from the two shards' owned rows (what gather_full_layers does over a process group)."""
    dev = runs[0].rp._dev()
    full = torch.empty((L + 1, N, d), dtype=torch.float32, device=dev)
    for r, run in enumerate(runs):
        rows = torch.stack(run.rp.get_random_projections(torch.arange(run.n_cap, dtype=torch.int64, device=dev)))
        n_own = (N - r + 1) // 2
        full[:, r::2] = rows[:, :n_own]
    return full


def main():
    cfgs = [(211, 128, 3, 700, 100, 2e-6), (150, 64, 2, 333, 50, 1e-6), (3000, 128, 3, 900, 60, 2e-6)]
    if os.environ.get("TPNET_LOOPBACK_ONE"):
        cfgs = cfgs[:1]
    if os.environ.get("TPNET_LOOPBACK_RANDOM"):            # "cases,seed": random shapes instead (a development soak, not the test)
        ncase, seed = (int(x) for x in os.environ["TPNET_LOOPBACK_RANDOM"].split(","))
        rg = np.random.RandomState(seed)
        cfgs = []
        for _ in range(ncase):
            B = int(rg.choice([1, 3, 16, 50, 100, 257]))
            nbr = int(rg.randint(1, 30))
            cfgs.append((int(rg.choice([60, 211, 1000, 5000])), int(rg.choice([32, 64, 128, 256, 120])), int(rg.choice([1, 2, 3, 3, 4])),
                         max(1, nbr * B - int(rg.randint(0, B))), B, float(rg.choice([1e-6, 2e-6, 1e-7]))))
        print("random shapes:", cfgs, flush=True)
    lib = _lib.load()
    so = os.path.join(HERE, "librccl_loopback.so").encode()
    loop = C.CDLL(so.decode())
    for fn in ("tpnet_loopback_count", "tpnet_loopback_pending"):
        getattr(loop, fn).restype = C.c_long
    dev = torch.device("cuda:0")
    D = lambda x: torch.from_numpy(x).to(dev)
    summary = []
    for ci, (N, d, L, E, B, lam) in enumerate(cfgs):
        src, dst, neg, t = stream_of(5 + ci, N, E)
        P0 = torch.from_numpy((np.random.RandomState(77).randn(N, d) / np.sqrt(d)).astype(np.float32))
        ds, dd, dn, dt = D(src), D(dst), D(neg), D(t)
        nb = (E + B - 1) // B
        # ---- (A) the RCCL branch through the stand-in: one communicator per shard, one stream per shard
        idbuf = (C.c_ubyte * 128)()
        assert lib.tpnet_rccl_unique_id(so, idbuf) == 0
        comms = []
        for r in range(2):
            h = C.c_void_p()
            assert lib.tpnet_rccl_comm_create(so, idbuf, 2, r, C.byref(h)) == 0
            comms.append(h)
        runs = shards(N, d, L, E, B, lam, t[0], P0, dev)
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        torch.cuda.synchronize()
        ctxs = []
        for r in range(2):
            with torch.cuda.stream(streams[r]):
                ctxs.append(runs[r].prepare_targeted(ds, dd, dn, dt, B, comm=comms[r]))
        before = [loop.tpnet_loopback_count(k) for k in range(4)]
        # one host thread per rank, as ranks of a real job run side by side: each issues its whole stream in ONE call
        # (tpnet_rows_stream_targeted); ctypes releases the interpreter lock for the duration of the call
        import threading
        failed = []

        def rank_thread(r):
            try:
                with torch.cuda.stream(streams[r]):
                    half = nb // 2
                    runs[r].steps_targeted(ctxs[r], 0, half)           # (two calls: the loop's clock hand-over across calls too)
                    runs[r].steps_targeted(ctxs[r], half, nb)
            except Exception as ex:            # noqa: BLE001
                failed.append((r, repr(ex)))
        ths = [threading.Thread(target=rank_thread, args=(r,)) for r in range(2)]
        for th in ths:
            th.start()
        for th in ths:
            th.join(timeout=120)
        assert not failed and not any(th.is_alive() for th in ths), failed
        outs = []
        for r in range(2):
            with torch.cuda.stream(streams[r]):
                outs.append(runs[r].finish_targeted(ctxs[r], merge_outputs=False))
        torch.cuda.synchronize()
        assert loop.tpnet_loopback_pending() == 0, "unmatched sends / receives left behind"
        cnt = [loop.tpnet_loopback_count(k) - before[k] for k in range(4)]
        assert cnt[0] == cnt[1] == cnt[2] and (cnt[0] > 0 or os.environ.get("TPNET_LOOPBACK_RANDOM")) and cnt[3] == 0, cnt   # (a one-edge stream may send nothing)
        for r in range(2):
            runs[r].rp.check_device_errors()
            assert lib.tpnet_rccl_comm_destroy(comms[r]) == 0
        fpA, fnA = outs[0][0] + outs[1][0], outs[0][1] + outs[1][1]      # disjoint rows: the sum is the merge
        layA = full_layers(runs, N, L, d)
        # ---- (B) the same shards, rows moved by plain copies on one stream (the gloo transport's data path)
        runs_b = shards(N, d, L, E, B, lam, t[0], P0, dev)
        cb = [runs_b[r].prepare_targeted(ds, dd, dn, dt, B, comm=None) for r in range(2)]
        for b in range(nb):
            now = cb[0]["now"] if b == 0 else float(cb[0]["t_last"][b - 1])
            for r in range(2):
                c = cb[r]
                _lib.check(lib.tpnet_pack_split(C.byref(c["st"]), c["R"]["pack_ids"].data_ptr() + 8 * int(c["sstart"][b]),
                                                int(c["stot"][b]), now, c["lam"], c["send_p0"].data_ptr(), c["send_q"].data_ptr(),
                                                runs_b[r].n_cap, int(c["rtot"][b]), c["stream"]), "pack_split")
            for r in range(2):                              # rank r receives what the other one packed for it
                c, o = cb[r], cb[1 - r]
                n = int(c["rcnt"][b][1 - r])
                assert n == int(o["scnt"][b][r])
                rp = runs_b[r].rp
                n_cap = runs_b[r].n_cap
                rp._plist()[0].data[n_cap:n_cap + n].copy_(o["send_p0"][:n])
                rp._eng["q"].view(2, rp.node_num, L * d)[0, n_cap:n_cap + n].copy_(o["send_q"][:n])
            for r in range(2):
                c = cb[r]
                ln, on = c["ln"], c["out_neg"]
                _lib.check(lib.tpnet_step_batch(C.byref(c["st"]), c["ls"].data_ptr(), c["ld"].data_ptr(), ln.data_ptr(), c["t"].data_ptr(),
                                                E, B, b, c["lam"], c["lid0"] + b, c["flags"], 0, runs_b[r].n_cap,
                                                c["out_pos"].data_ptr(), on.data_ptr(), c["ws"].data_ptr(), c["ws"].numel(), c["stream"]),
                           "step_batch")
        ob = [runs_b[r].finish_targeted(cb[r], merge_outputs=False) for r in range(2)]
        torch.cuda.synchronize()
        fpB, fnB = ob[0][0] + ob[1][0], ob[0][1] + ob[1][1]
        layB = full_layers(runs_b, N, L, d)
        assert torch.equal(fpA, fpB) and torch.equal(fnA, fnB), "features differ from the plain-copy transport"
        assert torch.equal(layA, layB), "tables differ from the plain-copy transport"
        # ---- (C) the single-GPU run
        ref = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=E, dim_factor=10, num_layer=L, time_decay_weight=lam,
                                               device="cuda:0", use_matrix=False, beginning_time=np.float64(t[0]),
                                               not_scale=False, enforce_dim=d)
        ref.random_projections[0].data = P0.clone()
        ref = ref.to(dev)
        rfp, rfn = ref.run_stream(ds, dd, dn, dt, B, schedule="batch")
        rl = torch.stack([ref.random_projections[i].detach() for i in range(0, L + 1)])
        scale = float(rl[1:].abs().max())
        e_f = max(float((fpA - rfp).abs().max()), float((fnA - rfn).abs().max()))
        e_s = float((layA[1:] - rl[1:]).abs().max()) / scale
        # (a feature is log(1 + G) of a Gram entry whose f32 sum carries ~1e-6 |R_a| |R_b| whatever its order: the random shapes of
        # the soak have hubs whose rows are long)
        tol_f = 2e-4 if not os.environ.get("TPNET_LOOPBACK_RANDOM") else 2e-4 + 2e-6 * float(rl.norm(dim=2).max()) ** 2
        assert e_f < tol_f and e_s < 1e-5 and torch.equal(layA[0], rl[0]), (e_f, tol_f, e_s)
        assert float(runs[0].rp.now_time.item()) == float(t[-1])
        summary.append((N, d, L, nb, cnt[0], e_f, e_s))
    print("LOOPBACK OK " + "; ".join(f"N={a} d={b_} L={c} batches={n}: {s_} messages, features {ef:.1e}, state {es:.1e} from one GPU"
                                      for a, b_, c, n, s_, ef, es in summary))


if __name__ == "__main__":
    main()
