#!/usr/bin/env python3
"""Developer tool: random SEQUENCES of module calls on one table -- run_stream on the auto / windowed schedule with plans replayed
where the library allows it, per-batch update / get_pair_wise_feature, backup / reload, reset to the epoch's start, an in-place
write to a stream tensor -- against the same sequence with every stream on the per-batch schedule and no replay.  The two differ
in f32 summation order only (1e-4 class); a plan replayed on a table state or a stream it was not built for would differ by far
more.  usage: python tools/soak_sequence.py [sequences] [seed]
TPNET_SOAK_BATCH_REPLAY=1 (round 5): module `a` runs its streams on the PER-BATCH schedule with plans replayed where the library allows
it (one-chunk streams), a third of the sequences in the exact mode; `b` the same without replay: features and state BIT FOR BIT."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DEV = "cuda:0"
replays = 0
BATCH_REPLAY = bool(os.environ.get("TPNET_SOAK_BATCH_REPLAY"))
# TPNET_SOAK_STREAM=1: every call under a stream of its own instead of the default one (the library takes torch's current stream)
side = torch.cuda.Stream() if os.environ.get("TPNET_SOAK_STREAM") else None
if side is not None:
    torch.cuda.synchronize()
    torch.cuda.set_stream(side)
for seq in range(nseq):
    d = int(rng.choice([32, 64, 128, 256])); L = int(rng.choice([2, 3, 3, 4])); N = int(rng.choice([90, 400, 3000]))
    B = int(rng.choice([16, 50, 100, 400])); nbt = int(rng.randint(40, 120)); E = nbt * B
    src, dst, neg, t = T._random_stream(rng, N, E, 4.0e5)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    ds, dd, dn, dt = dev(src), dev(dst), dev(neg), dev(t)
    exact = BATCH_REPLAY and rng.rand() < 0.33
    if BATCH_REPLAY and rng.rand() < 0.3:
        B = int(rng.choice([1500, 2500, 5000])); nbt = int(rng.randint(6, 14)); E = nbt * B   # (the chunk planner's batches)
        src, dst, neg, t = T._random_stream(rng, N, E, 4.0e5)
        ds, dd, dn, dt = dev(src), dev(dst), dev(neg), dev(t)
    a = T._module(N, d, L, 2e-6, t[0], P0=P0, exact=exact)          # the library's schedules, plans replayed
    b = T._module(N, d, L, 2e-6, t[0], P0=P0, exact=exact)          # per batch, planned every time
    if rng.rand() < 0.4:
        a.stream_log_cap_bytes = int(rng.randint(20, 60)) * B * 2 * L * d * 4
    pos = 0
    saved = None
    log = []
    lengths = {}

    def check(what):
        la, lb = T._layers(a), T._layers(b)
        sc = max(1e-30, float(np.abs(lb).max()))
        err = float(np.abs(la - lb).max()) / sc
        if not err <= (0.0 if BATCH_REPLAY else 1e-4) or float(a.now_time.item()) != float(b.now_time.item()):
            print(f"MISMATCH seq {seq} after {what}: state {err:.3g}; d={d} L={L} N={N} B={B}; ops: {log}", flush=True)
            sys.exit(1)
    for step in range(int(rng.randint(8, 24))):
        op = str(rng.choice(["stream", "stream", "stream", "restart", "restart", "update", "readout", "backup", "reload", "touch"]))
        if op == "stream" and pos < E:
            n = min(E - pos, int(rng.randint(1, 60)) * B - int(rng.randint(0, B)) * int(rng.rand() < 0.3))
            n = max(1, n)
            if pos in lengths and rng.rand() < 0.85:
                n = lengths[pos]                    # the slice an earlier epoch ran from here: a candidate for a replayed plan
            lengths[pos] = n
            sl = slice(pos, pos + n)
            sched = "batch" if BATCH_REPLAY else str(rng.choice(["auto", "windowed"]))
            fa = a.run_stream(ds[sl], dd[sl], dn[sl], dt[sl], B, schedule=sched)
            fb = b.run_stream(ds[sl], dd[sl], dn[sl], dt[sl], B, schedule="batch", replay=False)
            replays += int(a.last_stream_replayed)
            log.append(f"stream[{pos}:{pos + n}] {sched}{' R' if a.last_stream_replayed else ''}")
            for x, y in zip(fa, fb):
                e = float((x - y).abs().max())
                tol = 2e-4 + 2e-6 * float(torch.stack([b.random_projections[i].detach() for i in range(L + 1)]).norm(dim=2).max()) ** 2
                if BATCH_REPLAY:
                    tol = 0.0
                if not e <= tol:
                    print(f"MISMATCH seq {seq} features {e:.3g} (tol {tol:.3g}); ops: {log}", flush=True); sys.exit(1)
            pos += n
        elif op == "restart":
            for m in (a, b):
                m.reset_random_projections()
                m.random_projections[0].data.copy_(torch.from_numpy(P0))
            pos = 0; saved = None
            log.append("restart")
        elif op == "update" and pos + B <= E:
            sl = slice(pos, pos + B)
            for m in (a, b):
                m.update(src[sl], dst[sl], t[sl])
            pos += B
            log.append("update")
        elif op == "readout":
            u = rng.randint(0, N, 300).astype(np.int64); v = rng.randint(0, N, 300).astype(np.int64)
            with torch.no_grad():
                ga, gb = a.pair_gram(u, v), b.pair_gram(u, v)
            e = float((ga - gb).abs().max())
            tol = 2e-4 + 2e-6 * float(torch.stack([b.random_projections[i].detach() for i in range(L + 1)]).norm(dim=2).max()) ** 2
            if not e <= tol:
                print(f"MISMATCH seq {seq} readout {e:.3g} (tol {tol:.3g}); ops: {log}", flush=True); sys.exit(1)
            log.append("readout")
        elif op == "backup":
            saved = (a.backup_random_projections(), b.backup_random_projections(), pos)
            log.append("backup")
        elif op == "reload" and saved is not None:
            a.reload_random_projections(saved[0]); b.reload_random_projections(saved[1]); pos = saved[2]
            log.append("reload")
        elif op == "touch":
            k = int(rng.randint(0, E))
            dd[k] = dd[k]                       # (an in-place write: torch bumps the tensor's version, the plan is not replayed)
            log.append("touch")
        else:
            continue
        check(log[-1])
    a.check_device_errors(); b.check_device_errors()
    print(f"seq {seq}: d={d} L={L} N={N} B={B}{' exact' if exact else ''}: {' | '.join(log)}", flush=True)
print("sequences:", nseq, "replayed streams:", replays)
