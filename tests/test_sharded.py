"""Row-sharded multi-GPU path (tpnet_amd/sharded.py).

CPU tier (gloo, world_size 2, no GPU): the host logic of the N>1 path -- ownership, the per-batch lists of
touched-and-owned rows every rank derives from the stream, and the all-gather plumbing that moves row bundles.
GPU tier (-m gpu): two ranks (gloo) sharing cuda:0 run the sharded stream; features and owned rows must equal the
single-GPU run (the production backend is nccl = RCCL, one rank per GPU; the data path is identical)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _stream(seed, N, E):
    rng = np.random.RandomState(seed)
    src = rng.randint(1, N, E).astype(np.int64)
    dst = rng.randint(1, N, E).astype(np.int64)
    src[rng.rand(E) < 0.2] = 3
    dst[rng.rand(E) < 0.1] = 8
    neg = rng.randint(0, N, E).astype(np.int64)
    t = np.sort(rng.uniform(1.0e6, 1.4e6, E))
    return src, dst, neg, t


def _cpu_worker(rank, world, port, N, E, B, q):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        from tpnet_amd.sharded import plan_touched, exchange_rows
        src, dst, neg, t = _stream(0, N, E)
        ts = lambda x: torch.from_numpy(x)
        nodes, counts, offsets = plan_touched(ts(src), ts(dst), ts(neg), B, N, world)
        nb = (E + B - 1) // B
        assert counts.shape == (nb, world)
        ok = True
        for b in range(nb):
            s = slice(b * B, min((b + 1) * B, E))
            touched = np.unique(np.concatenate([src[s], dst[s], neg[s]]))
            got = []
            for r in range(world):
                ids = nodes[offsets[b, r]:offsets[b, r] + counts[b, r]].numpy()
                ok &= bool(np.all(ids % world == r)) and bool(np.all(np.diff(ids) > 0))
                got.append(ids)
            ok &= np.array_equal(np.sort(np.concatenate(got)), touched)       # a partition of the touched set
            # plumbing: every rank sends "its" rows (row k of node n = [n, n+0.5]); all must arrive unmixed
            maxc = int(counts[b].max())
            mine = nodes[offsets[b, rank]:offsets[b, rank] + counts[b, rank]].to(torch.float32)
            send = torch.zeros((maxc, 2))
            send[:len(mine), 0] = mine
            send[:len(mine), 1] = mine + 0.5
            recv = exchange_rows(send, maxc)
            for r in range(world):
                want = torch.from_numpy(got[r]).to(torch.float32)
                ok &= bool(torch.equal(recv[r, :len(want), 0], want)) and bool(torch.equal(recv[r, :len(want), 1], want + 0.5))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_sharding_plan_and_exchange_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cpu_worker, args=(r, 2, port, 97, 530, 64, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _agree_worker(rank, world, port, q):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        from tpnet_amd.sharded import ShardedStreamRunner

        class _R:                                    # what _agree_windowed touches of a runner
            detached = False
            group = dist.group.WORLD
            rp = None
        agree = lambda *a: ShardedStreamRunner._agree_windowed(_R(), *a)
        out = []
        out.append(agree(True, None))                                   # every rank planned: take the windowed shard
        out.append(agree(rank != 1, None))                              # one rank declined: none takes it
        released = []
        try:                                                            # rank 0's plan raised: it re-raises ITS error, the others
            agree(rank != 0, MemoryError("rank 0") if rank == 0 else None, lambda: released.append(1))   # free their plan and raise too
            out.append("no error")
        except MemoryError as ex:
            out.append(("own", str(ex)))
        except RuntimeError as ex:
            out.append(("other", "another rank failed" in str(ex), released == [1]))
        out.append(agree(True, None))                                   # (the group is still usable afterwards)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_windowed_shard_agreement_survives_a_rank_that_raises_gloo():
    """run_stream_targeted: all ranks take the windowed shard or none does, and a rank whose tpnet_wshard_plan raised (out of memory,
    a HIP error) tells the others BEFORE anyone enters the first exchange -- they raise too instead of hanging in it."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res[0] == [True, False, ("own", "rank 0"), True]
    assert res[1] == res[2] == [True, False, ("other", True, True), True]


def test_plan_touched_single_rank_and_no_neg():
    from tpnet_amd.sharded import plan_touched
    src, dst, neg, t = _stream(1, 50, 100)
    nodes, counts, offsets = plan_touched(torch.from_numpy(src), torch.from_numpy(dst), None, 30, 50, 1)
    assert counts.shape == (4, 1) and counts.sum() == len(nodes)
    np.testing.assert_array_equal(nodes[:counts[0, 0]].numpy(), np.unique(np.concatenate([src[:30], dst[:30]])))


# ---------------------------------------------------------------------------------------------------------
def _relabel_worker(rank, world, port, N, E, B, q):
    """CPU tier: the host logic of the compact row shard -- local ids of owned and halo rows, per batch."""
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        from tpnet_amd.sharded import ShardedStreamRunner

        class _Stub:                                   # stands in for the local module (no GPU in this tier)
            node_num = (N + world - 1) // world + 3 * B
        src, dst, neg, t = _stream(2, N, E)
        run = ShardedStreamRunner(_Stub(), N, 3 * B)
        ts = lambda x: torch.from_numpy(x)
        R = run.relabel(ts(src), ts(dst), ts(neg), B)
        n_cap, H = run.n_cap, run.H
        ok = True
        nb = (E + B - 1) // B
        for b in range(nb):
            s = slice(b * B, min((b + 1) * B, E))
            g = np.concatenate([src[s], dst[s], neg[s]])
            l = np.concatenate([R["src"][s].numpy(), R["dst"][s].numpy(), R["neg"][s].numpy()])
            own = g % world == rank
            ok &= bool(np.all(l[own] == g[own] // world))                          # owned rows: n // G
            ok &= bool(np.all((l[~own] >= n_cap) & (l[~own] < n_cap + H)))         # the rest: this batch's halo
            # one halo row per distinct remote node of the batch, and the unpack list agrees with the edge relabelling
            pairs = set(zip(g[~own].tolist(), l[~own].tolist()))
            ok &= len(pairs) == len({a for a, _ in pairs}) == len({c for _, c in pairs})
            k0, k1 = int(R["offsets"][b, 0]), int(R["offsets"][b, 0] + R["tot"][b])
            ent = R["nodes"][k0:k1].numpy(); lid = R["unpack_ids"][k0:k1].numpy()
            ok &= bool(np.all(lid[ent % world == rank] == -1))
            ok &= {(int(a), int(c)) for a, c in zip(ent, lid) if c >= 0} == pairs
            mine = ent[ent % world == rank]
            o, c = int(R["offsets"][b, rank]), int(R["counts"][b, rank])
            ok &= bool(np.array_equal(R["pack_ids"][o:o + c].numpy(), mine // world))
        q.put((rank, ok, n_cap + H))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_targeted_exchange_plan_is_complete_and_consistent(G):
    """plan_targeted (a row travels only to the ranks that read it), computed for EVERY rank: rank r receives
    exactly the remote nodes its own pairs and targets read (dst / neg of the edges whose src it owns, src of those whose dst
    it owns), ordered by (owner, node); and what owner s says it sends to r is, element for element, what r expects from s."""
    from tpnet_amd.sharded import plan_targeted
    rng = np.random.RandomState(G)
    N, E, B = 200, 530, 64
    src, dst, neg = (rng.randint(1, N, E).astype(np.int64) for _ in range(3))
    ts = torch.from_numpy
    plans = [plan_targeted(ts(src), ts(dst), ts(neg), B, N, G, me) for me in range(G)]
    for b in range((E + B - 1) // B):
        s = slice(b * B, min((b + 1) * B, E))
        for r in range(G):
            need = set()
            for e in range(s.start, s.stop):
                if src[e] % G == r:
                    need.update([int(dst[e]), int(neg[e])])
                if dst[e] % G == r:
                    need.add(int(src[e]))
            need = {x for x in need if x % G != r}
            P = plans[r]
            lo = int(P["recv_cnt"][:b].sum())
            got = P["recv_nodes"][lo:lo + int(P["recv_cnt"][b].sum())].tolist()
            assert set(got) == need and len(got) == len(need)
            o = 0
            for own in range(G):
                c = int(P["recv_cnt"][b][own])
                seg = got[o:o + c]
                o += c
                assert all(x % G == own for x in seg) and seg == sorted(seg)
                Ps = plans[own]
                assert int(Ps["send_cnt"][b][r]) == c
                slo = int(Ps["send_cnt"][:b].sum()) + int(Ps["send_cnt"][b][:r].sum())
                assert Ps["send_nodes"][slo:slo + c].tolist() == seg


def _targeted_relabel_worker(rank, world, port, q):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        from tpnet_amd.sharded import ShardedStreamRunner
        N, E, B = 97, 300, 40

        class _Stub:                                   # stands in for the local module (no GPU in this tier)
            node_num = (N + world - 1) // world + 3 * B
        src, dst, neg, t = _stream(2, N, E)
        run = ShardedStreamRunner(_Stub(), N, 3 * B)
        ts = lambda x: torch.from_numpy(x)
        R = run.relabel_targeted(ts(src), ts(dst), ts(neg), B)
        n_cap, H = run.n_cap, run.H
        ok = True
        for b in range((E + B - 1) // B):
            s = slice(b * B, min((b + 1) * B, E))
            k0, k1 = int(R["rstart"][b]), int(R["rstart"][b] + R["rtot"][b])
            slot_of = {int(n): int(l) for n, l in zip(R["recv_nodes"][k0:k1], R["unpack_ids"][k0:k1])}
            ok &= sorted(slot_of.values()) == list(range(n_cap, n_cap + len(slot_of)))          # one halo row per received node
            for which, g in (("src", src[s]), ("dst", dst[s]), ("neg", neg[s])):
                l = R[which][s].numpy()
                for x, lx in zip(g.tolist(), l.tolist()):
                    if x % world == rank:
                        ok &= lx == x // world                                                   # owned rows: n // G
                    elif x in slot_of:
                        ok &= lx == slot_of[x]                                                   # received this batch: its halo row
                    else:
                        ok &= lx == n_cap                                                        # never read by this rank
            # everything this rank's units read is owned or received
            for e in range(s.start, s.stop):
                if src[e] % world == rank:
                    ok &= all(x % world == rank or int(x) in slot_of for x in (dst[e], neg[e]))
                if dst[e] % world == rank:
                    ok &= src[e] % world == rank or int(src[e]) in slot_of
        q.put((rank, ok, n_cap + H))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_targeted_relabelling_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_targeted_relabel_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res


@pytest.mark.parametrize("world", [2, 3])
def test_row_shard_relabelling_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    N, E, B = 101, 470, 60
    procs = [ctx.Process(target=_relabel_worker, args=(r, world, port, N, E, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert all(rows == (N + world - 1) // world + 3 * B for _, _, rows in res)     # rows per rank: N/G + halo, not N


def _gpu_worker(rank, world, port, cfg, q, exchange="allgather"):
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        import tpnet_amd
        from tpnet_amd.sharded import ShardedStreamRunner
        N, d, L, E, B, lam = cfg
        dev = torch.device("cuda:0")
        src, dst, neg, t = _stream(5, N, E)
        P0 = torch.from_numpy((np.random.RandomState(77).randn(N, d) / np.sqrt(d)).astype(np.float32))
        runner = ShardedStreamRunner.create(node_num=N, edge_num=E, dim=d, num_layer=L, time_decay_weight=lam, device=dev,
                                            beginning_time=np.float64(t[0]), halo_rows=3 * B, seed=rank)
        runner.set_full_p0(P0)
        runner.exchange = exchange
        # the shard holds N/G + halo rows, not N
        n_cap = (N + world - 1) // world
        assert runner.rp.node_num == n_cap + 3 * B
        assert runner.rp.random_projections[0].shape[0] == n_cap + 3 * B
        full_bytes = N * d * 4 * (1 + 2 * L) + 32 * N
        D = lambda x: torch.from_numpy(x).to(dev)
        fp, fn = runner.run_stream(D(src), D(dst), D(neg), D(t), B)
        layers = runner.gather_full_layers().cpu()                       # [L+1, N, d], global row order
        runner.rp.check_device_errors()
        if rank == 0:
            ref = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=E, dim_factor=10, num_layer=L, time_decay_weight=lam,
                                                   device="cuda:0", use_matrix=False, beginning_time=np.float64(t[0]),
                                                   not_scale=False, enforce_dim=d)
            ref.random_projections[0].data = P0.clone()
            ref = ref.to(dev)
            rfp, rfn = ref.run_stream(D(src), D(dst), D(neg), D(t), B, schedule="batch")
            rl = torch.stack([ref.random_projections[i].detach() for i in range(0, L + 1)]).cpu()
            scale = float(rl[1:].abs().max())
            q.put(("cmp", float((fp - rfp).abs().max()), float((fn - rfn).abs().max()),
                   float((layers[1:] - rl[1:]).abs().max()) / scale, bool(torch.equal(layers[0], rl[0])),
                   float(runner.rp.now_time.item()) == float(t[-1]), runner.table_bytes(), full_bytes))
        q.put(("layers", rank, layers.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["allgather", "targeted"])
@pytest.mark.parametrize("cfg", [(211, 128, 3, 700, 100, 2e-6), (150, 64, 2, 333, 50, 1e-6), (3000, 128, 3, 900, 60, 2e-6)])
def test_sharded_stream_equals_single_gpu(cfg, exchange):
    """Two ranks (gloo) sharing cuda:0, each holding ONLY its rows (+ halo): features and the re-assembled table equal the
    single-GPU run; the per-rank table is N/G + halo rows."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, cfg, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    msgs = [q.get(timeout=300) for _ in range(3)]
    for p in procs:
        p.join(timeout=120)
    cmp = [m for m in msgs if m[0] == "cmp"][0]
    lay = {m[1]: m[2] for m in msgs if m[0] == "layers"}
    # features: same tolerance as the single-GPU parity tests; state: one extra f32 rounding of the pending decay
    assert cmp[1] < 2e-4 and cmp[2] < 2e-4, cmp
    assert cmp[3] < 1e-5 and cmp[4] and cmp[5], cmp
    np.testing.assert_array_equal(lay[0], lay[1])            # every rank re-assembles the same table
    N, d, L, E, B, lam = cfg
    if N >= 3000:                                            # memory: about half of the table per rank (+ 3B halo rows)
        assert cmp[6] < 0.6 * cmp[7], cmp


def _gpu_worker_windowed(rank, world, port, cfg, q):
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        import tpnet_amd
        from tpnet_amd.sharded import ShardedStreamRunner
        N, d, L, E, B, lam = cfg
        dev = torch.device("cuda:0")
        src, dst, neg, t = _stream(5, N, E)
        P0 = torch.from_numpy((np.random.RandomState(77).randn(N, d) / np.sqrt(d)).astype(np.float32))
        runner = ShardedStreamRunner.create(node_num=N, edge_num=E, dim=d, num_layer=L, time_decay_weight=lam, device=dev,
                                            beginning_time=np.float64(t[0]), halo_rows=max(3 * B, N), seed=rank)
        runner.set_full_p0(P0)
        runner.exchange = "targeted"
        D = lambda x: torch.from_numpy(x).to(dev)
        fp, fn = runner.run_stream(D(src), D(dst), D(neg), D(t), B)
        took = runner.last_stream_windowed
        layers = runner.gather_full_layers().cpu()
        runner.check_device_errors()
        # a second stream on the table the first one left (the chunk's halo rows now carry a pending decay)
        src2, dst2, neg2, t2 = _stream(6, N, E)
        t2 = t2 + (t[-1] - t2[0]) + 1.0
        fp2, fn2 = runner.run_stream(D(src2), D(dst2), D(neg2), D(t2), B)
        layers2 = runner.gather_full_layers().cpu()
        if rank == 0:
            ref = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=E, dim_factor=10, num_layer=L, time_decay_weight=lam,
                                                   device="cuda:0", use_matrix=False, beginning_time=np.float64(t[0]),
                                                   not_scale=False, enforce_dim=d)
            ref.random_projections[0].data = P0.clone()
            ref = ref.to(dev)
            rfp, rfn = ref.run_stream(D(src), D(dst), D(neg), D(t), B, schedule="windowed")
            rl = torch.stack([ref.random_projections[i].detach() for i in range(0, L + 1)]).cpu()
            rfp2, rfn2 = ref.run_stream(D(src2), D(dst2), D(neg2), D(t2), B, schedule="windowed")
            rl2 = torch.stack([ref.random_projections[i].detach() for i in range(0, L + 1)]).cpu()
            scale = float(rl2[1:].abs().max())
            q.put(("cmp", took, bool(torch.equal(fp, rfp)), bool(torch.equal(fn, rfn)), bool(torch.equal(layers, rl)),
                   max(float((fp2 - rfp2).abs().max()), float((fn2 - rfn2).abs().max())),
                   float((layers2[1:] - rl2[1:]).abs().max()) / scale, float(runner.rp.now_time.item()) == float(t2[-1])))
        q.put(("done", rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(211, 128, 3, 2000, 100, 2e-6), (600, 64, 2, 2990, 100, 1e-6)])
def test_windowed_shard_two_gloo_ranks_equal_single_gpu(cfg):
    """Two ranks (gloo) sharing cuda:0 through ShardedStreamRunner.run_stream: from 16 batches on the stream takes the WINDOWED
    shard (csrc/wshard.hip: one launch + one exchange per window of batches).  On a table fresh from a reset the two shards equal
    the single-GPU run on the windowed schedule BIT FOR BIT (features and every layer: a log slot travels as it is); a second
    stream on the table the first one left equals it within the per-batch shard's tolerances (the chunk's halo rows carry one
    extra f32 rounding of their pending decay)."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker_windowed, args=(r, 2, port, cfg, q)) for r in range(2)]
    for p in procs:
        p.start()
    msgs = [q.get(timeout=300) for _ in range(3)]
    for p in procs:
        p.join(timeout=120)
    cmp = [m for m in msgs if m[0] == "cmp"][0]
    assert cmp[1], "the stream did not take the windowed shard"
    assert cmp[2] and cmp[3] and cmp[4], cmp
    assert cmp[5] < 2e-4 and cmp[6] < 1e-5 and cmp[7], cmp


@pytest.mark.gpu
def test_windowed_shard_equals_single_gpu_bit_for_bit():
    """Two, three and four row shards in ONE process on the windowed pipeline (tests/loopback/run_wshard.py): the phases of
    tpnet_wshard_begin / tpnet_wshard_step with the rows moved by plain copies, and tpnet_wshard_run with every shard a host thread
    on the in-process RCCL stand-in -- features and tables BIT FOR BIT equal to the single-GPU run on the windowed schedule (hubs
    whose chains span windows, ragged tail, d = 64 / 128 / 256, L = 2 / 3, the C2 table with global batches of 2 000 edges)."""
    import subprocess
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "loopback")
    if not os.path.exists(os.path.join(here, "librccl_loopback.so")):
        subprocess.run(["make", "-C", here], check=True)
    res = subprocess.run([sys.executable, os.path.join(here, "run_wshard.py")], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "WSHARD OK" in res.stdout, (res.stdout[-2000:], res.stderr[-4000:])


def _c_loop_worker(port, cfg, use_c, q):
    """One rank on RCCL with the collectives forced: the batch loop as ONE C call per batch with RCCL called from C
    (tpnet_rows_step), or through torch.distributed -- same results as the plain single-GPU stream."""
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    os.environ["TPNET_DEV_FORCE_COLLECTIVES"] = "1"
    os.environ["TPNET_ROWS_C_LOOP"] = "1" if use_c else "0"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        import tpnet_amd
        from tpnet_amd.sharded import ShardedStreamRunner
        N, d, L, E, B, lam = cfg
        src, dst, neg, t = _stream(5, N, E)
        P0 = torch.from_numpy((np.random.RandomState(77).randn(N, d) / np.sqrt(d)).astype(np.float32))
        runner = ShardedStreamRunner.create(node_num=N, edge_num=E, dim=d, num_layer=L, time_decay_weight=lam, device=dev,
                                            beginning_time=np.float64(t[0]), halo_rows=3 * B)
        runner.set_full_p0(P0)
        D = lambda x: torch.from_numpy(x).to(dev)
        fp, fn = runner.run_stream(D(src), D(dst), D(neg), D(t), B)
        have_comm = bool(getattr(runner, "_comm", None))
        layers = runner.gather_full_layers()
        ref = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=E, dim_factor=10, num_layer=L, time_decay_weight=lam,
                                               device="cuda:0", use_matrix=False, beginning_time=np.float64(t[0]),
                                               not_scale=False, enforce_dim=d)
        ref.random_projections[0].data = P0.clone()
        ref = ref.to(dev)
        rfp, rfn = ref.run_stream(D(src), D(dst), D(neg), D(t), B, schedule="batch")
        rl = torch.stack([ref.random_projections[i].detach() for i in range(0, L + 1)])
        runner.rp.check_device_errors()
        runner.close()
        q.put((have_comm, bool(torch.equal(fp, rfp)), bool(torch.equal(fn, rfn)), bool(torch.equal(layers, rl))))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("use_c", [True, False])
def test_row_sharded_batch_loop_with_rccl_from_c(use_c):
    """tpnet_rows_step: pack -> RCCL all-gather (called from C through a communicator the library created from a broadcast
    unique id) -> unpack -> step, one FFI call per batch.  With one rank every row is owned, so the results must equal the
    plain stream BIT FOR BIT; the torch.distributed variant of the same loop is the control."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_c_loop_worker, args=(_free_port(), (500, 128, 3, 1200, 100, 2e-6), use_c, q))
    p.start()
    have_comm, e1, e2, e3 = q.get(timeout=300)
    p.join(timeout=120)
    assert have_comm == use_c
    assert e1 and e2 and e3


# =========================================================================================================
# column (dim) sharding: tpnet_amd.sharded.ColumnShardedRunner
# =========================================================================================================
class _FakeShard:
    """Stands in for the local module on the CPU tier: writes 'partial features' (rank+1)*(edge+1) so that the
    merged value of edge e is known: sum_r (r+1)(e+1)."""
    not_scale = True
    num_layer = 0                    # NN = 2: full rows of 4, packed rows of 3
    pair_wise_feature_dim = 4
    packed_feature_dim = 3
    dim = 8

    def __init__(self, rank):
        self.rank = rank
        self.calls = []
        self.e0 = 0

    def _dev(self):
        return torch.device("cpu")

    def run_stream(self, src, dst, neg, t, B, out_pos=None, out_neg=None, t_end=None, packed=False):
        assert packed
        n = src.numel()
        e = torch.arange(self.e0, self.e0 + n, dtype=torch.float32)
        out_pos.copy_(((self.rank + 1) * (e + 1))[:, None].expand(n, 3))
        if out_neg is not None:
            out_neg.copy_((-(self.rank + 1) * (e + 1))[:, None].expand(n, 3))
        self.calls.append((n, t_end))
        self.e0 += n


def _cpu_unpack(self, packed, out):
    """CPU stand-in of tpnet_gram_unpack for the fake shard (NN = 2: entries (0,0), (0,1), (1,1))."""
    out.copy_(packed[:, [0, 1, 1, 2]])
    return out


def _col_cpu_worker(rank, world, port, E, B, chunk_steps, q):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        from tpnet_amd.sharded import ColumnShardedRunner
        ColumnShardedRunner._unpack = _cpu_unpack            # the real one is a HIP kernel
        src, dst, neg, t = _stream(2, 40, E)
        ts = lambda x: torch.from_numpy(x)
        ok = True
        for merge in ("scatter", "all"):
            fake = _FakeShard(rank)
            runner = ColumnShardedRunner(fake, 8 * world)
            fp, fn, idx = runner.run_stream(ts(src), ts(dst), ts(neg), ts(t), B, chunk_steps=chunk_steps, merge=merge)
            tot = sum(r + 1 for r in range(world))
            if merge == "all":
                want = tot * (torch.arange(E, dtype=torch.float32) + 1)
                ok &= idx is None and bool(torch.equal(fp[:, 0], want)) and bool(torch.equal(fn[:, 2], -want))
            else:
                keep = idx >= 0
                want = tot * (idx[keep].to(torch.float32) + 1)
                ok &= bool(torch.equal(fp[keep][:, 1], want)) and bool(torch.equal(fn[keep][:, 3], -want))
                ok &= bool((fp[~keep] == 0).all())                       # padding rows carry zeros
                allidx = [torch.empty_like(idx) for _ in range(world)]
                dist.all_gather(allidx, idx)
                got = torch.cat(allidx)
                ok &= bool(torch.equal(torch.sort(got[got >= 0]).values, torch.arange(E)))   # every edge exactly once
            # chunks end on batch boundaries and carry their own clock
            ce = chunk_steps * B
            ok &= [c[0] for c in fake.calls] == [min(ce, E - a) for a in range(0, E, ce)]
            ok &= [c[1] for c in fake.calls] == [float(t[min(a + ce, E) - 1]) for a in range(0, E, ce)]
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("E,B,chunk_steps,world", [(530, 64, 3, 2), (96, 16, 2, 2), (7, 3, 1, 2), (531, 64, 3, 3)])
def test_column_runner_merge_logic_gloo(E, B, chunk_steps, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_col_cpu_worker, args=(r, world, port, E, B, chunk_steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)]


def _col_gpu_worker(rank, world, port, cfg, q):
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        import tpnet_amd
        from tpnet_amd.sharded import ColumnShardedRunner
        N, d, L, E, B, lam, exact = cfg
        dev = torch.device("cuda:0")
        src, dst, neg, t = _stream(6, N, E)
        P0 = torch.from_numpy((np.random.RandomState(3).randn(N, d) / np.sqrt(d)).astype(np.float32))
        runner = ColumnShardedRunner.create(node_num=N, edge_num=E, dim=d, num_layer=L, time_decay_weight=lam,
                                            device=dev, beginning_time=np.float64(t[0]), seed=1, exact=exact)
        own_draw = runner.rp.random_projections[0].detach().cpu().clone()
        runner.set_full_p0(P0)
        D = lambda x: torch.from_numpy(x).to(dev)
        half = (E // (2 * B)) * B
        fp, fn, idx = runner.run_stream(D(src[:half]), D(dst[:half]), D(neg[:half]), D(t[:half]), B, chunk_steps=2)
        # second part through the module-level calls (what a per-batch training loop does)
        fp2 = []
        for a in range(half, E, B):
            s = slice(a, min(a + B, E))
            fp2.append(runner.pair_gram(src[s], dst[s]))
            runner.update(src[s], dst[s], t[s])
        fp2 = torch.cat(fp2)
        full = runner.gather_full_layers().cpu()
        runner.rp.check_device_errors()
        allp = [torch.empty_like(fp) for _ in range(world)]; alli = [torch.empty_like(idx) for _ in range(world)]
        alln = [torch.empty_like(fn) for _ in range(world)]
        dist.all_gather(allp, fp); dist.all_gather(alli, idx); dist.all_gather(alln, fn)
        if rank == 0:
            ref = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=E, dim_factor=10, num_layer=L,
                                                   time_decay_weight=lam, device="cuda:0", use_matrix=False,
                                                   beginning_time=np.float64(t[0]), not_scale=False, enforce_dim=d,
                                                   exact=exact)
            ref.random_projections[0].data = P0.clone()
            ref = ref.to(dev)
            rfp, rfn = ref.run_stream(D(src[:half]), D(dst[:half]), D(neg[:half]), D(t[:half]), B)
            rfp2 = []
            for a in range(half, E, B):
                s = slice(a, min(a + B, E))
                rfp2.append(ref.pair_gram(src[s], dst[s]))
                ref.update(src[s], dst[s], t[s])
            rfp2 = torch.cat(rfp2)
            rl = torch.stack([ref.random_projections[i].detach() for i in range(L + 1)]).cpu()
            ii = torch.cat(alli); keep = ii >= 0
            gp = torch.empty_like(rfp); gn = torch.empty_like(rfn)
            gp[ii[keep]] = torch.cat(allp)[keep]; gn[ii[keep]] = torch.cat(alln)[keep]
            q.put(("cmp", int(keep.sum()) == half, float((gp - rfp).abs().max()), float((gn - rfn).abs().max()),
                   float((fp2 - rfp2).abs().max()), bool(torch.equal(full, rl)),
                   float((full - rl).abs().max() / rl.abs().max()), float(runner.rp.now_time.item()) == float(t[-1])))
        q.put(("p0", rank, own_draw.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(211, 128, 3, 700, 100, 2e-6, False), (150, 64, 2, 333, 50, 1e-6, True),
                                 (90, 32, 3, 420, 60, 2e-6, False)])
def test_column_sharded_stream_equals_single_gpu(cfg):
    """Two ranks (gloo) sharing cuda:0, each with half of the columns: features == the single-GPU features up to the
    association of the sum over d, the table == the single-GPU table (bit for bit in exact mode: a column's
    arithmetic does not depend on which other columns share its GPU)."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_col_gpu_worker, args=(r, 2, port, cfg, q)) for r in range(2)]
    for p in procs:
        p.start()
    msgs = [q.get(timeout=300) for _ in range(3)]
    for p in procs:
        p.join(timeout=120)
    cmp = [m for m in msgs if m[0] == "cmp"][0]
    p0s = {m[1]: m[2] for m in msgs if m[0] == "p0"}
    assert cmp[1] and cmp[7], cmp
    assert cmp[2] < 2e-4 and cmp[3] < 2e-4 and cmp[4] < 2e-4, cmp
    if cfg[-1]:
        assert cmp[5], cmp                                    # exact mode: identical bits
    assert cmp[6] < 1e-5, cmp
    # the ranks' own draws of P[0] are independent and scaled for the FULL width
    assert not np.array_equal(p0s[0], p0s[1])
    assert abs(np.std(np.concatenate([p0s[0], p0s[1]])) * np.sqrt(cfg[1]) - 1.0) < 0.05


@pytest.mark.parametrize("E,B,kw", [(8_000_000, 2000, {}), (4_100_000, 1000, {}), (7, 3, {}), (530, 64, {"chunk_steps": 3}),
                                    (1, 1, {}), (2_000_001, 1000, {}), (130_000, 1000, {"chunk_edges": 50_000}),
                                    (999, 1000, {}), (64 * 2000, 2000, {"chunk_edges": 10**9})])
def test_column_runner_chunk_bounds_properties(E, B, kw):
    """Chunks tile [0, E) in order, every chunk but the last is a whole number of batches, and without an explicit
    chunk_steps the tail is split so that the last collective (which nothing can hide) covers < 2 * 64 batches."""
    from tpnet_amd.sharded import ColumnShardedRunner
    bounds = ColumnShardedRunner.chunk_bounds(E, B, **kw)
    assert bounds[0][0] == 0 and bounds[-1][1] == E
    for (a, b), (c, d) in zip(bounds, bounds[1:]):
        assert b == c and a < b
    for a, b in bounds[:-1]:
        assert a % B == 0 and (b - a) % B == 0
    if "chunk_steps" not in kw:
        a, b = bounds[-1]
        assert (b - a + B - 1) // B < 2 * 64


# =========================================================================================================
# BASELINE config 4 at full table size: the row-sharded runner with two ranks on ONE GPU (2 x 36 GB + halos)
# =========================================================================================================
def _c4_worker(rank, world, port, exchange, q):
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        import tpnet_amd
        from tpnet_amd.sharded import ShardedStreamRunner
        from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
        c = CONFIGS["C4"]
        N, d, lam, L = c["U"] + c["I"] + 1, c["d"], c["lam"], 3
        B, E = 10_000, 3 * 10_000 + 2_500
        src, dst, t, N_ = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
        neg = synthetic_negatives(c["U"], N, E, B, 1)
        assert N_ == N == 10_000_001
        dev = torch.device("cuda:0")
        runner = ShardedStreamRunner.create(node_num=N, edge_num=c["E"], dim=d, num_layer=L, time_decay_weight=lam, device=dev,
                                            beginning_time=np.float64(0.0), halo_rows=3 * B, seed=11, draw_on_device=True)
        runner.exchange = exchange
        n_cap = (N + world - 1) // world
        assert runner.rp.node_num == n_cap + 3 * B                       # the shard holds N/G + halo rows, not N
        D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        fp, fn = runner.run_stream(D(src), D(dst), D(neg), D(t), B)
        runner.rp.check_device_errors()
        # a sample of the touched rows this rank owns, all layers (no 41 GB materialisation)
        touched = np.unique(np.concatenate([src, dst]))
        sample = touched[np.random.RandomState(3).permutation(len(touched))[:4000]]
        mine = sample[sample % world == rank]
        rows = torch.stack(runner.rp.get_random_projections(mine // world)).cpu().numpy()      # local row = id // G
        shard_bytes = runner.table_bytes()
        torch.cuda.empty_cache()                                         # (the shard's set-up temporaries: 5 GB of normals per rank)
        if rank == 0:
            ref = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=L, time_decay_weight=lam,
                                                   device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0),
                                                   not_scale=False, enforce_dim=d, alloc_device=dev).to(dev)
            for r in range(world):                                       # the full layer 0 = the ranks' rows interleaved
                own = ref.random_projections[0].data[r::world]
                own.copy_(ShardedStreamRunner.draw_rows(11, r, n_cap, d, dev)[:own.shape[0]])
            rfp, rfn = ref.run_stream(D(src), D(dst), D(neg), D(t), B, schedule="batch")
            rrows = torch.stack(ref.get_random_projections(sample)).cpu().numpy()
            scale = float(np.abs(rrows[1:]).max())
            q.put(("cmp", float((fp - rfp).abs().max()), float((fn - rfn).abs().max()), scale, shard_bytes,
                   N * d * 4 * (1 + 2 * L) + 32 * N, float(runner.rp.now_time.item()) == float(t[-1])))
            q.put(("ref", sample, rrows))
        q.put(("rows", rank, mine, rows))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["targeted"])
def test_row_sharded_c4_table_two_ranks_on_one_gpu(exchange):
    """BASELINE config 4 ("10 M nodes ... row-sharded"): N = 10 000 001 rows of d = 256, three batches of 10 000 edges and a
    ragged one, two gloo ranks sharing cuda:0 -- each holds 5 M rows of all layers (36 GB) + 30 000 halo rows -- against the
    single-GPU run on the full table: features of every edge, and all layers of a sample of the touched rows."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    if torch.cuda.get_device_properties(0).total_memory < 200e9:
        pytest.skip("needs ~150 GB of device memory (two 36 GB shards + the 72 GB control)")
    # the workers need 36 + 36 + 72 GB of the card: whatever the tests before this one left cached in THIS process's allocator
    # (the 72 GB table of the single-GPU C4 test, stream workspaces) goes back to the driver first
    import gc
    import queue as _queue
    gc.collect()
    torch.cuda.empty_cache()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_c4_worker, args=(r, 2, port, exchange, q)) for r in range(2)]
    for p in procs:
        p.start()
    msgs = []
    waited = 0
    while len(msgs) < 4:
        try:
            msgs.append(q.get(timeout=5))
        except _queue.Empty:
            waited += 5
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or waited > 600:                                    # a worker that died (out of memory, ...) must not hang the suite
                for p in procs:
                    if p.is_alive():
                        p.kill()
                pytest.fail(f"sharded C4 workers: exit codes {[p.exitcode for p in procs]} after {waited} s")
    for p in procs:
        p.join(timeout=300)
    cmp = [m for m in msgs if m[0] == "cmp"][0]
    ref = [m for m in msgs if m[0] == "ref"][0]
    assert cmp[1] < 2e-4 and cmp[2] < 2e-4 and cmp[6], cmp
    assert cmp[4] < 0.52 * cmp[5], cmp                                  # half the table (+ halo) per rank
    sample, rrows = ref[1], ref[2]
    for m in msgs:
        if m[0] == "rows":
            _, rank, mine, rows = m
            pos = {int(x): i for i, x in enumerate(sample)}
            want = rrows[:, [pos[int(x)] for x in mine]]
            np.testing.assert_array_equal(rows[0], want[0])            # layer 0: the same draw
            np.testing.assert_allclose(rows[1:], want[1:], rtol=1e-4, atol=1e-5 * cmp[3])


@pytest.mark.gpu
@pytest.mark.parametrize("G,N,E,B,with_neg,large", [(2, 211, 700, 100, True, False), (3, 3000, 905, 60, True, False),
                                                    (8, 9000, 8000 * 3 - 17, 8000, True, False), (4, 150, 333, 50, False, False),
                                                    (8, 1 << 20, 5000, 1000, True, False),
                                                    (2, 211, 700, 100, True, True), (4, 150, 333, 50, False, True), (1, 97, 250, 40, True, True),
                                                    (8, 3_000_000, 40000 * 3 - 17, 40000, True, False), (1, 97, 250, 40, True, False)])
def test_device_exchange_plan_equals_torch_plan(G, N, E, B, with_neg, large):
    """tpnet_xplan_targeted (csrc/xplan.hip: the targeted exchange's plan in two launches) against plan_targeted + the torch
    relabelling, for every rank of G: the same message sizes, the same pack order, the same halo slot for every endpoint --
    hubs, ragged last batch, a global batch of 8 000 edges (24 000 items per batch) and a table of 2^20 nodes included.  `large`
    (and batches of 40 000 edges on their own): tpnet_xplan_targeted_large, one device-wide sort of the call's keys -- the plan of
    C4-sized global batches, whose lists exceed one workgroup's LDS.  One rank: empty lists, every id its own row."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd import _lib
    from tpnet_amd.sharded import ShardedStreamRunner
    src, dst, neg, _ = _stream(G * 7 + B, N, E)
    dev = torch.device("cuda:0")
    ds, dd = torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev)
    dn = torch.from_numpy(neg).to(dev) if with_neg else None

    class _Rp:                                  # (the planners only ask the module for its stream)
        @staticmethod
        def _stream():
            return None
    cap = int(_lib.load().tpnet_xplan_capacity())
    for me in range(G):
        r = object.__new__(ShardedStreamRunner)
        r.G, r.me, r.N, r.n_cap, r.H, r.rp, r.group = G, me, N, (N + G - 1) // G, 3 * B, _Rp(), None
        r.xplan_force_large = large
        want = r.relabel_targeted(ds, dd, dn, B)
        got = r.relabel_targeted_device(ds, dd, dn, B)
        assert got is not None
        np.testing.assert_array_equal(got["send_cnt"], want["send_cnt"])
        np.testing.assert_array_equal(got["recv_cnt"], want["recv_cnt"])
        for k in ("src", "dst") + (("neg",) if with_neg else ()):
            assert torch.equal(got[k], want[k]), (me, k)
        stot = want["send_cnt"].sum(axis=1)
        s0 = np.concatenate([[0], np.cumsum(stot)[:-1]]).astype(np.int64)
        gp, wp = got["pack_ids"].cpu().numpy(), want["pack_ids"].cpu().numpy()
        g0 = got["sstart"]
        for b in range(len(stot)):
            np.testing.assert_array_equal(gp[g0[b]: g0[b] + stot[b]], wp[s0[b]: s0[b] + stot[b]])
    # an id outside [0, N) is an IndexError, as on the torch path
    bad = ds.clone()
    bad[5] = N
    r = object.__new__(ShardedStreamRunner)
    r.G, r.me, r.N, r.n_cap, r.H, r.rp, r.group = G, 0, N, (N + G - 1) // G, 3 * B, _Rp(), None
    r.xplan_force_large = large
    with pytest.raises(IndexError):
        r.relabel_targeted_device(bad, dd, dn, B)
        r._check_pending_status()               # (one rank: the status words are read back asynchronously)


@pytest.mark.gpu
def test_rccl_branch_through_loopback():
    """The `comm && G > 1` branch of tpnet_rows_step_targeted (csrc/rows_rccl.hip: grouped ncclSend / ncclRecv of two messages per
    peer, received straight into the halo rows) -- which every gloo test bypasses and a one-GPU box cannot reach with real RCCL --
    runs against an in-process stand-in for the eight RCCL entry points (tests/loopback/rccl_loopback.cpp): two shards in one
    process, one stream each, driven batch by batch through tpnet_rows_stream_targeted.  Bit for bit equal to the same shards
    with the rows moved by plain copies; equal to the single-GPU run within the tolerances of
    test_sharded_stream_equals_single_gpu.  In a process of its own: the library resolves RCCL once per process."""
    import subprocess
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "loopback")
    if not os.path.exists(os.path.join(here, "librccl_loopback.so")):
        subprocess.run(["make", "-C", here], check=True)
    res = subprocess.run([sys.executable, os.path.join(here, "run_loopback.py")], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "LOOPBACK OK" in res.stdout, (res.stdout[-2000:], res.stderr[-4000:])
