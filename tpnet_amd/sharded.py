"""Multi-GPU runners of the temporal-walk-matrix hot path (new capability: the reference is
single-device, utils/load_configs.py:88).  Two ways to cut the table, both one process per GPU:

* `ShardedStreamRunner`  -- ROW sharding, the contract of BASELINE.json's north_star: described below.
* `ColumnShardedRunner`  -- COLUMN (dim) sharding (SURVEY.md §8e "worth measuring"): every rank keeps all N rows but
  only d/G of the d columns.  The update is column-wise independent (P[i][u, :] += w * P[i-1][v, :]), so it needs NO
  exchange at all; only the readout's inner products are partial sums, and they are summed over the ranks when the
  features leave the path (one reduce-scatter per chunk of steps, overlapped with the next chunk's kernels).  No
  collective sits on the per-batch critical path, which is what a 7 us step needs: see DESIGN.md §6.

Row sharding
------------

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI in production; "gloo" in the tests).
Rows are owned cyclically, owner(n) = n % G (spreads the power-law head over the ranks).  Every rank holds the
whole edge stream (32 bytes per edge) but ONLY ITS OWN ROWS of the table -- all L+1 layers, in a compact local table
of ceil(N/G) rows -- plus a halo of H rows that per batch receive the other ranks' rows the batch reads.  Per batch:

  1. pack     the rows this rank owns that the batch touches (src, dst or neg endpoint): whole bundles, layer 0
              included, decayed to the batch's clock                                   (tpnet_pack_bundles)
  2. exchange ONE all-gather of those bundles ((L+1)*d floats per row; the lists are derived from the stream by
              every rank, so no request round is needed)                               (RCCL all_gather)
  3. unpack   the other ranks' bundles into the batch's halo rows, one launch           (tpnet_unpack_bundles)
  4. step     the fused kernel on the local table and relabelled node ids, restricted to the targets this rank owns
              and the pairs whose src it owns: all its reads are pre-batch rows that are either owned or just
              received                                                                 (tpnet_step_batch, own_mod = 0)

Additions are owner-local, so no reduction collective is needed; the readout outputs of the ranks are disjoint
rows and are merged by one all-reduce at the end of the stream.  Results equal the single-GPU results up to one
extra f32 rounding of the exchanged rows' pending decay (<= 1e-6 relative).
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib

# development: issue the collectives even with a single rank (exercises the RCCL calls on a one-GPU box)
_FORCE = os.environ.get("TPNET_DEV_FORCE_COLLECTIVES") == "1"


def plan_touched(src: torch.Tensor, dst: torch.Tensor, neg, batch: int, N: int, G: int):
    """For every batch and every rank: the sorted list of distinct nodes that rank owns and the batch touches.
    Pure torch (any device).  Returns (nodes, counts, offsets): `nodes` is one int64 tensor ordered by
    (batch, owner, node); counts/offsets are host int64 arrays of shape [nb, G] into it."""
    E = int(src.numel())
    nb = (E + batch - 1) // batch
    bidx = torch.arange(E, device=src.device, dtype=torch.int64) // batch
    parts = [src, dst] + ([neg] if neg is not None else [])
    nodes = torch.cat([p.to(torch.int64) for p in parts])
    b3 = torch.cat([bidx] * len(parts))
    uk = torch.unique(b3 * N + nodes)                       # sorted distinct (batch, node)
    ub, un = uk // N, uk % N
    seg = ub * G + (un % G)
    order = torch.argsort(seg * N + un)
    un, seg = un[order], seg[order]
    counts = torch.bincount(seg, minlength=nb * G).view(nb, G).cpu().numpy().astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(counts.reshape(-1))[:-1]]).reshape(nb, G)
    return un.contiguous(), counts, offsets


def plan_targeted(src: torch.Tensor, dst: torch.Tensor, neg, batch: int, N: int, G: int, me: int):
    """TARGETED exchange plan of rank `me` (every row travels only to the ranks that read it).
    Rank r reads, in batch b: dst and neg of the edges whose src it owns (their readouts are its, and dst is the partner of its
    target src), and src of the edges whose dst it owns (the partner of its target dst).  Every rank derives the same lists
    from the stream (no request round).  Pure torch on any device.  Returns a dict:
      recv_nodes   int64 [R]      the remote nodes this rank receives, ordered by (batch, owner, node)
      recv_cnt     int64 [nb, G]  how many of them per batch come from each owner (host)
      send_nodes   int64 [S]      the owned nodes this rank sends, ordered by (batch, reader, node) (a node repeats per reader)
      send_cnt     int64 [nb, G]  how many go to each reader per batch (host)
      recv_keys    int64 [R]      (batch * N + node), ascending inside a batch's owner runs: for the endpoint lookup
    """
    E = int(src.numel())
    nb = (E + batch - 1) // batch
    dev = src.device
    bidx = torch.arange(E, device=dev, dtype=torch.int64) // batch
    s64, d64 = src.to(torch.int64), dst.to(torch.int64)
    readers = [s64 % G, d64 % G]
    nodes = [d64, s64]
    bs = [bidx, bidx]
    if neg is not None:
        readers.append(s64 % G)
        nodes.append(neg.to(torch.int64))
        bs.append(bidx)
    rd, nd, bb = torch.cat(readers), torch.cat(nodes), torch.cat(bs)
    keep = (nd % G) != rd                                        # a rank does not send to itself
    rd, nd, bb = rd[keep], nd[keep], bb[keep]
    # distinct (batch, reader, owner, node), sorted in that order
    key = ((bb * G + rd) * G + (nd % G)) * N + nd
    key = torch.unique(key)
    k_node = key % N
    k_own = (key // N) % G
    k_rd = (key // (N * G)) % G
    k_b = key // (N * G * G)
    mine_r = k_rd == me
    recv_nodes = k_node[mine_r]
    recv_b, recv_o = k_b[mine_r], k_own[mine_r]
    recv_cnt = torch.bincount(recv_b * G + recv_o, minlength=nb * G).view(nb, G).cpu().numpy().astype(np.int64)
    mine_s = k_own == me
    send_nodes, send_b, send_r = k_node[mine_s], k_b[mine_s], k_rd[mine_s]          # already ordered by (batch, reader, node)
    send_cnt = torch.bincount(send_b * G + send_r, minlength=nb * G).view(nb, G).cpu().numpy().astype(np.int64)
    return dict(recv_nodes=recv_nodes.contiguous(), recv_cnt=recv_cnt, send_nodes=send_nodes.contiguous(), send_cnt=send_cnt,
                recv_keys=(recv_b * N + recv_nodes).contiguous(), recv_b=recv_b)


def exchange_rows(send: torch.Tensor, maxc: int, group=None):
    """All-gather of one padded [maxc, row] buffer per rank -> [G, maxc, row]."""
    G = dist.get_world_size(group)
    recv = torch.empty((G,) + tuple(send.shape), dtype=send.dtype, device=send.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(recv, send, group=group)
    else:
        dist.all_gather([recv[r] for r in range(G)], send, group=group)
    return recv


class ShardedStreamRunner:
    """ROW shard of the table (BASELINE.json north_star: "shards row-wise ... all-gather only for cross-shard neighbour
    rows").  Rank `me` of G holds ONLY the rows n with n % G == me -- all L+1 layers, local row n // G -- in a compact
    `RandomProjectionModule` of n_cap + H rows: n_cap = ceil(N / G) owned rows followed by H HALO rows, which per batch
    receive copies of the other ranks' rows that the batch reads (src, dst and neg endpoints).  The kernels run on the local
    table with relabelled node ids (owned: n // G, remote: n_cap + its slot in the batch's halo), restricted to the
    targets / pair sources the rank owns (tpnet_step_batch, own_mod = 0).  Memory per rank: (n_cap + H) rows instead of N.
    """

    exchange = "allgather"     # "allgather" (v1: every touched row to every rank) | "targeted" (v2: each row only to the ranks
                               # that read it, received in place: run_stream_targeted); both with RCCL called from C

    def __init__(self, rp_local, node_num: int, halo_rows: int, group=None, world=None, rank=None):
        self.rp = rp_local
        self.group = group
        # world / rank given: a shard with no process group behind it (several shards in ONE process: the loopback test of the
        # RCCL branch, tests/loopback) -- whoever drives it moves the rows and merges the outputs
        self.detached = world is not None
        self.G = int(world) if world is not None else dist.get_world_size(group)
        self.me = int(rank) if rank is not None else dist.get_rank(group)
        self.N = int(node_num)
        self.n_cap = (self.N + self.G - 1) // self.G
        self.H = int(halo_rows)
        if rp_local.node_num != self.n_cap + self.H:
            raise ValueError(f"local table has {rp_local.node_num} rows, expected n_cap + halo = {self.n_cap} + {self.H}")

    # ---- construction -------------------------------------------------------------------------------------------
    @classmethod
    def create(cls, node_num: int, edge_num: int, dim: int, num_layer: int, time_decay_weight: float, device,
               beginning_time, halo_rows: int, not_scale: bool = False, group=None, seed: int = 0,
               draw_on_device: bool = False, world=None, rank=None):
        """Build the local shard.  `halo_rows` >= the distinct remote nodes one batch can touch (3 * batch is always enough).
        P[0] ~ N(0, 1/sqrt(dim)) (models/TPNet.py:58): each rank draws its own rows from a generator seeded with (seed,
        rank); tests inject a full matrix with set_full_p0."""
        from .random_projection import RandomProjectionModule
        G = int(world) if world is not None else dist.get_world_size(group)
        n_cap = (node_num + G - 1) // G
        rp = RandomProjectionModule(node_num=n_cap + int(halo_rows), edge_num=edge_num, dim_factor=1, num_layer=num_layer,
                                    time_decay_weight=time_decay_weight, device=str(device), use_matrix=False,
                                    beginning_time=beginning_time, not_scale=not_scale, enforce_dim=dim,
                                    alloc_device=device)
        self = cls(rp.to(device), node_num, halo_rows, group, world=world, rank=rank)
        p0 = self.rp._plist()[0]
        if draw_on_device:         # (a 10 M-row shard: 5 GB of normals that need not pass through host memory)
            p0.data[:self.n_cap].copy_(cls.draw_rows(seed, self.me, self.n_cap, dim, p0.device))
        else:
            gen = torch.Generator().manual_seed(seed * 1021 + self.me)
            p0.data[:self.n_cap].copy_(torch.normal(0.0, 1.0 / np.sqrt(dim), (self.n_cap, dim), generator=gen).to(p0.device))
        p0.data[self.n_cap:].zero_()
        return self

    @staticmethod
    def draw_rows(seed: int, rank: int, n_rows: int, dim: int, device):
        """The layer-0 rows rank `rank` draws with draw_on_device (same device, seed and shape -> same values: what a
        single-GPU control run re-assembles the full matrix from)."""
        gen = torch.Generator(device=device).manual_seed(seed * 1021 + rank)
        return torch.normal(0.0, 1.0 / np.sqrt(dim), (n_rows, dim), generator=gen, device=device)

    def owned_nodes(self):
        return torch.arange(self.me, self.N, self.G, dtype=torch.int64, device=self.rp._dev())

    def set_full_p0(self, P0_full: torch.Tensor):
        """Take this rank's rows of a full [N, d] layer-0 matrix (tests, checkpoints of a single-GPU run)."""
        own = P0_full[self.me::self.G]
        p0 = self.rp._plist()[0]
        p0.data[:own.shape[0]].copy_(own.to(p0.device))
        self.rp._params_exposed = False

    def table_bytes(self) -> int:
        """Bytes of table state this rank holds (layer 0 + the two copies of layers 1..L + per-row records)."""
        rp = self.rp
        lib = _lib.load()
        return rp.node_num * rp.dim * 4 + lib.tpnet_q_bytes(rp.node_num, rp.dim, rp.num_layer) + lib.tpnet_meta_bytes(rp.node_num)

    # ---- RCCL from C -----------------------------------------------------------------------------------------------
    def _c_comm(self):
        """An RCCL communicator owned by the C library (tpnet_rccl_comm_create), so that a batch's pack -> all-gather ->
        unpack -> step is ONE FFI call (tpnet_rows_step) instead of three plus a torch.distributed collective.  The
        128-byte unique id travels over the existing process group.  None if the group is not on RCCL or RCCL cannot be
        resolved (the runner then issues the collective through torch.distributed)."""
        if hasattr(self, "_comm"):
            return self._comm
        self._comm = None
        if os.environ.get("TPNET_ROWS_C_LOOP", "1") == "0" or dist.get_backend(self.group) != "nccl":
            return None
        lib = _lib.load()
        dev = self.rp._dev()
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        path_b = path.encode() if os.path.exists(path) else None
        idbuf = (C.c_ubyte * 128)()
        ok = torch.ones(1, dtype=torch.int32, device=dev)
        if self.me == 0 and lib.tpnet_rccl_unique_id(path_b, idbuf) != 0:
            ok.zero_()
        t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device=dev)
        src_rank = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast(t, src=src_rank, group=self.group)
        dist.broadcast(ok, src=src_rank, group=self.group)
        if int(ok.item()) == 0:
            return None
        idb = (C.c_ubyte * 128)(*t.cpu().tolist())
        h = C.c_void_p()
        with torch.cuda.device(dev):
            rc = lib.tpnet_rccl_comm_create(path_b, idb, self.G, self.me, C.byref(h))
        good = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(good, op=dist.ReduceOp.MIN, group=self.group)      # all ranks take the same path
        if int(good.item()) == 1:
            self._comm = h
        elif rc == 0:
            lib.tpnet_rccl_comm_destroy(h)
        return self._comm

    def close(self):
        if getattr(self, "_comm", None):
            _lib.load().tpnet_rccl_comm_destroy(self._comm)
            self._comm = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown: the runtime may be gone already
            pass

    # ---- the stream -----------------------------------------------------------------------------------------------
    def relabel(self, src, dst, neg, batch_size: int):
        """Everything the per-batch loop needs, derived from the stream by every rank on its own (no request round):
        the touched-node lists, the local id of every list entry and of every edge endpoint, the rows to pack."""
        G, me, N, n_cap = self.G, self.me, self.N, self.n_cap
        E, B = int(src.numel()), int(batch_size)
        nb = (E + B - 1) // B
        dev = src.device
        nodes, counts, offsets = plan_touched(src, dst, neg, B, N, G)
        tot = counts.sum(axis=1)
        start = offsets[:, 0] if nb else np.zeros(0, dtype=np.int64)
        remote = tot - counts[:, me]
        if nb and int(remote.max()) > self.H:
            raise ValueError(f"a batch touches {int(remote.max())} rows of other ranks but the shard has {self.H} halo rows "
                             f"(3 * batch_size is always enough)")
        t_tot = torch.from_numpy(tot).to(dev)
        b_of = torch.repeat_interleave(torch.arange(nb, device=dev), t_tot)                  # batch of every list entry
        pos = torch.arange(nodes.numel(), device=dev) - torch.from_numpy(start).to(dev)[b_of]   # position inside its batch's list
        owner = nodes % G
        cnt_me = torch.from_numpy(counts[:, me]).to(dev)[b_of]
        slot = pos - torch.where(owner > me, cnt_me, torch.zeros_like(cnt_me))             # among the batch's REMOTE nodes
        entry_lid = torch.where(owner == me, nodes // G, n_cap + slot)                      # local row of every list entry
        unpack_ids = torch.where(owner == me, torch.full_like(entry_lid, -1), entry_lid).contiguous()
        lkeys = (b_of * G + owner) * N + nodes                                               # ascending by construction
        bidx = torch.arange(E, device=dev, dtype=torch.int64) // B

        def local(x):
            k = (bidx * G + (x % G)) * N + x
            return entry_lid[torch.searchsorted(lkeys, k)].contiguous()

        return dict(nodes=nodes, counts=counts, offsets=offsets, tot=tot, unpack_ids=unpack_ids,
                    pack_ids=(nodes // G).contiguous(), src=local(src), dst=local(dst),
                    neg=local(neg) if neg is not None else None)

    def run_stream(self, src, dst, neg, t, batch_size: int, t_host_last=None, merge_outputs: bool = True, timing=None,
                   out_pos=None, out_neg=None):
        """Same contract as RandomProjectionModule.run_stream (global node ids in, per-edge features out), over all ranks.
        Returns (feat_pos, feat_neg): complete on every rank if merge_outputs (one all-reduce of disjoint rows at the end),
        otherwise each rank holds the rows of the pairs whose src node it owns and zeros elsewhere (a caller's own out_pos /
        out_neg, targeted exchange only: the other rows keep what they held)."""
        if self.exchange == "targeted":
            return self.run_stream_targeted(src, dst, neg, t, batch_size, t_host_last, merge_outputs, timing, out_pos, out_neg)
        rp, G, me = self.rp, self.G, self.me
        rp._ensure_engine()
        lib = _lib.load()
        dev = rp._dev()
        E, B = int(src.numel()), int(batch_size)
        nb = (E + B - 1) // B
        L, d = rp.num_layer, rp.dim
        bundle = (L + 1) * d
        NG = rp.pair_wise_feature_dim
        lam = float(rp.time_decay_weight)
        out_pos = torch.zeros((E, NG), dtype=torch.float32, device=dev)
        out_neg = torch.zeros((E, NG), dtype=torch.float32, device=dev) if neg is not None else None
        if E == 0:
            return out_pos, out_neg
        ends = [src, dst] + ([neg] if neg is not None else [])      # (the lists key on batch * N + node: a bad id would alias)
        if int(torch.stack([x.min() for x in ends]).min()) < 0 or int(torch.stack([x.max() for x in ends]).max()) >= self.N:
            raise IndexError(f"node id out of range for {self.N} nodes")
        R = self.relabel(src, dst, neg, B)
        counts, offsets, tot = R["counts"], R["offsets"], R["tot"]
        last_idx = torch.clamp(torch.arange(1, nb + 1, device=dev) * B, max=E) - 1
        t_last = t[last_idx].cpu().numpy() if t_host_last is None else np.asarray(t_host_last, dtype=np.float64)
        ws = rp._workspace(E, B)
        st = rp._state()
        stream = rp._stream()
        flags = _lib.FLAG_NOT_SCALE if rp.not_scale else 0
        ls, ld, ln = R["src"], R["dst"], R["neg"]
        _lib.check(lib.tpnet_plan_stream(C.byref(st), ls.data_ptr(), ld.data_ptr(), t.data_ptr(), E, B, rp._now_host,
                                         lam, flags, ws.data_ptr(), ws.numel(), stream), "plan_stream")
        lid0 = rp._next_launch_ids(nb)
        maxc_all = max(int(counts.max()), 1)
        send = torch.zeros((maxc_all, bundle), dtype=torch.float32, device=dev)
        recv_all = torch.empty((G, maxc_all, bundle), dtype=torch.float32, device=dev)
        offs_rel = torch.from_numpy(offsets - offsets[:, :1]).to(dev).contiguous()          # [nb, G] owner runs inside a batch
        now = rp._now_host
        exchange = G > 1 or _FORCE
        nccl = exchange and dist.get_backend(self.group) == "nccl"
        # host-side loop: everything per batch is precomputed as plain ints / raw pointers (the loop issues 3 kernel
        # launches + 1 collective per batch and must not be the bottleneck)
        stp = C.byref(st)
        pack_ptr, unpack_ptr = R["pack_ids"].data_ptr(), R["unpack_ids"].data_ptr()
        send_ptr, recv_ptr, offs_ptr = send.data_ptr(), recv_all.data_ptr(), offs_rel.data_ptr()
        cnt_me = counts[:, me].tolist(); off_me = offsets[:, me].tolist(); off_0 = offsets[:, 0].tolist()
        maxcs = counts.max(axis=1).tolist(); tots = tot.tolist(); t_last_l = [float(x) for x in t_last]
        ls_p, ld_p, t_p = ls.data_ptr(), ld.data_ptr(), t.data_ptr()
        ln_p = ln.data_ptr() if ln is not None else None
        op_p = out_pos.data_ptr(); on_p = out_neg.data_ptr() if out_neg is not None else None
        ws_p, ws_n = ws.data_ptr(), ws.numel()
        pack, unpack, step = lib.tpnet_pack_bundles, lib.tpnet_unpack_bundles, lib.tpnet_step_batch
        flat_recv = recv_all.view(-1)
        n_own = self.n_cap
        comm = self._c_comm() if nccl else None
        if comm is not None:
            # RCCL from C: one call per batch enqueues pack -> all-gather -> unpack -> step on the current stream
            rows_step = lib.tpnet_rows_step
            for b in range(nb):
                rc = rows_step(stp, comm, pack_ptr + 8 * off_me[b], cnt_me[b], send_ptr, recv_ptr, maxcs[b],
                               unpack_ptr + 8 * off_0[b], tots[b], offs_ptr + 8 * G * b, G, now, ls_p, ld_p, ln_p, t_p, E, B, b,
                               lam, lid0 + b, flags, n_own, op_p, on_p, ws_p, ws_n, stream)
                if rc:
                    _lib.check(rc, "rows_step")
                now = t_last_l[b]
            nb_done = nb
        else:
            nb_done = 0
        for b in range(nb_done, nb):
            maxc = maxcs[b]
            if exchange and maxc > 0:
                rc = pack(stp, pack_ptr + 8 * off_me[b], cnt_me[b], now, lam, send_ptr, stream)
                if rc:
                    _lib.check(rc, "pack_bundles")
                # one all-gather of the padded [maxc, (L+1)*d] bundles -> recv[G][maxc][(L+1)*d] (a dense prefix of recv_all)
                recv = flat_recv[:G * maxc * bundle]
                if nccl:
                    dist.all_gather_into_tensor(recv, send[:maxc].view(-1), group=self.group)
                else:
                    dist.all_gather(list(recv.view(G, maxc * bundle).unbind(0)), send[:maxc].view(-1), group=self.group)
                # one launch writes every peer's bundles into this batch's halo rows
                rc = unpack(stp, unpack_ptr + 8 * off_0[b], tots[b], now, recv_ptr, maxc, offs_ptr + 8 * G * b, G, stream)
                if rc:
                    _lib.check(rc, "unpack_bundles")
            rc = step(stp, ls_p, ld_p, ln_p, t_p, E, B, b, lam, lid0 + b, flags, 0, n_own, op_p, on_p, ws_p, ws_n, stream)
            if rc:
                _lib.check(rc, "step_batch")
            now = t_last_l[b]
        rp._now_host = now
        rp._params_valid = False
        rp._now_dirty = True
        if G > 1 and merge_outputs:
            dist.all_reduce(out_pos, group=self.group)          # disjoint rows: the sum is a merge
            if out_neg is not None:
                dist.all_reduce(out_neg, group=self.group)
        return out_pos, out_neg

    # ---- targeted exchange (v2) ------------------------------------------------------------------------------------
    def relabel_targeted(self, src, dst, neg, batch_size: int):
        """plan_targeted + the local id of every edge endpoint: owned -> n // G; a remote node this rank receives in the
        edge's batch -> its halo row (n_cap + position in the batch's receive list); any other remote node -> the first halo
        row (never dereferenced by a unit this rank computes: it is neither a partner of one of its targets nor an endpoint of
        one of its pairs)."""
        G, me, N, n_cap = self.G, self.me, self.N, self.n_cap
        E, B = int(src.numel()), int(batch_size)
        nb = (E + B - 1) // B
        dev = src.device
        P = plan_targeted(src, dst, neg, B, N, G, me)
        rtot = P["recv_cnt"].sum(axis=1)
        if nb and int(rtot.max()) > self.H:
            raise ValueError(f"a batch reads {int(rtot.max())} rows of other ranks but the shard has {self.H} halo rows")
        rstart = np.concatenate([[0], np.cumsum(rtot)[:-1]]).astype(np.int64)
        slot = torch.arange(P["recv_nodes"].numel(), device=dev) - torch.from_numpy(rstart).to(dev)[P["recv_b"]]
        bidx = torch.arange(E, device=dev, dtype=torch.int64) // B
        keys = P["recv_keys"]
        # (recv_keys are ascending per batch only inside an owner's run: sort once for the lookup)
        order = torch.argsort(keys)
        skeys, sslot = keys[order], slot[order]

        def local(x):
            x = x.to(torch.int64)
            k = bidx * N + x
            pos = torch.clamp(torch.searchsorted(skeys, k), max=max(int(skeys.numel()) - 1, 0))
            hit = (skeys[pos] == k) if skeys.numel() else torch.zeros_like(k, dtype=torch.bool)
            halo = n_cap + (sslot[pos] if skeys.numel() else torch.zeros_like(k))
            return torch.where(x % G == me, x // G, torch.where(hit, halo, torch.full_like(k, n_cap))).contiguous()

        P.update(src=local(src), dst=local(dst), neg=local(neg) if neg is not None else None,
                 unpack_ids=(n_cap + slot).contiguous(), pack_ids=(P["send_nodes"] // G).contiguous(), rtot=rtot, rstart=rstart)
        return P

    device_plan = True         # the targeted exchange's plan by tpnet_xplan_targeted (two launches, one read-back) where it applies
    xplan_force_large = False  # tests: take tpnet_xplan_targeted_large (one device-wide sort) whatever the batch size

    @staticmethod
    def _stream_sig(src, dst, neg, t, E, B):
        """Identifies the CONTENTS of a call's stream tensors the way RandomProjectionModule.run_stream does (storage + torch's
        in-place version counters): an equal value = the same arrays, unchanged -- what lets a runner keep a stream's exchange plan
        and its kernels' plan across calls (every epoch of train_link_prediction.py:234-253 runs the same stream again)."""
        parts = [E, B]
        for x in (src, dst, neg, t):
            parts += [None, 0] if x is None else [x.data_ptr(), x._version]
        return tuple(parts)

    def _pinned(self, n_int64: int):
        """A pinned host buffer (+ the event of its last copy) for the one read-back of a cold exchange plan: `tensor.cpu()` was a
        pageable allocation + a blocking copy per call."""
        pin = self.__dict__.get("_pin")
        if pin is None or pin[0].numel() < n_int64:
            pin = (torch.empty(max(n_int64, 1024), dtype=torch.int64).pin_memory(), torch.cuda.Event())
            self.__dict__["_pin"] = pin
        return pin

    def _small_scratch(self, nb, E, have_neg, dev, cap):
        """Scratch of one exchange plan of the workgroup-per-batch kernel, kept across calls and only ever grown (torch.empty per call
        was five allocations -- and a call longer than the ones before it paid them inside its timed region)."""
        G = self.G
        sc = self.__dict__.get("_xplan_scratch")
        if sc is None or sc["dev"] != dev or sc["nb"] < nb or sc["E"] < E or sc["G"] != G or (have_neg and sc["lneg"] is None):
            nb_, E_ = max(nb, sc["nb"] if sc else 0), max(E, sc["E"] if sc else 0)
            sc = dict(dev=dev, nb=nb_, E=E_, G=G, recv_keys=torch.empty((nb_, cap), dtype=torch.int32, device=dev),
                      pack_ids=torch.empty((nb_, cap), dtype=torch.int64, device=dev),
                      tail=torch.empty(nb_ * 2 * G + 1, dtype=torch.int64, device=dev),
                      lsrc=torch.empty(E_, dtype=torch.int64, device=dev), ldst=torch.empty(E_, dtype=torch.int64, device=dev),
                      lneg=torch.empty(E_, dtype=torch.int64, device=dev) if (have_neg or (sc and sc["lneg"] is not None)) else None)
            self.__dict__["_xplan_scratch"] = sc
        return sc

    def reserve_stream(self, max_edges: int, batch_size: int, have_neg: bool = True):
        """Size the runner's scratch (exchange plan, read-back buffer) and the module's workspace for run_stream calls of up to
        max_edges edges in batches of batch_size now, so that the first call of that size allocates nothing."""
        rp = self.rp
        rp._ensure_engine()
        E, B = int(max_edges), int(batch_size)
        nb = (E + B - 1) // B
        dev = rp._dev()
        lib = _lib.load()
        cap = int(lib.tpnet_xplan_capacity())
        if not self.xplan_force_large and 3 * B <= 4 * cap:
            self._small_scratch(nb, E, have_neg, dev, cap)
        self._pinned(nb * 2 * self.G + 1)
        if self.G == 1 and self.single_rank_pipeline and not rp.exact:
            rp.reserve_stream(E, B)
        else:
            rp._workspace(E, B, keep_plan=True)
            if self.windowed and self.G > 1:
                need = int(lib.tpnet_wshard_workspace_bytes(rp.node_num, rp.dim, rp.num_layer, E, B, self.G, self.n_cap))
                eng = rp._engine()
                if need and (eng["ws"] is None or eng["ws"].numel() < need):
                    eng["ws"] = torch.empty(need, dtype=torch.uint8, device=dev)
                    rp._drop_plan()

    def _check_pending_status(self):
        """The status words of a one-rank exchange plan whose read-back was left in flight (relabel_targeted_device, G = 1)."""
        ps = self.__dict__.pop("_pending_status", None)
        if ps is not None:
            pin, ev, off = ps
            ev.synchronize()
            if int(pin[off:off + 1].numpy().view(np.uint32)[0]):
                self.__dict__["_xplan_cache"] = None
                raise IndexError(f"node id out of range for {self.N} nodes")

    def check_device_errors(self):
        """Raise IndexError if a call met a node id outside [0, node_num) since the last check (the shard's module included)."""
        self._check_pending_status()
        self.rp.check_device_errors()

    def relabel_targeted_device(self, src, dst, neg, batch_size: int):
        """relabel_targeted on the device (csrc/xplan.hip: one workgroup per batch sorts the batch's receive and send lists, one
        thread per endpoint relabels) -- None where it does not apply (ids not on a GPU, 32-bit keys too narrow for N and G, a
        batch whose lists exceed the kernel's capacity): the caller then takes the torch plan.  Same lists, same order.  One rank
        (G = 1) is served too: no lists, every id its own local row -- the path a forced single-rank run measures."""
        G, me, N, n_cap = self.G, self.me, self.N, self.n_cap
        E, B = int(src.numel()), int(batch_size)
        if G < 1 or G > 64 or not src.is_cuda or E == 0:
            return None
        if max(N - 1, 1).bit_length() + max(G - 1, 1).bit_length() > 31:
            return None
        for x in (src, dst) + ((neg,) if neg is not None else ()):
            if x.dtype != torch.int64 or not x.is_contiguous() or x.device != src.device:
                return None
        lib = _lib.load()
        nb = (E + B - 1) // B
        dev = src.device
        cap = int(lib.tpnet_xplan_capacity())
        n_tail = nb * 2 * G + 1

        def read_back(tail):
            """ONE read-back (message sizes + status) through a pinned buffer: an asynchronous copy + an event, no pageable staging"""
            pin, ev = self._pinned(n_tail)
            pin[:n_tail].copy_(tail, non_blocking=True)
            ev.record(torch.cuda.current_stream(dev))
            ev.synchronize()
            host = pin[:n_tail].numpy()
            st = host[nb * 2 * G:].view(np.uint32)
            if int(st[0]):
                raise IndexError(f"node id out of range for {N} nodes")
            c = host[: nb * 2 * G].reshape(nb, 2, G)
            return int(st[1]), np.ascontiguousarray(c[:, 0, :]), np.ascontiguousarray(c[:, 1, :])   # (copies: the buffer is reused)

        def result(lsrc, ldst, lneg, recv_cnt, send_cnt, pack_ids, sstart):
            rtot = recv_cnt.sum(axis=1)
            if nb and int(rtot.max()) > self.H:
                raise ValueError(f"a batch reads {int(rtot.max())} rows of other ranks but the shard has {self.H} halo rows")
            return dict(src=lsrc, dst=ldst, neg=lneg, send_cnt=send_cnt, recv_cnt=recv_cnt, rtot=rtot, pack_ids=pack_ids, sstart=sstart)

        # a workgroup per batch while a batch's lists can fit its LDS (at most 3 B items per list); else -- or when a batch
        # overflowed after all -- ONE device-wide sort of the call's keys (tpnet_xplan_targeted_large)
        small = not self.xplan_force_large and 3 * B <= 4 * cap
        if small:
            sc = self._small_scratch(nb, E, neg is not None, dev, cap)
            recv_keys, pack_ids, tail = sc["recv_keys"], sc["pack_ids"], sc["tail"][:n_tail]
            lsrc, ldst, lneg = sc["lsrc"][:E], sc["ldst"][:E], (sc["lneg"][:E] if neg is not None else None)
            cnt, status = tail[: nb * 2 * G], tail[nb * 2 * G:]
            rc = lib.tpnet_xplan_targeted(src.data_ptr(), dst.data_ptr(), neg.data_ptr() if neg is not None else None, E, B, N, G, me,
                                          n_cap, recv_keys.data_ptr(), pack_ids.data_ptr(), cnt.data_ptr(), status.data_ptr(),
                                          lsrc.data_ptr(), ldst.data_ptr(), lneg.data_ptr() if lneg is not None else None,
                                          self.rp._stream())
            if rc:
                return None
            if G == 1:
                # ONE rank: no message sizes to wait for -- the status words (ids out of range) are read back asynchronously and
                # looked at by check_device_errors() / the next call; a bad id was relabelled to a halo row, never dereferenced wildly
                pin, ev = self._pinned(n_tail)
                pin[:n_tail].copy_(tail, non_blocking=True)
                ev.record(torch.cuda.current_stream(dev))
                self.__dict__["_pending_status"] = (pin, ev, nb * 2 * G)
                z = np.zeros((nb, 1), dtype=np.int64)
                return result(lsrc, ldst, lneg, z, z.copy(), pack_ids.view(-1), np.arange(nb, dtype=np.int64) * cap)
            over, recv_cnt, send_cnt = read_back(tail)
            if not over:
                return result(lsrc, ldst, lneg, recv_cnt, send_cnt, pack_ids.view(-1), np.arange(nb, dtype=np.int64) * cap)
        if 6 * E >= (1 << 32):
            return None
        need = int(lib.tpnet_xplan_large_bytes(E, B, G))
        key = ("l", nb, G, E, neg is not None, dev)
        sc = self.__dict__.get("_xplan_scratch_l")
        if sc is None or sc[0] != key:
            sc = (key, torch.empty(need, dtype=torch.uint8, device=dev), torch.empty(3 * E, dtype=torch.int64, device=dev),
                  torch.empty(n_tail, dtype=torch.int64, device=dev), torch.empty(E, dtype=torch.int64, device=dev),
                  torch.empty(E, dtype=torch.int64, device=dev), torch.empty(E, dtype=torch.int64, device=dev) if neg is not None else None)
            self.__dict__["_xplan_scratch_l"] = sc
        _, scratch, pack_ids, tail, lsrc, ldst, lneg = sc
        cnt, status = tail[: nb * 2 * G], tail[nb * 2 * G:]
        rc = lib.tpnet_xplan_targeted_large(src.data_ptr(), dst.data_ptr(), neg.data_ptr() if neg is not None else None, E, B, N, G, me,
                                            n_cap, scratch.data_ptr(), scratch.numel(), pack_ids.data_ptr(), cnt.data_ptr(),
                                            status.data_ptr(), lsrc.data_ptr(), ldst.data_ptr(),
                                            lneg.data_ptr() if lneg is not None else None, self.rp._stream())
        if rc:
            return None
        _, recv_cnt, send_cnt = read_back(tail)
        stot = send_cnt.sum(axis=1)
        return result(lsrc, ldst, lneg, recv_cnt, send_cnt, pack_ids, np.concatenate([[0], np.cumsum(stot)[:-1]]).astype(np.int64))

    # a stream that is run again (same tensors, unchanged: _stream_sig) keeps its exchange plan, its relabelled ids and -- while
    # nothing else used the module's workspace and the clock it starts from is the same -- the plan of its kernels
    reuse_plans = True

    def _send_buffers(self, d, L, smax, dev):
        """The pack launch's two send buffers, kept across calls (torch.zeros of both per call sat inside every timed region)."""
        b = self.__dict__.get("_send_bufs")
        if b is None or b[0].shape[0] < smax or b[0].device != dev:
            b = (torch.zeros((smax, d), dtype=torch.float32, device=dev), torch.zeros((smax, L * d), dtype=torch.float32, device=dev))
            self.__dict__["_send_bufs"] = b
        return b

    def prepare_targeted(self, src, dst, neg, t, batch_size: int, t_host_last=None, comm="auto", out_pos=None, out_neg=None,
                         zero_outputs: bool = True):
        """Everything a stream's per-batch calls need (exchange plan, relabelled ids, per-batch plan of the local kernels, send
        buffers, outputs).  comm: "auto" = the C-side RCCL communicator of this shard's process group (None if it is not on
        RCCL); or a communicator handle the caller made (tests/loopback).  out_pos / out_neg: the caller's [E, (2L+2)^2] buffers
        (default: fresh zeroed ones); zero_outputs=False leaves a caller's buffers as they are -- the rows of pairs whose src
        another rank owns then keep what they held (enough where the outputs stay sharded; a merge sums disjoint rows over zeros)."""
        rp, G, me = self.rp, self.G, self.me
        rp._ensure_engine()
        lib = _lib.load()
        dev = rp._dev()
        E, B = int(src.numel()), int(batch_size)
        nb = (E + B - 1) // B
        L, d = rp.num_layer, rp.dim
        NG = rp.pair_wise_feature_dim
        lam = float(rp.time_decay_weight)
        if E == 0:
            z = torch.zeros((0, NG), dtype=torch.float32, device=dev)
            return dict(E=0, B=B, nb=0, out_pos=z, out_neg=z.clone() if neg is not None else None, now=rp._now_host)
        self._check_pending_status()
        if G == 1 and self.single_rank_pipeline and not rp.exact and src.is_cuda:
            # ONE rank: every id is its own local row and nothing travels -- no exchange plan; the stream runs on the module's own
            # schedules (the windowed pipeline from 16 batches on), which count ids out of range themselves (check_device_errors)
            for name, o in (("out_pos", out_pos), ("out_neg", out_neg)):
                if o is not None and (o.dtype != torch.float32 or o.device != dev or not o.is_contiguous() or tuple(o.shape) != (E, NG)):
                    raise ValueError(f"{name} must be a contiguous float32 tensor of shape ({E}, {NG}) on {dev}")
            if out_pos is None:
                out_pos = torch.empty((E, NG), dtype=torch.float32, device=dev)
            if neg is None:
                out_neg = None
            elif out_neg is None:
                out_neg = torch.empty((E, NG), dtype=torch.float32, device=dev)
            t_end = float(np.asarray(t_host_last, dtype=np.float64)[-1]) if t_host_last is not None else float(t[-1].item())
            return dict(E=E, B=B, nb=nb, out_pos=out_pos, out_neg=out_neg, now=rp._now_host, ls=src, ld=dst, ln=neg, t=t,
                        t_last=np.array([t_end], dtype=np.float64), windowed=True, comm=None, nccl=False)
        sig = self._stream_sig(src, dst, neg, t, E, B) if self.reuse_plans else None
        xc = self.__dict__.get("_xplan_cache")
        R = xc[1] if (sig is not None and xc is not None and xc[0] == sig) else None
        if R is None:
            R = self.relabel_targeted_device(src, dst, neg, B) if self.device_plan else None   # (checks the ids' range itself)
            if R is None:
                ends = [src, dst] + ([neg] if neg is not None else [])  # (the lists key on batch * N + node: a bad id would alias)
                if int(torch.stack([x.min() for x in ends]).min()) < 0 or int(torch.stack([x.max() for x in ends]).max()) >= self.N:
                    raise IndexError(f"node id out of range for {self.N} nodes")
                R = self.relabel_targeted(src, dst, neg, B)
            # (the device plan's lists live in the runner's scratch, which the NEXT cold plan overwrites -- and that plan replaces
            # this one in the one-entry cache at the same time; the module's own plan is dropped below for the same reason)
            R["scnt"] = np.ascontiguousarray(R["send_cnt"], dtype=np.int64)
            R["rcnt"] = np.ascontiguousarray(R["recv_cnt"], dtype=np.int64)
            R["stot"] = R["scnt"].sum(axis=1)
            R["sstart_"] = np.ascontiguousarray(R["sstart"] if "sstart" in R else
                                                np.concatenate([[0], np.cumsum(R["stot"])[:-1]]), dtype=np.int64)
            R["smax"] = max(int(R["stot"].max()), 1)
            if sig is not None:
                self.__dict__["_xplan_cache"] = (sig, R)
            # a plan the module's workspace still holds was built on OTHER relabelled arrays (a freed tensor's address can come back
            # with the same version counter: nothing may be replayed across a cold exchange plan)
            rp._drop_plan()
        if t_host_last is None:
            last_idx = torch.clamp(torch.arange(1, nb + 1, device=dev) * B, max=E) - 1
            t_last = t[last_idx].cpu().numpy()
        else:
            t_last = np.asarray(t_host_last, dtype=np.float64)
        for name, o in (("out_pos", out_pos), ("out_neg", out_neg)):
            if o is not None and (o.dtype != torch.float32 or o.device != dev or not o.is_contiguous() or tuple(o.shape) != (E, NG)):
                raise ValueError(f"{name} must be a contiguous float32 tensor of shape ({E}, {NG}) on {dev}")
        if out_pos is None:
            out_pos = torch.zeros((E, NG), dtype=torch.float32, device=dev)
        elif zero_outputs:
            out_pos.zero_()
        if neg is None:
            out_neg = None
        elif out_neg is None:
            out_neg = torch.zeros((E, NG), dtype=torch.float32, device=dev)
        elif zero_outputs:
            out_neg.zero_()
        send_p0, send_q = self._send_buffers(d, L, R["smax"], dev)
        ctx = dict(E=E, B=B, nb=nb, out_pos=out_pos, out_neg=out_neg, now=rp._now_host)
        flags = _lib.FLAG_NOT_SCALE if rp.not_scale else 0
        ls, ld, ln = R["src"], R["dst"], R["neg"]
        if comm == "auto":
            nccl = G > 1 and not self.detached and dist.get_backend(self.group) == "nccl"
            comm = self._c_comm() if nccl else None
        else:
            nccl = comm is not None
        ctx.update(R=R, t_last=np.ascontiguousarray(t_last, dtype=np.float64), flags=flags, lam=lam, ls=ls, ld=ld, ln=ln, t=t,
                   scnt=R["scnt"], rcnt=R["rcnt"], stot=R["stot"], rtot=R["rtot"], sstart=R["sstart_"], send_p0=send_p0, send_q=send_q,
                   comm=comm, nccl=nccl, windowed=False)
        if G == 1 and self.single_rank_pipeline and not rp.exact:
            # ONE rank: no halo, no exchange -- the stream takes the module's own schedules (the windowed pipeline from 16 batches on)
            # on the relabelled ids: RandomProjectionModule.run_stream in steps_targeted
            ctx["windowed"] = True
            return ctx
        ws = rp._workspace(E, B, keep_plan=True)
        st = rp._state()
        stream = rp._stream()
        psig = (sig, rp._now_host, lam, flags, ws.data_ptr(), ws.numel()) if sig is not None else None
        if psig is None or rp.__dict__.get("_rows_plan_sig") != psig:
            rp._drop_plan()
            _lib.check(lib.tpnet_plan_stream(C.byref(st), ls.data_ptr(), ld.data_ptr(), t.data_ptr(), E, B, rp._now_host,
                                             lam, flags, ws.data_ptr(), ws.numel(), stream), "plan_stream")
            rp.__dict__["_rows_plan_sig"] = psig
        ctx.update(ws=ws, st=st, stream=stream, lid0=rp._next_launch_ids(nb))
        return ctx

    single_rank_pipeline = True     # G = 1: the stream runs on the module's own schedules (no halo rows, no exchange)
    schedule = None                 # ... this one (RandomProjectionModule.run_stream's `schedule`; None: the module's default, "auto")

    def steps_targeted(self, ctx, b0: int, b1: int, timing=None):
        """Batches [b0, b1) of a prepared stream in ONE FFI call: per batch pack + grouped ncclSend / ncclRecv (comm; none with one
        rank) + step, all enqueued on the current stream (tpnet_rows_stream_targeted).  `timing` (a dict, measurement only): the
        same loop with HIP events around every batch's exchange and step (tpnet_time_rows_stream_targeted; synchronises) -- filled
        with total_ms, step_ms, exchange_ms."""
        if ctx["E"] == 0 or b0 >= b1:
            return
        if ctx.get("windowed"):
            if (b0, b1) != (0, ctx["nb"]):
                raise ValueError("a single-rank stream on the module's own schedule runs whole: steps_targeted(ctx, 0, nb)")
            rp = self.rp
            rp.run_stream(ctx["ls"], ctx["ld"], ctx["ln"], ctx["t"], ctx["B"], out_pos=ctx["out_pos"], out_neg=ctx["out_neg"],
                          want_neg=ctx["ln"] is not None, t_end=float(ctx["t_last"][-1]), schedule=self.schedule,
                          replay=None if self.reuse_plans else False)
            return
        lib = _lib.load()
        ln, on = ctx["ln"], ctx["out_neg"]
        args = (C.byref(ctx["st"]), ctx["comm"], ctx["R"]["pack_ids"].data_ptr(), ctx["sstart"].ctypes.data,
                ctx["send_p0"].data_ptr(), ctx["send_q"].data_ptr(), ctx["scnt"].ctypes.data,
                ctx["rcnt"].ctypes.data, self.G, self.me, ctx["now"], ctx["t_last"].ctypes.data,
                ctx["ls"].data_ptr(), ctx["ld"].data_ptr(), ln.data_ptr() if ln is not None else None,
                ctx["t"].data_ptr(), ctx["E"], ctx["B"], int(b0), int(b1), ctx["lam"], ctx["lid0"],
                ctx["flags"], self.n_cap, ctx["out_pos"].data_ptr(),
                on.data_ptr() if on is not None else None, ctx["ws"].data_ptr(), ctx["ws"].numel(),
                self.rp._stream())
        if timing is not None:
            tot, stp, xch = C.c_float(0), C.c_float(0), C.c_float(0)
            rc = lib.tpnet_time_rows_stream_targeted(*args, C.byref(tot), C.byref(stp), C.byref(xch))
            timing.update(total_ms=tot.value, step_ms=stp.value, exchange_ms=xch.value, batches=int(b1 - b0))
        else:
            rc = lib.tpnet_rows_stream_targeted(*args)
        if rc:
            _lib.check(rc, "rows_stream_targeted")

    def finish_targeted(self, ctx, merge_outputs: bool = True):
        rp = self.rp
        if ctx["E"] and not ctx.get("windowed"):
            rp._now_host = float(ctx["t_last"][-1])
            rp._params_valid = False
            rp._now_dirty = True
            rp._table_written()
        out_pos, out_neg = ctx["out_pos"], ctx["out_neg"]
        if self.G > 1 and merge_outputs and not self.detached:
            dist.all_reduce(out_pos, group=self.group)
            if out_neg is not None:
                dist.all_reduce(out_neg, group=self.group)
        return out_pos, out_neg

    def _agree_windowed(self, planned: bool, failed, release=None) -> bool:
        """All ranks of a call take the windowed shard or none does.  `planned`: this rank's tpnet_wshard_plan served the call;
        `failed`: the exception it raised instead (out of memory for its workspace, a HIP error), else None.  A rank that failed still
        takes part in the agreement -- with -1 -- and raises its error behind it; the others raise too (after `release()`, which
        frees their plan) instead of waiting for it in the first exchange."""
        ok = planned and failed is None
        if not self.detached:
            flag = torch.tensor([-1 if failed is not None else (1 if planned else 0)], dtype=torch.int32,
                                device=self.rp._dev() if dist.get_backend(self.group) == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            worst = int(flag.item())
            ok = worst > 0
            if worst < 0 and failed is None:
                if release is not None:
                    release()
                raise RuntimeError("run_stream: another rank failed while planning the windowed shard of this call")
        if failed is not None:
            raise failed
        return ok

    def run_stream_targeted(self, src, dst, neg, t, batch_size: int, t_host_last=None, merge_outputs: bool = True, timing=None,
                            out_pos=None, out_neg=None):
        """run_stream with the TARGETED exchange: per batch, pack the owned rows each peer reads (a row repeats per reader), ONE
        grouped ncclSend / ncclRecv issued from C (tpnet_rows_step_targeted) whose receives land straight in the halo rows of
        the local table, step -- two launches per batch, ONE FFI call per stream (tpnet_rows_stream_targeted).  (Without the C
        communicator: the same pack launch, torch.distributed.all_to_all_single into the same rows; gloo in the tests: an
        all-gather of the send buffers from which every rank takes its parts.)  Same results as the all-gather variant; each
        rank receives only what it reads."""
        rp, G, me = self.rp, self.G, self.me
        self.last_stream_windowed = False
        nb_ = (int(src.numel()) + int(batch_size) - 1) // int(batch_size)
        if self.windowed and G > 1 and nb_ >= self.windowed_min_batches and src.is_cuda:
            # every rank takes the same decision: tpnet_wshard_plan declines on shapes (all ranks alike) or on counts that all ranks
            # derive from the same stream -- except a batch whose OWNED contributions overflow one rank's sort: agreed on below
            self._check_pending_status()
            W, failed = None, None
            try:
                W = self.plan_windowed(src, dst, neg, t, batch_size)
            except Exception as ex:          # noqa: BLE001 -- re-raised by _agree_windowed, after every rank has heard of it
                failed = ex
            ok = self._agree_windowed(W is not None, failed, (lambda: _lib.load().tpnet_wshard_destroy(W["handle"])) if W is not None else None)
            if ok:
                return self.run_stream_windowed(src, dst, neg, t, batch_size, t_host_last, merge_outputs, out_pos, out_neg, plan=W,
                                                timing=timing)
            if W is not None:
                _lib.load().tpnet_wshard_destroy(W["handle"])
        ctx = self.prepare_targeted(src, dst, neg, t, batch_size, t_host_last, out_pos=out_pos, out_neg=out_neg,
                                    zero_outputs=merge_outputs and G > 1)
        if ctx["E"] == 0:
            return ctx["out_pos"], ctx["out_neg"]
        if ctx["comm"] is not None or G == 1:
            self.steps_targeted(ctx, 0, ctx["nb"], timing=timing)
            return self.finish_targeted(ctx, merge_outputs)
        # another transport (torch.distributed all_to_all on RCCL without the C communicator; gloo in the tests): the same
        # pack launch, the rows moved into the same halo rows, the same step
        lib = _lib.load()
        dev = rp._dev()
        E, B, nb = ctx["E"], ctx["B"], ctx["nb"]
        L, d = rp.num_layer, rp.dim
        lam, flags, stream, nccl = ctx["lam"], ctx["flags"], ctx["stream"], ctx["nccl"]
        scnt, rcnt, stot, rtot, sstart = ctx["scnt"], ctx["rcnt"], ctx["stot"], ctx["rtot"], ctx["sstart"]
        send_p0, send_q = ctx["send_p0"], ctx["send_q"]
        stp = C.byref(ctx["st"])
        pack_ptr = ctx["R"]["pack_ids"].data_ptr()
        ls_p, ld_p, t_p = ctx["ls"].data_ptr(), ctx["ld"].data_ptr(), ctx["t"].data_ptr()
        ln_p = ctx["ln"].data_ptr() if ctx["ln"] is not None else None
        op_p = ctx["out_pos"].data_ptr(); on_p = ctx["out_neg"].data_ptr() if ctx["out_neg"] is not None else None
        ws_p, ws_n = ctx["ws"].data_ptr(), ctx["ws"].numel()
        lid0, n_cap = ctx["lid0"], self.n_cap
        t_last_l = [float(x) for x in ctx["t_last"]]
        now = ctx["now"]
        p0_t = rp._plist()[0].data                                       # [n_cap + H, d]
        q_t = rp._eng["q"].view(2, rp.node_num, L * d)                   # copy 0 receives
        if not nccl:
            all_scnt = [None] * G
            dist.all_gather_object(all_scnt, scnt, group=self.group)
            gmax = max(max(int(a.sum(axis=1).max()) for a in all_scnt), 1)
            g_p0 = [torch.empty((gmax, d), dtype=torch.float32, device=dev) for _ in range(G)]
            g_q = [torch.empty((gmax, L * d), dtype=torch.float32, device=dev) for _ in range(G)]
            pad_p0 = torch.zeros((gmax, d), dtype=torch.float32, device=dev)
            pad_q = torch.zeros((gmax, L * d), dtype=torch.float32, device=dev)
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(nb)] if timing is not None else None
        cur = torch.cuda.current_stream(dev)
        for b in range(nb):
            ns, nr = int(stot[b]), int(rtot[b])
            if evs:
                evs[b][0].record(cur)
            _lib.check(lib.tpnet_pack_split(stp, pack_ptr + 8 * int(sstart[b]), ns, now, lam, send_p0.data_ptr(),
                                            send_q.data_ptr(), n_cap, nr, stream), "pack_split")
            if nccl:
                dist.all_to_all_single(p0_t[n_cap:n_cap + nr], send_p0[:ns], output_split_sizes=rcnt[b].tolist(),
                                       input_split_sizes=scnt[b].tolist(), group=self.group)
                dist.all_to_all_single(q_t[0, n_cap:n_cap + nr], send_q[:ns], output_split_sizes=rcnt[b].tolist(),
                                       input_split_sizes=scnt[b].tolist(), group=self.group)
            else:
                pad_p0[:ns].copy_(send_p0[:ns]); pad_q[:ns].copy_(send_q[:ns])
                dist.all_gather(g_p0, pad_p0, group=self.group)
                dist.all_gather(g_q, pad_q, group=self.group)
                o = 0
                for s_ in range(G):                  # rows owner s_ packed for me: after what it packed for readers < me
                    c = int(rcnt[b][s_])
                    if c:
                        a0 = int(all_scnt[s_][b][:me].sum())
                        p0_t[n_cap + o:n_cap + o + c].copy_(g_p0[s_][a0:a0 + c])
                        q_t[0, n_cap + o:n_cap + o + c].copy_(g_q[s_][a0:a0 + c])
                        o += c
            if evs:
                evs[b][1].record(cur)
            _lib.check(lib.tpnet_step_batch(stp, ls_p, ld_p, ln_p, t_p, E, B, b, lam, lid0 + b, flags, 0, n_cap, op_p, on_p,
                                            ws_p, ws_n, stream), "step_batch")
            if evs:
                evs[b][2].record(cur)
            now = t_last_l[b]
        if evs:
            evs[-1][2].synchronize()
            timing.update(total_ms=evs[0][0].elapsed_time(evs[-1][2]), batches=nb,
                          step_ms=sum(e[1].elapsed_time(e[2]) for e in evs) / nb,
                          exchange_ms=sum(e[0].elapsed_time(e[1]) for e in evs) / nb)
        return self.finish_targeted(ctx, merge_outputs)

    # ---- the shard on the windowed pipeline (csrc/wshard.hip) -------------------------------------------------------------------
    windowed = True                 # G > 1: one launch + one exchange per WINDOW of batches where tpnet_wshard_plan serves the call
    windowed_min_batches = 16       # (as the single-GPU schedule: below that the per-batch launches win)
    last_stream_windowed = False

    def _wshard_buffers(self, d, L, n_p0, n_send, n_recv, dev):
        """The windowed shard's exchange buffers, kept across calls and only ever grown."""
        b = self.__dict__.get("_wshard_bufs")
        need = (max(n_p0, 1), max(n_send, 1), max(n_recv, 1))
        if b is None or b["dev"] != dev or any(x < y for x, y in zip(b["n"], need)):
            n = tuple(max(x, y) for x, y in zip(b["n"], need)) if (b is not None and b["dev"] == dev) else need
            b = dict(dev=dev, n=n, send_p0=torch.empty((n[0], d), dtype=torch.float32, device=dev),
                     send_q=torch.empty((n[0], L * d), dtype=torch.float32, device=dev),
                     sendbuf=torch.empty((n[1], d), dtype=torch.float32, device=dev),
                     recvbuf=torch.empty((n[2], d), dtype=torch.float32, device=dev))
            self.__dict__["_wshard_bufs"] = b
        return b

    def plan_windowed(self, src, dst, neg, t, batch_size: int, want_pos: bool = True, want_neg: bool = True):
        """tpnet_wshard_plan for one call (the whole call is one chunk of the pipeline): a dict with the plan's handle, its counts
        and the exchange buffers -- or None where the windowed shard does not serve the call (the per-batch shard then does)."""
        rp, G, me = self.rp, self.G, self.me
        rp._ensure_engine()
        lib = _lib.load()
        dev = rp._dev()
        E, B = int(src.numel()), int(batch_size)
        L, d = rp.num_layer, rp.dim
        if E == 0 or rp.exact or (E + B - 1) // B < 4:
            return None
        for x, dt in ((src, torch.int64), (dst, torch.int64), (t, torch.float64)) + (((neg, torch.int64),) if neg is not None else ()):
            if x.device != dev or x.dtype != dt or not x.is_contiguous() or x.numel() != E:
                return None
        need = int(lib.tpnet_wshard_workspace_bytes(rp.node_num, d, L, E, B, G, self.n_cap))
        if need == 0:
            return None
        eng = rp._engine()
        if eng["ws"] is None or eng["ws"].numel() < need:
            eng["ws"] = torch.empty(need, dtype=torch.uint8, device=dev)
        rp._drop_plan()                                   # (the workspace is this call's now)
        ws = eng["ws"]
        h = C.c_void_p()
        flags = _lib.FLAG_NOT_SCALE if rp.not_scale else 0
        want_neg = want_neg and neg is not None
        rc = lib.tpnet_wshard_plan(rp._st_ref(), src.data_ptr(), dst.data_ptr(), neg.data_ptr() if neg is not None else None, t.data_ptr(),
                                   E, B, self.N, G, me, self.n_cap, rp._now_host, float(rp.time_decay_weight), flags,
                                   1 if want_pos else 0, 1 if want_neg else 0, ws.data_ptr(), ws.numel(), rp._stream(), C.byref(h))
        if rc == 1:
            return None
        _lib.check(rc, "wshard_plan")
        try:
            return self._plan_windowed_info(h, E, B, ws, want_pos, want_neg)
        except Exception:
            lib.tpnet_wshard_destroy(h)
            raise

    def _plan_windowed_info(self, h, E, B, ws, want_pos, want_neg):
        rp, G, me = self.rp, self.G, self.me
        lib = _lib.load()
        dev = rp._dev()
        L, d = rp.num_layer, rp.dim
        ns, halo, ms, mr = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        pc, psc, prc = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)()
        _lib.check(lib.tpnet_wshard_info(h, C.byref(ns), C.byref(halo), C.byref(ms), C.byref(mr), C.byref(pc), C.byref(psc), C.byref(prc)),
                   "wshard_info")
        nsteps = int(ns.value)
        chunk_cnt = np.ctypeslib.as_array(pc, shape=(G,)).copy()
        send_cnt = np.ctypeslib.as_array(psc, shape=(nsteps, G)).copy()
        recv_cnt = np.ctypeslib.as_array(prc, shape=(nsteps, G)).copy()
        bufs = self._wshard_buffers(d, L, int(chunk_cnt[me]), int(ms.value), int(mr.value), dev)
        _lib.check(lib.tpnet_wshard_set_buffers(h, bufs["send_p0"].data_ptr(), bufs["send_q"].data_ptr(), bufs["sendbuf"].data_ptr(),
                                                bufs["recvbuf"].data_ptr()), "wshard_set_buffers")
        hstart = np.concatenate([[0], np.cumsum([0 if o == me else int(chunk_cnt[o]) for o in range(G)])]).astype(np.int64)
        return dict(handle=h, nsteps=nsteps, halo=int(halo.value), chunk_cnt=chunk_cnt, send_cnt=send_cnt, recv_cnt=recv_cnt, bufs=bufs,
                    hstart=hstart, E=E, B=B, ws=ws, want_pos=want_pos, want_neg=want_neg)

    def run_stream_windowed(self, src, dst, neg, t, batch_size: int, t_host_last=None, merge_outputs: bool = True, out_pos=None,
                            out_neg=None, comm="auto", plan=None, timing=None):
        """run_stream on the windowed pipeline (see the module's doc and csrc/wshard.hip): one launch + one exchange per window of
        batches.  Returns (feat_pos, feat_neg), or None where tpnet_wshard_plan does not serve the call (nothing has run then)."""
        rp, G, me = self.rp, self.G, self.me
        W = plan if plan is not None else self.plan_windowed(src, dst, neg, t, batch_size)
        if W is None:
            return None
        lib = _lib.load()
        dev = rp._dev()
        E, NG = W["E"], rp.pair_wise_feature_dim
        h = W["handle"]
        try:
            zero = merge_outputs and G > 1
            for name, o in (("out_pos", out_pos), ("out_neg", out_neg)):
                if o is not None and (o.dtype != torch.float32 or o.device != dev or not o.is_contiguous() or tuple(o.shape) != (E, NG)):
                    raise ValueError(f"{name} must be a contiguous float32 tensor of shape ({E}, {NG}) on {dev}")
            if out_pos is None:
                out_pos = torch.zeros((E, NG), dtype=torch.float32, device=dev)
            elif zero:
                out_pos.zero_()
            if neg is None:
                out_neg = None
            elif out_neg is None:
                out_neg = torch.zeros((E, NG), dtype=torch.float32, device=dev)
            elif zero:
                out_neg.zero_()
            lid = rp._next_launch_ids(1)
            stream = rp._stream()
            op, on = out_pos.data_ptr(), (out_neg.data_ptr() if out_neg is not None else None)
            if comm == "auto":
                nccl = G > 1 and not self.detached and dist.get_backend(self.group) == "nccl"
                comm = self._c_comm() if nccl else None
            if comm is not None or G == 1:
                if timing is not None:
                    tot, la, xc = C.c_float(0), C.c_float(0), C.c_float(0)
                    _lib.check(lib.tpnet_time_wshard_run(h, comm, op, on, lid, stream, C.byref(tot), C.byref(la), C.byref(xc)),
                               "time_wshard_run")
                    timing.update(total_ms=tot.value, step_ms=la.value, exchange_ms=xc.value, batches=W["nsteps"], windowed=True)
                else:
                    _lib.check(lib.tpnet_wshard_run(h, comm, op, on, lid, stream), "wshard_run")
            else:
                self._wshard_run_gloo(W, op, on, lid, stream, timing)
            if timing is not None:
                timing.update(windowed=True, launches=W["nsteps"], rows_sent_per_launch=float(W["send_cnt"].sum()) / W["nsteps"],
                              rows_received_per_launch=float(W["recv_cnt"].sum()) / W["nsteps"],
                              chunk_rows_sent=int(W["chunk_cnt"][me]) * (G - 1), chunk_rows_received=int(W["halo"]))
            if t_host_last is not None:
                t_end = float(np.asarray(t_host_last, dtype=np.float64)[-1])
            else:
                t_end = float(t[-1].item())
            rp._now_host = t_end
            rp._params_valid = False
            rp._now_dirty = True
            rp._table_written()
        finally:
            lib.tpnet_wshard_destroy(h)
        self.last_stream_windowed = True
        if G > 1 and merge_outputs and not self.detached:
            dist.all_reduce(out_pos, group=self.group)
            if out_neg is not None:
                dist.all_reduce(out_neg, group=self.group)
        return out_pos, out_neg

    def _wshard_run_gloo(self, W, op, on, lid, stream, timing=None):
        """The windowed shard's loop with the rows moved by torch.distributed (gloo in the tests; a process group without the C-side
        RCCL communicator): pack launches, an all-gather of the padded send buffers from which every rank takes its parts, unpack
        launches -- the same pack / unpack kernels and the same buffers as the RCCL loop."""
        lib = _lib.load()
        rp, G, me = self.rp, self.G, self.me
        dev = rp._dev()
        L, d = rp.num_layer, rp.dim
        h, bufs = W["handle"], W["bufs"]
        all_plans = [None] * G
        dist.all_gather_object(all_plans, (W["chunk_cnt"], W["send_cnt"], W["recv_cnt"]), group=self.group)
        PH_LAUNCH, PH_PACK, PH_UNPACK = 1, 2, 8
        # ---- the chunk's halo rows
        _lib.check(lib.tpnet_wshard_begin(h, None, PH_PACK, stream), "wshard_begin")
        cmax = max(int(max(p[0])) for p in all_plans) or 1
        pad_p0 = torch.zeros((cmax, d), dtype=torch.float32, device=dev)
        pad_q = torch.zeros((cmax, L * d), dtype=torch.float32, device=dev)
        mine = int(W["chunk_cnt"][me])
        pad_p0[:mine].copy_(bufs["send_p0"][:mine]); pad_q[:mine].copy_(bufs["send_q"][:mine])
        g_p0 = [torch.empty_like(pad_p0) for _ in range(G)]
        g_q = [torch.empty_like(pad_q) for _ in range(G)]
        dist.all_gather(g_p0, pad_p0, group=self.group)
        dist.all_gather(g_q, pad_q, group=self.group)
        p0_t = rp._plist()[0].data
        q_t = rp._eng["q"].view(2, rp.node_num, L * d)
        for o in range(G):
            c = int(W["chunk_cnt"][o])
            if o != me and c:
                a0 = self.n_cap + int(W["hstart"][o])
                p0_t[a0:a0 + c].copy_(g_p0[o][:c])
                q_t[0, a0:a0 + c].copy_(g_q[o][:c])
        # ---- the pipeline
        smax = max(int(p[1].sum(axis=1).max()) for p in all_plans) or 1
        pad = torch.zeros((smax, d), dtype=torch.float32, device=dev)
        gath = [torch.empty_like(pad) for _ in range(G)]
        cur = torch.cuda.current_stream(dev)
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(W["nsteps"])] if timing is not None else None
        for j in range(W["nsteps"]):
            if evs:
                evs[j][0].record(cur)
            _lib.check(lib.tpnet_wshard_step(h, None, j, PH_LAUNCH, op, on, stream), "wshard_step")
            if evs:
                evs[j][1].record(cur)
            _lib.check(lib.tpnet_wshard_step(h, None, j, PH_PACK, op, on, stream), "wshard_step")
            if any(int(p[1][j].sum()) for p in all_plans):
                ns = int(W["send_cnt"][j].sum())
                pad[:ns].copy_(bufs["sendbuf"][:ns])
                dist.all_gather(gath, pad, group=self.group)
                ro = 0
                for o in range(G):                      # what owner o packed for me sits behind what it packed for the peers before me
                    c = int(W["recv_cnt"][j][o])
                    if c:
                        a0 = int(all_plans[o][1][j][:me].sum())
                        bufs["recvbuf"][ro:ro + c].copy_(gath[o][a0:a0 + c])
                    ro += c
                _lib.check(lib.tpnet_wshard_step(h, None, j, PH_UNPACK, op, on, stream), "wshard_step")
            if evs:
                evs[j][2].record(cur)
        _lib.check(lib.tpnet_wshard_finish(h, lid, stream), "wshard_finish")
        if evs:
            evs[-1][2].synchronize()
            n = len(evs)
            timing.update(total_ms=evs[0][0].elapsed_time(evs[-1][2]), step_ms=sum(e[0].elapsed_time(e[1]) for e in evs) / n,
                          exchange_ms=sum(e[1].elapsed_time(e[2]) for e in evs) / n, batches=n)

    def gather_full_layers(self):
        """All ranks' owned rows interleaved back into global order: [L+1, N, d] on every rank (tests / checkpoints of small
        tables: N * d * (L+1) * 4 bytes)."""
        rp, G, N, n_cap = self.rp, self.G, self.N, self.n_cap
        dev = rp._dev()
        rows = rp.get_random_projections(torch.arange(n_cap, dtype=torch.int64, device=dev))   # owned rows (+ padding)
        local = torch.stack(rows).contiguous()                                                  # [L+1, n_cap, d]
        if G == 1:
            return local[:, :N]
        parts = [torch.empty_like(local) for _ in range(G)]
        dist.all_gather(parts, local, group=self.group)
        full = torch.empty((local.shape[0], n_cap * G, local.shape[2]), dtype=local.dtype, device=dev)
        for r in range(G):
            full[:, r::G] = parts[r]
        return full[:, :N]


# =====================================================================================================================
class ColumnShardedRunner:
    """Column (dim) sharding of the table over the ranks of `group`: rank r holds columns [r*d/G, (r+1)*d/G) of every
    row of P[0..L] in an ordinary `RandomProjectionModule` of dim d/G.

    * update():  local, no communication (each column of models/TPNet.py:87-97 evolves on its own; the time weights
      and decay factors do not depend on the column).  The local slice is bit-identical to the same columns of a
      single-GPU table.
    * readout:   the local Gram entries are partial inner products (sum over the local columns); they are summed
      across ranks BEFORE the x<0 -> 0, log(x+1) tail (tpnet_gram_finish), so the result differs from the
      single-GPU one only in the association of the f32 sum over d.
    * stream:    `run_stream` walks the stream in chunks of steps; a chunk's raw features are reduce-scattered over
      the ranks (each rank ends up with the finished features of 1/G of the chunk's edges -- the shape a
      data-parallel decoder consumes) while the next chunk's kernels run.
    """

    def __init__(self, rp_local, full_dim: int, group=None):
        self.rp = rp_local
        self.group = group
        self.G = dist.get_world_size(group)
        self.me = dist.get_rank(group)
        self.full_dim = int(full_dim)
        if rp_local.dim * self.G != self.full_dim:
            raise ValueError(f"local dim {rp_local.dim} x {self.G} ranks != full dim {full_dim}")

    # ---- construction -------------------------------------------------------------------------------------------
    @classmethod
    def create(cls, node_num: int, edge_num: int, dim: int, num_layer: int, time_decay_weight: float, device,
               beginning_time, not_scale: bool = False, group=None, seed: int = 0, exact: bool = False):
        """Build the local module.  P[0] ~ N(0, 1/sqrt(dim)) over the FULL width (models/TPNet.py:58); each rank draws
        its own columns from a generator seeded with (seed, rank), so the slices are independent whatever the
        process-wide seed is."""
        from .random_projection import RandomProjectionModule
        G = dist.get_world_size(group)
        me = dist.get_rank(group)
        if dim % G:
            raise ValueError(f"dim {dim} is not divisible by the {G} ranks")
        rp = RandomProjectionModule(node_num=node_num, edge_num=edge_num, dim_factor=1, num_layer=num_layer,
                                    time_decay_weight=time_decay_weight, device=str(device), use_matrix=False,
                                    beginning_time=beginning_time, not_scale=not_scale, enforce_dim=dim // G,
                                    exact=exact)
        self = cls(rp, dim, group)
        self._seed = int(seed)
        self._draw_p0(0)
        self.rp = rp.to(device)
        return self

    def _draw_p0(self, epoch: int):
        gen = torch.Generator().manual_seed((self._seed * 1000003 + epoch) * 1021 + self.me)
        p0 = self.rp._plist()[0]
        fresh = torch.normal(0.0, 1.0 / np.sqrt(self.full_dim), (self.rp.node_num, self.rp.dim), generator=gen)
        p0.data.copy_(fresh.to(p0.device))

    def set_full_p0(self, P0_full: torch.Tensor):
        """Take this rank's columns of a full-width [N, d] layer-0 matrix (tests, checkpoints of a single-GPU run)."""
        dl = self.rp.dim
        self.rp._plist()[0].data.copy_(P0_full[:, self.me * dl:(self.me + 1) * dl].to(self.rp._plist()[0].device))

    def reset_random_projections(self, epoch: int = 0):
        """models/TPNet.py:132-139 on the shard: layers 1..L to zero, clock to the beginning, P[0] redrawn."""
        self.rp.reset_random_projections()
        if hasattr(self, "_seed"):
            self._draw_p0(epoch + 1)

    # ---- module-level API -----------------------------------------------------------------------------------------
    def update(self, src_node_ids, dst_node_ids, node_interact_times):
        self.rp.update(src_node_ids=src_node_ids, dst_node_ids=dst_node_ids, node_interact_times=node_interact_times)

    def _unpack(self, packed: torch.Tensor, out: torch.Tensor):
        """Summed packed rows -> full feature rows, with the element-wise tail (tpnet_gram_unpack)."""
        n = packed.shape[0]
        if n:
            flags = _lib.FLAG_NOT_SCALE if self.rp.not_scale else 0
            _lib.check(_lib.load().tpnet_gram_unpack(packed.data_ptr(), n, self.rp.num_layer, flags, out.data_ptr(),
                                                     self.rp._stream()), "gram_unpack")
        return out

    def pair_gram(self, src_node_ids, dst_node_ids) -> torch.Tensor:
        """Complete pre-mlp features on every rank: local partial Gram (distinct entries only), all-reduce, unpack."""
        pk = self.rp.pair_gram(src_node_ids, dst_node_ids, packed=True)
        if self.G > 1 or _FORCE:
            dist.all_reduce(pk, group=self.group)
        out = torch.empty((pk.shape[0], self.rp.pair_wise_feature_dim), dtype=torch.float32, device=pk.device)
        return self._unpack(pk, out)

    def get_pair_wise_feature(self, src_node_ids, dst_node_ids):
        return self.rp._apply_mlp(self.pair_gram(src_node_ids, dst_node_ids))

    # ---- device-resident stream -----------------------------------------------------------------------------------
    @staticmethod
    def chunk_bounds(E: int, B: int, chunk_steps=None, chunk_edges: int = 2_000_000, min_tail_steps: int = 64):
        """Edge ranges [a, b) of the chunks: whole batches, about `chunk_edges` edges each (a plan of a few million
        keys takes rocPRIM's one-sweep radix sort; much smaller ones fall to its merge sort, ~2x the time per key).  The
        LAST chunk is halved repeatedly down to `min_tail_steps` steps: its collective has no later kernels to hide
        behind, so it should be short.  `chunk_steps` fixes the size instead (no tail splitting)."""
        if chunk_steps is not None:
            ce = max(1, int(chunk_steps)) * B
            return [(a, min(a + ce, E)) for a in range(0, E, ce)]
        steps = max(1, int(chunk_edges) // B)
        ce = steps * B
        bounds = [(a, min(a + ce, E)) for a in range(0, E, ce)]
        a, b = bounds.pop()
        nb = (b - a + B - 1) // B
        while nb >= 2 * min_tail_steps:
            h = nb // 2
            bounds.append((a, a + h * B))
            a += h * B
            nb -= h
        bounds.append((a, b))
        return bounds

    def run_stream(self, src, dst, neg, t, batch_size: int, chunk_steps=None, merge: str = "scatter",
                   t_chunk_last=None, chunk_edges: int = 2_000_000):
        """Same per-batch semantics as RandomProjectionModule.run_stream, over all ranks.

        merge = "scatter": returns (feat_pos, feat_neg, edge_index): this rank's share of the finished features,
                           [M, (2L+2)^2] each, and the global edge number of every row (int64 [M], -1 = padding row of
                           a chunk whose edge count is not a multiple of the world size).
        merge = "all":     returns (feat_pos, feat_neg, None) with all E rows on every rank (all-reduce).
        On the wire: the (2L+2)(2L+3)/2 distinct raw Gram entries per pair (144 B at L=3), f32.
        Chunks: see chunk_bounds.  `t_chunk_last`: t at the last edge of every chunk, if the caller has it on the
        host (else one small device->host copy up front)."""
        rp, G, me = self.rp, self.G, self.me
        if merge not in ("scatter", "all"):
            raise ValueError("merge must be 'scatter' or 'all'")
        dev = rp._dev()
        E, B = int(src.numel()), int(batch_size)
        NG, NT = rp.pair_wise_feature_dim, rp.packed_feature_dim
        bounds = self.chunk_bounds(E, B, chunk_steps, chunk_edges)
        have_neg = neg is not None
        if E == 0:
            z = torch.empty((0, NG), dtype=torch.float32, device=dev)
            return z, (z.clone() if have_neg else None), (torch.empty(0, dtype=torch.int64, device=dev) if merge == "scatter" else None)
        if t_chunk_last is None:
            idx = torch.tensor([b - 1 for _, b in bounds], device=dev)
            t_chunk_last = t[idx].cpu().numpy()
        coll = G > 1 or _FORCE
        nccl = coll and dist.get_backend(self.group) == "nccl"
        K = 2 if have_neg else 1
        scatter = merge == "scatter"
        # rank r receives rows [r*m, (r+1)*m) of every (padded) chunk in scatter mode, all rows otherwise
        ms = [((b - a) + G - 1) // G if scatter else (b - a) for a, b in bounds]
        offs = np.concatenate([[0], np.cumsum(ms)]).astype(np.int64)
        M = int(offs[-1])
        outs = [torch.empty((M, NG), dtype=torch.float32, device=dev) for _ in range(K)]
        edge_index = torch.empty(M, dtype=torch.int64, device=dev) if scatter else None
        mmax = max(ms)
        rows_raw = mmax * G if scatter else mmax
        # two rotating sets of buffers: chunk c+1 is computed while chunk c is on the wire
        raws = [[torch.zeros((rows_raw, NT), dtype=torch.float32, device=dev) for _ in range(K)] for _ in range(2)]
        recvs = [[torch.empty((mmax, NT), dtype=torch.float32, device=dev) for _ in range(K)] for _ in range(2)] \
            if scatter else raws
        works = []

        def finish(c):                            # chunk c: wait for its collectives, then unpack + element-wise tail
            self._complete(works, c)
            for k in range(K):
                self._unpack(recvs[c % 2][k][:ms[c]], outs[k][offs[c]:offs[c + 1]])

        for c, (a, b) in enumerate(bounds):
            n, m = b - a, ms[c]
            raw = raws[c % 2]
            if scatter and n < m * G:             # padding rows of a ragged chunk: defined values on the wire
                for r_ in raw:
                    r_[n:m * G].zero_()
            rp.run_stream(src[a:b], dst[a:b], neg[a:b] if have_neg else None, t[a:b], B, out_pos=raw[0][:n],
                          out_neg=raw[1][:n] if have_neg else None, t_end=float(t_chunk_last[c]), packed=True)
            w = []
            for k in range(K):
                if not scatter:
                    if coll:
                        w.append(dist.all_reduce(raw[k][:n], group=self.group, async_op=True))
                elif nccl:
                    w.append(dist.reduce_scatter_tensor(recvs[c % 2][k][:m], raw[k][:m * G], group=self.group,
                                                        async_op=True))
                elif coll:                         # gloo (tests): all-reduce, keep the own share
                    dist.all_reduce(raw[k][:m * G], group=self.group)
                    recvs[c % 2][k][:m].copy_(raw[k][me * m:(me + 1) * m])
                else:
                    recvs[c % 2][k][:m].copy_(raw[k][:m])
            works.append(w)
            if scatter:
                ids = a + me * m + torch.arange(m, device=dev, dtype=torch.int64)
                edge_index[offs[c]:offs[c + 1]] = torch.where(ids < b, ids, torch.full_like(ids, -1))
            if c >= 1:                            # chunk c-1 has had chunk c's kernels to hide behind
                finish(c - 1)
        finish(len(bounds) - 1)
        return outs[0], (outs[1] if have_neg else None), edge_index

    @staticmethod
    def _complete(works, c):
        if c < len(works):
            for w in works[c]:
                if w is not None:
                    w.wait()                      # the compute stream waits for the collective (no host block with nccl)
            works[c] = []

    def gather_full_layers(self):
        """All ranks' column slices side by side: [L+1, N, d] on every rank (tests / checkpoints; N*d*(L+1)*4 bytes)."""
        rp = self.rp
        local = torch.stack([rp.random_projections[i].detach() for i in range(rp.num_layer + 1)])    # [L+1, N, dl]
        if self.G == 1:
            return local
        parts = [torch.empty_like(local) for _ in range(self.G)]
        dist.all_gather(parts, local.contiguous(), group=self.group)
        return torch.cat(parts, dim=2)
