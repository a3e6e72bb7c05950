"""Row-sharded multi-GPU runner of the temporal-walk-matrix hot path (new capability: the
reference is single-device, utils/load_configs.py:88).

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI in production; "gloo" in the tests).
Rows are owned cyclically, owner(n) = n % G (spreads the power-law head over the ranks).  Every rank holds the
whole edge stream (32 bytes per edge) and the static layer P[0]; layers 1..L of a row are authoritative on its
owner only.  Per batch:

  1. pack     the rows this rank owns that the batch touches (src, dst or neg endpoint), decayed to the batch's
              clock                                                                    (tpnet_pack_rows)
  2. exchange ONE all-gather of those row bundles (L*d floats per row; the lists are derived from the stream by
              every rank, so no request round is needed)                               (RCCL all_gather)
  3. unpack   the other ranks' rows into the local table, one launch                    (tpnet_unpack_gathered)
  4. step     the fused kernel restricted to the targets this rank owns and the pairs whose src it owns: all its
              reads are pre-batch rows that are either owned or just refreshed         (tpnet_step_batch)

Additions are owner-local, so no reduction collective is needed; the readout outputs of the ranks are disjoint
rows and are merged by one all-reduce at the end of the stream.  Results equal the single-GPU results up to one
extra f32 rounding of the exchanged rows' pending decay (<= 1e-6 relative).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


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
    """Drives a `RandomProjectionModule` (one per rank, identical construction and P[0]) as one row shard."""

    def __init__(self, rp, group=None):
        self.rp = rp
        self.group = group
        self.G = dist.get_world_size(group)
        self.me = dist.get_rank(group)
        # identical layer 0 everywhere (the reference draws it from the device RNG: TPNet.py:58,139)
        dist.broadcast(rp._plist()[0].data, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)

    def owned_nodes(self):
        return torch.arange(self.me, self.rp.node_num, self.G, dtype=torch.int64, device=self.rp._dev())

    def run_stream(self, src, dst, neg, t, batch_size: int, t_host_last=None, merge_outputs: bool = True):
        """Same contract as RandomProjectionModule.run_stream, over all ranks.  Returns (feat_pos, feat_neg): complete
        on every rank if merge_outputs (one all-reduce of disjoint rows at the end), otherwise each rank holds the
        rows of the pairs whose src node it owns and zeros elsewhere (a sharded consumer needs no merge)."""
        rp, G, me = self.rp, self.G, self.me
        rp._ensure_engine()
        lib = _lib.load()
        dev = rp._dev()
        E = int(src.numel())
        B = int(batch_size)
        nb = (E + B - 1) // B
        L, d, N = rp.num_layer, rp.dim, rp.node_num
        row = L * d
        NG = rp.pair_wise_feature_dim
        lam = float(rp.time_decay_weight)
        out_pos = torch.zeros((E, NG), dtype=torch.float32, device=dev)
        out_neg = torch.zeros((E, NG), dtype=torch.float32, device=dev) if neg is not None else None
        if E == 0:
            return out_pos, out_neg
        nodes, counts, offsets = plan_touched(src, dst, neg, B, N, G)
        # clock after each batch (host copy: one small device->host transfer per stream)
        last_idx = torch.clamp(torch.arange(1, nb + 1, device=dev) * B, max=E) - 1
        t_last = t[last_idx].cpu().numpy() if t_host_last is None else np.asarray(t_host_last, dtype=np.float64)
        ws = rp._workspace(E, B)
        st = rp._state()
        stream = rp._stream()
        flags = _lib.FLAG_NOT_SCALE if rp.not_scale else 0
        _lib.check(lib.tpnet_plan_stream(C.byref(st), src.data_ptr(), dst.data_ptr(), t.data_ptr(), E, B, rp._now_host,
                                         lam, flags, ws.data_ptr(), ws.numel(), stream), "plan_stream")
        lid0 = rp._next_launch_ids(nb)
        maxc_all = int(counts.max()) if counts.size else 0
        send = torch.zeros((max(maxc_all, 1), row), dtype=torch.float32, device=dev)
        recv_all = torch.empty((G, max(maxc_all, 1), row), dtype=torch.float32, device=dev)
        offs_rel = torch.from_numpy(offsets - offsets[:, :1]).to(dev).contiguous()   # [nb, G] owner runs inside a batch
        tot = counts.sum(axis=1)
        now = rp._now_host
        nccl = G > 1 and dist.get_backend(self.group) == "nccl"
        # host-side loop: everything per batch is precomputed as plain ints / raw pointers (the loop issues 3 kernel
        # launches + 1 collective per batch and must not be the bottleneck)
        stp = C.byref(st)
        nodes_ptr, send_ptr, recv_ptr = nodes.data_ptr(), send.data_ptr(), recv_all.data_ptr()
        offs_ptr = offs_rel.data_ptr()
        cnt_me = counts[:, me].tolist(); off_me = offsets[:, me].tolist(); off_0 = offsets[:, 0].tolist()
        maxcs = counts.max(axis=1).tolist(); tots = tot.tolist(); t_last_l = [float(x) for x in t_last]
        src_p, dst_p, t_p = src.data_ptr(), dst.data_ptr(), t.data_ptr()
        neg_p = neg.data_ptr() if neg is not None else None
        op_p = out_pos.data_ptr(); on_p = out_neg.data_ptr() if out_neg is not None else None
        ws_p, ws_n = ws.data_ptr(), ws.numel()
        pack, unpack, step = lib.tpnet_pack_rows, lib.tpnet_unpack_gathered, lib.tpnet_step_batch
        flat_recv = recv_all.view(-1)
        for b in range(nb):
            maxc = maxcs[b]
            if G > 1 and maxc > 0:
                rc = pack(stp, nodes_ptr + 8 * off_me[b], cnt_me[b], now, lam, send_ptr, stream)
                if rc:
                    _lib.check(rc, "pack_rows")
                # one all-gather of the padded [maxc, L*d] bundles -> recv[G][maxc][L*d] (a dense prefix of recv_all)
                recv = flat_recv[:G * maxc * row]
                if nccl:
                    dist.all_gather_into_tensor(recv, send[:maxc].view(-1), group=self.group)
                else:
                    dist.all_gather(list(recv.view(G, maxc * row).unbind(0)), send[:maxc].view(-1), group=self.group)
                # one launch scatters every peer's rows into the local table
                rc = unpack(stp, nodes_ptr + 8 * off_0[b], tots[b], now, recv_ptr, maxc, offs_ptr + 8 * G * b, G, me, stream)
                if rc:
                    _lib.check(rc, "unpack_gathered")
            rc = step(stp, src_p, dst_p, neg_p, t_p, E, B, b, lam, lid0 + b, flags, G, me, op_p, on_p, ws_p, ws_n, stream)
            if rc:
                _lib.check(rc, "step_batch")
            now = t_last_l[b]
        rp._now_host = now
        rp._params_valid = False
        rp._parameters["now_time"].data.fill_(now)
        if G > 1 and merge_outputs:
            dist.all_reduce(out_pos, group=self.group)          # disjoint rows: the sum is a merge
            if out_neg is not None:
                dist.all_reduce(out_neg, group=self.group)
        return out_pos, out_neg

    def sync_full_state(self):
        """Make every rank's table complete (all-gather of all owned rows), e.g. before a checkpoint."""
        rp, G, me = self.rp, self.G, self.me
        rp._ensure_engine()
        if G == 1:
            return
        lib = _lib.load()
        st = rp._state()
        stream = rp._stream()
        dev = rp._dev()
        row = rp.num_layer * rp.dim
        lam = float(rp.time_decay_weight)
        cnt = [(rp.node_num - r + G - 1) // G for r in range(G)]
        maxc = max(cnt)
        send = torch.zeros((maxc, row), dtype=torch.float32, device=dev)
        ids_me = self.owned_nodes()
        _lib.check(lib.tpnet_pack_rows(C.byref(st), ids_me.data_ptr(), ids_me.numel(), rp._now_host, lam,
                                       send.data_ptr(), stream), "pack_rows")
        recv = exchange_rows(send, maxc, self.group)
        for r in range(G):
            if r == me:
                continue
            ids_r = torch.arange(r, rp.node_num, G, dtype=torch.int64, device=dev)
            _lib.check(lib.tpnet_unpack_rows(C.byref(st), ids_r.data_ptr(), ids_r.numel(), rp._now_host,
                                             recv[r].data_ptr(), stream), "unpack_rows")
        rp._params_valid = False
