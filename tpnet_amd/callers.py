"""Callers either side of the hot path: what the reference's training / evaluation
loop does around `RandomProjectionModule`, restated so that the drop-in can be driven -- and pinned by fixtures G7 /
G8 -- without the reference's files (they never travel to the GPU box).

* `LinkPredictor_v1`        -- the decoder that consumes the pairwise feature (models/modules.py:73-117).
* `RecentNeighborSampler`   -- the 'recent' strategy of NeighborSampler (utils/utils.py:82-224), host side.
* `RandomNegativeSampler`   -- NegativeEdgeSampler.random_sample (utils/utils.py:388-400): draw order matters.
* `encoder_pair_indices`    -- the 4*B*K index pattern of the encoder's readout (models/TPNet.py:206-217,311-316).
* `link_prediction_batch` / `run_epoch` -- the per-batch order of train_link_prediction.py:253-373 (negatives, two
  encoder readouts, two decoder readouts, THEN update) and the epoch-level reset (:246-248).
The dense encoder (MLP-Mixer) itself is out of scope (stock dense layers); `encoder` / `decoder`
are plug-in callables here.
"""
from typing import Callable, Optional

import numpy as np
import torch
import torch.nn as nn


class LinkPredictor_v1(nn.Module):
    """concat[src_emb, dst_emb, pair feature] -> fc1 -> ReLU -> fc2 (models/modules.py:75-117); same constructor and
    forward keywords as the reference, same state-dict keys (fc1.*, fc2.*, random_projections.*)."""

    def __init__(self, input_dim1: int, input_dim2: int, hidden_dim: int, output_dim: int, random_projections,
                 not_encode: bool):
        super().__init__()
        self.random_projections = random_projections
        self.not_encode = not_encode
        self.random_feature_dim = 0 if random_projections is None else random_projections.pair_wise_feature_dim
        self.fc1 = nn.Linear(input_dim1 + input_dim2 + self.random_feature_dim, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, output_dim)
        self.act = nn.ReLU()
        self.fused = False      # opt-in: both layers in one bf16 matrix-core kernel (tpnet_amd/fused_decoder.py)

    def forward(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray, src_node_embeddings: torch.Tensor,
                dst_node_embeddings: torch.Tensor):
        if self.not_encode:                                        # modules.py:106-108
            src_node_embeddings = torch.zeros_like(src_node_embeddings)
            dst_node_embeddings = torch.zeros_like(dst_node_embeddings)
        feat = None
        if self.random_projections is not None:                    # modules.py:112-114
            feat = self.random_projections.get_pair_wise_feature(src_node_ids=src_node_ids, dst_node_ids=dst_node_ids)
        if self.fused and src_node_embeddings.is_cuda:
            from . import fused_decoder as fd
            D, F = src_node_embeddings.shape[1], (0 if feat is None else feat.shape[1])
            if dst_node_embeddings.shape[1] == D and fd.supported(self.fc1, self.fc2, D, F):
                return fd.fused_decoder(self, self.fc1, self.fc2, src_node_embeddings, dst_node_embeddings, feat,
                                        self.not_encode)
        parts = [src_node_embeddings, dst_node_embeddings] + ([feat] if feat is not None else [])
        return self.fc2(self.act(self.fc1(torch.cat(parts, dim=1))))


class RecentNeighborSampler:
    """Most-recent-K historical neighbours, strictly before the query time, left-padded with id 0
    (utils/utils.py:140-152 searchsorted side='left'; :205-213 'recent'; adjacency built as get_neighbor_sampler
    does, :293-312: undirected, per node sorted by time, stable)."""

    def __init__(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray, node_interact_times: np.ndarray,
                 edge_ids: Optional[np.ndarray] = None):
        src = np.asarray(src_node_ids, dtype=np.int64)
        dst = np.asarray(dst_node_ids, dtype=np.int64)
        t = np.asarray(node_interact_times, dtype=np.float64)
        eid = np.arange(1, len(src) + 1, dtype=np.int64) if edge_ids is None else np.asarray(edge_ids, dtype=np.int64)
        n = int(max(src.max(), dst.max())) + 1 if len(src) else 1
        # interleave (src->dst, dst->src) per edge to keep the reference's append order, then a stable sort by
        # (node, time) gives every node's list sorted by time with ties in append order
        node = np.stack([src, dst], axis=1).reshape(-1)
        nbr = np.stack([dst, src], axis=1).reshape(-1)
        tt = np.repeat(t, 2)
        ee = np.repeat(eid, 2)
        order = np.lexsort((np.arange(len(node)), tt, node))
        self._nbr, self._t, self._e = nbr[order], tt[order], ee[order]
        self._start = np.searchsorted(node[order], np.arange(n + 1))

    def get_historical_neighbors(self, node_ids: np.ndarray, node_interact_times: np.ndarray, num_neighbors: int = 20):
        assert num_neighbors > 0
        node_ids = np.asarray(node_ids, dtype=np.int64)
        times = np.asarray(node_interact_times, dtype=np.float64)
        n = len(node_ids)
        out_ids = np.zeros((n, num_neighbors), dtype=np.int64)
        out_eid = np.zeros((n, num_neighbors), dtype=np.int64)
        out_t = np.zeros((n, num_neighbors), dtype=np.float64)
        for i in range(n):
            nid = node_ids[i]
            if nid + 1 >= len(self._start):
                continue
            lo, hi = self._start[nid], self._start[nid + 1]
            cut = lo + np.searchsorted(self._t[lo:hi], times[i])       # interactions with time < query time
            first = max(lo, cut - num_neighbors)
            k = cut - first
            if k > 0:
                out_ids[i, num_neighbors - k:] = self._nbr[first:cut]
                out_eid[i, num_neighbors - k:] = self._e[first:cut]
                out_t[i, num_neighbors - k:] = self._t[first:cut]
        return out_ids, out_eid, out_t


class RandomNegativeSampler:
    """'random' strategy of NegativeEdgeSampler (utils/utils.py:388-400): src indices are drawn first, then dst
    indices, from the sorted unique id arrays; unseeded samplers use the GLOBAL numpy RNG like the reference."""

    def __init__(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray, seed: Optional[int] = None):
        self.unique_src_node_ids = np.unique(src_node_ids)
        self.unique_dst_node_ids = np.unique(dst_node_ids)
        self.seed = seed
        self.reset_random_state()

    def reset_random_state(self):
        self.random_state = np.random.RandomState(self.seed) if self.seed is not None else None

    def sample(self, size: int):
        rng = np.random if self.random_state is None else self.random_state
        si = rng.randint(0, len(self.unique_src_node_ids), size)
        di = rng.randint(0, len(self.unique_dst_node_ids), size)
        return self.unique_src_node_ids[si], self.unique_dst_node_ids[di]


def encoder_pair_indices(neighbor_node_ids: np.ndarray, src_node_ids: np.ndarray, dst_node_ids: np.ndarray):
    """Index arrays of the encoder's readout for one (src, dst) batch of B edges (models/TPNet.py:206-217, 311-316):
    nodes = [src; dst] (2B), their K neighbours each; every neighbour w is paired with the edge's src AND dst:
    u = tile(neigh.reshape(-1), 2), v = [repeat(tile(src,2), K); repeat(tile(dst,2), K)]  -> 4*B*K pairs."""
    K = neighbor_node_ids.shape[1]
    u = np.tile(neighbor_node_ids.reshape(-1), 2)
    v = np.concatenate([np.repeat(np.tile(src_node_ids, 2), K), np.repeat(np.tile(dst_node_ids, 2), K)])
    return u, v


def link_prediction_batch(rp, neighbor_sampler, src: np.ndarray, dst: np.ndarray, neg_dst: np.ndarray, t: np.ndarray,
                          num_neighbors: int, encoder: Optional[Callable] = None, decoder: Optional[Callable] = None):
    """One batch in the reference's order (train_link_prediction.py:325-373, evaluate_models_utils.py:132-184):
    encoder readout for (src,dst), encoder readout for (src,neg), decoder readout (src,dst), decoder readout
    (src,neg), THEN update(src,dst,t).  `encoder(pair_features[2B,K,2*F], node_ids, times) -> [2B, D]` and
    `decoder(src_ids, dst_ids, src_emb, dst_emb) -> logits` are optional plug-ins; without them the four pairwise
    feature tensors are returned."""
    B = len(src)
    feats = []
    embs = []
    on_device = hasattr(neighbor_sampler, "sample_device") and hasattr(rp, "get_pair_wise_feature_shared")
    if on_device:      # ids stay on the GPU from the sampler to the readout (same pairs, same order as the host path)
        dev = neighbor_sampler.device
        src_t = torch.from_numpy(np.ascontiguousarray(src, dtype=np.int64)).to(dev)
        t2_t = torch.from_numpy(np.tile(np.asarray(t, dtype=np.float64), 2)).to(dev)
    for other in (dst, neg_dst):
        node_ids = np.concatenate([src, other])
        if on_device:
            other_t = torch.from_numpy(np.ascontiguousarray(other, dtype=np.int64)).to(dev)
            neigh, _, _ = neighbor_sampler.sample_device(torch.cat([src_t, other_t]), t2_t, num_neighbors, with_edges=False)
            f = rp.get_pair_wise_feature_shared(neigh.reshape(-1), src_t.repeat(2).repeat_interleave(num_neighbors),
                                                other_t.repeat(2).repeat_interleave(num_neighbors))
        else:
            neigh, _, _ = neighbor_sampler.get_historical_neighbors(node_ids=node_ids,
                                                                    node_interact_times=np.tile(t, 2),
                                                                    num_neighbors=num_neighbors)
            u, v = encoder_pair_indices(neigh, src, other)
            f = rp.get_pair_wise_feature(src_node_ids=u, dst_node_ids=v)             # [4BK, F]
        half = 2 * B * num_neighbors
        f = torch.cat([f[:half], f[half:]], dim=1).reshape(2 * B, num_neighbors, -1)  # TPNet.py:318-321
        feats.append(f)
        embs.append(encoder(f, node_ids, np.tile(t, 2)) if encoder is not None else None)
    outs = []
    for other, e in zip((dst, neg_dst), embs):
        if decoder is not None and e is not None:
            outs.append(decoder(src_node_ids=src, dst_node_ids=other, src_node_embeddings=e[:B], dst_node_embeddings=e[B:]))
        else:
            outs.append(rp.get_pair_wise_feature(src_node_ids=src, dst_node_ids=other))
    rp.update(src_node_ids=src, dst_node_ids=dst, node_interact_times=t)             # after both readouts (:372)
    return feats, outs


def run_epoch(rp, neighbor_sampler, negative_sampler, src: np.ndarray, dst: np.ndarray, t: np.ndarray,
              batch_size: int, num_neighbors: int, encoder=None, decoder=None, on_batch: Optional[Callable] = None,
              after_reset: Optional[Callable] = None):
    """reset at epoch start (:246-248), chronological `range(0, E, B)` batches with a ragged tail
    (utils/DataLoader.py:31-45, shuffle=False), negatives drawn per batch (:259)."""
    rp.reset_random_projections()
    if after_reset is not None:
        after_reset(rp)           # e.g. inject a fixed P[0] (reset redraws it from the device RNG, TPNet.py:138-139)
    results = []
    for b0 in range(0, len(src), batch_size):
        s = slice(b0, min(b0 + batch_size, len(src)))
        _, neg = negative_sampler.sample(size=s.stop - s.start)
        res = link_prediction_batch(rp, neighbor_sampler, src[s], dst[s], neg, t[s], num_neighbors, encoder, decoder)
        if on_batch is not None:
            on_batch(b0 // batch_size, neg, res)
        results.append(res)
    return results


def evaluate_with_restore(rp, evaluate: Callable):
    """The backup / evaluate / reload pattern of train_link_prediction.py:403-494: run `evaluate()` (which updates the
    state while it streams the split) and put the pre-evaluation state back."""
    saved = rp.backup_random_projections()
    try:
        return evaluate()
    finally:
        rp.reload_random_projections(saved)
