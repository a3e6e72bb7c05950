"""Generator of the golden vectors in tests/golden/*.npz  (SURVEY.md §8c, G1..G7).

Runs ONLY in the build container, where the read-only reference checkout is mounted at /root/reference:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's own `models.TPNet.RandomProjectionModule` / `models.modules.LinkPredictor_v1`
(CPU device), feeds them seeded inputs and stores INPUTS AND OUTPUTS (data only).  The reference never
travels to the GPU box; the .npz fixtures do.  P[0] is stored in every fixture because the reference draws
it from the device RNG (models/TPNet.py:58,139).
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("TPNET_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
from models.TPNet import RandomProjectionModule  # noqa: E402
from models.modules import LinkPredictor_v1  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def make_rp(N, E, d, L, lam, t0, not_scale=False, use_matrix=False, seed=0):
    torch.manual_seed(seed)
    rp = RandomProjectionModule(node_num=N, edge_num=E, dim_factor=10, num_layer=L, time_decay_weight=lam,
                                device='cpu', use_matrix=use_matrix, beginning_time=np.float64(t0),
                                not_scale=not_scale, enforce_dim=-1 if use_matrix else d)
    return rp


def edge_stream(rng, N, nb, B, t_lo, t_hi, hub=True):
    """nb batches of B edges; ids in 1..N-1 (0 = padding, never in an edge); duplicates inside a batch,
    a hub node, and self-pairs (u,u) in some batches; sorted f64 timestamps."""
    E = nb * B
    src = rng.randint(1, N, E).astype(np.int64)
    dst = rng.randint(1, N, E).astype(np.int64)
    if hub:
        src[rng.rand(E) < 0.25] = 3          # hub: many contributions per batch
        dst[rng.rand(E) < 0.15] = 5
    for b in range(0, nb, 3):                 # self pairs
        dst[b * B + 1] = src[b * B + 1]
    t = np.sort(rng.uniform(t_lo, t_hi, E)).astype(np.float64)
    return src, dst, t


def traj(name, N, d, L, lam, nb, B, seed, t_lo=1.0e6, t_hi=2.678e6, use_matrix=False):
    """G1/G2/G6: update trajectory; P[1..L] + now_time after every batch."""
    rng = np.random.RandomState(seed)
    src, dst, t = edge_stream(rng, N, nb, B, t_lo, t_hi)
    rp = make_rp(N, nb * B, d, L, lam, t[0], use_matrix=use_matrix, seed=seed)
    out = dict(N=N, d=rp.dim, L=L, lam=lam, B=B, t0=t[0], src=src, dst=dst, t=t,
               P0=rp.random_projections[0].detach().numpy().copy(), use_matrix=use_matrix)
    with torch.no_grad():
        for b in range(nb):
            s = slice(b * B, (b + 1) * B)
            rp.update(src[s], dst[s], t[s])
            out[f"P_after_{b}"] = np.stack([rp.random_projections[i].detach().numpy() for i in range(1, L + 1)])
            out[f"now_after_{b}"] = rp.now_time.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, name), **out)
    return rp, (src, dst, t)


def readout(name, N, d, L, lam, nb, B, seed, n_pairs=96):
    """G3/G4: readout pre-MLP (mlp := Identity) for not_scale False/True, and post-MLP with stored weights.
    Pairs include id 0 (padding row) and u == v."""
    rng = np.random.RandomState(seed)
    src, dst, t = edge_stream(rng, N, nb, B, 1.0e6, 2.678e6)
    rp = make_rp(N, nb * B, d, L, lam, t[0], seed=seed)
    with torch.no_grad():
        for b in range(nb):
            s = slice(b * B, (b + 1) * B)
            rp.update(src[s], dst[s], t[s])
    u = rng.randint(0, N, n_pairs).astype(np.int64)
    v = rng.randint(0, N, n_pairs).astype(np.int64)
    u[:4] = 0
    v[2:6] = 0
    v[8:12] = u[8:12]
    u[12:20] = src[:8]; v[12:20] = dst[:8]          # pairs that really interacted
    out = dict(N=N, d=d, L=L, lam=lam, B=B, t0=t[0], src=src, dst=dst, t=t,
               P=np.stack([rp.random_projections[i].detach().numpy() for i in range(L + 1)]),
               now_time=rp.now_time.detach().numpy().copy(), u=u, v=v)
    mlp = rp.mlp
    with torch.no_grad():
        out["feat_mlp_scaled"] = rp.get_pair_wise_feature(u, v).numpy().copy()
        rp.mlp = torch.nn.Identity()
        rp.not_scale = False
        out["gram_scaled"] = rp.get_pair_wise_feature(u, v).numpy().copy()
        rp.not_scale = True
        out["gram_raw"] = rp.get_pair_wise_feature(u, v).numpy().copy()
        rp.mlp = mlp
        out["feat_mlp_raw"] = rp.get_pair_wise_feature(u, v).numpy().copy()
        rows = rp.get_random_projections(u)
        out["rows_u"] = np.stack([r.numpy() for r in rows])
    for k, p in mlp.state_dict().items():
        out["mlp." + k] = p.numpy().copy()
    np.savez_compressed(os.path.join(OUT, name), **out)


def backup_reload(name, N, d, L, lam, B, seed):
    """G5: backup -> updates -> reload round trip; reset zeroing."""
    rng = np.random.RandomState(seed)
    src, dst, t = edge_stream(rng, N, 6, B, 10.0, 5.0e5)
    rp = make_rp(N, 6 * B, d, L, lam, t[0], seed=seed)
    out = dict(N=N, d=d, L=L, lam=lam, B=B, t0=t[0], src=src, dst=dst, t=t,
               P0=rp.random_projections[0].detach().numpy().copy())
    snap = lambda: np.stack([rp.random_projections[i].detach().numpy().copy() for i in range(1, L + 1)])
    with torch.no_grad():
        for b in range(3):
            s = slice(b * B, (b + 1) * B); rp.update(src[s], dst[s], t[s])
        bk = rp.backup_random_projections()
        out["bk_now"] = bk[0].numpy().copy(); out["bk_P"] = np.stack([x.numpy() for x in bk[1]])
        for b in range(3, 6):
            s = slice(b * B, (b + 1) * B); rp.update(src[s], dst[s], t[s])
        out["P_after_6"] = snap(); out["now_after_6"] = rp.now_time.numpy().copy()
        rp.reload_random_projections(bk)
        out["P_after_reload"] = snap(); out["now_after_reload"] = rp.now_time.numpy().copy()
        s = slice(3 * B, 4 * B); rp.update(src[s], dst[s], t[s])       # continue from the restored state
        out["P_after_reload_update"] = snap()
        rp.reset_random_projections()
        out["P_after_reset"] = snap(); out["now_after_reset"] = rp.now_time.numpy().copy()
        out["P0_changed_by_reset"] = np.array(not np.array_equal(out["P0"], rp.random_projections[0].numpy()))
    sd = rp.state_dict()
    out["state_dict_keys"] = np.array(sorted(sd.keys()))
    out["state_dict_dtypes"] = np.array([str(sd[k].dtype) + str(tuple(sd[k].shape)) for k in sorted(sd.keys())])
    np.savez_compressed(os.path.join(OUT, name), **out)


def decoder(name, N, d, L, lam, B, seed):
    """G7: LinkPredictor_v1.forward (models/modules.py:95-117) with fixed weights/embeddings."""
    rng = np.random.RandomState(seed)
    src, dst, t = edge_stream(rng, N, 4, B, 1.0e6, 2.0e6)
    rp = make_rp(N, 4 * B, d, L, lam, t[0], seed=seed)
    torch.manual_seed(seed + 1)
    dec = LinkPredictor_v1(input_dim1=20, input_dim2=20, hidden_dim=20, output_dim=1, random_projections=rp,
                           not_encode=False)
    with torch.no_grad():
        for b in range(4):
            s = slice(b * B, (b + 1) * B); rp.update(src[s], dst[s], t[s])
        u = src[:B]; v = dst[:B]
        se = torch.randn(B, 20); de = torch.randn(B, 20)
        logits = dec(src_node_ids=u, dst_node_ids=v, src_node_embeddings=se, dst_node_embeddings=de)
    out = dict(N=N, d=d, L=L, lam=lam, B=B, t0=t[0], src=src, dst=dst, t=t, u=u, v=v,
               P=np.stack([rp.random_projections[i].detach().numpy() for i in range(L + 1)]),
               now_time=rp.now_time.detach().numpy().copy(),
               src_emb=se.numpy(), dst_emb=de.numpy(), logits=logits.numpy())
    for k, p in dec.state_dict().items():
        out["dec." + k] = p.numpy().copy()
    out["dec_keys"] = np.array(sorted(dec.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, name), **out)


def call_sequence(name, seed=11):
    """G8: the reference's own training-step call sequence on a toy graph (train_link_prediction.py:246-261,
    321-373): TPNet backbone + LinkPredictor_v1 + 'recent' NeighborSampler + random NegativeEdgeSampler, with a
    recorder around the RandomProjectionModule methods.  Stores the stream, the sampler seeds, and per batch the
    index arrays of the four get_pair_wise_feature calls, the update arguments and P / now_time afterwards."""
    from models.TPNet import TPNet
    from utils.utils import get_neighbor_sampler, NegativeEdgeSampler
    from utils.DataLoader import Data
    rng = np.random.RandomState(seed)
    U, I, E, B, K, L, d, lam = 12, 9, 70, 16, 4, 3, 32, 1e-6
    src = rng.randint(1, U + 1, E).astype(np.int64)
    dst = (U + rng.randint(1, I + 1, E)).astype(np.int64)
    t = np.sort(rng.uniform(0.0, 5.0e5, E)).astype(np.float64)
    eid = np.arange(1, E + 1).astype(np.int64)
    N = U + I + 1
    data = Data(src, dst, t, eid, np.zeros(E))
    sampler = get_neighbor_sampler(data, sample_neighbor_strategy='recent', time_scaling_factor=0.0, seed=0)
    neg_sampler = NegativeEdgeSampler(src_node_ids=src, dst_node_ids=dst)            # train sampler: unseeded
    node_feat = np.zeros((N, 16), dtype=np.float32)
    edge_feat = rng.randn(E + 1, 16).astype(np.float32)
    rp = make_rp(N, E, d, L, lam, t[0], seed=seed)
    out = dict(N=N, U=U, I=I, E=E, B=B, K=K, L=L, d=d, lam=lam, t0=t[0], src=src, dst=dst, t=t, eid=eid,
               P0=rp.random_projections[0].detach().numpy().copy(), np_seed=1234)
    calls = []
    orig_pair, orig_upd = rp.get_pair_wise_feature, rp.update

    def rec_pair(src_node_ids, dst_node_ids):
        calls.append(("pair", np.array(src_node_ids).copy(), np.array(dst_node_ids).copy()))
        return orig_pair(src_node_ids=src_node_ids, dst_node_ids=dst_node_ids)

    def rec_upd(src_node_ids, dst_node_ids, node_interact_times):
        calls.append(("update", np.array(src_node_ids).copy(), np.array(dst_node_ids).copy(),
                      np.array(node_interact_times).copy()))
        return orig_upd(src_node_ids=src_node_ids, dst_node_ids=dst_node_ids, node_interact_times=node_interact_times)

    rp.get_pair_wise_feature, rp.update = rec_pair, rec_upd
    torch.manual_seed(seed)
    backbone = TPNet(node_raw_features=node_feat, edge_raw_features=edge_feat, neighbor_sampler=sampler,
                     time_feat_dim=8, random_projections=rp, num_neighbors=K, num_layers=1, dropout=0.0, device='cpu')
    decoder = LinkPredictor_v1(input_dim1=16, input_dim2=16, hidden_dim=16, output_dim=1, random_projections=rp,
                               not_encode=False)
    np.random.seed(int(out["np_seed"]))             # the unseeded negative sampler draws from the global numpy RNG
    rp.reset_random_projections()                   # epoch start (train_link_prediction.py:246-248)
    rp.random_projections[0].data = torch.from_numpy(out["P0"]).clone()   # re-inject P[0] (reset redraws it)
    nb = 0
    with torch.no_grad():
        for b0 in range(0, E, B):                   # shuffle=False loader: chronological, ragged tail
            s = slice(b0, min(b0 + B, E))
            bs, bd, bt = src[s], dst[s], t[s]
            _, bneg = neg_sampler.sample(size=len(bs))
            n0 = len(calls)
            es, ed = backbone.compute_src_dst_node_temporal_embeddings(src_node_ids=bs, dst_node_ids=bd, node_interact_times=bt)
            ns, nd = backbone.compute_src_dst_node_temporal_embeddings(src_node_ids=bs, dst_node_ids=bneg, node_interact_times=bt)
            decoder(src_node_ids=bs, dst_node_ids=bd, src_node_embeddings=es, dst_node_embeddings=ed)
            decoder(src_node_ids=bs, dst_node_ids=bneg, src_node_embeddings=ns, dst_node_embeddings=nd)
            rp.update(src_node_ids=bs, dst_node_ids=bd, node_interact_times=bt)
            batch_calls = calls[n0:]
            assert [c[0] for c in batch_calls] == ["pair"] * 4 + ["update"]
            for k, c in enumerate(batch_calls[:4]):
                out[f"b{nb}_pair{k}_u"] = c[1]; out[f"b{nb}_pair{k}_v"] = c[2]
            out[f"b{nb}_neg"] = bneg
            out[f"b{nb}_upd_src"], out[f"b{nb}_upd_dst"], out[f"b{nb}_upd_t"] = batch_calls[4][1:]
            out[f"b{nb}_P"] = np.stack([rp.random_projections[i].detach().numpy() for i in range(1, L + 1)])
            out[f"b{nb}_now"] = rp.now_time.detach().numpy().copy()
            nb += 1
    out["nb"] = nb
    np.savez_compressed(os.path.join(OUT, name), **out)


if __name__ == "__main__":
    torch.set_num_threads(1)
    traj("g1_update_d16_L3.npz", N=64, d=16, L=3, lam=1e-6, nb=8, B=16, seed=1)
    traj("g2_update_d120_L2.npz", N=80, d=120, L=2, lam=1e-6, nb=6, B=24, seed=2)
    traj("g2b_update_d64_L3_fastdecay.npz", N=48, d=64, L=3, lam=2e-6, nb=10, B=32, seed=7, t_lo=0.0, t_hi=1.0e6)
    readout("g3g4_readout_d16_L3.npz", N=64, d=16, L=3, lam=1e-6, nb=6, B=16, seed=3)
    readout("g3g4_readout_d128_L3.npz", N=200, d=128, L=3, lam=1e-6, nb=8, B=50, seed=4)
    readout("g3g4_readout_d140_L1.npz", N=90, d=140, L=1, lam=1e-7, nb=4, B=30, seed=5)
    backup_reload("g5_backup_reload.npz", N=50, d=32, L=3, lam=1e-6, B=20, seed=6)
    traj("g6_use_matrix_N30.npz", N=30, d=-1, L=3, lam=1e-6, nb=5, B=10, seed=8, use_matrix=True)
    decoder("g7_decoder.npz", N=60, d=32, L=3, lam=1e-6, B=12, seed=9)
    call_sequence("g8_call_sequence.npz")
    print("golden vectors written to", OUT)
