"""Callers of the hot path against fixtures recorded from the reference's own classes:
G8 = the index arrays its TPNet backbone + LinkPredictor_v1 + samplers pass to RandomProjectionModule during one
training epoch on a toy graph; G7 = decoder logits.  CPU tier: host logic with a recording stand-in for the module
(no compute); GPU tier: the real module."""
import os

import numpy as np
import pytest
import torch

from tpnet_amd.callers import (LinkPredictor_v1, RandomNegativeSampler, RecentNeighborSampler, encoder_pair_indices,
                               evaluate_with_restore, run_epoch)


class _Recorder:
    """Duck-typed stand-in: records the calls, computes nothing."""
    pair_wise_feature_dim = 64

    def __init__(self):
        self.calls = []

    def reset_random_projections(self):
        self.calls.append(("reset",))

    def get_pair_wise_feature(self, src_node_ids, dst_node_ids):
        self.calls.append(("pair", np.array(src_node_ids), np.array(dst_node_ids)))
        return torch.zeros((len(src_node_ids), 64))

    def update(self, src_node_ids, dst_node_ids, node_interact_times):
        self.calls.append(("update", np.array(src_node_ids), np.array(dst_node_ids), np.array(node_interact_times)))

    def backup_random_projections(self):
        self.calls.append(("backup",))
        return "saved"

    def reload_random_projections(self, saved):
        self.calls.append(("reload", saved))


def _g8(golden_dir):
    return np.load(os.path.join(golden_dir, "g8_call_sequence.npz"))


def test_call_sequence_matches_reference(golden_dir):
    g = _g8(golden_dir)
    src, dst, t = g["src"], g["dst"], g["t"]
    B, K, nb = int(g["B"]), int(g["K"]), int(g["nb"])
    rec = _Recorder()
    sampler = RecentNeighborSampler(src, dst, t, g["eid"])
    negs = RandomNegativeSampler(src, dst)                        # unseeded: global numpy RNG, like the train sampler
    np.random.seed(int(g["np_seed"]))
    run_epoch(rec, sampler, negs, src, dst, t, B, K)
    assert rec.calls[0] == ("reset",)
    calls = rec.calls[1:]
    assert len(calls) == 5 * nb
    for b in range(nb):
        c = calls[5 * b:5 * b + 5]
        assert [x[0] for x in c] == ["pair"] * 4 + ["update"]       # readouts BEFORE the update
        n_b = len(g[f"b{b}_upd_src"])
        assert [len(x[1]) for x in c[:4]] == [4 * n_b * K, 4 * n_b * K, n_b, n_b]
        for k in range(4):
            np.testing.assert_array_equal(c[k][1], g[f"b{b}_pair{k}_u"], err_msg=f"batch {b} call {k} src ids")
            np.testing.assert_array_equal(c[k][2], g[f"b{b}_pair{k}_v"], err_msg=f"batch {b} call {k} dst ids")
        np.testing.assert_array_equal(c[4][1], g[f"b{b}_upd_src"])
        np.testing.assert_array_equal(c[4][2], g[f"b{b}_upd_dst"])
        np.testing.assert_array_equal(c[4][3], g[f"b{b}_upd_t"])
        np.testing.assert_array_equal(c[3][2], g[f"b{b}_neg"])      # negative draw order (src indices first)
    assert len(g[f"b{nb - 1}_upd_src"]) == int(g["E"]) - (nb - 1) * B < B   # ragged tail


def test_recent_sampler_and_index_pattern():
    src = np.array([1, 1, 2, 1, 3]); dst = np.array([4, 5, 4, 4, 5]); t = np.array([1.0, 2.0, 2.0, 3.0, 4.0])
    s = RecentNeighborSampler(src, dst, t)
    ids, eids, ts = s.get_historical_neighbors(np.array([1, 4, 1, 0]), np.array([3.0, 2.0, 10.0, 5.0]), 2)
    np.testing.assert_array_equal(ids, [[4, 5], [0, 1], [5, 4], [0, 0]])     # strictly-before, most recent last
    np.testing.assert_array_equal(eids, [[1, 2], [0, 1], [2, 4], [0, 0]])
    u, v = encoder_pair_indices(np.array([[7, 8], [9, 0]]), np.array([1]), np.array([2]))
    np.testing.assert_array_equal(u, [7, 8, 9, 0, 7, 8, 9, 0])
    np.testing.assert_array_equal(v, [1, 1, 1, 1, 2, 2, 2, 2])


def test_negative_sampler_draw_order_and_seed():
    src = np.array([3, 1, 2, 3]); dst = np.array([9, 8, 9, 7])
    a = RandomNegativeSampler(src, dst, seed=5)
    rs = np.random.RandomState(5)
    si = rs.randint(0, 3, 6); di = rs.randint(0, 3, 6)
    s, d = a.sample(6)
    np.testing.assert_array_equal(s, np.array([1, 2, 3])[si])
    np.testing.assert_array_equal(d, np.array([7, 8, 9])[di])
    a.reset_random_state()
    np.testing.assert_array_equal(a.sample(6)[1], d)


def test_backup_reload_choreography():
    rec = _Recorder()
    assert evaluate_with_restore(rec, lambda: 42) == 42
    assert [c[0] for c in rec.calls] == ["backup", "reload"] and rec.calls[1][1] == "saved"


def test_decoder_state_dict_keys(golden_dir):
    g = np.load(os.path.join(golden_dir, "g7_decoder.npz"))
    from tpnet_amd import RandomProjectionModule
    rp = RandomProjectionModule(node_num=int(g["N"]), edge_num=100, dim_factor=10, num_layer=3, time_decay_weight=1e-6,
                                device="cpu", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False,
                                enforce_dim=int(g["d"]))
    dec = LinkPredictor_v1(input_dim1=20, input_dim2=20, hidden_dim=20, output_dim=1, random_projections=rp,
                           not_encode=False)
    assert sorted(dec.state_dict().keys()) == [str(k) for k in g["dec_keys"]]


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_epoch_state_trajectory_matches_reference(golden_dir):
    """The same epoch on the GPU module: P[1..L] and now_time after every batch as the reference left them."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd import RandomProjectionModule
    g = _g8(golden_dir)
    src, dst, t = g["src"], g["dst"], g["t"]
    B, K, L = int(g["B"]), int(g["K"]), int(g["L"])
    rp = RandomProjectionModule(node_num=int(g["N"]), edge_num=int(g["E"]), dim_factor=10, num_layer=L,
                                time_decay_weight=float(g["lam"]), device="cuda:0", use_matrix=False,
                                beginning_time=np.float64(g["t0"]), not_scale=False, enforce_dim=int(g["d"]),
                                exact=True).to("cuda:0")
    P0 = torch.from_numpy(g["P0"]).to("cuda:0")

    def inject(m):
        m.random_projections[0].data.copy_(P0)

    seen = []

    def on_batch(b, neg, res):
        got = np.stack([rp.random_projections[i].detach().cpu().numpy() for i in range(1, L + 1)])
        want = g[f"b{b}_P"]
        np.testing.assert_allclose(got, want, rtol=5e-6, atol=1e-6 * max(1e-30, float(np.abs(want).max())))
        assert float(rp.now_time.item()) == float(g[f"b{b}_now"])
        feats, outs = res
        assert feats[0].shape == (2 * len(neg), K, 128) and outs[0].shape == (len(neg), 64)
        seen.append(b)

    np.random.seed(int(g["np_seed"]))
    run_epoch(rp, RecentNeighborSampler(src, dst, t, g["eid"]), RandomNegativeSampler(src, dst), src, dst, t, B, K,
              on_batch=on_batch, after_reset=inject)
    assert seen == list(range(int(g["nb"])))


@pytest.mark.gpu
def test_decoder_logits_match_reference(golden_dir):
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd import RandomProjectionModule
    g = np.load(os.path.join(golden_dir, "g7_decoder.npz"))
    L, N, d = int(g["L"]), int(g["N"]), int(g["d"])
    now = float(g["now_time"])
    rp = RandomProjectionModule(node_num=N, edge_num=100, dim_factor=10, num_layer=L, time_decay_weight=float(g["lam"]),
                                device="cuda:0", use_matrix=False, beginning_time=np.float64(now), not_scale=False,
                                enforce_dim=d)
    rp.random_projections[0].data = torch.from_numpy(g["P"][0])
    dec = LinkPredictor_v1(input_dim1=20, input_dim2=20, hidden_dim=20, output_dim=1, random_projections=rp,
                           not_encode=False).to("cuda:0")
    dec.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("dec.")})
    rp.reload_random_projections((torch.tensor(now, dtype=torch.float64, device="cuda:0"),
                                  [torch.from_numpy(g["P"][i]).to("cuda:0") for i in range(1, L + 1)]))
    out = dec(src_node_ids=g["u"], dst_node_ids=g["v"], src_node_embeddings=torch.from_numpy(g["src_emb"]).to("cuda:0"),
              dst_node_embeddings=torch.from_numpy(g["dst_emb"]).to("cuda:0"))
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["logits"], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,N,E,K", [(0, 40, 300, 5), (1, 500, 5000, 20), (2, 7, 60, 3), (3, 30, 0, 4)])
def test_gpu_recent_sampler_matches_host_restatement(seed, N, E, K):
    """f-3: the device sampler against RecentNeighborSampler (itself pinned to the reference's NeighborSampler through
    fixture G8): ties in time, nodes without history, queries before the first edge, ids outside the graph."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd.sampler import GpuRecentNeighborSampler
    rng = np.random.RandomState(seed)
    src = rng.randint(1, N, E).astype(np.int64)
    dst = rng.randint(1, N, E).astype(np.int64)
    t = np.sort(np.round(rng.uniform(0, 50.0, E), 0 if seed != 1 else 3))       # many equal timestamps
    if E:
        dst[::11] = src[::11]                                                    # self loops
    gpu = GpuRecentNeighborSampler(src, dst, t, num_nodes=N + 3)
    host = RecentNeighborSampler(src, dst, t) if E else None
    q_nodes = rng.randint(0, N + 3, 400).astype(np.int64)
    q_times = np.concatenate([rng.uniform(-1.0, 60.0, 300), rng.choice(t, 100) if E else rng.uniform(0, 1, 100)])
    got = gpu.get_historical_neighbors(q_nodes, q_times, K)
    if host is None:
        assert all(np.all(g == 0) for g in got)
        return
    want = host.get_historical_neighbors(q_nodes, q_times, K)
    for g, w, name in zip(got, want, ("neighbour ids", "edge ids", "times")):
        np.testing.assert_array_equal(g, w, err_msg=name)


@pytest.mark.gpu
def test_gpu_sampler_reproduces_reference_call_sequence(golden_dir):
    """The G8 epoch again, with the device sampler in the loop instead of the host one."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd.sampler import GpuRecentNeighborSampler
    g = _g8(golden_dir)
    src, dst, t = g["src"], g["dst"], g["t"]
    rec = _Recorder()
    np.random.seed(int(g["np_seed"]))
    run_epoch(rec, GpuRecentNeighborSampler(src, dst, t, g["eid"]), RandomNegativeSampler(src, dst), src, dst, t,
              int(g["B"]), int(g["K"]))
    calls = rec.calls[1:]
    for b in range(int(g["nb"])):
        for k in range(4):
            np.testing.assert_array_equal(calls[5 * b + k][1], g[f"b{b}_pair{k}_u"])
            np.testing.assert_array_equal(calls[5 * b + k][2], g[f"b{b}_pair{k}_v"])


@pytest.mark.gpu
def test_device_id_path_equals_host_id_path(golden_dir):
    """The on-device caller path (device sampler -> device ids -> shared readout) produces the same encoder features
    as the host-index path the reference's call sequence was pinned with."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd import RandomProjectionModule
    from tpnet_amd.sampler import GpuRecentNeighborSampler
    from tpnet_amd.callers import link_prediction_batch
    g = _g8(golden_dir)
    src, dst, t = g["src"], g["dst"], g["t"]
    B, K, L = int(g["B"]), int(g["K"]), int(g["L"])
    mk = lambda: RandomProjectionModule(node_num=int(g["N"]), edge_num=int(g["E"]), dim_factor=10, num_layer=L,
                                        time_decay_weight=float(g["lam"]), device="cuda:0", use_matrix=False,
                                        beginning_time=np.float64(g["t0"]), not_scale=False,
                                        enforce_dim=int(g["d"])).to("cuda:0")
    a, b = mk(), mk()
    b.load_state_dict(a.state_dict())
    host_s, dev_s = RecentNeighborSampler(src, dst, t, g["eid"]), GpuRecentNeighborSampler(src, dst, t, g["eid"])
    for k in range(int(g["nb"])):
        s = slice(k * B, min((k + 1) * B, len(src)))
        neg = g[f"b{k}_neg"]
        fa, oa = link_prediction_batch(a, host_s, src[s], dst[s], neg, t[s], K)
        fb, ob = link_prediction_batch(b, dev_s, src[s], dst[s], neg, t[s], K)
        for x, y in zip(fa + oa, fb + ob):
            # (the two paths take different kernels of the fp32 class for self.mlp: fmaf on the vector ALUs for short host lists,
            # split-bf16 products on the matrix cores behind the device readout -- ~1e-5 of the output scale apart, contract 1e-4)
            x, y = x.detach().cpu().numpy(), y.detach().cpu().numpy()
            np.testing.assert_allclose(x, y, rtol=1e-4, atol=3e-5 * float(np.abs(x).max()))
