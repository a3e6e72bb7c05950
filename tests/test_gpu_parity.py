"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI by the drop-in module,
against (1) the golden vectors produced by the reference itself, (2) the numpy oracle on seeded random inputs,
(3) size-independent properties at BASELINE.json's full C2 size.

Tolerances (north_star: 1e-4 relative fp32):
  state P[i]      rtol 1e-4, atol 1e-6 * max|P[i]|         (exact mode: rtol 2e-6 -- only expf differs)
  raw Gram        |delta| <= 1e-4 * ||R_a|| * ||R_b||       (entries are near-cancelling sums of d products)
  log(relu(G)+1)  rtol 1e-4, atol 1e-5
"""
import os

import numpy as np
import pytest
import torch

from oracle import tpnet_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: run on the MI355X box (python -m pytest -m gpu)")


def _module(N, d, L, lam, t0, P0=None, not_scale=False, use_matrix=False, exact=False, E=1000):
    from tpnet_amd import RandomProjectionModule
    rp = RandomProjectionModule(node_num=N, edge_num=E, dim_factor=10, num_layer=L, time_decay_weight=lam, device=DEV,
                                use_matrix=use_matrix, beginning_time=np.float64(t0), not_scale=not_scale,
                                enforce_dim=-1 if use_matrix else d, exact=exact)
    if P0 is not None:
        rp.random_projections[0].data = torch.from_numpy(np.ascontiguousarray(P0))
    return rp.to(DEV)


def _layers(rp):
    return np.stack([rp.random_projections[i].detach().cpu().numpy() for i in range(1, rp.num_layer + 1)])


def _assert_state(got, want, rtol, what=""):
    for i in range(want.shape[0]):
        scale = max(1e-30, float(np.abs(want[i]).max()))
        np.testing.assert_allclose(got[i], want[i], rtol=rtol, atol=1e-6 * scale, err_msg=f"{what} layer {i + 1}")


def _gram_bound(P, u, v, L, rel):
    R = np.stack([P[i][u] for i in range(L + 1)] + [P[i][v] for i in range(L + 1)], axis=1).astype(np.float64)
    nrm = np.linalg.norm(R, axis=2)
    return (rel * nrm[:, :, None] * nrm[:, None, :]).reshape(len(u), -1) + 1e-30


def _assert_features(got, st, u, v, what=""):
    """log(relu(G)+1) features against the oracle.  rtol 1e-4 (north_star) + an absolute term for entries whose raw
    Gram is a near-cancelling sum of d signed products: the f32 dot-product error is ~1e-6 * ||R_a|| * ||R_b||
    whatever the summation order, and d/dx log(1+x) = 1/(1+x) carries it into the feature."""
    L = st.L
    raw = O.pair_gram(st, u, v, not_scale=True)
    want = O.pair_gram(st, u, v)
    atol = 1e-5 + _gram_bound(st.P, u, v, L, 1e-6) / (1.0 + np.maximum(raw, 0))
    bad = np.abs(got - want) > 1e-4 * np.abs(want) + atol
    assert not bad.any(), f"{what}: {int(bad.sum())} features off, worst |delta| {np.abs(got - want)[bad].max():.3e}"
    return float(np.abs(got - want).max())


# ---------------------------------------------------------------------------------------------------------
# (1) golden vectors from the reference
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("name", ["g1_update_d16_L3.npz", "g2_update_d120_L2.npz",
                                  "g2b_update_d64_L3_fastdecay.npz", "g6_use_matrix_N30.npz"])
def test_golden_update_trajectory(golden_dir, name, exact):
    _need_gpu()
    g = np.load(os.path.join(golden_dir, name))
    L, B, N = int(g["L"]), int(g["B"]), int(g["N"])
    rp = _module(N, int(g["d"]), L, float(g["lam"]), float(g["t0"]), P0=g["P0"], use_matrix=bool(g["use_matrix"]),
                 exact=exact)
    nb = len(g["src"]) // B
    for b in range(nb):
        s = slice(b * B, (b + 1) * B)
        rp.update(src_node_ids=g["src"][s], dst_node_ids=g["dst"][s], node_interact_times=g["t"][s])
        _assert_state(_layers(rp), g[f"P_after_{b}"], 5e-6 if exact else 1e-4, f"{name} batch {b}")
        assert float(rp.now_time.item()) == float(g[f"now_after_{b}"])


@pytest.mark.parametrize("name", ["g3g4_readout_d16_L3.npz", "g3g4_readout_d128_L3.npz",
                                  "g3g4_readout_d140_L1.npz"])
def test_golden_readout(golden_dir, name):
    _need_gpu()
    g = np.load(os.path.join(golden_dir, name))
    L, N, d = int(g["L"]), int(g["N"]), int(g["d"])
    now = float(g["now_time"])
    rp = _module(N, d, L, float(g["lam"]), now, P0=g["P"][0])
    # load the reference's state the way its callers do: reload_random_projections (TPNet.py:149-157)
    rp.reload_random_projections((torch.tensor(now, dtype=torch.float64, device=DEV),
                                  [torch.from_numpy(g["P"][i]).to(DEV) for i in range(1, L + 1)]))
    u, v = g["u"], g["v"]
    got = rp.pair_gram(u, v).cpu().numpy()
    np.testing.assert_allclose(got, g["gram_scaled"], rtol=1e-4, atol=1e-5)
    rp.not_scale = True
    raw = rp.pair_gram(u, v).cpu().numpy()
    assert np.all(np.abs(raw - g["gram_raw"]) <= _gram_bound(g["P"], u, v, L, 1e-5))
    rp.not_scale = False
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("mlp.")}
    rp.mlp.load_state_dict(sd)
    feat = rp.get_pair_wise_feature(src_node_ids=u, dst_node_ids=v)
    np.testing.assert_allclose(feat.detach().cpu().numpy(), g["feat_mlp_scaled"], rtol=1e-4, atol=1e-5)
    rows = rp.get_random_projections(u)
    assert len(rows) == L + 1
    np.testing.assert_array_equal(np.stack([r.cpu().numpy() for r in rows]), g["rows_u"])
    # gradient reaches the mlp only
    feat.sum().backward()
    assert rp.mlp[0].weight.grad is not None and all(p.grad is None for p in rp.random_projections)


@pytest.mark.parametrize("exact", [True, False])
def test_golden_backup_reload_reset(golden_dir, exact):
    _need_gpu()
    g = np.load(os.path.join(golden_dir, "g5_backup_reload.npz"))
    L, B, N = int(g["L"]), int(g["B"]), int(g["N"])
    rp = _module(N, int(g["d"]), L, float(g["lam"]), float(g["t0"]), P0=g["P0"], exact=exact)
    upd = lambda b: rp.update(g["src"][b * B:(b + 1) * B], g["dst"][b * B:(b + 1) * B], g["t"][b * B:(b + 1) * B])
    rtol = 5e-6 if exact else 1e-4
    for b in range(3):
        upd(b)
    bk = rp.backup_random_projections()
    assert float(bk[0].item()) == float(g["bk_now"]) and len(bk[1]) == L
    _assert_state(np.stack([x.cpu().numpy() for x in bk[1]]), g["bk_P"], rtol, "backup")
    for b in range(3, 6):
        upd(b)
    _assert_state(_layers(rp), g["P_after_6"], rtol, "after 6")
    rp.reload_random_projections(bk)
    _assert_state(_layers(rp), g["P_after_reload"], rtol, "after reload")
    assert float(rp.now_time.item()) == float(g["now_after_reload"])
    upd(3)
    _assert_state(_layers(rp), g["P_after_reload_update"], rtol, "after reload+update")
    p0_before = rp.random_projections[0].detach().clone()
    rp.reset_random_projections()
    assert float(rp.now_time.item()) == float(g["now_after_reset"])
    assert np.all(_layers(rp) == 0)
    assert not torch.equal(p0_before, rp.random_projections[0])       # P[0] is redrawn (TPNet.py:138-139)
    # and the module keeps working after a reset
    upd(0)
    assert np.abs(_layers(rp)[0]).max() > 0
    # state-dict round trip through a fresh module (checkpoint save / load, utils/EarlyStopping.py:64-87)
    sd = {k: v.clone() for k, v in rp.state_dict().items()}
    rp2 = _module(N, int(g["d"]), L, float(g["lam"]), 0.0, exact=exact)
    rp2.load_state_dict(sd)
    upd(1)
    rp2.update(g["src"][B:2 * B], g["dst"][B:2 * B], g["t"][B:2 * B])
    # (the ragged tail runs on 256-thread workgroups in the fused stream and on 512-thread ones here: the hub's partial sums
    # associate differently)
    _assert_state(_layers(rp2), _layers(rp), 1e-5, "update-only vs fused")


# ---------------------------------------------------------------------------------------------------------
# (2) seeded random inputs against the oracle: geometries, layer counts, hubs, duplicates, ragged tail
# ---------------------------------------------------------------------------------------------------------
def _random_stream(rng, N, E, span, hub_frac=0.2):
    src = rng.randint(1, N, E).astype(np.int64)
    dst = rng.randint(1, N, E).astype(np.int64)
    src[rng.rand(E) < hub_frac] = 1 + rng.randint(0, 3)          # hubs: long contribution lists
    dst[rng.rand(E) < hub_frac / 2] = 7
    dst[::17] = src[::17]                                         # self pairs
    t = np.sort(rng.uniform(1.0e6, 1.0e6 + span, E))
    neg = rng.randint(0, N, E).astype(np.int64)                   # includes the padding row 0
    return src, dst, neg, t


@pytest.mark.parametrize("d,L,N,B", [(64, 3, 300, 50), (128, 3, 500, 128), (256, 3, 200, 96), (512, 2, 150, 64),
                                     (120, 3, 260, 40), (140, 1, 90, 33), (30, 4, 64, 20), (1024, 1, 80, 16),
                                     (128, 4, 333, 77), (64, 2, 1000, 500), (16, 3, 400, 200), (32, 3, 300, 150),
                                     (24, 2, 200, 64), (8, 4, 128, 100), (512, 4, 100, 40), (256, 4, 120, 500),
                                     (64, 4, 150, 450), (128, 1, 90, 30)])
@pytest.mark.parametrize("mode", ["exact", "auto", "batch", "windowed"])
def test_stream_matches_oracle(d, L, N, B, mode):
    """run_stream (ragged last batch) == oracle loop readout, readout, update -- in the exact mode and on every schedule
    of the default mode: "auto" (what the shipped default picks for a stream this short: the per-batch kernels), "batch",
    "windowed"."""
    _need_gpu()
    exact = mode == "exact"
    rng = np.random.RandomState(d * 7 + L)
    E = 4 * B + B // 3 + 1
    lam = 2e-6
    src, dst, neg, t = _random_stream(rng, N, E, 4.0e5)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    rp = _module(N, d, L, lam, t[0], P0=P0, exact=exact)
    st = O.OracleState(P0, L, lam, t[0])
    dev = lambda x: torch.from_numpy(x).to(DEV)
    fp, fn = rp.run_stream(dev(src), dev(dst), dev(neg), dev(t), B, schedule=None if exact else mode)
    fp = fp.cpu().numpy(); fn = fn.cpu().numpy()
    for b in range(0, E, B):
        s = slice(b, min(b + B, E))
        _assert_features(fp[s], st, src[s], dst[s], f"pos batch {b // B}")
        _assert_features(fn[s], st, src[s], neg[s], f"neg batch {b // B}")
        O.update(st, src[s], dst[s], t[s])
    _assert_state(_layers(rp), np.stack(st.P[1:]), 5e-6 if exact else 1e-4, "final state")
    assert float(rp.now_time.item()) == float(t[-1])
    rp.check_device_errors()


def test_exact_mode_is_bit_exact_when_times_are_batch_constant():
    """With one timestamp per batch every time weight is exp(0) = 1 and the decay factor comes from the host in
    f64 (as in the reference), so the exact mode must reproduce the oracle's f32 sums BIT FOR BIT: same products,
    same index-ordered additions (src side then dst side), no fused multiply-add."""
    _need_gpu()
    rng = np.random.RandomState(11)
    N, d, L, B, nb = 120, 128, 3, 64, 6
    src, dst, _, _ = _random_stream(rng, N, B * nb, 1.0)
    t = np.repeat(1000.0 * np.arange(1, nb + 1), B)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    rp = _module(N, d, L, 1e-4, 0.0, P0=P0, exact=True)
    st = O.OracleState(P0, L, 1e-4, 0.0)
    for b in range(nb):
        s = slice(b * B, (b + 1) * B)
        rp.update(src[s], dst[s], t[s])
        O.update(st, src[s], dst[s], t[s])
        np.testing.assert_array_equal(_layers(rp), np.stack(st.P[1:]))


@pytest.mark.parametrize("B", [1, 2, 33, 255, 256, 257, 511, 512, 513, 1000, 1023, 1024, 1025])
@pytest.mark.parametrize("law", ["one-node", "two-nodes", "groups-of-16-17", "distinct", "mixed"])
def test_single_batch_planner_group_shapes_bit_for_bit(B, law):
    """The per-batch planner groups a batch's contributions by target without a sort (k_plan_one_h, B <= 1 024: hash table,
    rank by contribution number -- small groups by counting, hubs by a bitmap; B = 1 025: the sorting planner).  Whatever
    the shape of the groups -- every contribution on ONE target, groups on both sides of the small / hub limit, all distinct
    -- each target's sum must keep the reference's order: with batch-constant times the exact mode is bit for bit the oracle
    (order-sensitive f32 sums), and the default mode (hubs summed by workgroups) is the oracle within the stream tolerance."""
    _need_gpu()
    rng = np.random.RandomState(B * 5 + len(law))
    N, d, L, nb = 2100, 64, 2, 2
    E = B * nb
    if law == "one-node":
        src = np.full(E, 5, np.int64); dst = np.full(E, 5, np.int64)
        dst[1::2] = rng.randint(1, N, len(dst[1::2]))               # (half of the dst side elsewhere: the hub keeps ~1.5 B terms)
    elif law == "two-nodes":
        src = np.where(rng.rand(E) < 0.5, 3, 4).astype(np.int64); dst = np.where(rng.rand(E) < 0.5, 3, 4).astype(np.int64)
    elif law == "groups-of-16-17":
        src = (1 + (np.arange(E) % B) // 16).astype(np.int64)       # 16 src-side terms per node ...
        dst = (1 + N // 2 + (np.arange(E) % B) // 17).astype(np.int64)   # ... 17 dst-side terms per node of another range
        dst[::40] = src[::40]                                       # (and a seventeenth / eighteenth for some)
    elif law == "distinct":
        src = (1 + np.arange(E) % B).astype(np.int64); dst = (1 + B + np.arange(E) % B).astype(np.int64)
    else:
        src, dst, _, _ = _random_stream(rng, N, E, 1.0, hub_frac=0.5)
    t = np.repeat(1000.0 * np.arange(1, nb + 1), B)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    for exact in (True, False):
        rp = _module(N, d, L, 1e-4, 0.0, P0=P0, exact=exact)
        st = O.OracleState(P0, L, 1e-4, 0.0)
        for b in range(nb):
            s = slice(b * B, (b + 1) * B)
            rp.update(src[s], dst[s], t[s])
            O.update(st, src[s], dst[s], t[s])
            if exact:
                np.testing.assert_array_equal(_layers(rp), np.stack(st.P[1:]), err_msg=f"batch {b}")
            else:
                _assert_state(_layers(rp), np.stack(st.P[1:]), 1e-4, f"batch {b}")
        rp.check_device_errors()


def test_module_calls_interleave_like_the_training_loop():
    """Caller order of train_link_prediction.py:325-373: readouts (big and small), then update, per batch; plus
    encoder-style index patterns (np.tile / np.repeat, padding id 0: models/TPNet.py:313-316)."""
    _need_gpu()
    rng = np.random.RandomState(2)
    N, d, L, B, K = 400, 128, 3, 40, 5
    lam = 1e-6
    src, dst, neg, t = _random_stream(rng, N, 6 * B, 3.0e5)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    rp = _module(N, d, L, lam, t[0], P0=P0)
    st = O.OracleState(P0, L, lam, t[0])
    for b in range(6):
        s = slice(b * B, (b + 1) * B)
        neigh = rng.randint(0, N, (2 * B, K))
        neigh[rng.rand(2 * B, K) < 0.3] = 0
        big_u = np.tile(neigh.reshape(-1), 2)
        big_v = np.concatenate([np.repeat(src[s], K), np.repeat(dst[s], K), np.repeat(src[s], K), np.repeat(dst[s], K)])
        for (u, v) in ((big_u, big_v), (src[s], dst[s]), (src[s], neg[s])):
            _assert_features(rp.pair_gram(u, v).cpu().numpy(), st, u, v)
        rp.update(src[s], dst[s], t[s])
        O.update(st, src[s], dst[s], t[s])
    _assert_state(_layers(rp), np.stack(st.P[1:]), 1e-4)


def test_edge_cases_and_errors():
    _need_gpu()
    N, d, L = 50, 64, 3
    rp = _module(N, d, L, 1e-6, 0.0)
    ids = np.array([1, 2, 3])
    with pytest.raises(IndexError):
        rp.update(np.array([], dtype=np.int64), np.array([], dtype=np.int64), np.array([]))
    with pytest.raises(IndexError):
        rp.update(np.array([N]), np.array([1]), np.array([1.0]))
    with pytest.raises(IndexError):
        rp.get_pair_wise_feature(np.array([1]), np.array([-N - 1]))
    with pytest.raises(ValueError):
        rp.update(ids, ids[:2], np.array([1.0, 2.0, 3.0]))
    assert rp.pair_gram(np.array([], dtype=np.int64), np.array([], dtype=np.int64)).shape == (0, 64)
    # a single edge, a self pair, the padding row
    rp.update(np.array([4]), np.array([4]), np.array([10.0]))
    P0 = rp.random_projections[0].detach().cpu().numpy()
    np.testing.assert_allclose(rp.random_projections[1][4].cpu().numpy(), 2 * P0[4], rtol=1e-6)
    g = rp.pair_gram(np.array([0]), np.array([0])).cpu().numpy().reshape(8, 8)
    assert np.isfinite(g).all() and g[0, 0] > 0 and g[1, 1] == 0
    # device-side id check of the stream path: bad ids are skipped and reported, never dereferenced
    src = torch.tensor([1, 2, N + 5, 3], dtype=torch.int64, device=DEV)
    dst = torch.tensor([2, 3, 4, -1], dtype=torch.int64, device=DEV)
    t = torch.tensor([11.0, 12.0, 13.0, 14.0], dtype=torch.float64, device=DEV)
    rp.run_stream(src, dst, None, t, 2, want_neg=False)
    with pytest.raises(IndexError):
        rp.check_device_errors()
    rp.check_device_errors()     # cleared


# ---------------------------------------------------------------------------------------------------------
# (3) BASELINE.json's config C2 at full size: oracle on the whole stream + size-independent properties
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c2():
    from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
    c = CONFIGS["C2"]
    src, dst, t, N = synthetic_stream(c["U"], c["I"], c["E"], c["span"], 0)
    neg = synthetic_negatives(c["U"], N, c["E"], c["B"], 1)
    torch.manual_seed(0)
    P0 = torch.normal(0, 1 / np.sqrt(c["d"]), (N, c["d"])).numpy()
    return dict(c=c, src=src, dst=dst, neg=neg, t=t, N=N, P0=P0)


def _run_c2(c2, P0, exact=False, E=None):
    c = c2["c"]
    E = E or c["E"]
    rp = _module(c2["N"], c["d"], 3, c["lam"], 0.0, P0=P0, exact=exact, E=c["E"])
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x[:E])).to(DEV)
    fp, fn = rp.run_stream(dev(c2["src"]), dev(c2["dst"]), dev(c2["neg"]), dev(c2["t"]), c["B"],
                           t_end=float(c2["t"][E - 1]))
    return rp, fp, fn


def test_c2_full_stream_against_oracle(c2):
    """Wikipedia-shaped stream (157 474 edges, d=128, B=1000, 158 batches incl. a ragged tail): every readout of
    every batch and the final state, lazy-decay mode, against the oracle's eager arithmetic."""
    _need_gpu()
    c = c2["c"]
    rp, fp, fn = _run_c2(c2, c2["P0"])
    fp = fp.cpu().numpy(); fn = fn.cpu().numpy()
    st = O.OracleState(c2["P0"], 3, c["lam"], 0.0)
    E, B = c["E"], c["B"]
    worst = 0.0
    for b in range(0, E, B):
        s = slice(b, min(b + B, E))
        for got, v in ((fp[s], c2["dst"][s]), (fn[s], c2["neg"][s])):
            worst = max(worst, _assert_features(got, st, c2["src"][s], v, f"C2 batch {b // B}"))
        O.update(st, c2["src"][s], c2["dst"][s], c2["t"][s])
    _assert_state(_layers(rp), np.stack(st.P[1:]), 1e-4, "C2 final state")
    rp.check_device_errors()
    print(f"C2 full stream: worst |feature delta| = {worst:.3e}")


def test_c2_properties(c2):
    """Size-independent properties on the full stream: (a) run-to-run determinism, bit for bit; (b) linearity in
    P[0]: scaling P[0] by 2 scales every layer by exactly 2 (power-of-two scaling commutes with f32 rounding) and
    shifts nothing else; (c) chunked == unchunked: the stream in two run_stream calls equals one call."""
    _need_gpu()
    c = c2["c"]
    E = 40 * c["B"] + 123
    rp1, fp1, fn1 = _run_c2(c2, c2["P0"], E=E)
    rp2, fp2, fn2 = _run_c2(c2, c2["P0"], E=E)
    assert torch.equal(fp1, fp2) and torch.equal(fn1, fn2)
    np.testing.assert_array_equal(_layers(rp1), _layers(rp2))
    rp3, _, _ = _run_c2(c2, 2.0 * c2["P0"], E=E)
    np.testing.assert_array_equal(_layers(rp3), 2.0 * _layers(rp1))
    # two calls
    rp4 = _module(c2["N"], c["d"], 3, c["lam"], 0.0, P0=c2["P0"], E=c["E"])
    dev = lambda x, a, b: torch.from_numpy(np.ascontiguousarray(x[a:b])).to(DEV)
    cut = 17 * c["B"]
    outs = []
    for a, b in ((0, cut), (cut, E)):
        outs.append(rp4.run_stream(dev(c2["src"], a, b), dev(c2["dst"], a, b), dev(c2["neg"], a, b),
                                   dev(c2["t"], a, b), c["B"]))
    assert torch.equal(torch.cat([outs[0][0], outs[1][0]]), fp1)
    np.testing.assert_array_equal(_layers(rp4), _layers(rp1))


# ---------------------------------------------------------------------------------------------------------
# shared-first-node readout (the encoder's tile/repeat pattern, models/TPNet.py:313-316)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d,L", [(128, 3), (64, 2), (256, 3), (120, 3), (512, 1), (30, 4)])
def test_shared_first_node_readout(d, L):
    """pair_gram_shared(w, a, b) == (pair_gram(w, a), pair_gram(w, b)); get_pair_wise_feature takes that path
    automatically for the encoder's index pattern and returns the reference's row order."""
    _need_gpu()
    rng = np.random.RandomState(d + L)
    N, B, K = 300, 24, 5
    src, dst, neg, t = _random_stream(rng, N, 6 * B, 2.0e5)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    rp = _module(N, d, L, 2e-6, t[0], P0=P0)
    st = O.OracleState(P0, L, 2e-6, t[0])
    for b in range(5):
        s = slice(b * B, (b + 1) * B)
        rp.update(src[s], dst[s], t[s]); O.update(st, src[s], dst[s], t[s])
    s = slice(5 * B, 6 * B)
    neigh = rng.randint(0, N, (2 * B, K)); neigh[rng.rand(2 * B, K) < 0.3] = 0
    w = neigh.reshape(-1)
    a = np.repeat(np.tile(src[s], 2), K); b_ = np.repeat(np.tile(dst[s], 2), K)
    g1, g2 = rp.pair_gram_shared(w, a, b_)
    r1, r2 = rp.pair_gram(w, a), rp.pair_gram(w, b_)
    # (the two kernels add the lanes' partial sums in different orders -- LDS rows vs recursive halving)
    np.testing.assert_allclose(g1.cpu().numpy(), r1.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(g2.cpu().numpy(), r2.cpu().numpy(), rtol=1e-5, atol=1e-5)
    _assert_features(g1.cpu().numpy(), st, w, a)
    _assert_features(g2.cpu().numpy(), st, w, b_)
    # the reference's call: src = tile(neigh, 2), dst = [repeat(tile(src,2),K); repeat(tile(dst,2),K)]
    rp.mlp = torch.nn.Identity()
    full = rp.get_pair_wise_feature(np.tile(w, 2), np.concatenate([a, b_]))
    if d > 128:                                       # the shared-first-node kernel itself served the call
        assert torch.equal(full, torch.cat([g1, g2], dim=0))
    else:                                             # the generic kernel did (16-lane x 2-vector geometry)
        np.testing.assert_allclose(full.cpu().numpy(), torch.cat([g1, g2], dim=0).cpu().numpy(), rtol=1e-5, atol=1e-5)
    # a pair list that only looks similar must take the generic path and still be right
    odd = rp.get_pair_wise_feature(np.concatenate([w, w[::-1]]), np.concatenate([a, b_]))
    _assert_features(odd.cpu().numpy()[len(w):], st, w[::-1], b_)


# ---------------------------------------------------------------------------------------------------------
# the other BASELINE.json configs: oracle on a prefix (the numpy oracle needs ~1 s per 10 000-edge batch),
# size-independent properties on the whole stream
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,nb_oracle", [("C1", 40), ("C3", 8), ("C5", 8)])
def test_other_configs_prefix_against_oracle(cfg, nb_oracle):
    _need_gpu()
    from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
    c = CONFIGS[cfg]
    B, d = c["B"], c["d"]
    E = nb_oracle * B + B // 2                              # ragged tail
    src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
    neg = synthetic_negatives(c["U"], N, E, B, 1)
    torch.manual_seed(0)
    P0 = torch.normal(0, 1 / np.sqrt(d), (N, d)).numpy()
    rp = _module(N, d, 3, c["lam"], 0.0, P0=P0, E=c["E"])
    dev = lambda x: torch.from_numpy(x).to(DEV)
    fp, fn = rp.run_stream(dev(src), dev(dst), dev(neg), dev(t), B)
    fp = fp.cpu().numpy(); fn = fn.cpu().numpy()
    st = O.OracleState(P0, 3, c["lam"], 0.0)
    for b in range(0, E, B):
        s = slice(b, min(b + B, E))
        _assert_features(fp[s], st, src[s], dst[s], f"{cfg} pos batch {b // B}")
        _assert_features(fn[s], st, src[s], neg[s], f"{cfg} neg batch {b // B}")
        O.update(st, src[s], dst[s], t[s])
    _assert_state(_layers(rp), np.stack(st.P[1:]), 1e-4, f"{cfg} state")
    rp.check_device_errors()


def test_c3_full_stream_properties():
    """Reddit-shaped stream at full size (672 447 edges, d=256, B=10 000, 68 batches): bit-for-bit run-to-run
    determinism of every feature and of the final state, and exact linearity in P[0]."""
    _need_gpu()
    from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
    c = CONFIGS["C3"]
    src, dst, t, N = synthetic_stream(c["U"], c["I"], c["E"], c["span"], 0)
    neg = synthetic_negatives(c["U"], N, c["E"], c["B"], 1)
    torch.manual_seed(0)
    P0 = torch.normal(0, 1 / np.sqrt(c["d"]), (N, c["d"])).numpy()
    dev = lambda x: torch.from_numpy(x).to(DEV)
    ds, dd, dn, dt = dev(src), dev(dst), dev(neg), dev(t)
    outs = []
    for scale in (1.0, 1.0, 2.0):
        rp = _module(N, c["d"], 3, c["lam"], 0.0, P0=(scale * P0).astype(np.float32), E=c["E"])
        fp, fn = rp.run_stream(ds, dd, dn, dt, c["B"], t_end=float(t[-1]))
        outs.append((fp, fn, _layers(rp)))
        rp.check_device_errors()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[0][2], outs[1][2])
    np.testing.assert_array_equal(outs[2][2], 2.0 * outs[0][2])
    assert np.isfinite(outs[0][2]).all() and float(np.abs(outs[0][2][2]).max()) > 0


def test_deepcopy_pickle_and_launch_id_wrap():
    """Plumbing robustness: deepcopy / pickle of a module whose truth lives in the engine; the launch-id counter
    wrapping (consolidation through export + import) must not change results."""
    _need_gpu()
    import copy, pickle
    from tpnet_amd import random_projection as rpmod
    rng = np.random.RandomState(3)
    N, d, L, B = 200, 128, 3, 50
    src, dst, neg, t = _random_stream(rng, N, 6 * B, 1.0e5)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    a = _module(N, d, L, 1e-6, t[0], P0=P0)
    b = _module(N, d, L, 1e-6, t[0], P0=P0)
    old = rpmod._MAX_LAUNCH_ID
    try:
        for k in range(6):
            s = slice(k * B, (k + 1) * B)
            a.update(src[s], dst[s], t[s])
            if k == 2:
                rpmod._MAX_LAUNCH_ID = b._launch_id + 1          # force the consolidation path on the next call
            b.update(src[s], dst[s], t[s])
            if k == 3:
                assert b._launch_id <= 3                          # counter restarted
                rpmod._MAX_LAUNCH_ID = old
    finally:
        rpmod._MAX_LAUNCH_ID = old
    # consolidation materialises the pending decay: one extra f32 rounding per element
    _assert_state(_layers(b), _layers(a), 1e-5, "after launch-id consolidation")
    c = copy.deepcopy(a)
    e = pickle.loads(pickle.dumps(a))
    for m in (c, e):
        np.testing.assert_array_equal(_layers(m), _layers(a))
        m.update(src[:B], dst[:B], t[-B:] + 10.0)
    np.testing.assert_array_equal(_layers(c), _layers(e))
    assert not np.array_equal(_layers(c), _layers(a))            # the copies own their state


# ---------------------------------------------------------------------------------------------------------
# odd shapes: tiny dims / graphs / batches, every edge identical, every layer count
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [1, 2, 3, 4, 8, 12, 16, 20, 32, 36, 100, 260])
@pytest.mark.parametrize("L", [1, 2, 3, 4])
def test_odd_shapes_fuzz(d, L):
    _need_gpu()
    rng = np.random.RandomState(1000 * d + L)
    for trial in range(3):
        N = int(rng.randint(2, 7))
        B = int(rng.randint(1, 5))
        nb = int(rng.randint(1, 6))
        E = nb * B - int(rng.randint(0, B))                 # ragged tail, possibly a single edge
        E = max(E, 1)
        src = rng.randint(0, N, E).astype(np.int64)         # id 0 allowed as an endpoint here: it is a legal row
        dst = rng.randint(0, N, E).astype(np.int64)
        if trial == 2:
            src[:] = src[0]; dst[:] = dst[0]                # every edge the same pair
        neg = rng.randint(0, N, E).astype(np.int64)
        t = np.sort(rng.uniform(5.0, 5.0e5, E))
        if trial == 1:
            t[:] = t[0]                                     # all timestamps equal
        lam = float(rng.choice([0.0, 1e-6, 3e-5]))
        P0 = rng.randn(N, d).astype(np.float32)
        for exact in (True, False):
            rp = _module(N, d, L, lam, t[0], P0=P0, exact=exact)
            st = O.OracleState(P0, L, lam, t[0])
            dev = lambda x: torch.from_numpy(x).to(DEV)
            fp, fn = rp.run_stream(dev(src), dev(dst), dev(neg), dev(t), B)
            fp = fp.cpu().numpy(); fn = fn.cpu().numpy()
            for b in range(0, E, B):
                s = slice(b, min(b + B, E))
                _assert_features(fp[s], st, src[s], dst[s], f"d={d} L={L} trial={trial} pos")
                _assert_features(fn[s], st, src[s], neg[s], f"d={d} L={L} trial={trial} neg")
                O.update(st, src[s], dst[s], t[s])
            _assert_state(_layers(rp), np.stack(st.P[1:]), 5e-6 if exact else 1e-4, f"d={d} L={L} trial={trial}")
            # the module API on the same final state
            rows = rp.get_random_projections(np.arange(N))
            np.testing.assert_allclose(np.stack([r.cpu().numpy() for r in rows[1:]]), _layers(rp), rtol=1e-6, atol=1e-30)


def test_internal_chunking_with_a_small_workspace():
    """tpnet_run_stream plans the stream in chunks when the caller's workspace cannot hold it at once; the chunked
    run must equal the one-chunk run bit for bit (same launches, same clocks: the chunk's first batch reads the
    previous chunk's last timestamp on the device).  Exact mode: the per-batch schedule (the windowed one has its own
    test below)."""
    _need_gpu()
    import ctypes as C
    from tpnet_amd import _lib
    rng = np.random.RandomState(9)
    N, d, L, B = 400, 128, 3, 64
    E = 23 * B + 17
    src, dst, neg, t = _random_stream(rng, N, E, 3.0e5)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    ds, dd, dn, dt = dev(src), dev(dst), dev(neg), dev(t)
    XF = _lib.FLAG_EAGER_DECAY | _lib.FLAG_SEQUENTIAL
    a = _module(N, d, L, 2e-6, t[0], P0=P0, exact=True)
    fa, na = a.run_stream(ds, dd, dn, dt, B)
    b = _module(N, d, L, 2e-6, t[0], P0=P0, exact=True)
    b._ensure_engine()
    lib = _lib.load()
    full = lib.tpnet_workspace_bytes(E, B)
    small = lib.tpnet_workspace_bytes(5 * B, B) + 4096          # room for 5 batches per chunk -> 5 chunks
    assert small < full
    ws = torch.empty(small, dtype=torch.uint8, device=DEV)
    fb = torch.empty_like(fa); nb_ = torch.empty_like(na)
    st = b._state()
    t_end = C.c_double(0.0)
    rc = lib.tpnet_run_stream(C.byref(st), ds.data_ptr(), dd.data_ptr(), dn.data_ptr(), dt.data_ptr(), E, B, float(t[0]),
                              2e-6, 1, XF, fb.data_ptr(), nb_.data_ptr(), ws.data_ptr(), small, C.byref(t_end),
                              C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0 and t_end.value == float(t[-1])
    b._now_host = t_end.value
    b._params_valid = False
    assert torch.equal(fa, fb) and torch.equal(na, nb_)
    np.testing.assert_array_equal(_layers(a), _layers(b))
    # packed feature rows through the same chunked path (a chunk's outputs start at c0 * 36 floats, not c0 * 64)
    c = _module(N, d, L, 2e-6, t[0], P0=P0, exact=True)
    c._ensure_engine()
    NT = c.packed_feature_dim
    pp = torch.empty((E, NT), dtype=torch.float32, device=DEV); pn = torch.empty_like(pp)
    st_c = c._state()
    rc = lib.tpnet_run_stream(C.byref(st_c), ds.data_ptr(), dd.data_ptr(), dn.data_ptr(), dt.data_ptr(), E, B, float(t[0]),
                              2e-6, 1, XF | _lib.FLAG_PACKED | _lib.FLAG_NOT_SCALE, pp.data_ptr(), pn.data_ptr(), ws.data_ptr(),
                              small, None, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    for full_rows, pk in ((fa, pp), (na, pn)):
        out = torch.empty_like(full_rows)
        _lib.check(lib.tpnet_gram_unpack(pk.data_ptr(), E, L, 0, out.data_ptr(), c._stream()), "gram_unpack")
        assert torch.equal(out, full_rows)
    # and a workspace that cannot hold even one batch is refused, not overrun
    tiny = torch.empty(1024, dtype=torch.uint8, device=DEV)
    rc = lib.tpnet_run_stream(C.byref(st), ds.data_ptr(), dd.data_ptr(), dn.data_ptr(), dt.data_ptr(), E, B, float(t[0]),
                              2e-6, 100, 0, fb.data_ptr(), nb_.data_ptr(), tiny.data_ptr(), 1024, None,
                              C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == -2


# ---------------------------------------------------------------------------------------------------------
# the windowed schedule of tpnet_run_stream (L+1 launches per window of batches)
# ---------------------------------------------------------------------------------------------------------
def _raw_run_stream(rp, ds, dd, dn, dt, B, now, lam, ws, flags=0):
    import ctypes as C
    from tpnet_amd import _lib
    E = ds.numel()
    NG = rp.packed_feature_dim if (flags & _lib.FLAG_PACKED) else rp.pair_wise_feature_dim
    fp = torch.empty((E, NG), dtype=torch.float32, device=DEV)
    fn = torch.empty_like(fp)
    rp._ensure_engine()
    st = rp._state()
    t_end = C.c_double(0.0)
    rc = _lib.load().tpnet_run_stream(C.byref(st), ds.data_ptr(), dd.data_ptr(), dn.data_ptr(), dt.data_ptr(), E, B, now, lam,
                                      rp._next_launch_ids((E + B - 1) // B), flags | _lib.FLAG_SCHED_WINDOWED, fp.data_ptr(), fn.data_ptr(), ws.data_ptr(),
                                      ws.numel(), C.byref(t_end), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    rp._now_host = t_end.value
    rp._params_valid = False
    return fp, fn


@pytest.mark.parametrize("d,L,N,B,nb", [(128, 3, 400, 64, 330), (64, 2, 3000, 200, 90), (256, 3, 500, 500, 40),
                                        (120, 3, 260, 40, 200), (16, 4, 300, 100, 70), (512, 1, 200, 1000, 9)])
def test_windowed_schedule_is_independent_of_how_the_stream_is_cut(d, L, N, B, nb):
    """The arithmetic of a (node, batch) run is fixed (blocks of 8 contributions in index order), so the windowed
    schedule gives the same bits whatever the windows are: one plan for the whole stream, a workspace that holds only two
    windows per plan, and two separate calls cut at an arbitrary batch.  And it matches the oracle."""
    _need_gpu()
    from tpnet_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(d + B)
    E = nb * B - B // 3
    lam = 2e-6
    src, dst, neg, t = _random_stream(rng, N, E, 4.0e5)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    ds, dd, dn, dt = dev(src), dev(dst), dev(neg), dev(t)
    a = _module(N, d, L, lam, t[0], P0=P0)
    fa, na = a.run_stream(ds, dd, dn, dt, B)
    # (1) against the oracle
    st = O.OracleState(P0, L, lam, t[0])
    fa_h, na_h = fa.cpu().numpy(), na.cpu().numpy()
    for o in range(0, E, B):
        s = slice(o, min(o + B, E))
        _assert_features(fa_h[s], st, src[s], dst[s], f"pos batch {o // B}")
        _assert_features(na_h[s], st, src[s], neg[s], f"neg batch {o // B}")
        O.update(st, src[s], dst[s], t[s])
    _assert_state(_layers(a), np.stack(st.P[1:]), 1e-4, "final state")
    a.check_device_errors()
    # (2) a workspace for two windows per plan
    K = min(64, 16384 // B)
    full = lib.tpnet_stream_workspace_bytes(N, d, L, E, B)
    small = lib.tpnet_stream_workspace_bytes(N, d, L, 2 * K * B, B)
    if small < full:
        b = _module(N, d, L, lam, t[0], P0=P0)
        ws = torch.empty(small, dtype=torch.uint8, device=DEV)
        fb, nb_ = _raw_run_stream(b, ds, dd, dn, dt, B, float(t[0]), lam, ws)
        assert torch.equal(fa, fb) and torch.equal(na, nb_)
        np.testing.assert_array_equal(_layers(a), _layers(b))
    # (3) two calls, cut in the middle of a window
    c = _module(N, d, L, lam, t[0], P0=P0)
    cut = (nb // 2 + 1) * B
    f1, n1 = c.run_stream(ds[:cut], dd[:cut], dn[:cut], dt[:cut], B)
    f2, n2 = c.run_stream(ds[cut:], dd[cut:], dn[cut:], dt[cut:], B)
    assert torch.equal(torch.cat([f1, f2]), fa) and torch.equal(torch.cat([n1, n2]), na)
    np.testing.assert_array_equal(_layers(c), _layers(a))
    # (4) packed rows and update-only streams through the same schedule
    e = _module(N, d, L, lam, t[0], P0=P0)
    pp, pn = e.run_stream(ds, dd, dn, dt, B, packed=True)
    out = torch.empty_like(fa)
    _lib.check(lib.tpnet_gram_unpack(pp.data_ptr(), E, L, 0, out.data_ptr(), e._stream()), "gram_unpack")
    assert torch.equal(out, fa)
    f = _module(N, d, L, lam, t[0], P0=P0)
    f.run_stream(ds, dd, None, dt, B, want_pos=False, want_neg=False)
    np.testing.assert_array_equal(_layers(f), _layers(a))
    # (5) the per-batch schedule (a workspace without room for a version log) agrees within f32 summation order
    g = _module(N, d, L, lam, t[0], P0=P0)
    wsb = torch.empty(lib.tpnet_workspace_bytes(E, B), dtype=torch.uint8, device=DEV)
    if wsb.numel() < lib.tpnet_stream_workspace_bytes(N, d, L, K * B, B):
        fg, ng = _raw_run_stream(g, ds, dd, dn, dt, B, float(t[0]), lam, wsb)
        _assert_state(_layers(g), _layers(a), 1e-5, "per-batch vs windowed")
        # (features: all but the near-cancelling Gram entries of hub rows, whose f32 error scales with the rows' norms)
        assert float(((fg - fa).abs() > 1e-4).float().mean()) < 2e-3


# ---------------------------------------------------------------------------------------------------------
# packed raw readout (the wire format of the column-sharded table) + tpnet_gram_unpack / tpnet_gram_finish
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d,L,N,B", [(128, 3, 150, 64), (16, 3, 150, 64), (64, 2, 150, 64), (30, 1, 150, 64),
                                     (256, 4, 150, 64), (32, 3, 5000, 2600), (128, 3, 7000, 2600), (64, 3, 300, 11)])
def test_packed_readout_unpacks_to_the_full_features(d, L, N, B):
    """(B = 2600: the plan diverts lone contributions to the edge-fused path, which must work with packed rows too)"""
    _need_gpu()
    import ctypes as C
    from tpnet_amd import _lib
    rng = np.random.RandomState(d + L)
    E = 3 * B + 5 if B < 1000 else 2 * B + 300
    if B == 11:
        E = 150 * B + 3                                    # a longer stream of small batches
    src, dst, neg, t = _random_stream(rng, N, E, 2.0e5)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    a = _module(N, d, L, 2e-6, t[0], P0=P0)
    b = _module(N, d, L, 2e-6, t[0], P0=P0)
    fp, fn = a.run_stream(dev(src), dev(dst), dev(neg), dev(t), B)
    pp, pn = b.run_stream(dev(src), dev(dst), dev(neg), dev(t), B, packed=True)
    NN = 2 * L + 2
    assert pp.shape == (E, NN * (NN + 1) // 2) == (E, b.packed_feature_dim)
    lib = _lib.load()
    for full, pk in ((fp, pp), (fn, pn)):
        out = torch.empty_like(full)
        _lib.check(lib.tpnet_gram_unpack(pk.data_ptr(), E, L, 0, out.data_ptr(), b._stream()), "gram_unpack")
        assert torch.equal(out, full)                      # same sums, same tail: identical bits
    np.testing.assert_array_equal(_layers(a), _layers(b))
    # module-level readout: packed -> unpack(NOT_SCALE) == raw; raw -> finish == scaled
    raw = a.pair_gram(src[:B], neg[:B], raw=True)
    pk = a.pair_gram(src[:B], neg[:B], packed=True)
    out = torch.empty_like(raw)
    _lib.check(lib.tpnet_gram_unpack(pk.data_ptr(), B, L, _lib.FLAG_NOT_SCALE, out.data_ptr(), a._stream()), "gram_unpack")
    assert torch.equal(out, raw)
    assert torch.equal(raw, raw.transpose(0, 1).reshape(NN, NN, B).permute(1, 0, 2).reshape(NN * NN, B).transpose(0, 1))  # symmetric
    _lib.check(lib.tpnet_gram_finish(raw.data_ptr(), raw.numel(), a._stream()), "gram_finish")
    assert torch.equal(raw, a.pair_gram(src[:B], neg[:B]))


# ---------------------------------------------------------------------------------------------------------
# edge-fused updates (batches > 1024 edges: a target's lone contribution is applied by its edge's readout group)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d,L,N,B", [(64, 3, 6000, 2600), (128, 2, 3000, 2600), (256, 3, 9000, 2600), (16, 3, 5000, 2600),
                                     (36, 1, 4000, 2600), (512, 4, 3000, 2600), (128, 4, 5000, 2600),
                                     (128, 3, 2500, 1100), (64, 3, 1500, 1025)])
def test_large_batches_with_edge_fused_updates_match_oracle(d, L, N, B):
    """B > 1024 on a sparse graph: most targets have exactly one contribution per batch and take the fused path, hubs
    and duplicates stay on the item lists; features and state must still equal the oracle's."""
    _need_gpu()
    rng = np.random.RandomState(d + N)
    E = 2 * B + 700                                           # ragged third batch (below the threshold: unfused)
    src = rng.randint(1, N, E).astype(np.int64)
    dst = rng.randint(1, N, E).astype(np.int64)
    src[rng.rand(E) < 0.05] = 2                               # a hub
    dst[::41] = src[::41]                                     # self pairs (two contributions on one target)
    neg = rng.randint(0, N, E).astype(np.int64)
    t = np.sort(rng.uniform(1.0e6, 1.5e6, E))
    lam = 2e-6
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    rp = _module(N, d, L, lam, t[0], P0=P0)
    st = O.OracleState(P0, L, lam, t[0])
    dev = lambda x: torch.from_numpy(x).to(DEV)
    fp, fn = rp.run_stream(dev(src), dev(dst), dev(neg), dev(t), B)
    fp = fp.cpu().numpy(); fn = fn.cpu().numpy()
    for b in range(0, E, B):
        s = slice(b, min(b + B, E))
        _assert_features(fp[s], st, src[s], dst[s], f"pos batch {b // B}")
        _assert_features(fn[s], st, src[s], neg[s], f"neg batch {b // B}")
        O.update(st, src[s], dst[s], t[s])
    _assert_state(_layers(rp), np.stack(st.P[1:]), 1e-4, "final state")
    rp.check_device_errors()
    # update-only stream (no readout outputs): the plan must not divert anything to the fused path
    rp2 = _module(N, d, L, lam, t[0], P0=P0)
    rp2.run_stream(dev(src), dev(dst), None, dev(t), B, want_pos=False, want_neg=False)
    # (the ragged tail runs on 256-thread workgroups in the fused stream and on 512-thread ones here: the hub's partial sums
    # associate differently)
    _assert_state(_layers(rp2), _layers(rp), 1e-5, "update-only vs fused")



@pytest.mark.parametrize("d,B,N", [(64, 1500, 2000), (128, 3000, 1200), (32, 1100, 5000)])
def test_module_api_with_batches_above_1024_edges(d, B, N):
    """update() / get_pair_wise_feature() with batches of more than 1024 edges: the update-only launch takes the 256-thread
    step kernel with 128 hub workgroups (no readout in the launch: nothing is diverted to the edge-fused path)."""
    _need_gpu()
    rng = np.random.RandomState(d + B)
    L, lam = 3, 2e-6
    E = 3 * B
    src, dst, neg, t = _random_stream(rng, N, E, 3.0e5)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    rp = _module(N, d, L, lam, t[0], P0=P0)
    rp.mlp = torch.nn.Identity()
    st = O.OracleState(P0, L, lam, t[0])
    for b in range(3):
        s = slice(b * B, (b + 1) * B)
        f = rp.get_pair_wise_feature(src[s], neg[s]).cpu().numpy()
        _assert_features(f, st, src[s], neg[s], f"batch {b}")
        rp.update(src[s], dst[s], t[s]); O.update(st, src[s], dst[s], t[s])
    _assert_state(_layers(rp), np.stack(st.P[1:]), 1e-4, "final state")
    rp.check_device_errors()


# ---------------------------------------------------------------------------------------------------------
# BASELINE.json config C4: the 10 M-node table (d=256: 10 GB per layer, 72 GB of engine state), power-law stream
# S(5 000 000, 5 000 000, E, ...).  The numpy oracle cannot hold a 10 M x 256 x 4 table, but rows are independent:
# the touched node ids are relabelled to a dense range and the oracle runs on the compact graph with the gathered
# P[0] rows; every readout and the touched rows of the final state of the REAL 10 M-row module are compared.
# ---------------------------------------------------------------------------------------------------------
def _c4_module(N, d, lam):
    """The 10 M-row module built directly on the GPU (the reference builds it on the host and moves it: 41 GB of
    host memory that the test box need not have), P[0] ~ N(0, 1/sqrt(d)) from the device generator."""
    from tpnet_amd import RandomProjectionModule
    torch.manual_seed(0)
    rp = RandomProjectionModule(node_num=N, edge_num=200_000_000, dim_factor=10, num_layer=3, time_decay_weight=lam,
                                device=DEV, use_matrix=False, beginning_time=np.float64(0.0), not_scale=False,
                                enforce_dim=d, alloc_device=DEV).to(DEV)
    assert rp.random_projections[0].device.type == "cuda" and rp.random_projections[0].shape == (N, d)
    return rp


def _compact(touched, *arrays):
    return [np.searchsorted(touched, a).astype(np.int64) for a in arrays]


def test_c4_ten_million_rows_against_compact_oracle():
    """C4 (10 000 001 rows, d=256, lambda=1e-7): 8 batches of 10 000 edges + a ragged tail, then ONE call with batches of
    100 000 edges (one full + a ragged one) on the same state.  Exercises the 64-bit row offsets
    ((copy*N + node) * L*d floats: up to 1.5e10), 24-bit node keys in the plan, the edge-fused update path at this
    sparsity (almost every target has a single contribution per batch) and HBM-miss-bound gathers."""
    _need_gpu()
    from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
    c = CONFIGS["C4"]
    N, d, lam = c["U"] + c["I"] + 1, c["d"], c["lam"]
    assert N == 10_000_001 and d == 256
    B1, E1 = 10_000, 8 * 10_000 + 3_000
    B2, E2 = 100_000, 100_000 + 20_000
    E = E1 + E2
    src, dst, t, N_ = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
    assert N_ == N
    neg = np.concatenate([synthetic_negatives(c["U"], N, E1, B1, 1), synthetic_negatives(c["U"], N, E2, B2, 2)])
    rp = _c4_module(N, d, lam)
    touched = np.unique(np.concatenate([[0], src, dst, neg]))
    assert touched[-1] > 9_000_000 and len(touched) > 300_000          # ids really span the whole table
    P0c = rp.random_projections[0][torch.from_numpy(touched).to(DEV)].cpu().numpy()
    st = O.OracleState(P0c, 3, lam, 0.0)
    cs, cd, cn = _compact(touched, src, dst, neg)
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    worst = 0.0
    for (a, b, B) in ((0, E1, B1), (E1, E, B2)):
        fp, fn = rp.run_stream(dev(src[a:b]), dev(dst[a:b]), dev(neg[a:b]), dev(t[a:b]), B, t_end=float(t[b - 1]))
        fp = fp.cpu().numpy(); fn = fn.cpu().numpy()
        for o in range(a, b, B):
            s = slice(o, min(o + B, b))
            for q in range(s.start, s.stop, 20_000):                      # (slices bound the oracle's [n, 8, d] temporaries)
                s2 = slice(q, min(q + 20_000, s.stop))
                r2 = slice(s2.start - a, s2.stop - a)
                worst = max(worst, _assert_features(fp[r2], st, cs[s2], cd[s2], f"C4 pos batch at {o} (B={B})"))
                worst = max(worst, _assert_features(fn[r2], st, cs[s2], cn[s2], f"C4 neg batch at {o} (B={B})"))
            O.update(st, cs[s], cd[s], t[s])
    rp.check_device_errors()
    # touched rows of the final state, through the row gather (no 30 GB materialisation), all layers incl. layer 0
    rows = rp.get_random_projections(touched)
    got = np.stack([r.cpu().numpy() for r in rows])
    np.testing.assert_array_equal(got[0], P0c)
    _assert_state(got[1:], np.stack(st.P[1:]), 1e-4, "C4 touched rows")
    assert float(np.abs(got[3]).max()) > 0                              # three-hop walks exist: layer 3 is populated
    # untouched rows stayed zero (a sample next to touched ids and at the far end of the table)
    cand = np.unique(np.concatenate([touched[:2000] + 1, touched[-2000:] - 1, [N - 1, N - 2, 1]]))
    untouched = cand[(cand > 0) & (cand < N) & ~np.isin(cand, touched)]
    z = rp.get_random_projections(untouched)
    assert all(float(r.abs().max()) == 0.0 for r in z[1:])
    assert float(rp.now_time.item()) == float(t[-1])
    print(f"C4: {len(touched)} touched rows of {N}, worst |feature delta| = {worst:.3e}")


def test_c4_wide_keys_chunked_equals_unchunked():
    """10 M rows with MANY batches in one plan: 24 node bits + 10 batch bits exceed 32, so the plan sorts 64-bit
    (batch, node) keys; the same stream cut into three calls (each plan then fits 32-bit keys) must give the same
    features and the same touched rows -- the two key widths are two sorts of the same contributions -- and a second
    one-call run must reproduce the first bit for bit."""
    _need_gpu()
    from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
    c = CONFIGS["C4"]
    N, d, lam = c["U"] + c["I"] + 1, 64, c["lam"]                         # narrow rows: the keys are what is under test
    B, nb = 4_500, 600                                                    # 2B > 8192: device-wide sort, 600 batches: 10 bits
    E = nb * B - 777
    src, dst, t, _ = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 3)
    neg = synthetic_negatives(c["U"], N, E, B, 4)
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    ds, dd, dn, dt = dev(src), dev(dst), dev(neg), dev(t)
    touched = torch.unique(torch.cat([ds, dd, dn]))
    outs = []
    for cuts in ((0, E), (0, E), (0, 200 * B, 410 * B, E)):
        rp = _c4_module(N, d, lam)
        fps, fns = [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            fp, fn = rp.run_stream(ds[a:b], dd[a:b], dn[a:b], dt[a:b], B, t_end=float(t[b - 1]))
            fps.append(fp); fns.append(fn)
        rp.check_device_errors()
        rows = torch.stack(rp.get_random_projections(touched)[1:])
        outs.append((torch.cat(fps), torch.cat(fns), rows))
        del rp
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1]) and torch.equal(outs[0][2], outs[2][2])
    assert float(outs[0][2][2].abs().max()) > 0


def test_in_place_writes_through_data_are_seen():
    """p.data.copy_() / p.data[idx] = ... on a handed-out Parameter change neither data_ptr nor _version; the
    engine must still pick them up (the hand-out itself forces the re-import)."""
    _need_gpu()
    rng = np.random.RandomState(5)
    N, d, L, B = 120, 64, 3, 32
    src, dst, neg, t = _random_stream(rng, N, 4 * B, 1.0e5)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    rp = _module(N, d, L, 1e-6, t[0], P0=P0)
    st = O.OracleState(P0, L, 1e-6, t[0])
    for b in range(2):
        s = slice(b * B, (b + 1) * B)
        rp.update(src[s], dst[s], t[s]); O.update(st, src[s], dst[s], t[s])
    new1 = (rng.randn(N, d)).astype(np.float32)
    rp.random_projections[1].data.copy_(torch.from_numpy(new1))          # in place, through .data
    rp.random_projections[2].data[5] = 7.0
    st.P[1] = new1.copy(); st.P[2][5] = 7.0
    s = slice(2 * B, 3 * B)
    _assert_features(rp.pair_gram(src[s], dst[s]).cpu().numpy(), st, src[s], dst[s], "after in-place writes")
    rp.update(src[s], dst[s], t[s]); O.update(st, src[s], dst[s], t[s])
    _assert_state(_layers(rp), np.stack(st.P[1:]), 1e-4, "after in-place writes + update")
    # a state_dict() hands out aliases too
    sd = rp.state_dict()
    sd["random_projections.3"].zero_()
    st.P[3][...] = 0
    s = slice(3 * B, 4 * B)
    rp.update(src[s], dst[s], t[s]); O.update(st, src[s], dst[s], t[s])
    _assert_state(_layers(rp), np.stack(st.P[1:]), 1e-4, "after a write through state_dict()")


@pytest.mark.parametrize("d,B,N", [(16, 100000, 48), (64, 60000, 40), (128, 40000, 64)])
def test_version_protocol_under_multi_pass_grids(d, B, N):
    """The ping-pong / version protocol under the worst interleaving the per-batch kernel can produce: a batch far larger
    than the grid (several grid-stride rounds) on a graph so small that EVERY node is rewritten early in the launch (by
    the hub workgroups, which lead the grid) and read by pairs that start many rounds later, when the writers are long
    done -- the case the launch-id check of a reader exists for (device_common.hpp meta_view: a ver stamped by THIS launch
    means the other copy is the pre-batch one).  Every readout must see pre-batch rows (oracle), run to run bit-identical."""
    _need_gpu()
    L, lam = 3, 1e-6
    rng = np.random.RandomState(d + N)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    E = 2 * B
    src = rng.randint(1, N, E).astype(np.int64)
    dst = rng.randint(1, N, E).astype(np.int64)
    neg = rng.randint(1, N, E).astype(np.int64)
    t = np.sort(rng.uniform(0, 5e5, E))
    D = lambda x: torch.from_numpy(x).to(DEV)
    outs = []
    for rep in range(2):
        rp = _module(N, d, L, lam, 0.0, P0)
        fp, fn = rp.run_stream(D(src), D(dst), D(neg), D(t), B, schedule="batch")
        outs.append((fp.clone(), fn.clone(), _layers(rp)))
        rp.check_device_errors()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[0][2], outs[1][2])
    st = O.OracleState(P0, L, lam, 0.0)
    for b in range(2):
        s = slice(b * B, (b + 1) * B)
        sub = rng.choice(B, 3000, replace=False) + b * B           # a sample of the batch's pairs against the oracle
        _assert_features(outs[0][0][sub].cpu().numpy(), st, src[sub], dst[sub], f"batch {b} pos")
        _assert_features(outs[0][1][sub].cpu().numpy(), st, src[sub], neg[sub], f"batch {b} neg")
        O.update(st, src[s], dst[s], t[s])
    _assert_state(outs[0][2], np.stack(st.P[1:]), 2e-4, "state")


# ---------------------------------------------------------------------------------------------------------
# the three-launch planner (wplan3.hip) and plan replay across epochs
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d,L,N,B,nb", [(128, 3, 400, 64, 50), (64, 2, 3000, 200, 90), (128, 3, 9000, 1000, 20),
                                        (256, 3, 500, 500, 13), (16, 4, 300, 100, 70), (512, 1, 200, 2048, 5),
                                        (120, 3, 260, 40, 200), (128, 3, 50, 1000, 12),
                                        (16, 3, 60000, 100, 40)])      # (many nodes, few edges: the chain table is hashed)
def test_three_launch_planner_equals_chunk_planner(d, L, N, B, nb):
    """All three planners of the windowed schedule (the one-launch dense planner where the table is small against the stream --
    every case but the last --, the hashed planner, the sorted chunk planner) describe the same runs (blocks of 8 contributions
    in index order, the same versions read): features and state are equal bit for bit, whatever the window length each picks."""
    _need_gpu()
    rng = np.random.RandomState(d + B + nb)
    E = nb * B - B // 3
    lam = 2e-6
    src, dst, neg, t = _random_stream(rng, N, E, 4.0e5)
    src[5] = N + 3                                               # a bad edge and a bad negative: skipped and counted
    neg[7] = -2
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    ds, dd, dn, dt = dev(src), dev(dst), dev(neg), dev(t)
    a = _module(N, d, L, lam, t[0], P0=P0)
    fa, na = a.run_stream(ds, dd, dn, dt, B, schedule="windowed")
    b = _module(N, d, L, lam, t[0], P0=P0)
    fb, nb_ = b.run_stream(ds, dd, dn, dt, B, schedule="windowed-sorted")
    ok = torch.ones(E, dtype=torch.bool, device=DEV)
    ok[5] = False                                                # (rows of a bad id are not defined)
    c = _module(N, d, L, lam, t[0], P0=P0)
    fc, nc_ = c.run_stream(ds, dd, dn, dt, B, schedule="windowed-hashed")
    assert torch.equal(fa[ok], fb[ok]) and torch.equal(fc[ok], fb[ok])
    ok[7] = False
    assert torch.equal(na[ok], nb_[ok]) and torch.equal(nc_[ok], nb_[ok])
    np.testing.assert_array_equal(_layers(a), _layers(b))
    np.testing.assert_array_equal(_layers(c), _layers(b))
    for m in (a, b, c):
        with pytest.raises(IndexError):
            m.check_device_errors()


@pytest.mark.parametrize("d,L,N,B,nb", [(128, 3, 400, 64, 50), (64, 2, 3000, 200, 30), (128, 3, 9000, 1000, 20),
                                        (64, 3, 2000, 100, 2048)])     # (2 048 batches: 86 windows, ONE chunk -- replayed too)
def test_plan_replay_across_epochs(d, L, N, B, nb):
    """train_link_prediction.py:234-253: every epoch resets the projections and runs the SAME stream with new negatives.
    The second epoch replays the first one's plan (only the negatives' readout references are formed again) and gives the
    bits a cold plan gives; anything that invalidates the plan (an in-place write to the stream, another use of the
    workspace, a different table state) is noticed."""
    _need_gpu()
    rng = np.random.RandomState(d + B + nb)
    E = nb * B - B // 3
    lam = 2e-6
    src, dst, neg, t = _random_stream(rng, N, E, 4.0e5)
    neg2 = rng.randint(0, N, E).astype(np.int64)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    ds, dd, dn, dn2, dt = dev(src), dev(dst), dev(neg), dev(neg2), dev(t)

    def epoch(rp, negs, **kw):
        rp.reset_random_projections()
        rp.random_projections[0].data.copy_(torch.from_numpy(P0))
        fp, fn = rp.run_stream(ds, dd, negs, dt, B, schedule="windowed", **kw)
        return fp.clone(), fn.clone(), _layers(rp), rp.last_stream_replayed

    cold = _module(N, d, L, lam, t[0], P0=P0)
    c1 = epoch(cold, dn, replay=False)
    c2_ = epoch(cold, dn2, replay=False)
    assert not c1[3] and not c2_[3]
    rp = _module(N, d, L, lam, t[0], P0=P0)
    e1 = epoch(rp, dn)
    e2 = epoch(rp, dn2)
    e3 = epoch(rp, dn)
    assert not e1[3] and e2[3] and e3[3]
    for got, want in ((e1, c1), (e2, c2_), (e3, c1)):
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        np.testing.assert_array_equal(got[2], want[2])
    # an in-place write to the stream is seen (torch bumps _version) ...
    dd[3] = dd[3]
    assert not epoch(rp, dn)[3]
    assert epoch(rp, dn)[3]
    # ... so is another user of the workspace (update() plans in it) ...
    rp.reset_random_projections()
    rp.update(src[:B], dst[:B], t[:B])
    assert not epoch(rp, dn)[3]
    # ... and a table that is not in the state the plan was built for (no reset: the second call continues the first)
    rp.run_stream(ds, dd, dn, dt + (t[-1] - t[0] + 1.0), B, schedule="windowed")
    assert not rp.last_stream_replayed
    e4 = epoch(rp, dn)
    assert torch.equal(e4[0], c1[0]) and torch.equal(e4[1], c1[1])
    np.testing.assert_array_equal(e4[2], c1[2])


@pytest.mark.parametrize("d,L,N,B,nb,schedule", [(128, 3, 9000, 1000, 20, "auto"), (64, 2, 700, 100, 9, "auto"), (256, 3, 900, 3000, 5, "batch")])
def test_prepared_stream_call_equals_run_stream(d, L, N, B, nb, schedule):
    """rp.prepare_stream(...) checks run_stream's arguments once; the prepared call then runs the stream every time it is called:
    the same bits as run_stream, the plan of an earlier run replayed under the same rules (a new `neg` written into the held tensor
    does not stop it, a write to src does), another use of the workspace in between is fine, bad arguments are refused at prepare."""
    _need_gpu()
    rng = np.random.RandomState(d + nb)
    E = nb * B - B // 4
    src, dst, neg, t = _random_stream(rng, N, E, 4.0e5)
    neg2 = rng.randint(0, N, E).astype(np.int64)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    ds, dd, dn, dn2, dt = dev(src), dev(dst), dev(neg), dev(neg2), dev(t)
    a = _module(N, d, L, 2e-6, t[0], P0=P0)
    fa = a.run_stream(ds, dd, dn, dt, B, schedule=schedule, replay=False)
    fa = (fa[0].clone(), fa[1].clone())
    la = _layers(a)
    fa2 = _module(N, d, L, 2e-6, t[0], P0=P0).run_stream(ds, dd, dn2, dt, B, schedule=schedule, replay=False)
    b = _module(N, d, L, 2e-6, t[0], P0=P0)
    held_neg = dn.clone()
    call = b.prepare_stream(ds, dd, held_neg, dt, B, schedule=schedule, t_end=float(t[-1]))

    def epoch():
        b.reset_random_projections()
        b.random_projections[0].data.copy_(torch.from_numpy(P0))
        fp, fn = call()
        assert fp is call.out_pos and fn is call.out_neg
        return fp.clone(), fn.clone(), _layers(b), b.last_stream_replayed

    e1 = epoch()
    assert not e1[3] and torch.equal(e1[0], fa[0]) and torch.equal(e1[1], fa[1])
    np.testing.assert_array_equal(e1[2], la)
    held_neg.copy_(dn2)                                       # the next epoch's negatives, in place: the plan is still replayed
    e2 = epoch()
    assert e2[3] and torch.equal(e2[0], fa2[0]) and torch.equal(e2[1], fa2[1])
    np.testing.assert_array_equal(e2[2], la)
    b.reset_random_projections()
    b.update(src[:B], dst[:B], t[:B])                         # another user of the workspace: the call plans again, same bits
    e3 = epoch()
    assert not e3[3] and torch.equal(e3[0], fa2[0])
    dd[5] = dd[5]                                             # a write to the stream: noticed (torch's version counter)
    assert not epoch()[3]
    assert epoch()[3]
    assert float(b.now_time.item()) == float(t[-1])
    with pytest.raises(ValueError):
        b.prepare_stream(ds, dd.to(torch.int32), dn, dt, B)
    with pytest.raises(ValueError):
        b.prepare_stream(ds, dd, dn, dt, B, schedule="fastest")
    with pytest.raises(ValueError):
        b.prepare_stream(ds, dd, dn, dt, B, out_pos=torch.empty((E, 3), device=DEV))
    empty = b.prepare_stream(ds[:0], dd[:0], dn[:0], dt[:0], B)
    assert empty()[0].shape[0] == 0
    with pytest.raises(AttributeError):                       # a tensor swapped in behind the prepared pointers is refused
        call.neg = dn2


@pytest.mark.parametrize("d,L,N,B,nb", [(64, 2, 500, 1000, 5),      # 2 000 pairs in workgroups of 32 lane groups: half a workgroup left over
                                        (64, 2, 500, 2049, 3),      # 4 098 pairs in workgroups of 16
                                        (128, 3, 300, 1000, 4), (64, 2, 5000, 300, 6)])
def test_exact_mode_stream_is_reproducible(d, L, N, B, nb):
    """The exact mode of a stream is three launches per batch (readout, dense decay, update: models/TPNet.py:83-96).  Until round 5
    the lane groups behind the last pair of the READOUT launch walked the first update items (step_impl.hpp): the same nodes were
    then updated again behind the decay from the copy it had not scaled -- inside the mode's tolerance, but different bits run to
    run.  Same bits every run, and the same state as the per-batch calls of the same mode."""
    _need_gpu()
    rng = np.random.RandomState(1)
    E = nb * B
    src, dst, neg, t = _random_stream(rng, N, E, 4.0e5)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    ds, dd, dn, dt = dev(src), dev(dst), dev(neg), dev(t)
    rp = _module(N, d, L, 2e-6, t[0], P0=P0, exact=True)
    outs = []
    for _ in range(4):
        rp.reset_random_projections()
        rp.random_projections[0].data.copy_(torch.from_numpy(P0))
        fp, fn = rp.run_stream(ds, dd, dn, dt, B, replay=False)
        outs.append((fp.clone(), fn.clone(), _layers(rp)))
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])
        np.testing.assert_array_equal(o[2], outs[0][2])
    # the stream without readouts (no readout launch at all) gives the same state
    rp.reset_random_projections()
    rp.random_projections[0].data.copy_(torch.from_numpy(P0))
    rp.run_stream(ds, dd, None, dt, B, want_pos=False, want_neg=False, replay=False)
    np.testing.assert_array_equal(_layers(rp), outs[0][2])


@pytest.mark.parametrize("d,L,N,B,nb,schedule,exact", [(256, 3, 3000, 5000, 12, "auto", False),     # the chunk planner (B > 4 096: C3 / C5)
                                                        (128, 3, 2000, 10000, 6, "auto", False),
                                                        (128, 3, 900, 300, 9, "batch", False),       # one workgroup per batch
                                                        (64, 2, 500, 2500, 7, "batch", True)])       # the exact mode's plan
def test_per_batch_plan_replay_across_epochs(d, L, N, B, nb, schedule, exact):
    """The per-batch schedule's plan (item lists, coefficients, batch descriptors) depends on the stream and the clock at entry
    alone: a stream of one chunk replays it in every later epoch (train_link_prediction.py:234-253) -- whatever the table holds --
    and gives the bits a cold plan gives."""
    _need_gpu()
    rng = np.random.RandomState(d + B + nb)
    E = nb * B - B // 3
    lam = 2e-6
    src, dst, neg, t = _random_stream(rng, N, E, 4.0e5)
    neg2 = rng.randint(0, N, E).astype(np.int64)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    ds, dd, dn, dn2, dt = dev(src), dev(dst), dev(neg), dev(neg2), dev(t)

    def epoch(rp, negs, reset=True, **kw):
        if reset:
            rp.reset_random_projections()
            rp.random_projections[0].data.copy_(torch.from_numpy(P0))
        fp, fn = rp.run_stream(ds, dd, negs, dt, B, schedule=schedule, **kw)
        return fp.clone(), fn.clone(), _layers(rp), rp.last_stream_replayed

    make = lambda: _module(N, d, L, lam, t[0], P0=P0, exact=exact)

    cold = make()
    c1 = epoch(cold, dn, replay=False)
    c2_ = epoch(cold, dn2, replay=False)
    assert not c1[3] and not c2_[3]
    rp = make()
    e1 = epoch(rp, dn)
    e2 = epoch(rp, dn2)
    e3 = epoch(rp, dn)
    assert not e1[3] and e2[3] and e3[3]
    for got, want in ((e1, c1), (e2, c2_), (e3, c1)):
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        np.testing.assert_array_equal(got[2], want[2])
    # an in-place write to the stream is seen, and so is another user of the workspace
    dd[3] = dd[3]
    assert not epoch(rp, dn)[3]
    assert epoch(rp, dn)[3]
    rp.reset_random_projections()
    rp.update(src[:B], dst[:B], t[:B])
    assert not epoch(rp, dn)[3]
    # a different clock at entry is another plan (the first batch's descriptor carries it)
    rp.reset_random_projections()
    rp.random_projections[0].data.copy_(torch.from_numpy(P0))
    rp.run_stream(ds[:B], dd[:B], dn[:B], dt[:B], B, schedule=schedule)
    assert not epoch(rp, dn, reset=False)[3]
    e4 = epoch(rp, dn)
    assert not e4[3]                                       # (the clock is back at t[0]: not what the plan in place was built for)
    e5 = epoch(rp, dn)
    assert e5[3]
    for got in (e4, e5):
        assert torch.equal(got[0], c1[0]) and torch.equal(got[1], c1[1])
        np.testing.assert_array_equal(got[2], c1[2])


@pytest.mark.parametrize("d,L,N,B,nb,per,packed", [(64, 3, 2000, 100, 300, 130, False), (128, 3, 9000, 1000, 70, 30, False),
                                                   (64, 2, 3000, 200, 150, 70, False),
                                                   (64, 2, 4000, 1170, 90, 45, True)])      # (packed rows of 21 floats: chunks of whole 16 bytes)
def test_plan_replay_of_a_stream_of_several_chunks(d, L, N, B, nb, per, packed):
    """A stream longer than one chunk (here: the version log capped at `per` batches' worth, `stream_log_cap_bytes`; without a cap:
    beyond 16 GiB of log or 256 windows) keeps the plan of EVERY chunk -- one region per chunk in front of the one log they share
    (api.hip, window_chunk / run_stream_windowed) -- so the second epoch replays all of them: same bits as the cold one-chunk run,
    and the replay is invalidated as a one-chunk plan is."""
    _need_gpu()
    from tpnet_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(d + B + nb)
    E = nb * B - B // 3
    lam = 2e-6
    src, dst, neg, t = _random_stream(rng, N, E, 4.0e5)
    neg2 = rng.randint(0, N, E).astype(np.int64)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    ds, dd, dn, dn2, dt = dev(src), dev(dst), dev(neg), dev(neg2), dev(t)

    def epoch(rp, negs, **kw):
        rp.reset_random_projections()
        rp.random_projections[0].data.copy_(torch.from_numpy(P0))
        fp, fn = rp.run_stream(ds, dd, negs, dt, B, schedule="windowed", packed=packed, **kw)
        return fp.clone(), fn.clone(), _layers(rp), rp.last_stream_replayed

    one = _module(N, d, L, lam, t[0], P0=P0)
    c1 = epoch(one, dn, replay=False)
    c2_ = epoch(one, dn2, replay=False)
    rp = _module(N, d, L, lam, t[0], P0=P0)
    rp.stream_log_cap_bytes = per * B * 2 * L * d * 4
    e1 = epoch(rp, dn)
    assert rp._eng["ws"].numel() < one._eng["ws"].numel()       # (several chunks: less log than the whole stream's)
    e2 = epoch(rp, dn2)
    e3 = epoch(rp, dn)
    assert not e1[3] and e2[3] and e3[3]
    for got, want in ((e1, c1), (e2, c2_), (e3, c1)):
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        np.testing.assert_array_equal(got[2], want[2])
    # update-only epochs replay too (a plan with readouts is not one without: planned once more, then replayed)
    assert not epoch_update_only(rp, ds, dd, dt, B, P0)
    assert epoch_update_only(rp, ds, dd, dt, B, P0)
    np.testing.assert_array_equal(_layers(rp), c1[2])
    # an in-place write to the stream is seen, so is another user of the workspace
    dd[3] = dd[3]
    assert not epoch(rp, dn)[3]
    assert epoch(rp, dn)[3]
    rp.reset_random_projections()
    rp.update(src[:B], dst[:B], t[:B])
    e4 = epoch(rp, dn)
    assert not e4[3]
    assert torch.equal(e4[0], c1[0]) and torch.equal(e4[1], c1[1])
    # a workspace sized for ONE capped chunk: chunk by chunk all the same, replayed or not as the chunks' plans find room in it
    small = _module(N, d, L, lam, t[0], P0=P0)
    ws = torch.empty(lib.tpnet_stream_workspace_bytes(N, d, L, per * B, B), dtype=torch.uint8, device=DEV)
    small._ensure_engine()
    small._eng["ws"] = ws
    small._ws_need = {(E, B, True, 0, 0): ws.numel()}
    epoch(small, dn)
    s2 = epoch(small, dn2)
    assert small._eng["ws"] is ws
    assert torch.equal(s2[0], c2_[0]) and torch.equal(s2[1], c2_[1])
    np.testing.assert_array_equal(s2[2], c2_[2])


def epoch_update_only(rp, ds, dd, dt, B, P0):
    rp.reset_random_projections()
    rp.random_projections[0].data.copy_(torch.from_numpy(P0))
    rp.run_stream(ds, dd, None, dt, B, schedule="windowed", want_pos=False, want_neg=False)
    return rp.last_stream_replayed


@pytest.mark.parametrize("d,L,N,B,nb", [(120, 3, 9000, 20, 67), (128, 3, 500, 37, 40), (64, 2, 300, 50, 25), (256, 3, 200, 13, 30)])
@pytest.mark.parametrize("schedule", ["windowed", "batch"])
def test_stream_with_no_negatives_at_all(d, L, N, B, nb, schedule):
    """run_stream(src, dst, None, t): the (src, dst) features and the update alone -- no negative ids exist, the pointer the
    kernels get is null.  (Found by tools/soak_chunks.py: the pipeline's readout let the lane groups PAST the end of a window's
    pair list form the address neg[e0] -- a fault at address 0 whenever the window's pairs did not fill its last workgroup.)
    Same features and state as the run that also reads out the negatives."""
    _need_gpu()
    rng = np.random.RandomState(d + B)
    E = nb * B - B // 2
    src, dst, neg, t = _random_stream(rng, N, E, 4.0e5)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    ds, dd, dn, dt = dev(src), dev(dst), dev(neg), dev(t)
    a = _module(N, d, L, 2e-6, t[0], P0=P0)
    pa, na = a.run_stream(ds, dd, dn, dt, B, schedule=schedule)
    b = _module(N, d, L, 2e-6, t[0], P0=P0)
    pb, nb_ = b.run_stream(ds, dd, None, dt, B, schedule=schedule)
    assert nb_ is None and torch.equal(pa, pb)
    np.testing.assert_array_equal(_layers(a), _layers(b))
    b.check_device_errors()


@pytest.mark.parametrize("schedule", ["windowed", "batch"])
@pytest.mark.parametrize("d,L,N,B,nb", [(128, 3, 500, 37, 40), (64, 2, 300, 50, 25), (256, 3, 2000, 1100, 6)])
def test_every_optional_pointer_null_on_both_schedules(d, L, N, B, nb, schedule):
    """tpnet_run_stream / tpnet_run_stream_tagged through the raw C ABI with each optional pointer NULL in turn, on BOTH schedules
    (round 4's fault lived where only one schedule had been tried: `neg == NULL` on the pipeline): out_pos alone, out_neg alone,
    neither (the update alone), neg + out_neg NULL, t_end_out NULL / given, tag NULL / given.  Every output that exists equals the
    full run's bit for bit (batches > 1 024 edges on the per-batch schedule: 2e-5, the edge-fused update sums in another order when
    the (src, dst) readout is missing), and so does the state; out_neg without neg is TPNET_ERR_BAD_ARG."""
    import ctypes as C
    from tpnet_amd import _lib
    _need_gpu()
    lib = _lib.load()
    rng = np.random.RandomState(7 * d + B)
    E = nb * B - B // 3
    src, dst, neg, t = _random_stream(rng, N, E, 4.0e5)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    ds, dd, dn, dt = dev(src), dev(dst), dev(neg), dev(t)
    flag = {"windowed": _lib.FLAG_SCHED_WINDOWED, "batch": _lib.FLAG_SCHED_BATCH}[schedule]
    NG = (2 * L + 2) ** 2

    def run(pos, negs, with_neg_ids, t_out, tagged):
        rp = _module(N, d, L, 2e-6, t[0], P0=P0)
        rp._ensure_engine()
        ws = rp._workspace(E, B, stream=True)
        st = rp._state()
        op = torch.full((E, NG), -7.0, device=DEV) if pos else None
        on = torch.full((E, NG), -7.0, device=DEV) if negs else None
        tv = C.c_double(-1.0)
        tag = _lib.PlanTag() if tagged else None
        args = [C.byref(st), ds.data_ptr(), dd.data_ptr(), dn.data_ptr() if with_neg_ids else None, dt.data_ptr(), E, B, float(t[0]),
                2e-6, rp._next_launch_ids((E + B - 1) // B), flag, op.data_ptr() if pos else None, on.data_ptr() if negs else None,
                ws.data_ptr(), ws.numel(), C.byref(tv) if t_out else None, rp._stream()]
        rc = lib.tpnet_run_stream_tagged(*args, C.byref(tag) if tagged else None) if tagged is not None else lib.tpnet_run_stream(*args)
        if rc == 0:
            torch.cuda.synchronize()
            rp._now_host = float(t[-1]); rp._params_valid = False; rp._now_dirty = True; rp._table_written()
            if t_out:
                assert tv.value == float(t[-1])
            rp.check_device_errors()
        return rc, op, on, (_layers(rp) if rc == 0 else None)

    rc, fp, fn, lay = run(True, True, True, True, None)
    assert rc == 0 and float(fp.min()) > -7.0 and float(fn.min()) > -7.0
    fused = schedule == "batch" and B > 1024            # (the edge-fused update needs the (src, dst) readout in the same launch)
    for pos, negs, ids, t_out, tagged in [(True, False, True, False, None), (False, True, True, True, None), (False, False, True, False, None),
                                          (True, False, False, True, None), (False, False, False, False, None),
                                          (True, True, True, False, True), (True, False, False, False, True), (False, False, True, True, True)]:
        rc, op, on, lay2 = run(pos, negs, ids, t_out, tagged)
        assert rc == 0, (pos, negs, ids, t_out, tagged, rc)
        if fused and not pos:
            # (without the (src, dst) readout the per-batch kernel's edge-fused update is off: the same sums in another order, so the
            # state -- and every later batch's features -- agree to rounding, not to the bit)
            if negs:
                np.testing.assert_allclose(on.cpu().numpy(), fn.cpu().numpy(), rtol=1e-4, atol=2e-5)
            np.testing.assert_allclose(lay2, lay, rtol=2e-5, atol=2e-5 * float(np.abs(lay).max()))
            continue
        if pos:
            assert torch.equal(op, fp), (pos, negs, ids)
        if negs:
            assert torch.equal(on, fn), (pos, negs, ids)
        np.testing.assert_array_equal(lay2, lay)
    rc, _, _, _ = run(False, True, False, False, None)             # out_neg without neg
    assert rc == -1


@pytest.mark.parametrize("nb,d,N,B", [(15, 64, 500, 100), (16, 64, 500, 100), (60, 64, 500, 100),
                                      (20, 128, 9228, 1000)])     # (the driver's timed shape: C2, 20 batches)
def test_auto_schedule_on_both_sides_of_its_threshold(nb, d, N, B):
    """The shipped default ("auto") takes the per-batch kernels below 16 batches and the windowed pipeline from there
    (tpnet_amd/csrc/api.hip, window_chunk): both sides against the oracle, and the choice itself (tpnet_stream_schedule; either
    schedule leaves a plan that the next epoch replays -- the per-batch one since round 5)."""
    _need_gpu()
    L, lam = 3, 2e-6
    rng = np.random.RandomState(nb)
    E = nb * B
    src, dst, neg, t = _random_stream(rng, N, E, 4.0e5)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    ds, dd, dn, dt = dev(src), dev(dst), dev(neg), dev(t)
    rp = _module(N, d, L, lam, t[0], P0=P0)
    fp, fn = rp.run_stream(ds, dd, dn, dt, B, schedule="auto")
    st = O.OracleState(P0, L, lam, t[0])
    fp_h, fn_h = fp.cpu().numpy(), fn.cpu().numpy()
    for o in range(0, E, B):
        s = slice(o, o + B)
        _assert_features(fp_h[s], st, src[s], dst[s], f"pos batch {o // B}")
        _assert_features(fn_h[s], st, src[s], neg[s], f"neg batch {o // B}")
        O.update(st, src[s], dst[s], t[s])
    _assert_state(_layers(rp), np.stack(st.P[1:]), 1e-4, "final state")
    rp.reset_random_projections()
    rp.random_projections[0].data.copy_(torch.from_numpy(P0))
    rp.run_stream(ds, dd, dn, dt, B, schedule="auto")
    assert rp.last_stream_replayed
    from tpnet_amd import _lib
    assert (_lib.load().tpnet_stream_schedule(N, d, L, E, B, 0, rp._eng["ws"].numel()) == 1) == (nb >= 16)


def test_multi_chunk_packed_rows_with_odd_row_length():
    """Packed rows of 21 floats (L = 2) and a workspace that holds only a few windows: every chunk's output must start on a
    16-byte boundary, which a chunk of an odd number of edges would break: the chunk length is rounded."""
    _need_gpu()
    from tpnet_amd import _lib
    lib = _lib.load()
    d, L, N, B, nb, lam = 64, 2, 4000, 1170, 90, 2e-6            # windows of 21 batches = 24 570 edges: not a multiple of 4
    rng = np.random.RandomState(5)
    E = nb * B
    src, dst, neg, t = _random_stream(rng, N, E, 4.0e5)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    ds, dd, dn, dt = dev(src), dev(dst), dev(neg), dev(t)
    a = _module(N, d, L, lam, t[0], P0=P0)
    pa, na = a.run_stream(ds, dd, dn, dt, B, packed=True)
    b = _module(N, d, L, lam, t[0], P0=P0)
    small = lib.tpnet_stream_workspace_bytes_capped(N, d, L, E, B, 2 * 21 * B * 2 * L * d * 4)     # two windows per chunk
    assert small < lib.tpnet_stream_workspace_bytes(N, d, L, E, B)
    ws = torch.empty(small, dtype=torch.uint8, device=DEV)
    pb, nb_ = _raw_run_stream(b, ds, dd, dn, dt, B, float(t[0]), lam, ws, flags=_lib.FLAG_PACKED | _lib.FLAG_NOT_SCALE)
    assert torch.equal(pa, pb) and torch.equal(na, nb_)
    np.testing.assert_array_equal(_layers(a), _layers(b))
