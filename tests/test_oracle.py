"""CPU tests: the oracle (oracle/tpnet_oracle.py) against the golden vectors generated from the reference itself
(tests/golden/make_golden.py) and against reference-independent known answers (G1-G7, K1-K4)."""
import glob
import os

import numpy as np
import pytest

from oracle import tpnet_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _replay(g):
    L, B = int(g["L"]), int(g["B"])
    st = O.OracleState(g["P0"], L, float(g["lam"]), float(g["t0"]))
    nb = len(g["src"]) // B
    for b in range(nb):
        s = slice(b * B, (b + 1) * B)
        O.update(st, g["src"][s], g["dst"][s], g["t"][s])
        yield b, st


@pytest.mark.parametrize("name", ["g1_update_d16_L3.npz", "g2_update_d120_L2.npz",
                                  "g2b_update_d64_L3_fastdecay.npz", "g6_use_matrix_N30.npz"])
def test_update_trajectory_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    for b, st in _replay(g):
        ref = g[f"P_after_{b}"]
        mine = np.stack(st.P[1:])
        # the only difference to the reference is the f32 exp of libm vs ATen (<= 1 ulp of the time weight)
        np.testing.assert_allclose(mine, ref, rtol=2e-6, atol=1e-7 * max(1.0, np.abs(ref).max()))
        assert st.now_time == g[f"now_after_{b}"]


@pytest.mark.parametrize("name", ["g3g4_readout_d16_L3.npz", "g3g4_readout_d128_L3.npz",
                                  "g3g4_readout_d140_L1.npz"])
def test_readout_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    L = int(g["L"])
    st = O.OracleState(g["P"][0], L, float(g["lam"]), 0.0)
    st.P = [g["P"][i] for i in range(L + 1)]
    raw = O.pair_gram(st, g["u"], g["v"], not_scale=True)
    # raw Gram entries are sums of d signed products: compare with a norm-scaled atol
    R = np.stack([st.P[i][g["u"]] for i in range(L + 1)] + [st.P[i][g["v"]] for i in range(L + 1)], axis=1)
    nrm = np.linalg.norm(R.astype(np.float64), axis=2)
    bound = (1e-5 * nrm[:, :, None] * nrm[:, None, :]).reshape(len(g["u"]), -1) + 1e-12
    assert np.all(np.abs(raw - g["gram_raw"]) <= bound)
    np.testing.assert_allclose(O.pair_gram(st, g["u"], g["v"]), g["gram_scaled"], rtol=1e-5, atol=1e-5)
    feat = O.mlp(O.pair_gram(st, g["u"], g["v"]), g["mlp.0.weight"], g["mlp.0.bias"], g["mlp.2.weight"],
                 g["mlp.2.bias"])
    np.testing.assert_allclose(feat, g["feat_mlp_scaled"], rtol=1e-4, atol=1e-5)
    rows = O.get_random_projections(st, g["u"])
    np.testing.assert_array_equal(np.stack(rows), g["rows_u"])
    # layout facts the kernels rely on: row-major (a, b) -> a*(2L+2)+b, symmetric, id 0 is a legal row
    n = 2 * L + 2
    G = g["gram_raw"].reshape(-1, n, n)
    np.testing.assert_allclose(G, np.swapaxes(G, 1, 2), rtol=1e-6, atol=1e-6)
    assert (g["u"] == 0).any() and (g["u"] == g["v"]).any()


def test_backup_reload_reset(golden_dir):
    g = _load(golden_dir, "g5_backup_reload.npz")
    L, B = int(g["L"]), int(g["B"])
    st = O.OracleState(g["P0"], L, float(g["lam"]), float(g["t0"]))
    upd = lambda b: O.update(st, g["src"][b * B:(b + 1) * B], g["dst"][b * B:(b + 1) * B], g["t"][b * B:(b + 1) * B])
    tol = dict(rtol=2e-6, atol=2e-7 * max(1.0, float(np.abs(g['P_after_6']).max())))
    for b in range(3):
        upd(b)
    bk = st.backup()
    np.testing.assert_allclose(np.stack(bk[1]), g["bk_P"], **tol)
    assert bk[0] == g["bk_now"]
    for b in range(3, 6):
        upd(b)
    np.testing.assert_allclose(np.stack(st.P[1:]), g["P_after_6"], **tol)
    st.reload(bk)
    np.testing.assert_allclose(np.stack(st.P[1:]), g["P_after_reload"], **tol)
    assert st.now_time == g["now_after_reload"]
    upd(3)
    np.testing.assert_allclose(np.stack(st.P[1:]), g["P_after_reload_update"], **tol)
    st.reset()
    assert np.all(np.stack(st.P[1:]) == 0) and np.all(g["P_after_reset"] == 0)
    assert st.now_time == g["now_after_reset"] == g["t0"]
    assert bool(g["P0_changed_by_reset"])          # the reference redraws P[0] on reset (TPNet.py:138-139)


def test_state_dict_contract(golden_dir):
    g = _load(golden_dir, "g5_backup_reload.npz")
    keys = [str(k) for k in g["state_dict_keys"]]
    assert keys == sorted(["begging_time", "now_time", "random_projections.0", "random_projections.1",
                           "random_projections.2", "random_projections.3", "mlp.0.weight", "mlp.0.bias",
                           "mlp.2.weight", "mlp.2.bias"])


# ---------------------------------------------------------------------------------------------------------
# reference-independent known answers
# ---------------------------------------------------------------------------------------------------------
def _toy(seed, N=30, E=120, B=10):
    rng = np.random.RandomState(seed)
    src = rng.randint(0, N, E)
    dst = rng.randint(0, N, E)
    t = np.repeat(np.arange(1, E // B + 1), B).astype(np.float64)   # batch-constant times (notebook cell 10)
    return src, dst, t


@pytest.mark.parametrize("lam", [1e-2, 0.0])
def test_K1_bruteforce_walk_matrices(lam):
    """K1 (notebook cells 2/6/10): explicit-matrix update == brute-force enumeration of temporal walks.
    lam = 0 is K4a: unweighted walk counts (integers)."""
    N, L, B = 30, 3, 10
    src, dst, t = _toy(0)
    st = O.OracleState(np.eye(N, dtype=np.float32), L, lam, 0.0)
    for b in range(0, len(src), B):
        O.update(st, src[b:b + B], dst[b:b + B], t[b:b + B])
    A = O.brute_force_walk_matrices(N, src, dst, t, L, lam, t[-1])
    for k in range(1, L + 1):
        np.testing.assert_allclose(st.P[k], A[k], rtol=1e-5, atol=1e-5)
    if lam == 0.0:
        assert np.all(st.P[3] == np.round(st.P[3]))


def test_K2_projection_estimates_walk_matrix():
    """K2 (notebook cell 8, get_error_ratio): the estimate <h_u^(k), h_v^(0)> = (P[k] P[0]^T)[u, v] of A^(k)[u, v]
    has mean epsilon = |est - A| / (0.5 (||A_u^(k)||^2 + 1)) < 0.2 at d = 50 on the notebook's toy graph."""
    N, L, B, d, E, lam = 100, 3, 10, 50, 500, 1e-4
    rng = np.random.RandomState(5)
    src = rng.randint(0, N, E); dst = rng.randint(0, N, E)
    t = np.repeat(np.arange(1, E // B + 1), B).astype(np.float64)
    ex = O.OracleState(np.eye(N, dtype=np.float32), L, lam, 0.0)
    P0 = (rng.randn(N, d) / np.sqrt(d)).astype(np.float32)
    pr = O.OracleState(P0, L, lam, 0.0)
    for b in range(0, E, B):
        O.update(ex, src[b:b + B], dst[b:b + B], t[b:b + B])
        O.update(pr, src[b:b + B], dst[b:b + B], t[b:b + B])
    for k in range(1, L + 1):
        A = ex.P[k].astype(np.float64)
        est = pr.P[k].astype(np.float64) @ P0.T.astype(np.float64)
        eps = np.abs(est - A) / (0.5 * ((A ** 2).sum(axis=1)[:, None] + 1.0))
        assert eps.mean() < 0.2, (k, eps.mean())


def test_K3_batch_equals_single_edge_updates():
    """K3 (notebook cell 5): with one timestamp per batch and no node repeated inside a batch across hops,
    the batch update equals edge-by-edge updates."""
    N, L = 40, 2
    rng = np.random.RandomState(3)
    P0 = np.eye(N, dtype=np.float32)
    a = O.OracleState(P0, L, 1e-3, 0.0)
    b = O.OracleState(P0, L, 1e-3, 0.0)
    for step in range(1, 9):
        nodes = rng.permutation(N)[:12]          # 6 disjoint edges: no walk of length 2 inside the batch
        src, dst = nodes[:6], nodes[6:]
        t = np.full(6, float(step))
        O.update(a, src, dst, t)
        for e in range(6):
            O.update(b, src[e:e + 1], dst[e:e + 1], t[e:e + 1])
    for k in range(1, L + 1):
        np.testing.assert_allclose(a.P[k], b.P[k], rtol=1e-5, atol=1e-6)


def test_K4_time_rescaling_invariance():
    """K4b: scaling all times by c and lambda by 1/c leaves the state unchanged up to f32 time rounding."""
    N, L, B = 25, 3, 8
    rng = np.random.RandomState(4)
    P0 = (rng.randn(N, 16) / 4).astype(np.float32)
    src = rng.randint(0, N, 64); dst = rng.randint(0, N, 64)
    t = np.sort(rng.uniform(0, 1000.0, 64))
    a = O.OracleState(P0, L, 1e-3, 0.0)
    b = O.OracleState(P0, L, 1e-3 / 4.0, 0.0)
    for s in range(0, 64, B):
        O.update(a, src[s:s + B], dst[s:s + B], t[s:s + B])
        O.update(b, src[s:s + B], dst[s:s + B], 4.0 * t[s:s + B])
    for k in range(1, L + 1):
        np.testing.assert_allclose(a.P[k], b.P[k], rtol=1e-4, atol=1e-6)


def test_oracle_edge_cases():
    st = O.OracleState(np.eye(4, dtype=np.float32), 2, 1e-3, 0.0)
    with pytest.raises(IndexError):
        O.update(st, np.array([], dtype=np.int64), np.array([], dtype=np.int64), np.array([]))
    with pytest.raises(IndexError):
        O.update(st, np.array([4]), np.array([0]), np.array([1.0]))
    # self-pair (u, u): both scatter-adds land on the same row
    O.update(st, np.array([1]), np.array([1]), np.array([1.0]))
    assert st.P[1][1, 1] == 2.0
    # t[-1] (not max(t)) is the new clock (TPNet.py:76)
    O.update(st, np.array([2, 3]), np.array([3, 2]), np.array([5.0, 4.0]))
    assert st.now_time == 4.0


def test_decoder_fixture_consistent(golden_dir):
    """G7: LinkPredictor_v1.forward = fc2(relu(fc1(cat[src_emb, dst_emb, pair_feature])))  (modules.py:112-117);
    pair feature from the oracle + mlp must reproduce the stored logits."""
    g = _load(golden_dir, "g7_decoder.npz")
    L = int(g["L"])
    st = O.OracleState(g["P"][0], L, float(g["lam"]), 0.0)
    st.P = [g["P"][i] for i in range(L + 1)]
    pre = "dec.random_projections.mlp."
    feat = O.mlp(O.pair_gram(st, g["u"], g["v"]), g[pre + "0.weight"], g[pre + "0.bias"], g[pre + "2.weight"],
                 g[pre + "2.bias"])
    x = np.concatenate([g["src_emb"], g["dst_emb"], feat], axis=1)
    h = np.maximum(x @ g["dec.fc1.weight"].T + g["dec.fc1.bias"], 0)
    logits = h @ g["dec.fc2.weight"].T + g["dec.fc2.bias"]
    np.testing.assert_allclose(logits, g["logits"], rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------------------------------------------------
# the torch-CPU port (oracle/torch_port.py: bench.py's cpu_baseline) pinned to the same golden vectors
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g1_update_d16_L3.npz", "g2_update_d120_L2.npz", "g2b_update_d64_L3_fastdecay.npz"])
def test_torch_port_update_trajectory_matches_reference(golden_dir, name):
    """The port issues the reference's ATen op sequence (models/TPNet.py:67-99): bit-identical to its outputs."""
    from oracle.torch_port import TorchPort
    g = _load(golden_dir, name)
    L, B = int(g["L"]), int(g["B"])
    port = TorchPort(g["P0"], L, float(g["lam"]), float(g["t0"]))
    for b in range(len(g["src"]) // B):
        s = slice(b * B, (b + 1) * B)
        port.update(g["src"][s], g["dst"][s], g["t"][s])
        mine = np.stack([p.numpy() for p in port.P[1:]])
        np.testing.assert_array_equal(mine, g[f"P_after_{b}"])
        assert port.now == g[f"now_after_{b}"]


@pytest.mark.parametrize("name", ["g3g4_readout_d16_L3.npz", "g3g4_readout_d128_L3.npz", "g3g4_readout_d140_L1.npz"])
def test_torch_port_readout_matches_reference(golden_dir, name):
    import torch
    from oracle.torch_port import TorchPort
    g = _load(golden_dir, name)
    L = int(g["L"])
    port = TorchPort(g["P"][0], L, float(g["lam"]), 0.0)
    port.P = [torch.from_numpy(np.array(g["P"][i])) for i in range(L + 1)]
    np.testing.assert_array_equal(port.pair_gram(g["u"], g["v"], not_scale=True).numpy(), g["gram_raw"])
    np.testing.assert_array_equal(port.pair_gram(g["u"], g["v"]).numpy(), g["gram_scaled"])


def test_mlp_weight_gradients_fall_back_to_the_torch_expressions_on_the_host():
    """tpnet_amd/fused_feature.weight_grads_f32 serves the module's backward: long lists on a GPU take the matrix-core kernel
    (tested with -m gpu), everything else -- CPU tensors, short lists, a stale prepared entry -- the fp32 torch expressions, which
    must equal autograd's gradients of Linear(64, 256) -> ReLU -> Linear(256, 64) (models/TPNet.py:63-65)."""
    import torch
    from tpnet_amd import fused_feature as ff
    torch.manual_seed(3)
    mlp = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 64))
    x = torch.rand(500, 64) * 7.0
    gy = torch.randn(500, 64)
    got = ff.weight_grads_f32(x, gy, mlp[0].weight.detach(), mlp[0].bias.detach(), mlp[2].weight.detach(), None)
    mlp(x).backward(gy)
    want = [p.grad for p in mlp.parameters()]
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-4 * float(b.abs().max()))
