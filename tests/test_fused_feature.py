"""get_pair_wise_feature in one launch (tpnet_pair_feature / tpnet_host_pair_feature: readout + self.mlp fused in fp32) and
the host-array update (tpnet_host_update: one single-workgroup plan kernel + the step kernel), against the separate
kernels, the torch layers and the numpy oracle.  models/TPNet.py:63-65, 67-99, 112-129."""
import ctypes as C

import numpy as np
import pytest
import torch


def _module(N, d, L, lam=1e-6, **kw):
    from tpnet_amd import RandomProjectionModule
    # P[0] and self.mlp are drawn from torch's global generator (models/TPNet.py:49-65): seeded, so that a failing comparison
    # can be reproduced (the round-3 gradient tests drew different weights in every run)
    torch.manual_seed(1234 + 7 * N + 13 * d + L)
    args = dict(node_num=N, edge_num=4 * N, dim_factor=10, num_layer=L, time_decay_weight=lam, device="cuda:0",
                use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=d)
    args.update(kw)
    return RandomProjectionModule(**args).to("cuda:0")


def _stream(rng, N, B, nb, hubs=True):
    out, t0 = [], 0.0
    for _ in range(nb):
        src = rng.randint(1, N, B)
        dst = rng.randint(1, N, B)
        if hubs:
            src[rng.rand(B) < 0.3] = 3                       # a hub: > heavy threshold contributions per batch
            dst[rng.rand(B) < 0.1] = 5
        t = np.sort(rng.uniform(t0, t0 + 2e5, B))
        t0 = t[-1]
        out.append((src.astype(np.int64), dst.astype(np.int64), t))
    return out


def _assert_mlp_grads_close(mlp, gram, gy, got, want):
    """Gradients of self.mlp's four tensors from two implementations of the same forward.  The derivative of ReLU at a
    pre-activation that is zero up to GEMM rounding is either 0 or 1, and the two implementations (torch's Linear; the fused
    forward's backward, which recomputes the hidden layer with another GEMM call) may land on different sides: such a (pair,
    unit) moves row `unit` of the first layer's gradients by at most |dL/dh| * |x|.  That budget is added to the tolerance --
    without it the comparison fails in one run of five to ten, on a different row each time."""
    with torch.no_grad():
        w1, b1, w2 = mlp[0].weight.double(), mlp[0].bias.double(), mlp[2].weight.double()
        x = gram.double()
        pre = x @ w1.t() + b1
        near = (pre.abs() < 1e-4 * (1.0 + x.abs().max())).double()               # [n, 256]
        gh = (gy.double() @ w2).abs() * near                                      # what a flip there moves
        amb_w1 = gh.t() @ x.abs()                                                # [256, 64]
        amb_b1 = gh.sum(0)                                                       # [256]
        # every gradient is a sum over the n rows.  Short lists: fp32 torch GEMMs on both sides -- two evaluations (other blocking,
        # features that differ in their last bits) differ by rounding noise that grows with the sum of the terms' magnitudes, not
        # with the sum itself: 2e-7 of that sum.  From fused_feature.MFMA_BWD_FROM rows the module's backward is ONE matrix-core
        # launch on two-piece bf16 operands (tpnet_mlp64_bwd_f32: 2^-16 per product, also in the recomputed gH = W2^T gY and
        # H = relu(W1 x + b1), whose own terms set the scale): 4e-5 (tests/test_fused_mlp.py::test_mlp_backward_in_the_fp32_class
        # holds the kernel to that against float64, short lists and ragged tiles included)
        on = (pre > 0).double()
        ghm = (gy.double().abs() @ w2.abs()) * on
        hm = (x.abs() @ w1.abs().t() + b1.abs()) * on
        from tpnet_amd import fused_feature as _ffm
        fac = 4e-5 if x.shape[0] >= _ffm.MFMA_BWD_FROM else 2e-7
        noise = [fac * (ghm.t() @ x.abs()), fac * ghm.sum(0), fac * (gy.double().abs().t() @ hm), fac * gy.double().abs().sum(0)]
    extra = [amb_w1, amb_b1, None, None]
    for a, b, e, nz in zip(got, want, extra, noise):
        tol = 1e-4 + 1e-4 * b.abs().double() + nz
        if e is not None:
            tol = tol + e
        assert bool(((a.double() - b.double()).abs() <= tol).all()), float((a.double() - b.double()).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("d,L", [(128, 3), (64, 3), (256, 3), (512, 3), (16, 3), (32, 2), (120, 3), (128, 1), (128, 4),
                                 (256, 4), (30, 3)])
def test_fused_feature_equals_readout_then_torch_mlp(d, L):
    """One launch (readout + both dense layers, fp32 fmaf) vs tpnet_pair_gram followed by the torch layers: the features
    entering the mlp are the same bits, so only the f32 summation order of the two GEMVs differs."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from oracle import tpnet_oracle as O
    rng = np.random.RandomState(d * 7 + L)
    N = 300
    use_matrix = d == 30
    rp = _module(30 if use_matrix else N, d, L, use_matrix=use_matrix)
    N = rp.node_num
    st = O.OracleState(rp.random_projections[0].detach().cpu().numpy(), L, 1e-6, 0.0)
    for src, dst, t in _stream(rng, N, 100, 3):
        rp.update(src, dst, t)
        O.update(st, src, dst, t)
    for n in (1, 7, 8, 9, 64, 333):
        u, v = rng.randint(0, N, n).astype(np.int64), rng.randint(0, N, n).astype(np.int64)
        with torch.no_grad():
            gram = rp.pair_gram(u, v)
            want = rp.mlp(gram)
            got = rp.get_pair_wise_feature(u, v)
        # the pre-mlp features against the oracle (models/TPNet.py:119-128), the rest against the torch layers
        np.testing.assert_allclose(gram.cpu().numpy(), O.pair_gram(st, u, v), rtol=1e-4, atol=1e-4)
        assert got.shape == want.shape
        err = (got - want).abs().max().item()
        assert err <= 2e-5 * max(1.0, want.abs().max().item()), (n, err)


@pytest.mark.gpu
def test_fused_feature_gradients_match_autograd():
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    rng = np.random.RandomState(3)
    rp = _module(200, 128, 3)
    for src, dst, t in _stream(rng, 200, 64, 2):
        rp.update(src, dst, t)
    u, v = rng.randint(0, 200, 500).astype(np.int64), rng.randint(0, 200, 500).astype(np.int64)
    got = rp.get_pair_wise_feature(u, v)                       # grad mode: fused forward + saved pre-mlp features
    assert got.requires_grad
    gram = rp.pair_gram(u, v)
    want = rp.mlp(gram)
    gy = torch.randn_like(want)
    gw = torch.autograd.grad(want, list(rp.mlp.parameters()), gy)
    gf = torch.autograd.grad(got, list(rp.mlp.parameters()), gy)
    _assert_mlp_grads_close(rp.mlp, gram, gy, gf, gw)
    # an optimizer step changes the weights in place: the fused path must pick the new ones up
    with torch.no_grad():
        for p in rp.mlp.parameters():
            p.add_(0.01)
        np.testing.assert_allclose(rp.get_pair_wise_feature(u, v).cpu().numpy(), rp.mlp(rp.pair_gram(u, v)).cpu().numpy(),
                                   rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
def test_non_reference_mlp_keeps_torch():
    """self.mlp replaced by something the fused kernel does not implement (the goldens use Identity): torch applies it."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    rp = _module(100, 64, 3)
    rp.mlp = torch.nn.Identity()
    u = np.arange(1, 50, dtype=np.int64)
    with torch.no_grad():
        assert torch.equal(rp.get_pair_wise_feature(u, u[::-1].copy()), rp.pair_gram(u, u[::-1].copy()))


@pytest.mark.gpu
def test_host_entry_points_check_ids_and_wrap_negatives():
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    N = 120
    rp = _module(N, 64, 3)
    rng = np.random.RandomState(0)
    src, dst, t = _stream(rng, N, 50, 1)[0]
    rp.update(src, dst, t)
    before = [rp.random_projections[i].detach().clone() for i in range(1, 4)]
    bad = src.copy(); bad[7] = N
    with pytest.raises(IndexError):
        rp.update(bad, dst, t + 1e6)
    with pytest.raises(IndexError):
        rp.get_pair_wise_feature(bad, dst)
    with pytest.raises(IndexError):
        rp.pair_gram(src, np.full_like(dst, -N - 1))
    for i in range(3):                                          # a rejected call changes nothing
        assert torch.equal(rp.random_projections[i + 1].detach(), before[i])
    rp.check_device_errors()
    with torch.no_grad():                                       # python-style negative ids index from the end, as ATen does
        a = rp.get_pair_wise_feature(src - N, dst)
        b = rp.get_pair_wise_feature(src, dst)
    assert torch.equal(a, b)
    rp2 = _module(N, 64, 3)
    rp2.random_projections[0].data.copy_(rp.random_projections[0].data)
    rp2.update(src - N, dst - N, t)
    for i in range(1, 4):
        assert torch.equal(rp2.random_projections[i].detach(), before[i - 1])


@pytest.mark.gpu
@pytest.mark.parametrize("d,L,B", [(128, 3, 1000), (64, 3, 200), (256, 2, 2048), (16, 3, 700), (512, 3, 300), (120, 3, 257),
                                   (128, 4, 64), (128, 3, 1)])
def test_single_batch_plan_equals_chunk_planner_and_oracle(d, L, B):
    """update() from host arrays (tpnet_host_update: k_plan_one) and from device tensors (tpnet_update: k_plan_one) against
    run_stream on the per-batch schedule (chunk planner: keys + device sort + k_finish): identical item contents, so
    identical bits; and against the numpy oracle (models/TPNet.py:67-99)."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from oracle import tpnet_oracle as O
    rng = np.random.RandomState(B + d)
    N = 400
    rp_h, rp_d, rp_s = _module(N, d, L), _module(N, d, L), _module(N, d, L)
    P0 = rp_h.random_projections[0].data
    rp_d.random_projections[0].data.copy_(P0)
    rp_s.random_projections[0].data.copy_(P0)
    st = O.OracleState(P0.cpu().numpy(), L, 1e-6, 0.0)
    batches = _stream(rng, N, B, 4)
    for src, dst, t in batches:
        rp_h.update(src, dst, t)
        rp_d.update(torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda(), t)
        O.update(st, src, dst, t)
    cat = lambda k: torch.from_numpy(np.concatenate([b[k] for b in batches])).cuda()
    rp_s.run_stream(cat(0), cat(1), None, cat(2), B, want_pos=False, want_neg=False, schedule="batch")
    for i in range(1, L + 1):
        h = rp_h.random_projections[i].detach()
        assert torch.equal(h, rp_d.random_projections[i].detach())
        assert torch.equal(h, rp_s.random_projections[i].detach())
        np.testing.assert_allclose(h.cpu().numpy(), st.P[i], rtol=1e-4, atol=1e-6 * max(1.0, float(np.abs(st.P[i]).max())))
    assert float(rp_h.now_time) == batches[-1][2][-1]
    rp_h.check_device_errors()


@pytest.mark.gpu
def test_staging_ring_wraps_under_load():
    """More calls in flight than the ring has slots, results consumed late: every call must have read ITS ids."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    N = 500
    rp = _module(N, 128, 3)
    rng = np.random.RandomState(1)
    for src, dst, t in _stream(rng, N, 300, 2):
        rp.update(src, dst, t)
    pairs = [(rng.randint(0, N, 777).astype(np.int64), rng.randint(0, N, 777).astype(np.int64)) for _ in range(40)]
    with torch.no_grad():
        outs = [rp.pair_gram(u, v) for u, v in pairs]           # 40 launches enqueued back to back, ring of 8 slots
        torch.cuda.synchronize()
        for (u, v), o in zip(pairs, outs):
            assert torch.equal(o, rp.pair_gram(torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()))


@pytest.mark.gpu
@pytest.mark.parametrize("d,L,K", [(256, 3, 20), (128, 3, 20), (64, 3, 7), (512, 3, 10), (128, 2, 4), (256, 4, 5), (128, 1, 32)])
def test_anchored_encoder_readout(d, L, K):
    """tpnet_pair_gram_anchored (one lane group per row, anchors in registers) against the oracle on the reference's pair
    list get_pair_wise_feature(tile(neigh, 2), concat(repeat(src, K), repeat(dst, K))) (models/TPNet.py:311-321), incl.
    padding id 0 among the neighbours, rows whose two anchors coincide, and equal anchors in consecutive rows."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from oracle import tpnet_oracle as O
    rng = np.random.RandomState(K * 31 + d)
    N = 250
    rp = _module(N, d, L)
    st = O.OracleState(rp.random_projections[0].detach().cpu().numpy(), L, 1e-6, 0.0)
    for src, dst, t in _stream(rng, N, 120, 3):
        rp.update(src, dst, t)
        O.update(st, src, dst, t)
    n = 37
    neigh = rng.randint(0, N, (n, K)).astype(np.int64)
    neigh[rng.rand(n, K) < 0.2] = 0                          # left zero-padding of short histories (utils/utils.py:211-219)
    a1 = rng.randint(1, N, n).astype(np.int64)
    a2 = rng.randint(1, N, n).astype(np.int64)
    a2[3] = a1[3]
    a1[10] = a1[9]; a1[11] = a1[9]                           # runs of equal anchors across rows
    u = np.tile(neigh.reshape(-1), 2)
    v = np.concatenate([np.repeat(a1, K), np.repeat(a2, K)])
    want = O.pair_gram(st, u, v)
    got = rp.pair_gram_anchored(neigh, a1, a2).view(-1, rp.pair_wise_feature_dim).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
    # and equal to the generic kernel far below the parity bar (same products, another order of the partial sums)
    gen = rp.pair_gram(torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, gen, rtol=2e-5, atol=2e-5)
    rp.check_device_errors()


@pytest.mark.gpu
def test_encoder_pattern_takes_the_anchored_kernel_and_matches():
    """get_pair_wise_feature recognises the encoder's index pattern on a long pair list and must return what the generic
    path returns for the same arrays."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    rng = np.random.RandomState(5)
    N, K, n = 400, 20, 300
    rp = _module(N, 256, 3)
    for src, dst, t in _stream(rng, N, 200, 3):
        rp.update(src, dst, t)
    neigh = rng.randint(0, N, (n, K)).astype(np.int64)
    a1, a2 = rng.randint(1, N, n).astype(np.int64), rng.randint(1, N, n).astype(np.int64)
    u = np.tile(neigh.reshape(-1), 2)
    v = np.concatenate([np.repeat(a1, K), np.repeat(a2, K)])
    assert u.size > 8192
    runs = rp._anchor_runs(v[: n * K])
    assert runs is not None and runs[1] >= K and runs[1] % K == 0     # runs of equal anchors can only merge whole blocks of K
    with torch.no_grad():
        got = rp.get_pair_wise_feature(u, v)
        want = rp.mlp(rp.pair_gram(torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()))
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-4)
    # a pair list that only LOOKS tiled (second half of dst not made of runs) keeps the generic path and is right too
    v2 = v.copy(); v2[n * K + 1] = (v2[n * K + 1] % (N - 1)) + 1
    with torch.no_grad():
        got2 = rp.get_pair_wise_feature(u, v2)
        want2 = rp.mlp(rp.pair_gram(torch.from_numpy(u).cuda(), torch.from_numpy(v2).cuda()))
    np.testing.assert_allclose(got2.cpu().numpy(), want2.cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("d", [64, 128, 256, 512, 120, 36])
def test_feature_on_the_fp32_matrix_cores(d):
    """From 2 048 pairs get_pair_wise_feature is ONE launch with self.mlp on the fp32 matrix cores (v_mfma_f32_32x32x2_f32:
    fp32 products and sums, so it stays in the fp32 parity class): against the readout + torch layers on host ids (staged),
    on a list longer than the staging slot (device copy of the ids), exact on integer data, gradients, and the dense layers
    alone (tpnet_mlp64_f32) behind a separate readout."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd import fused_feature as ff
    rng = np.random.RandomState(d + 1)
    N = 300
    rp = _module(N, d, 3)
    for src, dst, t in _stream(rng, N, 150, 3):
        rp.update(src, dst, t)
    for n in (2048, 3001, 16384, 20011):
        u, v = rng.randint(0, N, n).astype(np.int64), rng.randint(0, N, n).astype(np.int64)
        with torch.no_grad():
            gram = rp.pair_gram(torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda())
            want = rp.mlp(gram)
            got = rp.get_pair_wise_feature(u, v)
            alone = ff.mlp_f32(rp.mlp, gram)
        assert got.shape == want.shape and alone is not None
        for x in (got, alone):
            err = (x - want).abs().max().item()
            assert err <= 2e-5 * max(1.0, want.abs().max().item()), (n, err)
    # gradients through the one-launch forward (the kernel also writes the pre-mlp features)
    u, v = rng.randint(0, N, 2500).astype(np.int64), rng.randint(0, N, 2500).astype(np.int64)
    got = rp.get_pair_wise_feature(u, v)
    gram = rp.pair_gram(u, v)
    want = rp.mlp(gram)
    gy = torch.randn_like(want)
    _assert_mlp_grads_close(rp.mlp, gram, gy, torch.autograd.grad(got, list(rp.mlp.parameters()), gy),
                            torch.autograd.grad(want, list(rp.mlp.parameters()), gy))
    # integer data: fp32 products and sums are exact whatever their order
    rq = _module(N, d, 3, not_scale=True)
    g = torch.Generator().manual_seed(d)
    P0 = torch.zeros(N, d)
    for i in range(N):
        idx = torch.randperm(d, generator=g)[:4]
        P0[i, idx] = torch.randint(0, 2, (4,), generator=g).float() * 2 - 1
    rq.random_projections[0].data.copy_(P0.cuda())
    with torch.no_grad():
        rq.mlp[0].weight.copy_(torch.randint(-3, 4, (256, 64), generator=g).float())
        rq.mlp[0].bias.copy_(torch.randint(-8, 9, (256,), generator=g).float())
        rq.mlp[2].weight.copy_(torch.randint(-2, 3, (64, 256), generator=g).float())
        rq.mlp[2].bias.copy_(torch.randint(-8, 9, (64,), generator=g).float())
        u, v = rng.randint(0, N, 4099).astype(np.int64), rng.randint(0, N, 4099).astype(np.int64)
        assert torch.equal(rq.get_pair_wise_feature(u, v), rq.mlp(rq.pair_gram(u, v)))
    rp.check_device_errors()


@pytest.mark.gpu
@pytest.mark.parametrize("d,L,K", [(128, 3, 20), (256, 3, 7), (64, 2, 4)])
def test_encoder_call_with_device_resident_ids(d, L, K):
    """encoder_pair_features (row set-up + device sampler + anchored readout in one call, self.mlp behind it) against the
    reference's own sequence on host arrays: NeighborSampler('recent').get_historical_neighbors(concat(src, other), tile(t, 2))
    then get_pair_wise_feature(tile(neigh, 2), concat(repeat(tile(src, 2), K), repeat(tile(other, 2), K)))
    (models/TPNet.py:280-316) -- same neighbours, same features, same row order."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd.callers import RecentNeighborSampler, encoder_pair_indices
    from tpnet_amd.sampler import GpuRecentNeighborSampler
    rng = np.random.RandomState(d + K)
    N, E, B = 300, 900, 60
    src = rng.randint(1, N, E).astype(np.int64)
    dst = rng.randint(1, N, E).astype(np.int64)
    t = np.sort(rng.uniform(1.0e6, 1.4e6, E))
    rp = _module(N, d, L)
    host = RecentNeighborSampler(src, dst, t)
    gpu = GpuRecentNeighborSampler(src, dst, t, device="cuda:0", num_nodes=N)
    for b0 in range(0, E - B, B):
        s = slice(b0, b0 + B)
        if b0 >= 5 * B:
            other = rng.randint(1, N, B).astype(np.int64)
            neigh_h, _, _ = host.get_historical_neighbors(np.concatenate([src[s], other]), np.tile(t[s], 2), K)
            u, v = encoder_pair_indices(neigh_h, src[s], other)
            want = rp.get_pair_wise_feature(u, v)
            dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
            got, neigh_d = rp.encoder_pair_features(gpu, dev(src[s]), dev(other), dev(t[s]), K)
            np.testing.assert_array_equal(neigh_d.cpu().numpy(), neigh_h)
            np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().cpu().numpy(), rtol=2e-4, atol=2e-4)
            # the batch's arrays handed over as the reference's loop holds them (host numpy slices, staged through the pinned
            # ring, no copy enqueued): the same kernels on the same ids -> the same bits
            got_h, neigh_dh = rp.encoder_pair_features(gpu, src[s], other, t[s], K)
            np.testing.assert_array_equal(neigh_dh.cpu().numpy(), neigh_h)
            np.testing.assert_array_equal(got_h.detach().cpu().numpy(), got.detach().cpu().numpy())
        rp.update(src[s], dst[s], t[s])
    rp.check_device_errors()
    with pytest.raises(IndexError):
        rp.encoder_pair_features(gpu, np.array([1, N], dtype=np.int64), np.array([1, 2], dtype=np.int64), np.array([2e6, 2e6]), K)


@pytest.mark.gpu
def test_staging_ring_wraps_without_reusing_a_slot_in_flight():
    """The host-array calls carve their arrays from the slot of the pinned ring being filled and record ONE event when a slot is
    left (host_api.hip): 700 calls of mixed sizes walk the 8 slots several times over; every call's result must be the one its
    own ids give (a slot handed out again before the kernel that read it finished would show as another call's features)."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    rng = np.random.RandomState(11)
    N = 500
    rp = _module(N, 64, 3)
    for src, dst, t in _stream(rng, N, 200, 3, hubs=False):
        rp.update(src, dst, t)
    sizes = [1000, 7, 1024, 333, 2048, 64]
    calls = []
    with torch.no_grad():
        for i in range(700):
            n = sizes[i % len(sizes)]
            u, v = rng.randint(0, N, n).astype(np.int64), rng.randint(0, N, n).astype(np.int64)
            calls.append((u, v, rp.pair_gram(u, v)))                       # enqueued back to back: no synchronisation in between
        torch.cuda.synchronize()
        for i in range(0, 700, 7):
            u, v, got = calls[i]
            want = rp.pair_gram(torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda())      # ids resident on the device: no ring
            assert torch.equal(got, want), i
    rp.check_device_errors()


@pytest.mark.gpu
@pytest.mark.parametrize("d,K,m", [(128, 20, 250), (64, 7, 700), (256, 20, 220), (120, 20, 250)])
def test_encoder_pattern_on_host_arrays_is_recognised(d, K, m):
    """The reference's encoder calls get_pair_wise_feature(tile(neigh, 2), concat(repeat(a1, K), repeat(a2, K))) on host arrays
    (models/TPNet.py:311-316).  The module recognises the pattern in C (tpnet_host_encoder_pattern), ships neigh + the anchors
    only and runs the anchored readout + dense layers as one call: same features as the general path on the same pairs --
    consecutive equal anchors (runs longer than K), a list that is NOT the pattern, gradients, ids out of range."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd import _lib
    rng = np.random.RandomState(d + K)
    N = 400
    rp = _module(N, d, 3)
    for src, dst, t in _stream(rng, N, 120, 3):
        rp.update(src, dst, t)
    neigh = rng.randint(0, N, (m, K)).astype(np.int64)
    a1, a2 = rng.randint(1, N, m).astype(np.int64), rng.randint(1, N, m).astype(np.int64)
    a1[5] = a1[4]                                               # two rows with the same anchor: a run of 2 K
    a2[9:12] = 7
    u = np.tile(neigh.reshape(-1), 2)
    v = np.concatenate([np.repeat(a1, K), np.repeat(a2, K)])
    n = u.size
    lib = _lib.load()
    assert lib.tpnet_host_encoder_pattern(u.ctypes.data, v.ctypes.data, n, N) % K == 0          # (K or a multiple that still fits)
    with torch.no_grad():
        got = rp.get_pair_wise_feature(u, v)
        gram = rp.pair_gram(torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda())               # the general kernel, device ids
        want = rp.mlp(gram)
    assert got.shape == want.shape
    err = (got - want).abs().max().item()
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), err
    # not the pattern: one neighbour of the second half changed -> the general path, same result as the torch layers
    u2 = u.copy()
    u2[n // 2 + 3] = (u2[n // 2 + 3] + 1) % N
    assert lib.tpnet_host_encoder_pattern(u2.ctypes.data, v.ctypes.data, n, N) == 0
    with torch.no_grad():
        got2 = rp.get_pair_wise_feature(u2, v)
        want2 = rp.mlp(rp.pair_gram(torch.from_numpy(u2).cuda(), torch.from_numpy(v).cuda()))
    assert (got2 - want2).abs().max().item() <= 2e-5 * max(1.0, want2.abs().max().item())
    # gradients through the one-call path
    got = rp.get_pair_wise_feature(u, v)
    assert got.requires_grad
    gram = rp.pair_gram(u, v)
    want = rp.mlp(gram)
    gy = torch.randn_like(want)
    _assert_mlp_grads_close(rp.mlp, gram, gy, torch.autograd.grad(got, list(rp.mlp.parameters()), gy),
                            torch.autograd.grad(want, list(rp.mlp.parameters()), gy))
    # python-style negative ids (ATen indexing, as the reference's P[i][ids]): the one-call path declines, the general path wraps
    un = u.copy()
    un[0] = un[n // 2] = -1
    uw = u.copy()
    uw[0] = uw[n // 2] = N - 1
    with torch.no_grad():
        assert torch.equal(rp.get_pair_wise_feature(un, v), rp.get_pair_wise_feature(uw, v))
    # an id out of range is still an IndexError (the detector declines, the general path reports)
    bad = v.copy()
    bad[3] = N
    with pytest.raises(IndexError):
        rp.get_pair_wise_feature(u, bad)
    rp.check_device_errors()


@pytest.mark.gpu
def test_mlp_prepare_layouts_follow_the_parameters():
    """tpnet_mlp_prepare: the derived layouts of self.mlp in one launch (what a training loop pays after every optimizer step)
    equal the torch expressions they replace -- two transposes and the gathered w2f of include/tpnet_hip.h -- and follow an
    in-place update of the Parameters (the cache key is (data_ptr, _version): an optimizer step bumps the version)."""
    from tpnet_amd import fused_feature as ff
    rp = _module(50, 128, 3)
    mlp = rp.mlp
    prep = ff.prepared(mlp, 64)
    w1t, w2t, w2f, img = prep[3]
    torch.cuda.synchronize()
    rows, cols = ff._w2f_index(mlp[2].weight.device)
    assert torch.equal(w1t, mlp[0].weight.detach().t()) and torch.equal(w2t, mlp[2].weight.detach().t())
    assert torch.equal(w2f, mlp[2].weight.detach()[rows, cols])
    assert prep[1].w1 == mlp[0].weight.data_ptr() and prep[1].b1 == mlp[0].bias.data_ptr()
    with torch.no_grad():
        for p in mlp.parameters():
            p.add_(0.25)                                  # (what an optimizer step does: in place, _version bumped)
    prep2 = ff.prepared(mlp, 64)
    torch.cuda.synchronize()
    assert prep2[3][0].data_ptr() == w1t.data_ptr()        # same buffers, rewritten

    # the weight image of the encoder's one-launch kernel (tpnet_mlp_prepare_image, include/tpnet_hip.h): hi + lo of every element at
    # the documented operand position reproduces the Parameter to 2^-16, biases behind the four matrices
    def bf16(a):
        return (a.astype(np.uint32) << 16).view(np.float32)
    raw = prep2[3][3].cpu().numpy()
    w1 = mlp[0].weight.detach().cpu().numpy(); w2 = mlp[2].weight.detach().cpu().numpy()
    parts = [bf16(raw[o:o + 32768].view(np.uint16)).reshape(-1, 64, 8) for o in (0, 32768, 65536, 98304)]
    lane = np.arange(64); j = np.arange(8)
    for w in range(16):
        for s_ in range(2):
            want = w1[(16 * w + (lane & 15))[:, None], (32 * s_ + 8 * (lane >> 4))[:, None] + j[None, :]]
            got = parts[0][w * 2 + s_] + parts[1][w * 2 + s_]
            assert np.all(np.abs(got - want) <= 2.0 ** -15 * np.abs(want))
    for k2 in range(8):
        for t_ in range(4):
            hid = 16 * (2 * k2 + (j >> 2))[None, :] + 4 * (lane >> 4)[:, None] + (j & 3)[None, :]
            want = w2[(16 * t_ + (lane & 15))[:, None], hid]
            got = parts[2][k2 * 4 + t_] + parts[3][k2 * 4 + t_]
            assert np.all(np.abs(got - want) <= 2.0 ** -15 * np.abs(want))
    assert np.array_equal(raw[131072:131072 + 1024].view(np.float32), mlp[0].bias.detach().cpu().numpy())
    assert np.array_equal(raw[132096:132096 + 256].view(np.float32), mlp[2].bias.detach().cpu().numpy())
    assert torch.equal(prep2[3][0], mlp[0].weight.detach().t()) and torch.equal(prep2[3][2], mlp[2].weight.detach()[rows, cols])
    # and the fused forward uses them: against the torch layers on the same features
    rng = np.random.RandomState(0)
    u, v = rng.randint(1, 50, 300), rng.randint(1, 50, 300)
    with torch.no_grad():
        got = rp.get_pair_wise_feature(u, v)
        want = mlp(rp.pair_gram(u, v))
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("d,K,n", [(128, 20, 37), (128, 4, 9), (128, 5, 13), (128, 33, 3), (64, 20, 11), (64, 7, 50), (128, 20, 2000),
                                   (128, 16, 1), (128, 6, 1)])
def test_encoder_readout_on_the_matrix_cores(d, K, n):
    """csrc/encoder_mfma.hip (tpnet_pair_gram_anchored on rows of 64 / 128 floats, L = 3, K >= 4: 16 x 16 x 32 bf16 products on
    operands split into three bf16 pieces) against the oracle on the reference's pair list (models/TPNet.py:311-321) and against
    the vector-ALU walk: lists whose length is no multiple of 4 or 16 slots, tiles that straddle two rows, padding id 0,
    coinciding anchors, raw Gram entries (not_scale) inside 1e-6 |R_a| |R_b| (SURVEY Appendix C asks 1e-4)."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from oracle import tpnet_oracle as O
    rng = np.random.RandomState(K * 131 + d + n)
    N, L = 260, 3
    for not_scale in (False, True):
        rp = _module(N, d, L, not_scale=not_scale)
        st = O.OracleState(rp.random_projections[0].detach().cpu().numpy(), L, 1e-6, 0.0)
        for src, dst, t in _stream(rng, N, 150, 3):
            rp.update(src, dst, t)
            O.update(st, src, dst, t)
        neigh = rng.randint(0, N, (n, K)).astype(np.int64)
        neigh[rng.rand(n, K) < 0.2] = 0
        a1 = rng.randint(1, N, n).astype(np.int64)
        a2 = rng.randint(1, N, n).astype(np.int64)
        a2[n // 2] = a1[n // 2]
        u = np.tile(neigh.reshape(-1), 2)
        v = np.concatenate([np.repeat(a1, K), np.repeat(a2, K)])
        got = rp.pair_gram_anchored(neigh, a1, a2).view(-1, 64).cpu().numpy()
        valu = rp.pair_gram_anchored(neigh, a1, a2, matrix_cores=False).view(-1, 64).cpu().numpy()
        if not_scale:
            # raw entries against the float64 Gram: |delta| <= 1e-6 |R_a| |R_b| + the fp32 rounding of the entry itself (the
            # vector-ALU walk holds the same bound)
            want = O.pair_gram(st, u, v, not_scale=True, accumulate=np.float64).reshape(-1, 8, 8)
            diag = np.sqrt(np.abs(want[:, np.arange(8), np.arange(8)]))                              # |R_a| per row of the pair
            bound = 1e-6 * diag[:, :, None] * diag[:, None, :] + 2e-7 * np.abs(want)
            assert np.all(np.abs(got.reshape(-1, 8, 8) - want) <= bound), float(np.max(np.abs(got.reshape(-1, 8, 8) - want) / (bound + 1e-30)))
            assert np.all(np.abs(valu.reshape(-1, 8, 8) - want) <= bound)
        else:
            want = O.pair_gram(st, u, v)
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(got, valu, rtol=2e-5, atol=2e-6)
        rp.check_device_errors()
    # ids out of range (device-resident ids are checked by the kernel): NaN features for the pairs they are in, counted
    rp = _module(N, d, L)
    neigh = rng.randint(1, N, (n, K)).astype(np.int64)
    a1 = rng.randint(1, N, n).astype(np.int64)
    a2 = rng.randint(1, N, n).astype(np.int64)
    neigh[0, K - 1] = N + 5
    bad_row = n - 1
    a2[bad_row] = -3
    dev = lambda x: torch.from_numpy(x).cuda()
    got = rp.pair_gram_anchored(dev(neigh), dev(a1), dev(a2)).cpu().numpy()              # [2, n*K, 64]
    nanrow = np.isnan(got).all(axis=2)
    exp = np.zeros((2, n * K), dtype=bool)
    exp[:, K - 1] = True
    exp[:, bad_row * K:(bad_row + 1) * K] = True
    assert np.array_equal(nanrow, exp)
    assert not np.isnan(got[~exp]).any()
    with pytest.raises(IndexError):
        rp.check_device_errors()


@pytest.mark.gpu
@pytest.mark.parametrize("d,K,n", [(128, 20, 200), (128, 5, 13), (64, 8, 77), (128, 4, 1), (128, 7, 3), (128, 20, 2000)])
def test_encoder_readout_and_dense_layers_in_one_launch(d, K, n):
    """The encoder's call (models/TPNet.py:311-324, 129) as ONE launch on the matrix cores (csrc/encoder_mfma.hip, MLP = true:
    Gram tiles, then self.mlp on their 16 feature rows in the same wave, weights from the prepared image in LDS) against the torch
    fp32 layers on the vector-ALU readout's features: fp32 class (<= 2e-5 of the output scale), wave shares with an odd number
    of tiles, lists shorter than one tile pair; with gradients recorded the pre-mlp features are written too, the outputs keep
    their bits and the gradients of self.mlp's four tensors match autograd on the torch layers."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd import _lib
    rng = np.random.RandomState(K * 17 + d + n)
    N = 300
    rp = _module(N, d, 3)
    for src, dst, t in _stream(rng, N, 150, 3):
        rp.update(src, dst, t)
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    neigh = rng.randint(0, N, (n, K)).astype(np.int64)
    neigh[rng.rand(n, K) < 0.2] = 0
    a1, a2 = rng.randint(1, N, n).astype(np.int64), rng.randint(1, N, n).astype(np.int64)
    prep = rp._overlapped_mlp()
    assert prep is not None and _lib.load().tpnet_encoder_fused_supported(rp._st_ref(), n, K, prep[2]) == 1
    with torch.no_grad():
        got = rp.get_pair_wise_feature_anchored(dev(neigh), dev(a1), dev(a2))
        feats = rp.pair_gram_anchored(neigh, a1, a2, matrix_cores=False).view(-1, 64)
        want = rp.mlp(feats)
    scale = float(want.abs().max())
    err = (got - want).abs()
    assert bool((err <= 2e-5 * scale + 1e-4 * want.abs()).all()), (float(err.max()), scale)
    # with gradients recorded: same outputs, pre-mlp features kept for the backward pass
    got2 = rp.get_pair_wise_feature_anchored(dev(neigh), dev(a1), dev(a2))
    assert got2.requires_grad and torch.equal(got2.detach(), got)
    gy = torch.from_numpy(rng.randn(*got.shape).astype(np.float32)).cuda()
    got2.backward(gy)
    grads = [p.grad.clone() for p in rp.mlp.parameters()]
    for p in rp.mlp.parameters():
        p.grad = None
    ref = rp.mlp(feats)
    ref.backward(gy)
    want_g = [p.grad.clone() for p in rp.mlp.parameters()]
    _assert_mlp_grads_close(rp.mlp, feats, gy, grads, want_g)
    rp.check_device_errors()


@pytest.mark.gpu
def test_one_launch_encoder_call_on_random_shapes():
    """k_encoder_fused over forty random (rows, K, d) shapes -- shares of one tile and of many, odd tile counts (a half-empty last
    pass of the dense layers), producers without tiles, K from 4 to 45, rows of 64 and 128 floats -- against the two-launch path
    (vector-ALU walk + torch layers): the hand-off between the producer and consumer waves must neither lose nor repeat a tile."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    rng = np.random.RandomState(2024)
    N = 500
    mods = {}
    for d in (64, 128):
        rp = _module(N, d, 3)
        for src, dst, t in _stream(rng, N, 200, 2):
            rp.update(src, dst, t)
        mods[d] = rp
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    for trial in range(40):
        d = (64, 128)[trial % 2]
        rp = mods[d]
        K = int(rng.randint(4, 46))
        n = int(rng.choice([1, 2, 3, 5, 17, 64, 257, 1000, 3000]))
        neigh = rng.randint(0, N, (n, K)).astype(np.int64)
        a1, a2 = rng.randint(1, N, n).astype(np.int64), rng.randint(1, N, n).astype(np.int64)
        with torch.no_grad():
            got = rp.get_pair_wise_feature_anchored(dev(neigh), dev(a1), dev(a2))
            want = rp.mlp(rp.pair_gram_anchored(neigh, a1, a2, matrix_cores=False).view(-1, 64))
        scale = float(want.abs().max())
        err = float((got - want).abs().max())
        assert err <= 2e-5 * scale + 1e-6, (trial, d, K, n, err, scale)
    for rp in mods.values():
        rp.check_device_errors()
