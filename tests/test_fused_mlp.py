"""f-1: self.mlp fused on the bf16 matrix cores (tpnet_mlp64_bf16).  bf16 operands: tolerance 2e-2, NOT the 1e-4
parity bar of the fp32 path -- which is why it is opt-in."""
import numpy as np
import pytest
import torch


def test_w2_permutation_is_a_permutation_per_tile():
    from tpnet_amd.fused_mlp import permute_w2
    w2 = torch.arange(64 * 256, dtype=torch.float32).reshape(64, 256)
    p = permute_w2(w2)
    assert p.shape == (64, 256)
    for ht in range(8):
        assert sorted(p[0, ht * 32:(ht + 1) * 32].tolist()) == list(range(ht * 32, ht * 32 + 32))
    # position 16 s + 8 h + j  <-  hidden 16 s + 8 (j>>2) + 4 h + (j&3)
    assert p[0, 8 * 1 + 5].item() == 8 * 1 + 4 * 1 + 1      # s=0,h=1,j=5 -> 8 + 4 + 1
    assert p[3, 32 + 16 + 2].item() == 3 * 256 + 32 + 16 + 2


@pytest.mark.gpu
def test_fused_mlp_exact_on_bf16_representable_integers():
    """Asymmetric small-integer weights/inputs: every product and sum is exact in bf16 x bf16 -> fp32, so the kernel
    must equal the fp32 reference bit for bit -- catches any wrong lane/register/permutation mapping."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd.fused_mlp import fused_mlp
    g = torch.Generator().manual_seed(0)
    mlp = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 64)).cuda()
    with torch.no_grad():
        mlp[0].weight.copy_(torch.randint(-3, 4, (256, 64), generator=g).float())
        mlp[0].bias.copy_(torch.randint(-8, 9, (256,), generator=g).float())
        mlp[2].weight.copy_(torch.randint(-2, 3, (64, 256), generator=g).float())
        mlp[2].bias.copy_(torch.randint(-8, 9, (64,), generator=g).float())
    for n in (1, 31, 32, 33, 1000, 4097):
        x = torch.randint(0, 4, (n, 64), generator=g).float().cuda()
        with torch.no_grad():
            want = mlp(x)
            got = fused_mlp(mlp, x)
        assert torch.equal(got, want), f"n={n}: max |delta| {(got - want).abs().max().item()}"


@pytest.mark.gpu
def test_fused_mlp_numerics_and_gradients():
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd.fused_mlp import fused_mlp
    torch.manual_seed(1)
    mlp = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 64)).cuda()
    x = (torch.rand(5000, 64, device="cuda") * 12.0)            # log(relu(G)+1) features: non-negative, O(10)
    want = mlp(x)
    got = fused_mlp(mlp, x)
    err = (got - want).abs().max().item()
    assert err < 2e-2 * max(1.0, want.abs().max().item()), err
    gy = torch.randn_like(want)
    gw = torch.autograd.grad(want, list(mlp.parameters()), gy)
    gf = torch.autograd.grad(got, list(mlp.parameters()), gy)
    # the matrix-core backward differentiates the bf16 forward it belongs to: its ReLU mask is the one of the bf16 hidden
    # layer, which differs from the fp32 one wherever |pre| is below bf16 resolution -- so against fp32 autograd the bar is
    # a norm-wise one (the exact lane / tile mapping is pinned by the integer test below)
    for a, b in zip(gf, gw):
        assert float((a - b).norm()) < 5e-2 * float(b.norm()), (float((a - b).norm()), float(b.norm()))
    import tpnet_amd.fused_mlp as _fm
    _fm.BACKWARD = "torch"                                        # the fp32 expressions behind the same forward: tight
    try:
        gt = torch.autograd.grad(fused_mlp(mlp, x), list(mlp.parameters()), gy)
    finally:
        _fm.BACKWARD = "mfma"
    for a, b in zip(gt, gw):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.gpu
def test_module_uses_fused_mlp_when_asked():
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd import RandomProjectionModule
    rp = RandomProjectionModule(node_num=100, edge_num=500, dim_factor=10, num_layer=3, time_decay_weight=1e-6,
                                device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False,
                                enforce_dim=128).to("cuda:0")
    rng = np.random.RandomState(0)
    src, dst = rng.randint(1, 100, 64), rng.randint(1, 100, 64)
    rp.update(src, dst, np.sort(rng.uniform(0, 1e5, 64)))
    ref = rp.get_pair_wise_feature(src, dst)
    rp.fused_mlp = True
    out = rp.get_pair_wise_feature(src, dst)
    assert (out - ref).abs().max().item() < 2e-2 * max(1.0, ref.abs().max().item())
    out.sum().backward()
    assert rp.mlp[2].weight.grad is not None


@pytest.mark.gpu
def test_config5_end_to_end_bf16_mlp_on_its_own_features():
    """BASELINE config 5 (LastFM shape: N = 1981, d = 512, B = 10 000, lambda = 1e-7) end to end: batches of the C5 stream
    through update(), then get_pair_wise_feature on a whole batch with self.mlp on the bf16 matrix cores
    (fused_mlp = True) against the fp32 path, on the features this config really produces (heavy reuse: log(x+1) features
    reach far beyond the O(10) range of random inputs)."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd import RandomProjectionModule
    from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
    c = CONFIGS["C5"]
    B, nb = c["B"], 4
    E = nb * B
    src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
    neg = synthetic_negatives(c["U"], N, E, B, 1)
    torch.manual_seed(0)
    rp = RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
                                device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False,
                                enforce_dim=c["d"]).to("cuda:0")
    for b in range(nb - 1):
        s = slice(b * B, (b + 1) * B)
        rp.update(src[s], dst[s], t[s])
    s = slice((nb - 1) * B, nb * B)
    with torch.no_grad():
        feats = rp.pair_gram(src[s], dst[s])
        assert float(feats.max()) > 12.0                       # the range the bf16 inputs have to carry at this config
        for u, v in ((src[s], dst[s]), (src[s], neg[s])):
            rp.fused_mlp = False
            want = rp.get_pair_wise_feature(u, v)
            rp.fused_mlp = True
            got = rp.get_pair_wise_feature(u, v)
            assert got.shape == want.shape == (B, 64)
            err = float((got - want).abs().max())
            assert err < 2e-2 * max(1.0, float(want.abs().max())), err
    rp.check_device_errors()


def _again(rp, u, v):
    rp.fused_mlp = True
    try:
        return rp.get_pair_wise_feature(u, v)
    finally:
        rp.fused_mlp = False


@pytest.mark.gpu
@pytest.mark.parametrize("d", [64, 128, 256, 512, 120, 36])
def test_readout_with_mlp_on_the_matrix_cores_in_one_kernel(d):
    """tpnet_pair_feature_bf16 (features formed into LDS, consumed there by the bf16 MFMA layers) against the fp32 path:
    outputs within the bf16 tolerance, run-to-run identical bits, gradients of the four weight tensors (the exact
    lane / register / permutation mapping is pinned by the integer test below)."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd import RandomProjectionModule
    from tpnet_amd import fused_mlp as fm
    rng = np.random.RandomState(d)
    N = 300
    rp = RandomProjectionModule(node_num=N, edge_num=2000, dim_factor=10, num_layer=3, time_decay_weight=1e-6,
                                device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False,
                                enforce_dim=d).to("cuda:0")
    t0 = 0.0
    for _ in range(3):
        src, dst = rng.randint(1, N, 200), rng.randint(1, N, 200)
        t = np.sort(rng.uniform(t0, t0 + 1e5, 200)); t0 = t[-1]
        rp.update(src, dst, t)
    assert fm.readout_supported(rp)
    for n in (1, 31, 32, 33, 1000, 4099):
        u, v = rng.randint(0, N, n).astype(np.int64), rng.randint(0, N, n).astype(np.int64)
        with torch.no_grad():
            rp.fused_mlp = False
            want = rp.get_pair_wise_feature(u, v)
            gram = rp.pair_gram(u, v)
            rp.fused_mlp = True
            got = rp.get_pair_wise_feature(u, v)
            again = rp.get_pair_wise_feature(u, v)
        assert got.shape == (n, 64)
        assert torch.equal(got, again)                                        # fixed summation order over the 8 waves
        err = float((got - want).abs().max())
        assert err < 2e-2 * max(1.0, float(want.abs().max())), (n, err)
    # gradients (grad mode: the kernel also writes the pre-mlp features)
    rp.fused_mlp = True
    u, v = rng.randint(0, N, 500).astype(np.int64), rng.randint(0, N, 500).astype(np.int64)
    got = rp.get_pair_wise_feature(u, v)
    rp.fused_mlp = False
    want = rp.mlp(rp.pair_gram(u, v))
    gy = torch.randn_like(want)
    gw = torch.autograd.grad(want, list(rp.mlp.parameters()), gy)
    gf = torch.autograd.grad(got, list(rp.mlp.parameters()), gy)
    # the matrix-core backward differentiates the bf16 forward it belongs to: its ReLU mask is the one of the bf16 hidden
    # layer, which differs from the fp32 one wherever |pre| is below bf16 resolution -- so against fp32 autograd the bar is
    # a norm-wise one (the exact lane / tile mapping is pinned by the integer test below)
    for a, b in zip(gf, gw):
        assert float((a - b).norm()) < 5e-2 * float(b.norm()), (float((a - b).norm()), float(b.norm()))
    import tpnet_amd.fused_mlp as _fm
    _fm.BACKWARD = "torch"                                        # the fp32 expressions behind the same forward: tight
    try:
        gt = torch.autograd.grad(_again(rp, u, v), list(rp.mlp.parameters()), gy)
    finally:
        _fm.BACKWARD = "mfma"
    for a, b in zip(gt, gw):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-3)
    rp.check_device_errors()


@pytest.mark.gpu
@pytest.mark.parametrize("d", [64, 128, 256, 512, 120, 36])
def test_readout_with_mlp_in_one_kernel_is_exact_on_integers(d):
    """Sparse +-1 projections (raw Gram entries are small integers: not_scale=True), small-integer weights: every product and
    sum is exact in bf16 x bf16 -> fp32, so the fused kernel must equal the fp32 torch layers on the fp32 readout BIT FOR BIT
    -- any wrong lane / register / hidden-tile / permutation mapping shows."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd import RandomProjectionModule
    N = 200
    rp = RandomProjectionModule(node_num=N, edge_num=2000, dim_factor=10, num_layer=3, time_decay_weight=1e-6,
                                device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=True,
                                enforce_dim=d).to("cuda:0")
    g = torch.Generator().manual_seed(d)
    P0 = torch.zeros(N, d)
    for i in range(N):
        idx = torch.randperm(d, generator=g)[:4]
        P0[i, idx] = torch.randint(0, 2, (4,), generator=g).float() * 2 - 1
    rp.random_projections[0].data.copy_(P0.cuda())
    with torch.no_grad():
        rp.mlp[0].weight.copy_(torch.randint(-1, 2, (256, 64), generator=g).float())
        rp.mlp[0].bias.copy_(torch.randint(-8, 9, (256,), generator=g).float())
        rp.mlp[2].weight.copy_(torch.randint(-2, 3, (64, 256), generator=g).float())
        rp.mlp[2].bias.copy_(torch.randint(-8, 9, (64,), generator=g).float())
    rng = np.random.RandomState(d)
    for n in (1, 31, 32, 33, 1000, 4097):
        u, v = rng.randint(0, N, n).astype(np.int64), rng.randint(0, N, n).astype(np.int64)
        with torch.no_grad():
            rp.fused_mlp = False
            want = rp.get_pair_wise_feature(u, v)
            rp.fused_mlp = True
            got = rp.get_pair_wise_feature(u, v)
        assert torch.equal(got, want), f"n={n}: max |delta| {(got - want).abs().max().item()}"


@pytest.mark.gpu
def test_mlp_backward_on_the_matrix_cores_is_exact_on_integers():
    """tpnet_mlp64_bwd_bf16 on small-integer data (every product and sum exact in bf16 x bf16 -> fp32): the four weight
    gradients must equal autograd's through the fp32 torch layers BIT FOR BIT, for pair counts around the 32-pair tile and
    beyond one pass of the grid; run-to-run identical."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from tpnet_amd import fused_mlp as fm
    g = torch.Generator().manual_seed(3)
    mlp = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 64)).cuda()
    with torch.no_grad():
        mlp[0].weight.copy_(torch.randint(-1, 2, (256, 64), generator=g).float())
        mlp[0].bias.copy_(torch.randint(-8, 9, (256,), generator=g).float())
        mlp[2].weight.copy_(torch.randint(-2, 3, (64, 256), generator=g).float())
        mlp[2].bias.copy_(torch.randint(-8, 9, (64,), generator=g).float())
    prep = fm._prepared(mlp)
    for n in (1, 31, 32, 33, 1000, 9000):
        x = torch.randint(0, 3, (n, 64), generator=g).float().cuda()
        x[:, 8:] *= (torch.rand(n, 56, generator=g) < 0.2).float().cuda()        # sparse: |pre| stays far below 256
        gy = torch.randint(-1, 2, (n, 64), generator=g).float().cuda()
        gy *= (torch.rand(n, 64, generator=g) < 0.3).float().cuda()              # |gH| = |sum_o gy W2| <= 64 * 2 * 0.3 ...
        want = torch.autograd.grad(mlp(x), list(mlp.parameters()), gy)
        got = fm.weight_grads(x, gy, mlp[0].weight, mlp[0].bias, mlp[2].weight, prep)
        again = fm.weight_grads(x, gy, mlp[0].weight, mlp[0].bias, mlp[2].weight, prep)
        for a, b, c, name in zip(got, want, again, ("gW1", "gb1", "gW2", "gb2")):
            assert torch.equal(a, c), name
            assert torch.equal(a, b), f"{name}, n={n}: max |delta| {(a - b).abs().max().item()}"


@pytest.mark.gpu
@pytest.mark.parametrize("n", [33, 5000, 80000])
def test_mlp_backward_in_the_fp32_class(n):
    """tpnet_mlp64_bwd_f32 (csrc/mlp_bwd.hip, F32 = true: two-piece bf16 operands, three products, fp32 accumulation; one partial
    per workgroup, summed in a fixed order) against the float64 expressions of the gradients of self.mlp's four tensors
    (models/TPNet.py:64-65): within 4e-5 of the sum of the terms' magnitudes (two-piece operands: 2^-16 per product, in the
    recomputed hidden layer and in the contraction over the rows; observed 3-5e-6), ReLU flips at pre-activations that are zero
    up to rounding allowed for; run-to-run identical bits; through autograd the module's
    long calls take it (fused_feature.weight_grads_f32) and short ones the torch expressions."""
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    import ctypes as C
    from tpnet_amd import _lib, fused_feature as ff
    torch.manual_seed(n)
    mlp = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 64)).cuda()
    x = torch.rand(n, 64, device="cuda") * 9.0
    x[:, 5] = 0.0
    gy = torch.randn(n, 64, device="cuda")
    prep = ff.prepared(mlp, 64)
    lib = _lib.load()
    pf = int(lib.tpnet_mlp64_bwd_partial_floats())
    nblk = min(256, (n + 31) // 32)
    outs = []
    for _ in range(2):
        part = torch.empty((nblk, pf), device="cuda")
        rc = lib.tpnet_mlp64_bwd_f32(x.data_ptr(), gy.data_ptr(), n, prep[2], part.data_ptr(), nblk,
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == nblk
        outs.append(part.sum(0))
    assert torch.equal(outs[0], outs[1])
    tot = outs[0].double()
    gw1, gw2, gb1 = tot[:256 * 64].view(256, 64), tot[256 * 64:2 * 256 * 64].view(64, 256), tot[2 * 256 * 64:]
    w1, b1, w2 = mlp[0].weight.detach().double(), mlp[0].bias.detach().double(), mlp[2].weight.detach().double()
    xd, gd = x.double(), gy.double()
    pre = xd @ w1.t() + b1
    on = (pre > 0).double()
    hid = pre * on
    gh = (gd @ w2) * on
    near = (pre.abs() < 1e-4 * (1.0 + xd.abs().max())).double()           # a flip there moves row `unit` of the first layer's gradients
    gha = (gd @ w2).abs()
    # magnitudes the rounding of the recomputed operands scales with: gH = W2^T gY and H = relu(W1 x + b1) are themselves sums whose
    # error goes with the sum of their terms' magnitudes (a hidden unit that is barely on has a small H and a full-size error)
    ghm = (gd.abs() @ w2.abs()) * on
    hm = (xd.abs() @ w1.abs().t() + b1.abs()) * on
    checks = [(gw1, gh.t() @ xd, ghm.t() @ xd.abs(), (gha * near).t() @ xd.abs()),
              (gb1, gh.sum(0), ghm.sum(0), (gha * near).sum(0)),
              (gw2, gd.t() @ hid, gd.abs().t() @ hm, gd.abs().t() @ (pre.abs() * near))]
    for got, want, mag, amb in checks:
        tol = 4e-5 * mag + amb + 1e-6 * float(want.abs().max())
        assert bool(((got - want).abs() <= tol).all()), float(((got - want).abs() / tol).max())
    # through autograd: mlp_f32 on a long list takes the kernel, and agrees with the torch layers' autograd
    for p in mlp.parameters():
        p.grad = None
    y = ff.mlp_f32(mlp, x)
    y.backward(gy)
    g_mod = [p.grad.clone().double() for p in mlp.parameters()]
    for p in mlp.parameters():
        p.grad = None
    mlp(x).backward(gy)
    g_ref = [p.grad.clone().double() for p in mlp.parameters()]
    zero = torch.zeros(64, device="cuda", dtype=torch.float64)
    mags = [(checks[0][2], checks[0][3]), (checks[1][2], checks[1][3]), (checks[2][2], checks[2][3]), (gd.abs().sum(0), zero)]
    for a, b, (m, amb) in zip(g_mod, g_ref, mags):
        # (both sides may flip a ReLU at a zero pre-activation against the float64 evaluation, each its own way: twice the budget)
        assert bool(((a - b).abs() <= 5e-5 * m + 2 * amb + 1e-5).all()), float(((a - b).abs() / (5e-5 * m + 2 * amb + 1e-5)).max())
