"""LinkPredictor_v1 on the bf16 matrix cores (tpnet_amd/fused_decoder.py, C ABI tpnet_decoder_bf16).

CPU tier: the weight packing (input axis [src | pad | dst | pad | feature], hidden axis padded to whole tiles) emulated in
torch equals fc2(relu(fc1(concat))).  GPU tier: the fused forward against the fp32 layers (exact on bf16-representable
integer data, bf16 tolerance otherwise), fp32 gradients for weights AND inputs, not_encode mode, decoders without a
pairwise feature, ragged batch sizes."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tpnet_amd import fused_decoder as fd  # noqa: E402
from tpnet_amd.callers import LinkPredictor_v1  # noqa: E402


def _bf16_exact_(lin, rng, lo=-4, hi=5):
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(rng.randint(lo, hi, tuple(lin.weight.shape)).astype(np.float32)) / 4.0)
        lin.bias.copy_(torch.from_numpy(rng.randint(lo, hi, tuple(lin.bias.shape)).astype(np.float32)) / 2.0)


@pytest.mark.parametrize("D,F,H", [(172, 64, 172), (64, 64, 100), (8, 0, 40), (0, 64, 33), (20, 16, 256)])
def test_pack_weights_emulation(D, F, H):
    rng = np.random.RandomState(D + F + H)
    fc1, fc2 = torch.nn.Linear(2 * D + F, H), torch.nn.Linear(H, 1)
    _bf16_exact_(fc1, rng); _bf16_exact_(fc2, rng)
    assert fd.supported(fc1, fc2, D, F)
    w1p, b1p, w2p, b2, HT = fd.pack_weights(fc1, fc2, D, F)
    DP = (D + 15) // 16 * 16
    assert tuple(w1p.shape) == (32 * HT, 2 * DP + F) and HT == (H + 31) // 32
    x = torch.from_numpy(rng.randint(-3, 4, (37, 2 * D + F)).astype(np.float32))
    xp = torch.zeros(37, 2 * DP + F)
    xp[:, :D] = x[:, :D]; xp[:, DP:DP + D] = x[:, D:2 * D]
    if F:
        xp[:, 2 * DP:] = x[:, 2 * D:]
    with torch.no_grad():
        ref = fc2(torch.relu(fc1(x)))
        emu = (torch.relu(xp @ w1p.float().t() + b1p) * w2p).sum(1, keepdim=True) + b2
    assert torch.equal(ref, emu)
    assert not fd.supported(fc1, torch.nn.Linear(H, 2), D, F)             # one logit only


# ---------------------------------------------------------------------------------------------------------
class _FeatStub(torch.nn.Module):
    """stands in for RandomProjectionModule: hands out a fixed feature matrix that carries a gradient"""
    pair_wise_feature_dim = 64

    def __init__(self, feat):
        super().__init__()
        self.feat = torch.nn.Parameter(feat)

    def get_pair_wise_feature(self, src_node_ids, dst_node_ids):
        return self.feat[: len(src_node_ids)]


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU (the product has no CPU fallback)")


@pytest.mark.gpu
@pytest.mark.parametrize("D,H,n,with_feat,not_encode", [(172, 172, 1000, True, False), (172, 172, 77, True, True),
                                                         (64, 100, 333, True, False), (8, 40, 50, False, False),
                                                         (128, 256, 2000, True, False), (20, 64, 31, True, False)])
def test_fused_decoder_matches_fp32_layers(D, H, n, with_feat, not_encode):
    _need_gpu()
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(D + H + n)
    ids = np.arange(n)
    feat = torch.from_numpy(rng.randint(0, 5, (n, 64)).astype(np.float32)).to(dev)
    rp = _FeatStub(feat).to(dev) if with_feat else None
    dec = LinkPredictor_v1(D, D, H, 1, rp, not_encode).to(dev)
    _bf16_exact_(dec.fc1, rng, -2, 3); _bf16_exact_(dec.fc2, rng, -2, 3)
    src = torch.from_numpy(rng.randint(-3, 4, (n, D)).astype(np.float32)).to(dev).requires_grad_(True)
    dst = torch.from_numpy(rng.randint(-3, 4, (n, D)).astype(np.float32)).to(dev).requires_grad_(True)
    # exact case: small integers, every product and partial sum is representable
    ref = dec(ids, ids, src, dst)
    dec.fused = True
    out = dec(ids, ids, src, dst)
    assert out.shape == ref.shape == (n, 1)
    assert torch.equal(out, ref)
    # gradients: fp32 recompute in backward == autograd of the unfused module
    g = torch.from_numpy(rng.randn(n, 1).astype(np.float32)).to(dev)
    params = [dec.fc1.weight, dec.fc1.bias, dec.fc2.weight, dec.fc2.bias] + ([rp.feat] if with_feat else [])
    inputs = params + ([] if not_encode else [src, dst])
    dec.fused = False
    gr = torch.autograd.grad(dec(ids, ids, src, dst), inputs, g, allow_unused=True)
    dec.fused = True
    gf = torch.autograd.grad(dec(ids, ids, src, dst), inputs, g, allow_unused=True)
    for a, b in zip(gf, gr):
        assert (a is None) == (b is None)
        if a is not None:
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-4)
    # general data: bf16 operands, fp32 accumulate
    with torch.no_grad():
        dec.fc1.weight.normal_(0, 0.1); dec.fc1.bias.normal_(0, 0.1); dec.fc2.weight.normal_(0, 0.2); dec.fc2.bias.normal_()
        s2, d2 = torch.randn(n, D, device=dev), torch.randn(n, D, device=dev)
        if with_feat:
            rp.feat.copy_(torch.rand(n, 64, device=dev) * 3)
        dec.fused = False
        r2 = dec(ids, ids, s2, d2)
        dec.fused = True
        o2 = dec(ids, ids, s2, d2)
    scale = float(r2.abs().max()) + 1e-6
    assert float((o2 - r2).abs().max()) <= 3e-2 * scale
