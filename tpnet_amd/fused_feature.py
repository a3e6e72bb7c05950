"""get_pair_wise_feature in one launch (C ABI: tpnet_pair_feature / tpnet_host_pair_feature): the pairwise readout and
self.mlp = Linear(F, 4F) -> ReLU -> Linear(4F, F) (models/TPNet.py:63-65, 112-129) fused in fp32.

Forward = the fused kernel (the features stay in LDS between the readout and the dense layers).  When gradients are
being recorded, the kernel also writes the pre-mlp features and the backward pass is the fp32 torch expression of the two
layers' gradients (the projections carry no gradient: requires_grad=False in the reference, models/TPNet.py:49-62).
Differs from the torch layers in f32 summation order only, so it is the DEFAULT for the decoder's short pair lists."""
import ctypes as C
import os
import weakref

import torch

from . import _lib

_PREPARED = weakref.WeakKeyDictionary()     # mlp module -> cache entry (kept off the module: ctypes objects do not deepcopy)
# longer lists: the readout kernels + the dense layers as GEMMs (torch, or the bf16 kernel if opted in)
MAX_PAIRS = int(os.environ.get("TPNET_DEV_FUSED_MAX_PAIRS", "8192"))


def supported(mlp: torch.nn.Module, F: int) -> bool:
    return (isinstance(mlp, torch.nn.Sequential) and len(mlp) == 3 and isinstance(mlp[0], torch.nn.Linear)
            and isinstance(mlp[1], torch.nn.ReLU) and isinstance(mlp[2], torch.nn.Linear)
            and mlp[0].in_features == F and mlp[0].out_features == 4 * F and mlp[2].in_features == 4 * F
            and mlp[2].out_features == F and mlp[0].bias is not None and mlp[2].bias is not None
            and mlp[0].weight.dtype == torch.float32 and mlp[0].weight.is_cuda)


def prepared(mlp, F):
    """(key, tpnet_mlp struct, its byref, keep-alive tensors) or None if `mlp` is not the reference's Linear-ReLU-Linear on
    a GPU: transposed f32 copies of the weights, rebuilt only when a parameter changed (optimizer step, load_state_dict,
    .to()): keyed on (data_ptr, _version) of the four tensors."""
    try:
        l1, l2 = mlp._modules["0"], mlp._modules["2"]
        w1, b1, w2, b2 = l1.weight, l1.bias, l2.weight, l2.bias
    except (KeyError, AttributeError):
        return None
    key = (w1.data_ptr(), w1._version, b1.data_ptr(), b1._version, w2.data_ptr(), w2._version, b2.data_ptr(), b2._version)
    cache = _PREPARED.get(mlp)
    if cache is None or cache[0] != key:
        if not supported(mlp, F):
            return None
        with torch.no_grad():
            keep = (w1.detach().t().contiguous(), b1.detach().contiguous(), w2.detach().t().contiguous(),
                    b2.detach().contiguous())
        st = _lib.Mlp(w1t=keep[0].data_ptr(), b1=keep[1].data_ptr(), w2t=keep[2].data_ptr(), b2=keep[3].data_ptr(),
                      F=w1.shape[1], H=w1.shape[0])
        cache = (key, st, C.byref(st), keep, (w1, b1, w2, b2))
        _PREPARED[mlp] = cache
    return cache


class _FusedFeature(torch.autograd.Function):
    """forward(w1, b1, w2, b2, launch): `launch(out_gram)` enqueues the fused kernel and returns the features."""

    @staticmethod
    def forward(ctx, w1, b1, w2, b2, launch, n, F):
        gram = torch.empty((n, F), dtype=torch.float32, device=w1.device)
        out = launch(gram)
        ctx.save_for_backward(gram, w1, b1, w2)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, w1, b1, w2 = ctx.saved_tensors
        pre = torch.addmm(b1, x, w1.t())                 # fp32 recompute of the hidden layer
        hid = torch.relu(pre)
        gw2 = gy.t() @ hid
        gb2 = gy.sum(0)
        gh = (gy @ w2) * (pre > 0)
        gw1 = gh.t() @ x
        gb1 = gh.sum(0)
        return gw1, gb1, gw2, gb2, None, None, None


def needs_grad(keep_params) -> bool:
    return torch.is_grad_enabled() and any(p.requires_grad for p in keep_params)


def apply_with_grad(mlp, launch, n, F):
    return _FusedFeature.apply(mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, launch, n, F)
