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


_W2F_IDX = {}


def _w2f_index(dev):
    """Index tensors of the gather w2f[((w*2 + t)*64 + lane)*16 + s] = W2[32 t + (lane & 31)][32 w + (s & 3) + 8 (s >> 2) +
    4 (lane >> 5)] (w = 0..7, t = 0..1, lane = 0..63, s = 0..15), cached per device."""
    key = str(dev)
    if key not in _W2F_IDX:
        w = torch.arange(8).view(8, 1, 1, 1)
        t = torch.arange(2).view(1, 2, 1, 1)
        lane = torch.arange(64).view(1, 1, 64, 1)
        s_ = torch.arange(16).view(1, 1, 1, 16)
        rows = (32 * t + (lane & 31)).expand(8, 2, 64, 16).reshape(-1)
        cols = (32 * w + (s_ & 3) + 8 * (s_ >> 2) + 4 * (lane >> 5)).expand(8, 2, 64, 16).reshape(-1)
        _W2F_IDX[key] = (rows.to(dev), cols.to(dev))
    return _W2F_IDX[key]


def _w2f_rows(dev):
    return _w2f_index(dev)[0]


def _w2f_cols(dev):
    return _w2f_index(dev)[1]


# rows from which the matrix-core backward serves a call: one launch + a sum of partials against ten small torch launches -- at the
# decoder's 1 000 rows per call a training step went 704 -> 520-570 us (bench.py, dropin.train); below, the torch expressions
MFMA_BWD_FROM = 512
# "mfma" (default): weight gradients of self.mlp on the matrix cores from MFMA_BWD_FROM rows on -- two-piece bf16 operands, three
# products per term, fp32 accumulation: ~2^-16 per PRODUCT (3-5e-6 of the terms' magnitude observed, 4e-5 allowed by the tests)
# where the reference's fp32 GEMMs carry 2^-24; "torch": always the fp32 torch expressions (five GEMMs + three elementwise passes).
# A module overrides the process-wide value with an attribute of the same name on its self.mlp (rp.mlp.mlp_backward = "torch").
mlp_backward = "mfma"
_declined_once = set()


def weight_grads_f32(x, gy, w1, b1, w2, prep, mode=None):
    """Gradients of (mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias) given the pre-mlp features x [n, 64] and the gradient
    gy [n, 64] of self.mlp's output, fp32 class.  Long lists (the encoder's calls: 80 000 rows each at C2) take ONE launch on the
    matrix cores (tpnet_mlp64_bwd_f32: hidden layer recomputed, both weight-gradient products contracted over the rows with
    two-piece bf16 operands and fp32 accumulation, one partial result per workgroup, summed in a fixed order) instead of five
    fp32 GEMMs and three elementwise passes over n x 256 floats; `prep` = the prepared() entry the forward used -- if a Parameter
    changed since (never in a normal training step), or for short lists and CPU tensors, the torch expressions serve.
    mode: "mfma" | "torch" (None: the module-level `mlp_backward`)."""
    mode = mode or mlp_backward
    n = int(x.shape[0])
    if (mode == "mfma" and prep is not None and x.is_cuda and n >= MFMA_BWD_FROM and x.dtype == torch.float32 and prep[1].w1 and prep[1].w2t
            and prep[0][:6] == (w1.data_ptr(), w1._version, b1.data_ptr(), b1._version, w2.data_ptr(), w2._version)):
        lib = _lib.load()
        x = x.contiguous()
        gy = gy.contiguous().float()
        pf = int(lib.tpnet_mlp64_bwd_partial_floats())
        nblk = min(256, (n + 31) // 32)
        part = torch.empty((nblk, pf), dtype=torch.float32, device=x.device)
        rc = lib.tpnet_mlp64_bwd_f32(x.data_ptr(), gy.data_ptr(), n, prep[2], part.data_ptr(), nblk,
                                     C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc > 0:
            tot = part[:rc].sum(0)
            H, F = w1.shape
            return tot[:H * F].view(H, F), tot[2 * H * F:2 * H * F + H], tot[H * F:2 * H * F].view(F, H), gy.sum(0)
        if rc not in _declined_once:                    # (said once per code: the torch expressions below serve the call)
            _declined_once.add(rc)
            import warnings
            warnings.warn(f"tpnet_mlp64_bwd_f32 declined a call of {n} rows (rc {rc}): weight gradients by the fp32 torch expressions")
    pre = torch.addmm(b1, x, w1.t())                 # fp32 recompute of the hidden layer
    hid = torch.relu(pre)
    gh = (gy @ w2) * (pre > 0)
    return gh.t() @ x, gh.sum(0), gy.t() @ hid, gy.sum(0)


class _MlpF32(torch.autograd.Function):
    """self.mlp alone on the fp32 matrix cores (tpnet_mlp64_f32); backward = weight_grads_f32."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, mlp_ref, prep=None, mode=None):
        x = x.contiguous()
        y = torch.empty_like(x)
        if x.shape[0]:
            rc = _lib.load().tpnet_mlp64_f32(x.data_ptr(), x.shape[0], mlp_ref, y.data_ptr(),
                                             C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
            if rc:
                _lib.check(rc, "mlp64_f32")
        ctx.save_for_backward(x, w1, b1, w2)
        ctx.prep = prep
        ctx.mode = mode
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w1, b1, w2 = ctx.saved_tensors
        if ctx.needs_input_grad[0]:                            # (the readout's features carry no gradient; a caller's own x may)
            pre = torch.addmm(b1, x, w1.t())
            hid = torch.relu(pre)
            gh = (gy @ w2) * (pre > 0)
            return gh @ w1, gh.t() @ x, gh.sum(0), gy.t() @ hid, gy.sum(0), None, None, None
        gw1, gb1, gw2, gb2 = weight_grads_f32(x, gy, w1, b1, w2, ctx.prep, ctx.mode)
        return None, gw1, gb1, gw2, gb2, None, None, None


def mlp_f32(mlp, x):
    """mlp(x) through tpnet_mlp64_f32 if `mlp` is the reference's 64-256-64 on this GPU (else None)."""
    if x.dtype != torch.float32 or not x.is_cuda or x.dim() != 2 or x.shape[1] != 64:
        return None
    prep = prepared(mlp, 64)
    if prep is None or not prep[1].w1:
        return None
    if needs_grad(prep[4]):
        return _MlpF32.apply(x, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, prep[2], prep, getattr(mlp, "mlp_backward", None))
    return _MlpF32.forward(_NoCtx(), x, None, None, None, None, prep[2])


class _NoCtx:
    def save_for_backward(self, *a):
        pass


def invalidate(mlp=None):
    """Forget the prepared copies of `mlp`'s weights (all modules' if None).  The cache is keyed on (data_ptr, _version) of the
    four tensors; an in-place write THROUGH `.data` (p.data.copy_(), p.data.mul_(): some EMA / weight-averaging helpers) bumps
    neither, so whoever writes that way calls this afterwards.  Optimizer steps, load_state_dict, .to() and plain in-place ops
    on the Parameters are seen without it."""
    if mlp is None:
        _PREPARED.clear()
    else:
        _PREPARED.pop(mlp, None)


def prepared(mlp, F):
    """(key, tpnet_mlp struct, its byref, keep-alive tensors) or None if `mlp` is not the reference's Linear-ReLU-Linear on
    a GPU: transposed f32 copies of the weights, rebuilt only when a parameter changed (optimizer step, load_state_dict,
    .to()): keyed on (data_ptr, _version) of the four tensors.
    CONTRACT: the Parameters are updated through versioned ops (optimizer steps, copy_(), load_state_dict, .to()).  b1, b2 and
    tpnet_mlp::w1 are the Parameters' OWN storage while w1t / w2t / w2f / wimg are derived copies, so a write that bumps no version
    counter (p.data.clamp_(), a raw pointer) leaves a MIX of live and stale values until invalidate(mlp) is called; and the derived
    buffers are rewritten in place on the CURRENT stream -- update the Parameters on the stream that runs the module's kernels (the
    reference's loop has one stream)."""
    try:
        l1, l2 = mlp._modules["0"], mlp._modules["2"]
        p1, p2 = l1._parameters, l2._parameters      # (the dicts behind l1.weight ...: nn.Module.__getattr__ is ~0.4 us per read,
        w1, b1, w2, b2 = p1["weight"], p1["bias"], p2["weight"], p2["bias"]   #  and this runs on every call)
        if w1 is None or b1 is None or w2 is None or b2 is None:
            return None
    except (KeyError, AttributeError):
        return None
    key = (w1.data_ptr(), w1._version, b1.data_ptr(), b1._version, w2.data_ptr(), w2._version, b2.data_ptr(), b2._version)
    cache = _PREPARED.get(mlp)
    if cache is None or cache[0] != key:
        if not supported(mlp, F) or not (w1.is_contiguous() and w2.is_contiguous() and b1.is_contiguous() and b2.is_contiguous()):
            return None
        # the derived layouts (w1t, w2t and, for F = 64, the gathered w2f) live in buffers that stay with the module and are
        # rewritten by ONE launch (tpnet_mlp_prepare) when a parameter changed; b1, b2 and tpnet_mlp::w1 are the Parameters' own
        # storage.  (As torch expressions -- two transposes, a copy, an index gather -- this was ~100 us of every training step.)
        same_store = cache is not None and cache[5] == (w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr())
        if same_store:
            keep, st = cache[3], cache[1]
        else:
            H = w1.shape[0]
            x64 = F == 64 and H == 256
            keep = (torch.empty((F, H), dtype=torch.float32, device=w1.device), torch.empty((H, F), dtype=torch.float32, device=w1.device),
                    torch.empty(F * H, dtype=torch.float32, device=w1.device) if x64 else None,
                    # the weight image of the encoder's one-launch readout + dense layers (csrc/encoder_mfma.hip)
                    torch.empty(int(_lib.load().tpnet_mlp_image_bytes()), dtype=torch.uint8, device=w1.device) if x64 else None)
            st = _lib.Mlp(w1t=keep[0].data_ptr(), b1=b1.data_ptr(), w2t=keep[1].data_ptr(), b2=b2.data_ptr(), F=F, H=H,
                          w1=w1.data_ptr() if keep[2] is not None else None,
                          w2f=keep[2].data_ptr() if keep[2] is not None else None,
                          wimg=keep[3].data_ptr() if keep[3] is not None else None)
        _lib.check(_lib.load().tpnet_mlp_prepare(w1.data_ptr(), w2.data_ptr(), F, w1.shape[0], keep[0].data_ptr(), keep[1].data_ptr(),
                                                 keep[2].data_ptr() if keep[2] is not None else None,
                                                 C.c_void_p(torch.cuda.current_stream(w1.device).cuda_stream)), "mlp_prepare")
        if keep[3] is not None:
            _lib.check(_lib.load().tpnet_mlp_prepare_image(w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), keep[3].data_ptr(),
                                                           C.c_void_p(torch.cuda.current_stream(w1.device).cuda_stream)), "mlp_prepare_image")
        cache = (key, st, C.byref(st), keep, (w1, b1, w2, b2), (w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()))
        _PREPARED[mlp] = cache
    return cache


class _FusedFeature(torch.autograd.Function):
    """forward(w1, b1, w2, b2, launch): `launch(out_gram)` enqueues the fused kernel and returns the features."""

    @staticmethod
    def forward(ctx, w1, b1, w2, b2, launch, n, F, prep=None, mode=None):
        gram = torch.empty((n, F), dtype=torch.float32, device=w1.device)
        out = launch(gram)
        ctx.save_for_backward(gram, w1, b1, w2)
        ctx.prep = prep
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, gy):
        x, w1, b1, w2 = ctx.saved_tensors
        gw1, gb1, gw2, gb2 = weight_grads_f32(x, gy, w1, b1, w2, ctx.prep, ctx.mode)
        return gw1, gb1, gw2, gb2, None, None, None, None, None


def needs_grad(keep_params) -> bool:
    return torch.is_grad_enabled() and any(p.requires_grad for p in keep_params)


def apply_with_grad(mlp, launch, n, F):
    return _FusedFeature.apply(mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, launch, n, F, _PREPARED.get(mlp),
                               getattr(mlp, "mlp_backward", None))
