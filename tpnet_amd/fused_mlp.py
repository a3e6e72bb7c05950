"""self.mlp of RandomProjectionModule (Linear(64,256) -> ReLU -> Linear(256,64), models/TPNet.py:64-65,129) as one
bf16 MFMA kernel (C ABI: tpnet_mlp64_bf16; SURVEY.md §8 f-1, BASELINE config 5).

Opt-in (`RandomProjectionModule.fused_mlp = True`): bf16 operands with fp32 accumulation are NOT within the 1e-4
parity budget of the fp32 path (expect ~1e-2 relative), so the default stays the torch fp32 layers.  Forward runs
the fused kernel; backward recomputes the hidden layer with torch in fp32 and returns exact fp32 gradients for the
four parameter tensors (the input features carry no gradient: the projections are requires_grad=False)."""
import ctypes as C

import torch

from . import _lib

F, H = 64, 256


def permute_w2(w2: torch.Tensor) -> torch.Tensor:
    """[64][256] -> the hidden axis of every 32-tile in accumulator order: position 16 s + 8 h + j of a tile holds the
    hidden unit 16 s + 8 (j>>2) + 4 h + (j&3) (the order in which a 32x32 MFMA result tile lays its rows out over the
    16 registers of lane half h)."""
    idx = []
    for ht in range(H // 32):
        for s in range(2):
            for h in range(2):
                for j in range(8):
                    idx.append(ht * 32 + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3))
    return w2[:, torch.tensor(idx, device=w2.device)].contiguous()


def supported(mlp: torch.nn.Module) -> bool:
    return (isinstance(mlp, torch.nn.Sequential) and len(mlp) == 3 and isinstance(mlp[0], torch.nn.Linear)
            and isinstance(mlp[1], torch.nn.ReLU) and isinstance(mlp[2], torch.nn.Linear)
            and mlp[0].in_features == F and mlp[0].out_features == H and mlp[2].in_features == H
            and mlp[2].out_features == F and mlp[0].bias is not None and mlp[2].bias is not None)


def _prepared(mlp):
    """bf16 / permuted copies of the weights, rebuilt only when a parameter changed (optimizer step, load_state_dict,
    .to()): keyed on (data_ptr, _version) of the four tensors."""
    ps = (mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias)
    key = tuple((p.data_ptr(), p._version, p.device) for p in ps)
    cache = getattr(mlp, "_tpnet_prepared", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            prep = (ps[0].detach().to(torch.bfloat16).contiguous(), ps[1].detach().float().contiguous(),
                    permute_w2(ps[2].detach()).to(torch.bfloat16).contiguous(), ps[3].detach().float().contiguous(),
                    ps[2].detach().t().to(torch.bfloat16).contiguous())          # [4] = W2^T [256][64]: the backward's operand
        cache = (key, prep)
        mlp._tpnet_prepared = cache
    return cache[1]


BACKWARD = "mfma"       # "mfma": tpnet_mlp64_bwd_bf16 (bf16 operands, fp32 accumulation); "torch": the fp32 expressions


def weight_grads(x, gy, w1, b1, w2, prep=None):
    """Gradients of (mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias) given the pre-mlp features x [n, 64] and the
    gradient gy [n, 64] of the mlp's output.  The bf16 forward paths take the matrix-core kernel (one launch + a sum over the
    workgroups' partial results, deterministic); `BACKWARD = "torch"` or CPU tensors take the fp32 expressions."""
    if BACKWARD == "mfma" and x.is_cuda and x.shape[0] > 0 and prep is not None:
        lib = _lib.load()
        n = int(x.shape[0])
        x = x.contiguous()
        gy = gy.contiguous().float()
        pf = int(lib.tpnet_mlp64_bwd_partial_floats())
        nblk = min(256, (n + 31) // 32)
        part = torch.empty((nblk, pf), dtype=torch.float32, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        rc = lib.tpnet_mlp64_bwd_bf16(x.data_ptr(), gy.data_ptr(), n, prep[0].data_ptr(), prep[1].data_ptr(),
                                      prep[4].data_ptr(), part.data_ptr(), nblk, stream)
        if rc < 0:
            _lib.check(rc, "mlp64_bwd_bf16")
        tot = part[:rc].sum(0)
        return tot[:H * F].view(H, F), tot[2 * H * F:2 * H * F + H], tot[H * F:2 * H * F].view(F, H), gy.sum(0)
    pre = torch.addmm(b1, x, w1.t())                 # fp32 recompute of the hidden layer
    hid = torch.relu(pre)
    gh = (gy @ w2) * (pre > 0)
    return gh.t() @ x, gh.sum(0), gy.t() @ hid, gy.sum(0)


class _FusedMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, prep):
        if x.device.type != "cuda" or x.dtype != torch.float32:
            raise _lib.TPNetHipError("fused mlp needs float32 features on the GPU")
        x = x.contiguous()
        n = x.shape[0]
        y = torch.empty((n, F), dtype=torch.float32, device=x.device)
        w1b, b1c, w2p, b2c = prep[:4]
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(_lib.load().tpnet_mlp64_bf16(x.data_ptr(), n, w1b.data_ptr(), b1c.data_ptr(), w2p.data_ptr(),
                                                b2c.data_ptr(), y.data_ptr(), stream), "mlp64_bf16")
        ctx.save_for_backward(x, w1, b1, w2)
        ctx.prep = prep
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w1, b1, w2 = ctx.saved_tensors
        gw1, gb1, gw2, gb2 = weight_grads(x, gy, w1, b1, w2, ctx.prep)
        return None, gw1, gb1, gw2, gb2, None


def fused_mlp(mlp: torch.nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    return _FusedMLP.apply(x, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, _prepared(mlp))


# ---- readout + mlp in one kernel (tpnet_pair_feature_bf16): the features are formed into an LDS tile and consumed there ----
class _FusedReadoutMLP(torch.autograd.Function):
    """forward(w1, b1, w2, b2, launch, n): `launch(out_gram)` enqueues tpnet_pair_feature_bf16 and returns the features;
    backward as _FusedMLP (fp32 recompute of the hidden layer from the saved pre-mlp features)."""

    @staticmethod
    def forward(ctx, w1, b1, w2, b2, launch, n, prep):
        gram = torch.empty((n, F), dtype=torch.float32, device=w1.device)
        y = launch(gram)
        ctx.save_for_backward(gram, w1, b1, w2)
        ctx.prep = prep
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w1, b1, w2 = ctx.saved_tensors
        gw1, gb1, gw2, gb2 = weight_grads(x, gy, w1, b1, w2, ctx.prep)
        return gw1, gb1, gw2, gb2, None, None, None


def readout_supported(rp) -> bool:
    """tpnet_pair_feature_bf16 serves L = 3 and rows of whole 16-byte vectors from 64 floats (16, 32 or 64 lanes per pair)."""
    return rp.num_layer == 3 and rp.dim % 4 == 0 and rp.dim >= 36 and not rp.use_matrix and supported(rp.mlp)


def fused_readout_mlp(rp, u_dev: torch.Tensor, v_dev: torch.Tensor) -> torch.Tensor:
    """get_pair_wise_feature(u, v) (models/TPNet.py:112-129) for device-resident ids with self.mlp on the bf16 matrix cores
    inside the readout kernel."""
    mlp = rp.mlp
    prep = _prepared(mlp)
    w1b, b1c, w2p, b2c = prep[:4]
    n = int(u_dev.numel())
    lib = _lib.load()
    flags = _lib.FLAG_NOT_SCALE if rp.not_scale else 0

    def launch(gram):
        y = torch.empty((n, F), dtype=torch.float32, device=u_dev.device)
        _lib.check(lib.tpnet_pair_feature_bf16(rp._st_ref(), u_dev.data_ptr(), v_dev.data_ptr(), n, rp._now_host,
                                               float(rp.time_decay_weight), flags, w1b.data_ptr(), b1c.data_ptr(),
                                               w2p.data_ptr(), b2c.data_ptr(), gram.data_ptr() if gram is not None else None,
                                               y.data_ptr(), rp._stream()), "pair_feature_bf16")
        return y

    if torch.is_grad_enabled() and any(p.requires_grad for p in mlp.parameters()):
        return _FusedReadoutMLP.apply(mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, launch, n, prep)
    return launch(None)
