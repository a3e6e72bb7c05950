"""LinkPredictor_v1's two layers (models/modules.py:73-117: concat[src_emb, dst_emb, feature] -> fc1 -> ReLU -> fc2 -> one
logit) as one bf16 matrix-core kernel (C ABI: tpnet_decoder_bf16).

Opt-in (`LinkPredictor_v1.fused = True`): bf16 operands with fp32 accumulation are NOT within the 1e-4 parity budget of
the fp32 path (expect ~1e-2 relative on a logit), so the default stays the torch fp32 layers.  Forward runs the fused
kernel (no concatenated input, no hidden layer in memory); backward recomputes the hidden layer with torch in fp32 and
returns exact fp32 gradients for fc1 / fc2 AND for the three inputs (the embeddings come from the trainable encoder, the
feature from the trainable `rp.mlp`)."""
import ctypes as C

import torch

from . import _lib


def pack_weights(fc1: torch.nn.Linear, fc2: torch.nn.Linear, D: int, F: int):
    """fc1.weight [H][2D+F] -> bf16 [32*HT][2*DP+F] with the input axis laid out [src | 0-pad | dst | 0-pad | feature]
    (DP = 16*ceil(D/16)) and zero rows beyond H; fc1.bias, fc2.weight[0] zero-padded to 32*HT; HT = ceil(H/32)."""
    H = fc1.out_features
    HT = (H + 31) // 32
    DP = (D + 15) // 16 * 16
    dev = fc1.weight.device
    w1 = fc1.weight.detach().float()
    w1p = torch.zeros((32 * HT, 2 * DP + F), dtype=torch.float32, device=dev)
    w1p[:H, :D] = w1[:, :D]
    w1p[:H, DP:DP + D] = w1[:, D:2 * D]
    if F:
        w1p[:H, 2 * DP:] = w1[:, 2 * D:]
    b1p = torch.zeros(32 * HT, dtype=torch.float32, device=dev)
    b1p[:H] = fc1.bias.detach().float()
    w2p = torch.zeros(32 * HT, dtype=torch.float32, device=dev)
    w2p[:H] = fc2.weight.detach().float()[0]
    return w1p.to(torch.bfloat16).contiguous(), b1p, w2p, float(fc2.bias.detach().float()[0]), HT


def supported(fc1, fc2, D: int, F: int) -> bool:
    return (isinstance(fc1, torch.nn.Linear) and isinstance(fc2, torch.nn.Linear) and fc2.out_features == 1
            and fc1.bias is not None and fc2.bias is not None and fc1.in_features == 2 * D + F
            and fc2.in_features == fc1.out_features and fc1.out_features <= 256 and D % 4 == 0 and F % 16 == 0
            and (D > 0 or F > 0))


def _prepared(owner, fc1, fc2, D, F):
    """Packed copies of the weights, rebuilt only when a parameter changed (optimizer step, load_state_dict, .to())."""
    ps = (fc1.weight, fc1.bias, fc2.weight, fc2.bias)
    key = tuple((p.data_ptr(), p._version, p.device) for p in ps) + (D, F)
    cache = owner.__dict__.get("_tpnet_decoder_prepared")
    if cache is None or cache[0] != key:
        with torch.no_grad():
            cache = (key, pack_weights(fc1, fc2, D, F))
        owner.__dict__["_tpnet_decoder_prepared"] = cache
    return cache[1]


class _FusedDecoder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, dst, feat, w1, b1, w2, b2, prep, not_encode):
        ref = feat if feat is not None else src
        if ref.device.type != "cuda" or ref.dtype != torch.float32:
            raise _lib.TPNetHipError("fused decoder needs float32 inputs on the GPU")
        src = src.contiguous(); dst = dst.contiguous()
        n, D = src.shape
        F = 0 if feat is None else feat.shape[1]
        if feat is not None:
            feat = feat.contiguous()
        out = torch.empty((n, 1), dtype=torch.float32, device=ref.device)
        w1p, b1p, w2p, b2f, HT = prep
        stream = C.c_void_p(torch.cuda.current_stream(ref.device).cuda_stream)
        _lib.check(_lib.load().tpnet_decoder_bf16(
            None if not_encode else src.data_ptr(), None if not_encode else dst.data_ptr(), D,
            None if feat is None else feat.data_ptr(), F, n, w1p.data_ptr(), b1p.data_ptr(), w2p.data_ptr(), b2f, HT,
            out.data_ptr(), stream), "decoder_bf16")
        ctx.not_encode = not_encode
        ctx.has_feat = feat is not None
        ctx.save_for_backward(src, dst, feat if feat is not None else src.new_empty(0), w1, b1, w2)
        return out

    @staticmethod
    def backward(ctx, gout):
        src, dst, feat, w1, b1, w2 = ctx.saved_tensors
        D = src.shape[1]
        if ctx.not_encode:                                   # modules.py:106-108: the embeddings are replaced by zeros
            src = torch.zeros_like(src); dst = torch.zeros_like(dst)
        x = torch.cat([src, dst, feat], dim=1) if ctx.has_feat else torch.cat([src, dst], dim=1)
        pre = torch.addmm(b1, x, w1.t())                     # fp32 recompute of the hidden layer
        hid = torch.relu(pre)
        gw2 = gout.t() @ hid
        gb2 = gout.sum(0)
        gh = (gout @ w2) * (pre > 0)
        gw1 = gh.t() @ x
        gb1 = gh.sum(0)
        gx = gh @ w1
        gsrc = None if ctx.not_encode else gx[:, :D]
        gdst = None if ctx.not_encode else gx[:, D:2 * D]
        gfeat = gx[:, 2 * D:] if ctx.has_feat else None
        return gsrc, gdst, gfeat, gw1, gb1, gw2, gb2, None, None


def fused_decoder(owner, fc1, fc2, src_emb, dst_emb, feat, not_encode: bool):
    D = src_emb.shape[1]
    F = 0 if feat is None else feat.shape[1]
    return _FusedDecoder.apply(src_emb, dst_emb, feat, fc1.weight, fc1.bias, fc2.weight, fc2.bias,
                               _prepared(owner, fc1, fc2, D, F), bool(not_encode))
