// SURVEY.md §8 f-1, second half: the decoder right behind the pairwise feature, LinkPredictor_v1 =
// concat[src_emb, dst_emb, feature] -> fc1 -> ReLU -> fc2 -> 1 logit (models/modules.py:73-117), as ONE kernel on the
// bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate).  Same tile convention as mlp.hip: layer 1 is computed
// transposed, H^T = W1 . X^T, so a 32x32 result tile has the pair on the lane and 16 hidden units in the registers; fc2
// has ONE output, so it is a dot product over the hidden axis: every lane multiplies its hidden units by w2 and sums
// them over all tiles in fp32, and the two lane halves of a pair are added at the end.  Neither the concatenated input
// nor the hidden layer is ever written to memory.
//   X^T columns (the k axis) = [ src_emb (D, zero-padded to DP = 16*ceil(D/16)) | dst_emb (same) | feature (F, F % 16 == 0) ]
//   w1p: bf16 [32*HT][KP], KP = 2*DP + F, fc1.weight with its input axis laid out the same way and its output axis padded
//   with zero rows to 32*HT (tpnet_amd/fused_decoder.py::pack_weights); b1p / w2p: f32 [32*HT], zero beyond the hidden dim.
//   src_emb == dst_emb == NULL is the reference's not_encode mode (embeddings replaced by zeros, modules.py:106-108):
//   their k-steps are skipped.
// (Measured and not kept, round 2: the opposite arrangement -- a wave keeps the 2 x 26 A fragments of two hidden tiles in
// registers for every pair tile and the 32 x 416 input tile is staged through LDS once for all waves: at 350+ registers per
// wave a CU holds one 3-wave workgroup, whose staging and MFMA phases then run back to back: 518 us against 255 us at
// n = 200 000.)
#include "tpnet_common.h"

namespace tpnet {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

static constexpr int DEC_BLOCK = 256;

// 8 consecutive floats of a row starting at column c (c % 8 == 0), zero beyond `width`, as bf16
__device__ __forceinline__ bf16x8 load8_bf16(const float* __restrict__ row, int c, int width) {
    float v[8];
    if (c + 8 <= width) {
        const float4 lo = *reinterpret_cast<const float4*>(row + c);
        const float4 hi = *reinterpret_cast<const float4*>(row + c + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (c + i < width) ? row[c + i] : 0.0f;
    }
    bf16x8 b;
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = (__bf16)v[i];
    return b;
}

template <int HT>
__global__ __launch_bounds__(DEC_BLOCK) void k_decoder_bf16(const float* __restrict__ src_emb,
                                                            const float* __restrict__ dst_emb, int D,
                                                            const float* __restrict__ feat, int F, int64_t n,
                                                            const __bf16* __restrict__ w1p, const float* __restrict__ b1p,
                                                            const float* __restrict__ w2p, float b2,
                                                            float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int DP = (D + 15) / 16 * 16;
    const int KP = 2 * DP + F;
    const int DS = DP / 16, FS = F / 16;
    const int64_t ntiles = (n + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * (DEC_BLOCK / 64) + wave; tile < ntiles;
         tile += (int64_t)gridDim.x * (DEC_BLOCK / 64)) {
        const int64_t pair = tile * 32 + r;
        const bool valid = pair < n;
        const int64_t prow = valid ? pair : 0;
        f32x16 acc[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;
        // k-steps of 16 input columns: B fragment = X^T, lane (r, h) holds X[pair r][16 s + 8 h + j]; A = W1p rows
        auto step = [&](const bf16x8 bx, int kcol) {
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(w1p + (int64_t)(t * 32 + r) * KP + kcol + 8 * h);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bx, acc[t], 0, 0, 0);
            }
        };
        if (src_emb) {
            const float* xs = src_emb + prow * D;
            const float* xd = dst_emb + prow * D;
            for (int s = 0; s < DS; ++s) step(load8_bf16(xs, 16 * s + 8 * h, D), 16 * s);
            for (int s = 0; s < DS; ++s) step(load8_bf16(xd, 16 * s + 8 * h, D), DP + 16 * s);
        }
        if (feat) {
            const float* xf = feat + prow * F;
            for (int s = 0; s < FS; ++s) step(load8_bf16(xf, 16 * s + 8 * h, F), 2 * DP + 16 * s);
        }
        // register q of tile t = hidden unit 32 t + (q&3) + 8 (q>>2) + 4 h of pair r:  bias, ReLU, times w2, summed
        float part = 0.0f;
#pragma unroll
        for (int t = 0; t < HT; ++t) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int hid = 32 * t + (q & 3) + 8 * (q >> 2) + 4 * h;
                float v = acc[t][q] + b1p[hid];
                v = v > 0.0f ? v : 0.0f;
                part = fmaf(v, w2p[hid], part);
            }
        }
        part += __shfl_xor(part, 32);
        if (valid && h == 0) out[pair] = part + b2;
    }
}

}  // namespace tpnet

using namespace tpnet;

extern "C" int tpnet_decoder_bf16(const float* src_emb, const float* dst_emb, int32_t D, const float* feat, int32_t F,
                                  int64_t n, const void* w1p_bf16, const float* b1p, const float* w2p, float b2,
                                  int32_t hidden_tiles, float* out, void* stream) {
    if (n < 0 || D < 0 || F < 0 || F % 16 != 0 || (D % 4) != 0) return TPNET_ERR_BAD_ARG;
    if ((src_emb == nullptr) != (dst_emb == nullptr)) return TPNET_ERR_BAD_ARG;
    if (!src_emb && !feat) return TPNET_ERR_BAD_ARG;
    if (n > 0 && (!w1p_bf16 || !b1p || !w2p || !out)) return TPNET_ERR_BAD_ARG;
    if (n == 0) return TPNET_OK;
    const int64_t tiles = (n + 31) / 32;
    int64_t grid = (tiles + 3) / 4;
    if (grid > 2048) grid = 2048;
#define TPNET_DEC(HT_)                                                                                              \
    hipLaunchKernelGGL(k_decoder_bf16<HT_>, dim3((unsigned)grid), dim3(DEC_BLOCK), 0, (hipStream_t)stream, src_emb,  \
                       dst_emb, (int)D, feat, (int)F, n, (const __bf16*)w1p_bf16, b1p, w2p, b2, out)
    switch (hidden_tiles) {
        case 1: TPNET_DEC(1); break;
        case 2: TPNET_DEC(2); break;
        case 3: TPNET_DEC(3); break;
        case 4: TPNET_DEC(4); break;
        case 5: TPNET_DEC(5); break;
        case 6: TPNET_DEC(6); break;
        case 7: TPNET_DEC(7); break;
        case 8: TPNET_DEC(8); break;
        default: return TPNET_ERR_BAD_ARG;
    }
#undef TPNET_DEC
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}
