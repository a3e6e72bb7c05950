// Backward of self.mlp = Linear(64,256) -> ReLU -> Linear(256,64) (models/TPNet.py:64-65) with respect to its four weight
// tensors, on the bf16 matrix cores ("needs a backward for training (grad w.r.t. weights only -- P has no
// grad)").  One kernel: per tile of 32 pairs, wave w (of 8) owns hidden units [32w, 32w+32):
//   pre^T = W1[32w.., :] . X^T + b1,  H^T = relu(pre^T)          (recomputed, as the forward kernel forms it)
//   gH^T  = (W2^T[32w.., :] . gY^T) * (pre^T > 0)
//   gW1[32w.., :] += gH^T . X        gW2[:, 32w..] += gY^T . H      (contraction over the 32 pairs: the tiles go through LDS
//                                                                   transposed, [unit][pair], so that they load as operands)
//   gb1[32w..]   += sum_pairs gH                                     (fp32, per-lane partial sums, reduced once per workgroup;
//                                                                    gb2 = sum_pairs gY is a column sum the caller takes)
// The accumulators live in registers over all tiles of a workgroup; every workgroup writes ONE partial result, and the
// caller adds the partials (a fixed-order sum: run-to-run identical bits).
// Two arithmetic classes (template F32): bf16 operands with fp32 accumulation (the opt-in 1e-2 class), and -- round 4 -- the fp32
// class of the default forward paths: every operand split into two bf16 pieces (x = hi + lo) and every product formed as
// lo*hi + hi*lo + hi*hi (mlp_x3.hip's arithmetic), weights taken from the f32 Parameters (mlp[0].weight [256][64] and the
// prepared transpose of mlp[2].weight, tpnet_mlp::w2t [256][64]); the four LDS tiles exist twice (hi and lo planes).  What it
// replaces in a training step at the encoder level: five fp32 torch GEMMs and three elementwise passes over 80 000 x 256 floats
// per call (~1 ms).
#include "tpnet_common.h"

namespace tpnet {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

static constexpr int BB = 512;             // 8 waves = the 8 hidden tiles
static constexpr int BF = 64, BH = 256;
static constexpr int RS = 40;              // bf16 elements per LDS row of a [unit][32 pairs] tile (80 B: 16-byte aligned, bank spread)

__device__ __forceinline__ bf16x8 row8_bf16(const float* __restrict__ p) {
    const float4 lo = *reinterpret_cast<const float4*>(p);
    const float4 hi = *reinterpret_cast<const float4*>(p + 4);
    bf16x8 b;
    b[0] = (__bf16)lo.x; b[1] = (__bf16)lo.y; b[2] = (__bf16)lo.z; b[3] = (__bf16)lo.w;
    b[4] = (__bf16)hi.x; b[5] = (__bf16)hi.y; b[6] = (__bf16)hi.z; b[7] = (__bf16)hi.w;
    return b;
}
// 8 consecutive f32 -> (hi, lo) bf16 pieces
__device__ __forceinline__ void row8_split(const float* __restrict__ p, bf16x8& hi, bf16x8& lo) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 t = (__bf16)v[j];
        hi[j] = t;
        lo[j] = (__bf16)(v[j] - (float)t);
    }
}

// partial layout per workgroup (floats): gW1 [256][64] | gW2 [64][256] | gb1 [256]
static constexpr int P_W1 = 0, P_W2 = BH * BF, P_B1 = 2 * BH * BF, P_TOT = 2 * BH * BF + BH;

// c += A B over one 16-deep step; F32: operands in two pieces, the small terms first
template <bool F32>
__device__ __forceinline__ f32x16 mm3(const bf16x8* a, const bf16x8* b, f32x16 c) {
    if constexpr (F32) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
}

// W1S / W2S: bf16 [256][64] (F32 = false) or f32 [256][64] (F32 = true: mlp[0].weight, and mlp[2].weight transposed)
template <bool F32>
__global__ __launch_bounds__(BB) void k_mlp64_bwd(const float* __restrict__ X, const float* __restrict__ GY, int64_t n,
                                                  const void* __restrict__ w1v, const float* __restrict__ b1,
                                                  const void* __restrict__ w2tv, float* __restrict__ partial) {
    constexpr int NP = F32 ? 2 : 1;                                       // planes: hi (and lo)
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    __bf16* xt = lds;                                                     // X^T   [plane][feature][pair]
    __bf16* gyt = xt + NP * BF * RS;                                      // gY^T  [plane][output][pair]
    __bf16* ht_all = gyt + NP * BF * RS;                                  // H^T   [wave][plane][hidden of the wave's tile][pair]
    __bf16* ght_all = ht_all + 8 * NP * 32 * RS;                          // gH^T
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // this wave's rows of W1 and of W2^T as A operands (k = features / outputs)
    bf16x8 a1[4][NP], a2[4][NP];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int off = (wave * 32 + r) * BF + 16 * s + 8 * h;
        if constexpr (F32) {
            row8_split(reinterpret_cast<const float*>(w1v) + off, a1[s][0], a1[s][NP - 1]);
            row8_split(reinterpret_cast<const float*>(w2tv) + off, a2[s][0], a2[s][NP - 1]);
        } else {
            a1[s][0] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(w1v) + off);
            a2[s][0] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(w2tv) + off);
        }
    }
    // the bias of this wave's hidden tile in accumulator order, parked in LDS (16 registers the fp32 class does not have):
    // float 16 * (2 wave + h) + q = b1[32 wave + (q & 3) + 8 (q >> 2) + 4 h]
    float* bias_l = reinterpret_cast<float*>(ght_all + 8 * NP * 32 * RS);
    if (r == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) bias_l[16 * (2 * wave + h) + q] = b1[wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * h];
    }
    __syncthreads();
    const float4* bias4 = reinterpret_cast<const float4*>(bias_l + 16 * (2 * wave + h));
    f32x16 gw1[2], gw2[2];                   // gW1[32w + m][32 nt + r], nt = 0,1;  gW2[32 mt + m][32w + r], mt = 0,1
    float gb1[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { gw1[0][q] = 0.f; gw1[1][q] = 0.f; gw2[0][q] = 0.f; gw2[1][q] = 0.f; gb1[q] = 0.f; }

    const int64_t ntiles = (n + 31) / 32;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t pair = tile * 32 + r;
        const bool valid = pair < n;
        const float* xr = X + (valid ? pair : 0) * BF;
        const float* gr = GY + (valid ? pair : 0) * BF;
        __bf16* hrow = ht_all + wave * NP * 32 * RS;
        __bf16* grow = ght_all + wave * NP * 32 * RS;
        // one operand set at a time (X, then gY: the fp32 class holds 160 registers of weights and accumulators for the whole
        // launch): rows -> pieces, the transposed tile to LDS (wave 0: X^T, wave 1: gY^T; every wave holds the same values:
        // element [16 s + 8 h + j][pair r]), then this wave's hidden tile -- register q = hidden row (q&3) + 8 (q>>2) + 4 h,
        // column = pair r
        uint32_t on_mask = 0;
        auto load_rows = [&](const float* rowp, bf16x8 (&b)[4][NP], __bf16* dst, bool store) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if constexpr (F32) row8_split(rowp + 16 * s + 8 * h, b[s][0], b[s][NP - 1]);
                else b[s][0] = row8_bf16(rowp + 16 * s + 8 * h);
                if (!valid) {
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                        for (int j = 0; j < 8; ++j) b[s][pl][j] = (__bf16)0.0f;
                }
            }
            if (store) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int j = 0; j < 8; ++j) dst[pl * BF * RS + (16 * s + 8 * h + j) * RS + r] = b[s][pl][j];
            }
        };
        {
            bf16x8 bx[4][NP];
            load_rows(xr, bx, xt, wave == 0);
            f32x16 pre;
#pragma unroll
            for (int q = 0; q < 16; ++q) pre[q] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) pre = mm3<F32>(a1[s], bx[s], pre);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int hq = (q & 3) + 8 * (q >> 2) + 4 * h;
                const float4 bq = bias4[q >> 2];
                const float p = pre[q] + ((q & 3) == 0 ? bq.x : (q & 3) == 1 ? bq.y : (q & 3) == 2 ? bq.z : bq.w);
                const bool on = valid && p > 0.0f;
                on_mask |= (on ? 1u : 0u) << q;
                const float hv = on ? p : 0.0f;
                const __bf16 hh = (__bf16)hv;
                hrow[hq * RS + r] = hh;
                if constexpr (F32) hrow[32 * RS + hq * RS + r] = (__bf16)(hv - (float)hh);
            }
        }
        {
            bf16x8 bg[4][NP];
            load_rows(gr, bg, gyt, wave == 1);
            f32x16 gh;
#pragma unroll
            for (int q = 0; q < 16; ++q) gh[q] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) gh = mm3<F32>(a2[s], bg[s], gh);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int hq = (q & 3) + 8 * (q >> 2) + 4 * h;
                const float gv = ((on_mask >> q) & 1u) ? gh[q] : 0.0f;
                gb1[q] += gv;
                const __bf16 gg = (__bf16)gv;
                grow[hq * RS + r] = gg;
                if constexpr (F32) grow[32 * RS + hq * RS + r] = (__bf16)(gv - (float)gg);
            }
        }
        __syncthreads();
        // gW1[32w + m][32 nt + r] += sum_pairs gH^T[m][pair] X[pair][32 nt + r]:  A = gH^T rows, B[k = pair][n] = X^T[n][pair]
        // gW2[32 mt + m][32w + r] += sum_pairs gY^T[32 mt + m][pair] H[pair][32w + r]: A = gY^T rows, B = H^T[r][pair]
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ko = 16 * s + 8 * h;           // 8 consecutive pairs
            bf16x8 a_gh[NP], b_h[NP];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                a_gh[pl] = *reinterpret_cast<const bf16x8*>(grow + pl * 32 * RS + r * RS + ko);
                b_h[pl] = *reinterpret_cast<const bf16x8*>(hrow + pl * 32 * RS + r * RS + ko);
            }
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                bf16x8 b_x[NP], a_gy[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    b_x[pl] = *reinterpret_cast<const bf16x8*>(xt + pl * BF * RS + (32 * t2 + r) * RS + ko);
                    a_gy[pl] = *reinterpret_cast<const bf16x8*>(gyt + pl * BF * RS + (32 * t2 + r) * RS + ko);
                }
                gw1[t2] = mm3<F32>(a_gh, b_x, gw1[t2]);
                gw2[t2] = mm3<F32>(a_gy, b_h, gw2[t2]);
            }
        }
        __syncthreads();                          // the tiles are rewritten by the next tile
    }
    // ---- this workgroup's partial result
    float* P = partial + (int64_t)blockIdx.x * P_TOT;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = (q & 3) + 8 * (q >> 2) + 4 * h;
            P[P_W1 + (wave * 32 + m) * BF + 32 * t2 + r] = gw1[t2][q];
            P[P_W2 + (32 * t2 + m) * BH + wave * 32 + r] = gw2[t2][q];
        }
    }
    // gb1: sum over the 32 pairs (lanes r) of each half h; register q <-> hidden (q&3) + 8 (q>>2) + 4 h
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        float v = gb1[q];
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 32);
        if (r == 0) P[P_B1 + wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * h] = v;
    }
}

template <bool F32>
static constexpr size_t bwd_lds_bytes() { return (size_t)(F32 ? 2 : 1) * (2 * BF * RS + 2 * 8 * 32 * RS) * sizeof(__bf16) + 256 * sizeof(float); }

}  // namespace tpnet

using namespace tpnet;

extern "C" {

int64_t tpnet_mlp64_bwd_partial_floats(void) { return (int64_t)P_TOT; }

int tpnet_mlp64_bwd_bf16(const float* x, const float* gy, int64_t n, const void* w1_bf16, const float* b1,
                         const void* w2t_bf16, float* partial, int32_t n_partial, void* stream) {
    if (n < 1 || !x || !gy || !w1_bf16 || !b1 || !w2t_bf16 || !partial || n_partial < 1) return TPNET_ERR_BAD_ARG;
    const int64_t tiles = (n + 31) / 32;
    const int grid = (int)(tiles < n_partial ? tiles : n_partial);
    // (workgroups beyond `grid` do not exist: the caller sums only the first min(n_partial, ceil(n / 32)) partials)
    hipLaunchKernelGGL(k_mlp64_bwd<false>, dim3((unsigned)grid), dim3(BB), bwd_lds_bytes<false>(), (hipStream_t)stream, x, gy, n,
                       w1_bf16, b1, w2t_bf16, partial);
    TPNET_HIP_TRY(hipGetLastError());
    return grid;
}

int tpnet_mlp64_bwd_f32(const float* x, const float* gy, int64_t n, const tpnet_mlp* mlp, float* partial, int32_t n_partial,
                        void* stream) {
    if (n < 1 || !x || !gy || !mlp || !mlp->w1 || !mlp->w2t || !mlp->b1 || mlp->F != 64 || mlp->H != 256 || !partial || n_partial < 1)
        return TPNET_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(mlp->w1) |
         reinterpret_cast<uintptr_t>(mlp->w2t)) & 15)
        return TPNET_ERR_BAD_ARG;
    static int lds_ok = 0;
    if (lds_ok == 0)
        lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp64_bwd<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)bwd_lds_bytes<true>()) == hipSuccess ? 1 : -1;
    if (lds_ok < 0) { (void)hipGetLastError(); return TPNET_ERR_BAD_ARG; }
    const int64_t tiles = (n + 31) / 32;
    const int grid = (int)(tiles < n_partial ? tiles : n_partial);
    hipLaunchKernelGGL(k_mlp64_bwd<true>, dim3((unsigned)grid), dim3(BB), bwd_lds_bytes<true>(), (hipStream_t)stream, x, gy, n,
                       (const void*)mlp->w1, mlp->b1, (const void*)mlp->w2t, partial);
    TPNET_HIP_TRY(hipGetLastError());
    return grid;
}

}  // extern "C"
