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
// caller adds the partials (a fixed-order sum: run-to-run identical bits).  bf16 operands, fp32 accumulation.
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

// partial layout per workgroup (floats): gW1 [256][64] | gW2 [64][256] | gb1 [256]
static constexpr int P_W1 = 0, P_W2 = BH * BF, P_B1 = 2 * BH * BF, P_TOT = 2 * BH * BF + BH;

__global__ __launch_bounds__(BB) void k_mlp64_bwd_bf16(const float* __restrict__ X, const float* __restrict__ GY, int64_t n,
                                                       const __bf16* __restrict__ w1, const float* __restrict__ b1,
                                                       const __bf16* __restrict__ w2t, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) __bf16 xt[BF * RS];          // X^T   [feature][pair]
    __shared__ __attribute__((aligned(16))) __bf16 gyt[BF * RS];         // gY^T  [output][pair]
    __shared__ __attribute__((aligned(16))) __bf16 ht[8][32 * RS];       // H^T   [hidden of the wave's tile][pair]
    __shared__ __attribute__((aligned(16))) __bf16 ght[8][32 * RS];      // gH^T
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // this wave's rows of W1 and of W2^T as A operands (k = features / outputs)
    bf16x8 a1[4], a2[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        a1[s] = *reinterpret_cast<const bf16x8*>(w1 + (wave * 32 + r) * BF + 16 * s + 8 * h);
        a2[s] = *reinterpret_cast<const bf16x8*>(w2t + (wave * 32 + r) * BF + 16 * s + 8 * h);
    }
    float bias1[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) bias1[q] = b1[wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * h];
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
        bf16x8 bx[4], bg[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bx[s] = row8_bf16(xr + 16 * s + 8 * h);
            bg[s] = row8_bf16(gr + 16 * s + 8 * h);
            if (!valid) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { bx[s][j] = (__bf16)0.0f; bg[s][j] = (__bf16)0.0f; }
            }
        }
        // X^T and gY^T tiles -> LDS (waves 0 / 1; every wave holds the same values): element [16 s + 8 h + j][pair r]
        if (wave < 2) {
            __bf16* dst = wave == 0 ? xt : gyt;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) dst[(16 * s + 8 * h + j) * RS + r] = (wave == 0 ? bx[s][j] : bg[s][j]);
            }
        }
        // pre^T and gH^T of this wave's hidden tile: register q = hidden row (q&3) + 8 (q>>2) + 4 h, column = pair r
        f32x16 pre, gh;
#pragma unroll
        for (int q = 0; q < 16; ++q) { pre[q] = 0.f; gh[q] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            pre = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[s], bx[s], pre, 0, 0, 0);
            gh = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[s], bg[s], gh, 0, 0, 0);
        }
        __bf16* hrow = ht[wave];
        __bf16* grow = ght[wave];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int hq = (q & 3) + 8 * (q >> 2) + 4 * h;
            const float p = pre[q] + bias1[q];
            const bool on = valid && p > 0.0f;
            const float hv = on ? p : 0.0f;
            const float gv = on ? gh[q] : 0.0f;
            gb1[q] += gv;
            hrow[hq * RS + r] = (__bf16)hv;
            grow[hq * RS + r] = (__bf16)gv;
        }
        __syncthreads();
        // gW1[32w + m][32 nt + r] += sum_pairs gH^T[m][pair] X[pair][32 nt + r]:  A = gH^T rows, B[k = pair][n] = X^T[n][pair]
        // gW2[32 mt + m][32w + r] += sum_pairs gY^T[32 mt + m][pair] H[pair][32w + r]: A = gY^T rows, B = H^T[r][pair]
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ko = 16 * s + 8 * h;           // 8 consecutive pairs
            const bf16x8 a_gh = *reinterpret_cast<const bf16x8*>(grow + r * RS + ko);
            const bf16x8 b_h = *reinterpret_cast<const bf16x8*>(hrow + r * RS + ko);
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const bf16x8 b_x = *reinterpret_cast<const bf16x8*>(xt + (32 * t2 + r) * RS + ko);
                const bf16x8 a_gy = *reinterpret_cast<const bf16x8*>(gyt + (32 * t2 + r) * RS + ko);
                gw1[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_gh, b_x, gw1[t2], 0, 0, 0);
                gw2[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_gy, b_h, gw2[t2], 0, 0, 0);
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
    hipLaunchKernelGGL(k_mlp64_bwd_bf16, dim3((unsigned)grid), dim3(BB), 0, (hipStream_t)stream, x, gy, n,
                       (const __bf16*)w1_bf16, b1, (const __bf16*)w2t_bf16, partial);
    TPNET_HIP_TRY(hipGetLastError());
    return grid;
}

}  // extern "C"
