// SURVEY.md §8 f-1 / BASELINE config 5: the dense layer pair right behind the pairwise readout, self.mlp =
// Linear(64,256) -> ReLU -> Linear(256,64) (models/TPNet.py:64-65,129), as ONE kernel on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate).  This IS a dense GEMM (65.5 kflop per pair), the only MFMA use on
// the path.  The hidden activations never leave registers: layer 1 is computed transposed, H^T = W1 . X^T, so a
// 32x32 result tile has the pair on the lane and the hidden index in the registers -- exactly the B-operand shape
// of layer 2, Y^T = W2 . H^T, which sums over that index (cdna_hip_programming.md §3 'An accumulator tile as the
// next MFMA's operand'); W2 is pre-permuted on the host to the k order that operand implies.
// Only the (2L+2)^2 = 64 feature case (L = 3, the reference default) has an MFMA kernel; other L keep torch.
#include "tpnet_common.h"

namespace tpnet {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

static constexpr int MLP_F = 64, MLP_H = 256;
static constexpr int W1_STRIDE = MLP_F + 8;    // bf16 elements per LDS row: 144 B, ds_read_b128 conflict-free
static constexpr int W2_STRIDE = MLP_H + 8;    // 528 B
static constexpr int MLP_BLOCK = 256;

// X [n][64] f32 -> Y [n][64] f32.  w1: bf16 [256][64] (torch weight layout [out][in]); w2p: bf16 [64][256] with the
// hidden index of every 32-tile permuted to the accumulator order (see tpnet_amd/fused_mlp.py::permute_w2).
__global__ __launch_bounds__(MLP_BLOCK) void k_mlp64_bf16(const float* __restrict__ X, int64_t n,
                                                          const __bf16* __restrict__ w1, const float* __restrict__ b1,
                                                          const __bf16* __restrict__ w2p, const float* __restrict__ b2,
                                                          float* __restrict__ Y) {
    __shared__ __attribute__((aligned(16))) __bf16 w1s[MLP_H * W1_STRIDE];
    __shared__ __attribute__((aligned(16))) __bf16 w2s[MLP_F * W2_STRIDE];
    // stage the weights: 16-byte pieces
    for (int v = threadIdx.x; v < MLP_H * MLP_F / 8; v += MLP_BLOCK) {
        const int row = v / (MLP_F / 8), c = v % (MLP_F / 8);
        *reinterpret_cast<bf16x8*>(&w1s[row * W1_STRIDE + c * 8]) = reinterpret_cast<const bf16x8*>(w1)[v];
    }
    for (int v = threadIdx.x; v < MLP_F * MLP_H / 8; v += MLP_BLOCK) {
        const int row = v / (MLP_H / 8), c = v % (MLP_H / 8);
        *reinterpret_cast<bf16x8*>(&w2s[row * W2_STRIDE + c * 8]) = reinterpret_cast<const bf16x8*>(w2p)[v];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t ntiles = (n + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * (MLP_BLOCK / 64) + wave; tile < ntiles;
         tile += (int64_t)gridDim.x * (MLP_BLOCK / 64)) {
        const int64_t pair = tile * 32 + r;
        const bool valid = pair < n;
        const float* xr = X + (valid ? pair : 0) * MLP_F;
        // B operand of layer 1: X^T, k-step s, lane (r, h) holds X[pair r][16 s + 8 h + j]
        bf16x8 bx[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 lo = *reinterpret_cast<const float4*>(xr + 16 * s + 8 * h);
            const float4 hi = *reinterpret_cast<const float4*>(xr + 16 * s + 8 * h + 4);
            bx[s][0] = (__bf16)lo.x; bx[s][1] = (__bf16)lo.y; bx[s][2] = (__bf16)lo.z; bx[s][3] = (__bf16)lo.w;
            bx[s][4] = (__bf16)hi.x; bx[s][5] = (__bf16)hi.y; bx[s][6] = (__bf16)hi.z; bx[s][7] = (__bf16)hi.w;
        }
        f32x16 y0, y1;
#pragma unroll
        for (int q = 0; q < 16; ++q) { y0[q] = 0.0f; y1[q] = 0.0f; }
#pragma unroll 2
        for (int ht = 0; ht < MLP_H / 32; ++ht) {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&w1s[(ht * 32 + r) * W1_STRIDE + 16 * s + 8 * h]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bx[s], acc, 0, 0, 0);
            }
            // register q of the tile = hidden row ht*32 + (q&3) + 8*(q>>2) + 4*h, column = pair r
            bf16x8 bh[2];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float v = acc[q] + b1[ht * 32 + (q & 3) + 8 * (q >> 2) + 4 * h];
                v = v > 0.0f ? v : 0.0f;
                bh[q >> 3][q & 7] = (__bf16)v;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int off = ht * 32 + 16 * s2 + 8 * h;       // position inside the PERMUTED hidden axis
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&w2s[r * W2_STRIDE + off]);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&w2s[(32 + r) * W2_STRIDE + off]);
                y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bh[s2], y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bh[s2], y1, 0, 0, 0);
            }
        }
        if (valid) {
            float* yr = Y + pair * MLP_F;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = 8 * i + 4 * h;                      // output rows (q&3) + 8*(q>>2) + 4*h, q = 4i..4i+3
                const float4 bb0 = *reinterpret_cast<const float4*>(b2 + o);
                const float4 bb1 = *reinterpret_cast<const float4*>(b2 + 32 + o);
                *reinterpret_cast<float4*>(yr + o) =
                    make_float4(y0[4 * i] + bb0.x, y0[4 * i + 1] + bb0.y, y0[4 * i + 2] + bb0.z, y0[4 * i + 3] + bb0.w);
                *reinterpret_cast<float4*>(yr + 32 + o) =
                    make_float4(y1[4 * i] + bb1.x, y1[4 * i + 1] + bb1.y, y1[4 * i + 2] + bb1.z, y1[4 * i + 3] + bb1.w);
            }
        }
    }
}

}  // namespace tpnet

using namespace tpnet;

extern "C" int tpnet_mlp64_bf16(const float* x, int64_t n, const void* w1_bf16, const float* b1, const void* w2p_bf16,
                                const float* b2, float* y, void* stream) {
    if (n < 0 || (n > 0 && (!x || !w1_bf16 || !b1 || !w2p_bf16 || !b2 || !y))) return TPNET_ERR_BAD_ARG;
    if (n == 0) return TPNET_OK;
    const int64_t tiles = (n + 31) / 32;
    int64_t grid = (tiles + 3) / 4;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_mlp64_bf16, dim3((unsigned)grid), dim3(MLP_BLOCK), 0, (hipStream_t)stream, x, n,
                       (const __bf16*)w1_bf16, b1, (const __bf16*)w2p_bf16, b2, y);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}
