// get_pair_wise_feature in ONE launch (models/TPNet.py:112-129): the Gram of the 2(L+1) rows of every pair, x<0 -> 0,
// log(x+1), AND self.mlp = Linear(F, 4F) -> ReLU -> Linear(4F, F), F = (2L+2)^2 (models/TPNet.py:63-65), in fp32.
// The features of a workgroup's pairs stay in LDS between the two halves: they never round-trip HBM, and the per-batch
// call of the reference's training loop (train_link_prediction.py:359-368, models/modules.py:112) is one kernel instead
// of the readout + three dense-layer launches.  fp32 on the vector ALUs (fmaf): within fp32 rounding of the torch layers,
// any L; meant for the short pair lists of the decoder (a tile of 8 pairs re-reads the 2 * 4F * F weights from L2, which
// is the right trade for <= ~10^4 pairs; long lists take the readout kernels + the bf16 matrix-core mlp, mlp.hip).
#include "readout.hpp"

namespace tpnet {

static constexpr int FB = 256;        // threads per workgroup
static constexpr int FSUB = 8;        // pairs per pass of the dense layers

template <int LPP, int VPL, int W, int L, bool FULL>
__global__ __launch_bounds__(FB) void k_pair_feature(tpnet_state S, const int64_t* __restrict__ u,
                                                     const int64_t* __restrict__ v, int64_t n, double now, double lambda,
                                                     uint32_t flags, tpnet_mlp M, float* __restrict__ out_gram,
                                                     float* __restrict__ out) {
    using C = GramCfg<LPP, L>;
    constexpr int GPB = FB / LPP;                 // pairs per pass of the readout
    constexpr int NG = C::NG, H = 4 * NG;
    constexpr int SUB = GPB < FSUB ? GPB : FSUB;
    // (the readout's reduction through LDS only where its tile, the feature tile and the hidden tile fit 64 KB together)
    constexpr bool LR = C::template lds_reduce<FB>() &&
                        (C::template stage_floats<FB>() + GPB * NG + SUB * H) * 4 <= 64 * 1024;
    __shared__ __attribute__((aligned(16))) float stage[(LR || LPP < 16) ? C::template stage_floats<FB>() : 1];
    __shared__ __attribute__((aligned(16))) float feat[GPB * NG];
    __shared__ __attribute__((aligned(16))) float hid[SUB * H];
    const int tid = threadIdx.x;
    const int gl = tid % LPP, g = tid / LPP;
    const bool do_scale = !(flags & TPNET_FLAG_NOT_SCALE);
    const float* __restrict__ w1t = M.w1t;        // [NG][H]: w1t[k][j] = mlp[0].weight[j][k]
    const float* __restrict__ w2t = M.w2t;        // [H][NG]: w2t[k][o] = mlp[2].weight[o][k]
    for (int64_t base = (int64_t)blockIdx.x * GPB; base < n; base += (int64_t)gridDim.x * GPB) {
        const int64_t p = base + g;
        const bool valid = p < n;
        const int64_t uu = valid ? u[p] : 0, vv = valid ? v[p] : 0;
        gram_pair<LPP, VPL, W, L, FULL, false, false, LR>(S, uu, vv, valid, READER_BID, now, lambda, do_scale,
                                                          feat + g * NG, gl, nullptr, stage);
        __syncthreads();
        const int npair = (n - base < GPB) ? (int)(n - base) : GPB;
        if (out_gram) {                           // the pre-mlp features, for a backward pass (training)
            for (int i = tid; i < npair * NG; i += FB) out_gram[base * NG + i] = feat[i];
        }
        for (int p0 = 0; p0 < npair; p0 += SUB) {
            // ---- hidden = relu(W1 f + b1): thread j owns hidden unit j for the SUB pairs of this pass
            for (int j = tid; j < H; j += FB) {
                float acc[SUB];
#pragma unroll
                for (int q = 0; q < SUB; ++q) acc[q] = 0.0f;
#pragma unroll 4
                for (int k = 0; k < NG; ++k) {
                    const float w = w1t[k * H + j];
#pragma unroll
                    for (int q = 0; q < SUB; ++q) acc[q] = fmaf(feat[(p0 + q) * NG + k], w, acc[q]);
                }
                const float b = M.b1[j];
#pragma unroll
                for (int q = 0; q < SUB; ++q) {
                    const float h = acc[q] + b;
                    hid[q * H + j] = (h < 0.0f) ? 0.0f : h;          // (NaN < 0 is false: NaN passes, as torch.relu)
                }
            }
            __syncthreads();
            // ---- out = W2 hidden + b2: one output element per thread and round
            for (int idx = tid; idx < SUB * NG; idx += FB) {
                const int q = idx / NG, o = idx - q * NG;
                float acc = 0.0f;
#pragma unroll 8
                for (int k = 0; k < H; ++k) acc = fmaf(hid[q * H + k], w2t[k * NG + o], acc);
                if (p0 + q < npair) out[(base + p0 + q) * NG + o] = acc + M.b2[o];
            }
            __syncthreads();
        }
    }
}

int launch_pair_feature(const tpnet_state& st, const int64_t* u, const int64_t* v, int64_t n, double now, double lambda,
                        uint32_t flags, const tpnet_mlp& m, float* out_gram, float* out, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    const int NN = 2 * st.L + 2;
    if (m.F != NN * NN || m.H != 4 * NN * NN || !m.w1t || !m.b1 || !m.w2t || !m.b2) return TPNET_ERR_BAD_ARG;
    if (flags & TPNET_FLAG_PACKED) return TPNET_ERR_BAD_ARG;
    TPNET_DISPATCH(({
        const int grid = grid_for(n, FB / LPP, 256 * 8);
        hipLaunchKernelGGL((k_pair_feature<LPP, VPL, W, L, FULL>), dim3(grid), dim3(FB), 0, s, st, u, v, n, now, lambda,
                           flags, m, out_gram, out);
    }));
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

}  // namespace tpnet
