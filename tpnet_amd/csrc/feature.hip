// get_pair_wise_feature in ONE launch (models/TPNet.py:112-129): the Gram of the 2(L+1) rows of every pair, x<0 -> 0,
// log(x+1), AND self.mlp = Linear(F, 4F) -> ReLU -> Linear(4F, F), F = (2L+2)^2 (models/TPNet.py:63-65), in fp32.
// The features of a workgroup's pairs stay in LDS between the two halves: they never round-trip HBM, and the per-batch
// call of the reference's training loop (train_link_prediction.py:359-368, models/modules.py:112) is one kernel instead
// of the readout + three dense-layer launches.  fp32 on the vector ALUs (fmaf): within fp32 rounding of the torch layers,
// any L; meant for the short pair lists of the decoder (a tile of 8 pairs re-reads the 2 * 4F * F weights from L2, which
// is the right trade for <= ~10^4 pairs; long lists take the readout kernels + the bf16 matrix-core mlp, mlp.hip).
#include "readout.hpp"

#include <cstdlib>

namespace tpnet {

static constexpr int FB = 256;        // threads per workgroup
static constexpr int FSUB = 8;        // pairs per pass of the dense layers

template <int LPP, int VPL, int W, int L, bool FULL, int SUBP>
__global__ __launch_bounds__(FB) void k_pair_feature(tpnet_state S, const int64_t* __restrict__ u,
                                                     const int64_t* __restrict__ v, int64_t n, double now, double lambda,
                                                     uint32_t flags, tpnet_mlp M, float* __restrict__ out_gram,
                                                     float* __restrict__ out, int ppb_arg) {
#ifdef TPNET_DEV
    const int ppb = ppb_arg & 0xFF, dbg_skip = ppb_arg >> 8;       // diagnostic builds: stop after a phase (timing brackets)
#else
    const int ppb = ppb_arg;
#endif
    using C = GramCfg<LPP, L>;
    constexpr int GPB = FB / LPP;                 // pairs per pass of the readout
    constexpr int NG = C::NG, H = 4 * NG;
    constexpr int SUB = GPB < SUBP ? GPB : SUBP;     // pairs per pass of the dense layers (and per workgroup when the list is short)
    constexpr int KQ = FB / NG;                   // layer 2: the hidden axis is cut into KQ slices, one per group of NG threads
    constexpr int KCH = (H + KQ - 1) / KQ;        // hidden units per slice
    // (the readout's reduction through LDS only where its tile and the dense layers' tiles fit 64 KB together)
    constexpr int MLP_WORDS = GPB * NG + SUB * H + KQ * SUB * NG;
    constexpr bool LR = C::template lds_reduce<FB>() && (C::template stage_floats<FB>() + MLP_WORDS) * 4 <= 64 * 1024;
    __shared__ __attribute__((aligned(16))) float stage[(LR || LPP < 16) ? C::template stage_floats<FB>() : 1];
    __shared__ __attribute__((aligned(16))) float feat[GPB * NG];
    __shared__ __attribute__((aligned(16))) float hid[SUB * H];          // [hidden unit][pair of the pass]
    __shared__ __attribute__((aligned(16))) float part[KQ * SUB * NG];   // [slice][pair][output]
#ifdef TPNET_STAMPS
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(out_gram);   // diagnostic build: out_gram is the stamp buffer
    out_gram = nullptr;
#endif
    const int tid = threadIdx.x;
    const int gl = tid % LPP, g = tid / LPP;
    STAMP(0);
    const bool do_scale = !(flags & TPNET_FLAG_NOT_SCALE);
    const float* __restrict__ w1t = M.w1t;        // [NG][H]: w1t[k][j] = mlp[0].weight[j][k]
    const float* __restrict__ w2t = M.w2t;        // [H][NG]: w2t[k][o] = mlp[2].weight[o][k]
    // L = 3 (H == FB): thread j IS hidden unit j and thread (kq, o) owns one slice of one output, so a thread's 64 + 64 weights
    // are two bursts of independent loads into registers, issued ONCE per workgroup right behind its first readout (both
    // bursts before the first layer's arithmetic, so the second one's round trip hides behind it) and kept for every further
    // tile.  (Issued ahead of the first readout they queue up in front of its own loads -- vector loads return in order:
    // C2, 1000 pairs, 17.6 us against 13.9 us.)
    constexpr bool WREG = (H == FB) && (KQ * NG == FB);
    float w1r[WREG ? NG : 1], w2r[WREG ? KCH : 1];
    float b1r = 0.0f;
    // ppb = pairs per workgroup and pass (<= GPB): a short list is spread over more workgroups with idle lane groups rather
    // than over few full ones (200 pairs at d=64: 13 workgroups of 16 pairs and two dense passes each, or 25 of 8 and one)
    // (Measured and not kept: the first tile's ids, then the weights, then the readout -- the compiler sinks the second burst
    // to its first use whatever the scheduling barriers say, and the kernel is 12.4 us against 11.8.)
    bool have_w = false;
    for (int64_t base = (int64_t)blockIdx.x * ppb; base < n; base += (int64_t)gridDim.x * ppb) {
        const int64_t p = base + g;
        const bool valid = g < ppb && p < n;
        const int64_t uu = valid ? u[p] : 0, vv = valid ? v[p] : 0;
        gram_pair<LPP, VPL, W, L, FULL, false, false, LR>(S, uu, vv, valid, READER_BID, now, lambda, do_scale,
                                                          feat + g * NG, gl, nullptr, stage);
        STAMP(1);
        if constexpr (WREG) {
            if (!have_w) {
                const int kq = tid / NG, o = tid - kq * NG;
#pragma unroll
                for (int k = 0; k < NG; ++k) w1r[k] = w1t[k * H + tid];
#pragma unroll
                for (int i = 0; i < KCH; ++i) w2r[i] = w2t[(kq * KCH + i) * NG + o];
                b1r = M.b1[tid];
                have_w = true;
            }
        }
        __syncthreads();
        STAMP(2);
        const int npair = (n - base < ppb) ? (int)(n - base) : ppb;
        if (out_gram) {                           // the pre-mlp features, for a backward pass (training)
            for (int i = tid; i < npair * NG; i += FB) out_gram[base * NG + i] = feat[i];
        }
#ifdef TPNET_DEV
        if (dbg_skip == 1) { for (int i = tid; i < npair * NG; i += FB) out[base * NG + i] = feat[i]; __syncthreads(); continue; }
        if (dbg_skip == 2) { float a_ = b1r; if constexpr (WREG) { for (int k = 0; k < NG; ++k) a_ += w1r[k] + w2r[k]; } if (tid < npair * NG) out[base * NG + tid] = a_ + feat[tid]; __syncthreads(); continue; }
#endif
        for (int p0 = 0; p0 < npair; p0 += SUB) {
            // ---- hidden = relu(W1 f + b1): thread j owns hidden unit j for the SUB pairs of this pass.  Its NG weights are
            // fetched in bursts of 32 independent loads (a k loop with a load per step is a chain of L2 round trips)
            if constexpr (WREG) {
                float acc[SUB];
#pragma unroll
                for (int q = 0; q < SUB; ++q) acc[q] = 0.0f;
#pragma unroll
                for (int k = 0; k < NG; ++k) {
#pragma unroll
                    for (int q = 0; q < SUB; ++q) acc[q] = fmaf(feat[(p0 + q) * NG + k], w1r[k], acc[q]);
                    if (k % 8 == 7) __builtin_amdgcn_sched_barrier(0);     // (keeps the LDS reads of later k out of the live set)
                }
#pragma unroll
                for (int q = 0; q < SUB; ++q) {
                    const float h = acc[q] + b1r;
                    hid[tid * SUB + q] = (h < 0.0f) ? 0.0f : h;      // (NaN < 0 is false: NaN passes, as torch.relu)
                }
            } else
            for (int j = tid; j < H; j += FB) {
                const float b = M.b1[j];
                float acc[SUB];
#pragma unroll
                for (int q = 0; q < SUB; ++q) acc[q] = 0.0f;
                constexpr int W1CH = 32;          // weights per burst
#pragma unroll 1
                for (int kb = 0; kb < NG; kb += W1CH) {
                    float w[W1CH];
#pragma unroll
                    for (int i = 0; i < W1CH; ++i) w[i] = (kb + i < NG) ? w1t[(kb + i) * H + j] : 0.0f;
#pragma unroll
                    for (int i = 0; i < W1CH; ++i) {
                        const int k = (kb + i < NG) ? kb + i : 0;
#pragma unroll
                        for (int q = 0; q < SUB; ++q) acc[q] = fmaf(feat[(p0 + q) * NG + k], w[i], acc[q]);
                    }
                }
#pragma unroll
                for (int q = 0; q < SUB; ++q) {
                    const float h = acc[q] + b;
                    hid[j * SUB + q] = (h < 0.0f) ? 0.0f : h;        // (NaN < 0 is false: NaN passes, as torch.relu)
                }
            }
            __syncthreads();
            STAMP(3);
            // ---- out = W2 hidden + b2: thread (slice kq, output o) sums its slice of the hidden axis for all SUB pairs,
            // weights again in bursts of up to 32 independent loads; the KQ partial sums meet in LDS
            if constexpr (WREG) {
                const int kq = tid / NG, o = tid - kq * NG;
                float acc[SUB];
#pragma unroll
                for (int q = 0; q < SUB; ++q) acc[q] = 0.0f;
#pragma unroll
                for (int i = 0; i < KCH; ++i) {
#pragma unroll
                    for (int q = 0; q < SUB; ++q) acc[q] = fmaf(hid[(kq * KCH + i) * SUB + q], w2r[i], acc[q]);
                    if (i % 8 == 7) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int q = 0; q < SUB; ++q) part[(kq * SUB + q) * NG + o] = acc[q];
            } else if (tid < KQ * NG) {
                const int kq = tid / NG, o = tid - kq * NG;
                const int k0 = kq * KCH, k1 = (k0 + KCH < H) ? k0 + KCH : H;
                float acc[SUB];
#pragma unroll
                for (int q = 0; q < SUB; ++q) acc[q] = 0.0f;
                constexpr int WCH = KCH < 32 ? KCH : 32;
#pragma unroll 1
                for (int kb = k0; kb < k1; kb += WCH) {
                    float w[WCH];
#pragma unroll
                    for (int i = 0; i < WCH; ++i) w[i] = (kb + i < k1) ? w2t[(kb + i) * NG + o] : 0.0f;
#pragma unroll
                    for (int i = 0; i < WCH; ++i) {
                        const int k = (kb + i < k1) ? kb + i : k0;
#pragma unroll
                        for (int q = 0; q < SUB; ++q) acc[q] = fmaf(hid[k * SUB + q], w[i], acc[q]);
                    }
                }
#pragma unroll
                for (int q = 0; q < SUB; ++q) part[(kq * SUB + q) * NG + o] = acc[q];
            }
            __syncthreads();
            STAMP(4);
            for (int idx = tid; idx < SUB * NG; idx += FB) {
                const int q = idx / NG, o = idx - q * NG;
                float sum = part[idx];
#pragma unroll
                for (int kq = 1; kq < KQ; ++kq) sum += part[kq * SUB * NG + idx];
                if (p0 + q < npair) out[(base + p0 + q) * NG + o] = sum + M.b2[o];
            }
            __syncthreads();
            STAMP(5);
        }
    }
}

int launch_pair_feature(const tpnet_state& st, const int64_t* u, const int64_t* v, int64_t n, double now, double lambda,
                        uint32_t flags, const tpnet_mlp& m, float* out_gram, float* out, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    const int NN = 2 * st.L + 2;
    if (m.F != NN * NN || m.H != 4 * NN * NN || !m.w1t || !m.b1 || !m.w2t || !m.b2) return TPNET_ERR_BAD_ARG;
    if (flags & TPNET_FLAG_PACKED) return TPNET_ERR_BAD_ARG;
    // L = 3 on rows of whole 16-byte vectors: the fp32 matrix-core kernel (feature_mfma.hip) from `mfma_from` pairs on -- its
    // weights live in registers for the whole launch, so lists of any length are one launch; below that the vector-ALU
    // kernel's 8-pair workgroups spread a short list over more CUs
    static const int64_t mfma_from = (int64_t)TPNET_DEV_INT(FEATURE_MFMA_FROM, 2048);
    if (m.w1 && m.w2f && n >= mfma_from && pair_feature_mfma_supported(st) && !(reinterpret_cast<uintptr_t>(out) & 15))
        return launch_pair_feature_bf16(st, u, v, n, now, lambda, flags, m.w1, m.b1, m.w2f, m.b2, out_gram, out, s, mlp_f32_mode());
    // pairs per workgroup: a short list is spread over as many CUs as it has tiles of 2 / 4 / 8 pairs -- a workgroup's dense
    // layers are 2 * 64 * 256 fmaf per pair on ONE wave per SIMD plus 128 KB of weights through its CU's L2 port, so the call's
    // latency falls with the tile until the launch fills the chip (C2, 1000 pairs: 13.9 us with 8 pairs per workgroup)
    static const int sub_dev = TPNET_DEV_INT(FEATURE_SUB, 0);
    static const int skip_dev = TPNET_DEV_INT(FEATURE_SKIP, 0);
    const int sub = sub_dev ? sub_dev : (n <= 512 ? 2 : n <= 1536 ? 4 : 8);
#define TPNET_FEATURE_LAUNCH(SUBP_)                                                                                          \
    TPNET_DISPATCH(({                                                                                                        \
        constexpr int GPB = FB / LPP;                                                                                        \
        const int ppb = GPB < SUBP_ ? GPB : ((SUBP_ < FSUB || n <= 256 * FSUB) ? SUBP_ : GPB);                                   \
        const int grid = grid_for(n, ppb, 256 * 8);                                                                          \
        hipLaunchKernelGGL((k_pair_feature<LPP, VPL, W, L, FULL, SUBP_>), dim3(grid), dim3(FB), 0, s, st, u, v, n, now, lambda, \
                           flags, m, out_gram, out, ppb | (skip_dev << 8));                                                  \
    }))
    if (sub == 2) TPNET_FEATURE_LAUNCH(2);
    else if (sub == 4) TPNET_FEATURE_LAUNCH(4);
    else TPNET_FEATURE_LAUNCH(FSUB);
#undef TPNET_FEATURE_LAUNCH
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

}  // namespace tpnet
