// gfx950 kernels of the standalone pairwise readout: tpnet_pair_gram / tpnet_pair_gram_shared
// (models/TPNet.py:112-128 before self.mlp).
#include "readout.hpp"

namespace tpnet {

static constexpr int RB = BLOCK_SMALL;   // threads per workgroup of the standalone readout kernels

template <int LPP, int VPL, int W, int L, bool FULL>
__global__ __launch_bounds__(RB) void k_pair_gram(tpnet_state S, const int64_t* __restrict__ u,
                                                     const int64_t* __restrict__ v, int64_t n, double now,
                                                     double lambda, uint32_t flags, float* __restrict__ out) {
    constexpr int GPB = RB / LPP;
    constexpr int NG = GramCfg<LPP, L>::NG;
    const int gl = threadIdx.x % LPP;
    const int g = threadIdx.x / LPP;
    const bool packed = (flags & TPNET_FLAG_PACKED) != 0;
    const bool do_scale = !(flags & TPNET_FLAG_NOT_SCALE) && !packed;
    const int stride = packed ? GramCfg<LPP, L>::NT : NG;
    __shared__ float stage[GramCfg<LPP, L>::template stage_floats<RB>()];
    constexpr bool LR = GramCfg<LPP, L>::template lds_reduce<RB>();
    for (int64_t base = (int64_t)blockIdx.x * GPB; base < n; base += (int64_t)gridDim.x * GPB) {
        const int64_t p = base + g;
        const bool valid = p < n;
        const int64_t uu = valid ? u[p] : 0, vv = valid ? v[p] : 0;
        if (packed)
            gram_pair<LPP, VPL, W, L, FULL, true, false, LR>(S, uu, vv, valid, READER_BID, now, lambda, do_scale,
                                                             out + p * stride, gl, nullptr, stage);
        else
            gram_pair<LPP, VPL, W, L, FULL, false, false, LR>(S, uu, vv, valid, READER_BID, now, lambda, do_scale,
                                                              out + p * stride, gl, nullptr, stage);
    }
}


template <int LPP, int VPL, int W, int L, bool FULL>
__global__ __launch_bounds__(RB) void k_pair_gram_shared(tpnet_state S, const int64_t* __restrict__ u,
                                                            const int64_t* __restrict__ v1,
                                                            const int64_t* __restrict__ v2, int64_t n, double now,
                                                            double lambda, uint32_t flags, float* __restrict__ out1,
                                                            float* __restrict__ out2) {
    constexpr int GPB = RB / LPP;
    constexpr int NG = GramCfg<LPP, L>::NG;
    const int gl = threadIdx.x % LPP;
    const int g = threadIdx.x / LPP;
    const bool do_scale = !(flags & TPNET_FLAG_NOT_SCALE);
    for (int64_t base = (int64_t)blockIdx.x * GPB; base < n; base += (int64_t)gridDim.x * GPB) {
        const int64_t p = base + g;
        const bool valid = p < n;
        const int64_t uu = valid ? u[p] : 0, a1 = valid ? v1[p] : 0, a2 = valid ? v2[p] : 0;
        gram_shared<LPP, VPL, W, L, FULL>(S, uu, a1, a2, valid, READER_BID, now, lambda, do_scale, out1 + p * NG,
                                          out2 + p * NG, gl);
    }
}


int launch_pair_gram(const tpnet_state& st, const int64_t* u, const int64_t* v, int64_t n, double now, double lambda,
                     uint32_t flags, float* out, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    if (reinterpret_cast<uintptr_t>(out) & 15) return TPNET_ERR_BAD_ARG;   // feature rows leave as 16-byte vectors
    TPNET_DISPATCH_G(pick_geom(st.d, n >= 16384), ({
        const int grid = grid_for(n, RB / LPP, 256 * 16);
        hipLaunchKernelGGL((k_pair_gram<LPP, VPL, W, L, FULL>), dim3(grid), dim3(RB), 0, s, st, u, v, n, now, lambda,
                           flags, out);
    }));
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

// resident workgroups of a kernel on this device: occupancy (per CU, from the runtime) x CU count

int launch_pair_gram_shared(const tpnet_state& st, const int64_t* u, const int64_t* v1, const int64_t* v2, int64_t n,
                            double now, double lambda, uint32_t flags, float* out1, float* out2, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    TPNET_DISPATCH(({
        const int grid = grid_for(n, RB / LPP, 256 * 16);
        hipLaunchKernelGGL((k_pair_gram_shared<LPP, VPL, W, L, FULL>), dim3(grid), dim3(RB), 0, s, st, u, v1, v2, n,
                           now, lambda, flags, out1, out2);
    }));
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}


// rows of the encoder's call (models/TPNet.py:311-324): one lane group per row, anchors in registers for its K neighbours
template <int LPP, int VPL, int W, int L, bool FULL>
__global__ __launch_bounds__(RB) void k_pair_gram_anchored(tpnet_state S, const int64_t* __restrict__ neigh,
                                                           const int64_t* __restrict__ a1, const int64_t* __restrict__ a2,
                                                           int64_t n_rows, int K, int KC, double now, double lambda,
                                                           uint32_t flags, float* __restrict__ out1,
                                                           float* __restrict__ out2) {
    constexpr int GPB = RB / LPP;
    constexpr int NG = GramCfg<LPP, L>::NG;
    const int gl = threadIdx.x % LPP;
    const int g = threadIdx.x / LPP;
    const bool do_scale = !(flags & TPNET_FLAG_NOT_SCALE);
    // unit = (row, chunk of KC neighbours): a long list of rows takes KC = K (anchors fetched once per row); few rows are cut
    // into chunks so that the launch still fills the chip
    const int nch = (K + KC - 1) / KC;
    const int64_t units = n_rows * nch;
    if constexpr (FULL) {
        for (int64_t base = (int64_t)blockIdx.x * GPB; base < units; base += (int64_t)gridDim.x * GPB) {
            const int64_t un = base + g;
            const bool valid = un < units;
            const int64_t rr = valid ? un / nch : 0;
            const int ch = valid ? (int)(un - rr * nch) : 0;
            const int kb = ch * KC, ke = (kb + KC < K) ? kb + KC : K;
            gram_anchored<LPP, VPL, W, L, FULL>(S, neigh + rr * K, valid ? a1[rr] : 0, valid ? a2[rr] : 0, kb, ke, valid, now,
                                                lambda, do_scale, out1 + rr * K * NG, out2 + rr * K * NG, gl);
        }
    }
}

bool pair_gram_anchored_supported(const tpnet_state& st) {
    const Geom gm = pick_geom(st.d);
    return gm.w == 4 && st.d == gm.lpp * gm.vpl * 4 && gm.lpp >= 16;      // rows of exactly one chunk of 16-byte vectors
}

int launch_pair_gram_anchored(const tpnet_state& st, const int64_t* neigh, const int64_t* a1, const int64_t* a2,
                              int64_t n_rows, int K, double now, double lambda, uint32_t flags, float* out1, float* out2,
                              hipStream_t s) {
    if (n_rows == 0 || K == 0) return TPNET_OK;
    if (!pair_gram_anchored_supported(st)) return TPNET_ERR_BAD_ARG;
    // (measured and not kept, round 3: the 16-lane x 2-vector geometry for rows of 128 floats -- 80 000 pairs 40.4 us against 35.3)
    TPNET_DISPATCH(({
        if constexpr (FULL && LPP >= 16) {
            static const int kc_env = TPNET_DEV_INT(ANCHOR_KC, 0);
            int64_t kc = ((int64_t)K * n_rows + 8191) / 8192;            // >= ~8192 units in the launch
            kc = kc < 4 ? 4 : (kc > K ? K : kc);
            if (kc_env > 0) kc = kc_env > K ? K : kc_env;
            const int64_t units = n_rows * ((K + kc - 1) / kc);
            const int grid = grid_for(units, RB / LPP, 256 * 16);
            hipLaunchKernelGGL((k_pair_gram_anchored<LPP, VPL, W, L, FULL>), dim3(grid), dim3(RB), 0, s, st, neigh, a1, a2,
                               n_rows, K, (int)kc, now, lambda, flags, out1, out2);
        } else {
            return TPNET_ERR_BAD_ARG;
        }
    }));
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}


}  // namespace tpnet
