// The encoder's call (models/TPNet.py:311-324, 129) end to end in ONE C call: anchored readout, then self.mlp in the fp32 class
// (the dense-layer kernel of feature_mfma.hip) on the same stream.
// Measured and not kept (round 3): the dense layers of a chunk of rows on a helper stream BESIDE the readout of the next chunk.
// Side by side the two kernels slow each other more than the overlap wins -- 800 000 pairs at d=256: readout 355 us + dense
// layers 237 us = 592 us one after the other; 608 us in 2 chunks, 707 in 4, 752 in 8, 1 194 in 16 (tools/encoder_call_rate.py) --
// both live on the memory system (gathers there, a 410 MB tile stream here).  What would remove the dense layers' traffic is
// their fusion INTO the anchored kernel.
#include "tpnet_common.h"

namespace tpnet {

static int mlp_rows(const tpnet_mlp* mlp, const float* x, int64_t n, float* y, hipStream_t s) {
    // long lists: every wave its own tiles, split weights in LDS (mlp_x3.hip)
    if (n >= mlp_x3_from() && mlp_f32_mode() == 2 && mlp_x3_available() &&
        launch_mlp_rows_x3(x, n, reinterpret_cast<const float*>(mlp->w1), mlp->b1, reinterpret_cast<const float*>(mlp->w2f),
                           mlp->b2, y, s) == TPNET_OK)
        return TPNET_OK;
    tpnet_state st{};                      // (not dereferenced when the tile comes from `x`; geometry of d = 128 picks the 32-lane kernel)
    st.N = 1; st.d = 128; st.L = 3;
    return launch_pair_feature_bf16(st, nullptr, nullptr, n, 0.0, 0.0, 0, mlp->w1, mlp->b1, mlp->w2f, mlp->b2, nullptr, y, s,
                                    mlp_f32_mode(), x);
}

}  // namespace tpnet

using namespace tpnet;

extern "C" {

int tpnet_encoder_fused_supported(const tpnet_state* st, int64_t n_rows, int32_t K, const tpnet_mlp* mlp) {
    if (!st || !st->p0 || !st->q || !st->meta || st->N < 1 || st->d < 1) return 0;
    return encoder_fused_supported(*st, n_rows, K, mlp) ? 1 : 0;
}

int tpnet_anchored_features(const tpnet_state* st, const int64_t* neigh, const int64_t* a1, const int64_t* a2, int64_t n_rows,
                            int32_t K, double now_time, double lambda, uint32_t flags, const tpnet_mlp* mlp, float* gram,
                            float* out, void* stream) {
    if (!st || !st->p0 || !st->q || !st->meta || !st->err || st->N < 1 || st->d < 1 || st->L != 3) return TPNET_ERR_BAD_ARG;
    if (n_rows < 0 || K < 0 || (n_rows > 0 && K > 0 && (!neigh || !a1 || !a2 || !out || !mlp))) return TPNET_ERR_BAD_ARG;
    if (n_rows == 0 || K == 0) return TPNET_OK;
    if (flags & TPNET_FLAG_PACKED) return TPNET_ERR_BAD_ARG;
    if (mlp->F != 64 || mlp->H != 256 || !mlp->w1 || !mlp->w2f || !mlp->b1 || !mlp->b2) return TPNET_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(gram) | reinterpret_cast<uintptr_t>(out)) & 15) return TPNET_ERR_BAD_ARG;
    // readout + dense layers in ONE launch on the matrix cores (encoder_mfma.hip); a runtime that refuses it: two launches below
    if (!(flags & TPNET_FLAG_NO_MFMA_READOUT) && encoder_fused_supported(*st, n_rows, K, mlp) &&
        launch_encoder_fused(*st, neigh, a1, a2, n_rows, K, now_time, lambda, flags, mlp, gram, out, (hipStream_t)stream) == TPNET_OK)
        return TPNET_OK;
    if (!gram) return TPNET_ERR_NEED_GRAM;             // (the fused launch was refused after the caller had been told it may pass NULL)
    if (!pair_gram_anchored_supported(*st)) return TPNET_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t F = 64;
    const int64_t half = n_rows * (int64_t)K;                // pairs per anchor side
    int rc = tpnet_pair_gram_anchored(st, neigh, a1, a2, n_rows, K, now_time, lambda, flags, gram, gram + half * F, stream);
    if (rc) return rc;
    rc = mlp_rows(mlp, gram, 2 * half, out, s);
    if (rc) return rc;
    return TPNET_OK;
}

int tpnet_encoder_features(const tpnet_state* st, const void* sampler, int64_t E, int64_t num_nodes, const int64_t* src,
                           const int64_t* other, const double* t, int64_t B, int32_t K, double now_time, double lambda,
                           uint32_t flags, const tpnet_mlp* mlp, void* scratch, size_t scratch_bytes, float* gram, float* out,
                           void* stream) {
    if (!st || !sampler || B < 0 || K < 1 || (B > 0 && (!src || !other || !t || !scratch || !out || !mlp)))
        return TPNET_ERR_BAD_ARG;
    if (B == 0) return TPNET_OK;
    // rows + neighbours exactly as tpnet_encoder_gram lays them out in `scratch`
    int rc = tpnet_encoder_rows(st, sampler, E, num_nodes, src, other, t, B, K, scratch, scratch_bytes, stream);
    if (rc) return rc;
    int64_t* nodes = (int64_t*)(((size_t)scratch + 255) / 256 * 256);
    int64_t* a1 = nodes + 4 * B;
    int64_t* a2 = a1 + 2 * B;
    int64_t* neigh = a2 + 2 * B;
    if (gram && encoder_generic_readout(*st) && !encoder_mfma_supported(*st, 2 * B, K) && st->L == 3 && !(flags & TPNET_FLAG_PACKED) && mlp->F == 64 && mlp->H == 256 && mlp->w1 &&
        mlp->w2f && mlp->b1 && mlp->b2 && !((reinterpret_cast<uintptr_t>(gram) | reinterpret_cast<uintptr_t>(out)) & 15)) {
        // narrow rows: the generic readout on the pair lists the sampler kernel wrote behind the neighbour ids, then the dense layers
        const int64_t half = 2 * B * (int64_t)K;
        rc = tpnet_pair_gram(st, neigh + half, neigh + 3 * half, 2 * half, now_time, lambda, flags, gram, stream);
        if (rc) return rc;
        return mlp_rows(mlp, gram, 2 * half, out, (hipStream_t)stream);
    }
    return tpnet_anchored_features(st, neigh, a1, a2, 2 * B, K, now_time, lambda, flags, mlp, gram, out, stream);
}

}  // extern "C"
