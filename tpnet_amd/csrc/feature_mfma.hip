// SURVEY.md §8 f-1 as written / BASELINE config 5: self.mlp = Linear(64,256) -> ReLU -> Linear(256,64)
// (models/TPNet.py:64-65,129) on the bf16 matrix cores INSIDE the readout kernel: a workgroup of 8 waves forms the Gram
// features of 32 pairs (models/TPNet.py:119-128) into an LDS tile, which is then the B operand (X^T) of layer 1 --
// wave w owns hidden tile w (32 of the 256 units): H^T = W1[32w.., :] . X^T by v_mfma_f32_32x32x16_bf16, bias + ReLU in
// registers, and its accumulator tile is directly the B operand of its slice of layer 2 (Y^T += W2[:, 32w..] . H^T, W2
// pre-permuted as in mlp.hip).  The eight partial Y tiles are added in a FIXED order through LDS (run-to-run identical
// bits).  The features never leave the chip; the weights of a wave (its 32 rows of W1, its 32 columns of W2: 48 VGPRs)
// are loaded once per workgroup.  bf16 operands, fp32 accumulation: 2e-2 class like tpnet_mlp64_bf16 -- opt-in.  L = 3.
// MODE 2 (the default fp32-CLASS path): the bf16 matrix cores with SPLIT operands -- every f32 operand x = hi + lo with hi =
// bf16(x), lo = bf16(x - hi) (16 bits of significand between them), every product as hi*hi + hi*lo + lo*hi in fp32 accumulators:
// relative error ~2^-16 per product, the accuracy class of the fp32 paths (<= 2e-5 of the output scale against the torch
// layers, exact on small integers), at 24 bf16 MFMAs per wave and tile instead of 64 fp32 ones of four times the cycles.  Same
// inputs as MODE 1 (tpnet_mlp::w1, w2f: the split happens in registers).
// MODE 1 (F32): the same kernel on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: fp32 products, fp32 accumulation -- differs from
// the torch layers in summation order only), which makes ONE launch the default get_pair_wise_feature for lists of any length:
// k-step s of layer 1 takes features (s, s + 32) (lane half h reads 32 consecutive floats of its pair's LDS row and of its W1
// row), k-step s of layer 2 takes the hidden units that register s of the two lane halves holds (W2 gathered to match).
#include "readout.hpp"

namespace tpnet {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

static constexpr int MB = 512;            // threads per workgroup: 8 waves = the 8 hidden tiles
static constexpr int MF = 64, MH = 256;
static constexpr int TS = 68;             // floats per LDS row of the feature / partial tiles (64 + 4: bank spread)

// 8 consecutive floats (16-byte aligned) -> hi = bf16(x), lo = bf16(x - hi)
__device__ __forceinline__ void split8(const float* __restrict__ x, bf16x8& hi, bf16x8& lo) {
    const float4 a = *reinterpret_cast<const float4*>(x);
    const float4 b = *reinterpret_cast<const float4*>(x + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 t = (__bf16)v[j];
        hi[j] = t;
        lo[j] = (__bf16)(v[j] - (float)t);
    }
}

template <int LPP, int VPL, int W, bool FULL, int MODE>
__global__ __launch_bounds__(MB) void k_pair_feature_bf16(tpnet_state S, const int64_t* __restrict__ u,
                                                          const int64_t* __restrict__ v, int64_t n, double now,
                                                          double lambda, uint32_t flags, const void* __restrict__ w1v,
                                                          const float* __restrict__ b1, const void* __restrict__ w2v,
                                                          const float* __restrict__ b2, float* __restrict__ out_gram,
                                                          float* __restrict__ out, const float* __restrict__ feat_in, int tp) {
    // tp = pairs per tile: 32, or GPB (ONE readout pass per tile: a short list spreads over twice / four times the CUs and a
    // workgroup's readout is one memory round-trip chain deep instead of PT of them; columns of the matrix products beyond tp
    // compute on stale LDS rows and are never stored -- a pair is one column, nothing crosses columns)
    // feat_in != NULL: the dense layers alone on features that already exist ([n][64] f32): the tile is loaded, not formed
    // bf16: w1 = bf16 [256][64], w2p = bf16 [64][256] (hidden axis permuted per 32-tile, fused_mlp.permute_w2)
    // f32:  w1 = f32 [256][64] (mlp[0].weight as is), w2f = f32 [8 waves][2 output tiles][64 lanes][16 k-steps]
    const __bf16* __restrict__ w1 = reinterpret_cast<const __bf16*>(w1v);
    const __bf16* __restrict__ w2p = reinterpret_cast<const __bf16*>(w2v);
    const float* __restrict__ w1f = reinterpret_cast<const float*>(w1v);
    const float* __restrict__ w2f = reinterpret_cast<const float*>(w2v);
    constexpr int L = 3;
    constexpr bool F32 = MODE == 1, X3 = MODE == 2;
    constexpr int GPB = MB / LPP;             // pairs per readout pass
    constexpr int PT = 32 / GPB;              // passes per 32-pair tile
    static_assert(GPB * PT == 32 && LPP >= 16, "k_pair_feature_bf16: 16, 32 or 64 lanes per pair");
    __shared__ __attribute__((aligned(16))) float feat[32 * TS];
    __shared__ __attribute__((aligned(16))) float slab[4][32 * TS];
    __shared__ float stage1[1];
    const int tid = threadIdx.x;
    const int gl = tid % LPP, g = tid / LPP;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const bool do_scale = !(flags & TPNET_FLAG_NOT_SCALE);
    const int64_t ntiles = (n + tp - 1) / tp;
    // the first tile's ids BEFORE the weights: vector loads return in order, and the ids (host-mapped memory in the per-batch
    // calls: a PCIe round trip) are the head of the readout's dependent chain -- the weights' 128 KB arrive underneath it
    int64_t uu0 = 0, vv0 = 0;
    if (!feat_in && (int64_t)blockIdx.x < ntiles) {
        const int64_t p = (int64_t)blockIdx.x * tp + g;
        if (g < tp && p < n) { uu0 = u[p]; vv0 = v[p]; }
    }
    // ---- this wave's weights: rows [32 wave, 32 wave + 32) of W1 as A operand, the matching columns of (permuted) W2
    bf16x8 a1[4], a2[2][2];
    bf16x8 a1l[X3 ? 4 : 1], a2l[X3 ? 2 : 1][2];          // MODE 2: the low halves of the split weights
    float f1[F32 ? 32 : 1], f2[2][F32 ? 16 : 1];
    if constexpr (X3) {
        // the bf16 kernel's operand layout, taken from the f32 sources: A of layer 1 = W1[32 wave + r][16 s + 8 h + j]; A of layer 2
        // = W2[32 t + r][32 wave + 16 s2 + 8 (j>>2) + 4 h + (j&3)] = w2f[..][8 s2 + j] (the gathered f32 layout lists them in order)
#pragma unroll
        for (int s = 0; s < 4; ++s) split8(w1f + (wave * 32 + r) * MF + 16 * s + 8 * h, a1[s], a1l[s]);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) split8(w2f + (((wave * 2 + t2) * 64 + lane) * 16) + 8 * s2, a2[s2][t2], a2l[s2][t2]);
    } else if constexpr (F32) {
        // layer 1: A[m = hidden 32 wave + r][k] with k-step s <-> features (s, s + 32): lane half h holds W1[..][32 h + s]
#pragma unroll
        for (int q4 = 0; q4 < 8; ++q4) {
            const float4 x = *reinterpret_cast<const float4*>(w1f + (wave * 32 + r) * MF + 32 * h + 4 * q4);
            f1[4 * q4] = x.x; f1[4 * q4 + 1] = x.y; f1[4 * q4 + 2] = x.z; f1[4 * q4 + 3] = x.w;
        }
        // layer 2: A[m = output 32 t2 + r][k] with k-step s <-> hidden 32 wave + (s&3) + 8 (s>>2) + 4 h (gathered on the host)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 x = *reinterpret_cast<const float4*>(w2f + (((wave * 2 + t2) * 64 + lane) * 16) + 4 * q4);
                f2[t2][4 * q4] = x.x; f2[t2][4 * q4 + 1] = x.y; f2[t2][4 * q4 + 2] = x.z; f2[t2][4 * q4 + 3] = x.w;
            }
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) a1[s] = *reinterpret_cast<const bf16x8*>(w1 + (wave * 32 + r) * MF + 16 * s + 8 * h);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int off = wave * 32 + 16 * s2 + 8 * h;          // position inside the PERMUTED hidden axis
            a2[s2][0] = *reinterpret_cast<const bf16x8*>(w2p + r * MH + off);
            a2[s2][1] = *reinterpret_cast<const bf16x8*>(w2p + (32 + r) * MH + off);
        }
    }
    float bias1[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) bias1[q] = b1[wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * h];

    // features that already exist: a tile is 512 float4, one per thread; the NEXT tile's piece is fetched while this tile's
    // matrix products run (a tile is otherwise one global round trip + three barriers deep: 8.7 -> ~5 us per tile)
    static_assert(32 * (MF / 4) == MB, "one float4 of the feature tile per thread");
    const int frow = tid / (MF / 4), fcol = tid % (MF / 4);
    float4 fnext = make_float4(0.f, 0.f, 0.f, 0.f);
    if (feat_in && (int64_t)blockIdx.x < ntiles) {
        const int64_t p = (int64_t)blockIdx.x * 32 + frow;
        if (p < n) fnext = *reinterpret_cast<const float4*>(feat_in + p * MF + 4 * fcol);
    }
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // ---- the features of the tile's 32 pairs -> LDS
        if (feat_in) {
            *reinterpret_cast<float4*>(feat + frow * TS + 4 * fcol) = fnext;
            const int64_t pn = (tile + gridDim.x) * 32 + frow;
            fnext = (tile + gridDim.x < ntiles && pn < n) ? *reinterpret_cast<const float4*>(feat_in + pn * MF + 4 * fcol)
                                                          : make_float4(0.f, 0.f, 0.f, 0.f);
        } else
#pragma unroll 1
        for (int pass = 0; pass * GPB < tp; ++pass) {
            const int pidx = pass * GPB + g;
            const int64_t p = tile * tp + pidx;
            const bool valid = pidx < tp && p < n;
            const bool first = pass == 0 && tile == (int64_t)blockIdx.x;
            const int64_t uu = first ? uu0 : (valid ? u[p] : 0), vv = first ? vv0 : (valid ? v[p] : 0);
            gram_pair<LPP, VPL, W, L, FULL, false, false, false>(S, uu, vv, valid, READER_BID, now, lambda, do_scale,
                                                                 feat + pidx * TS, gl, nullptr, stage1);
        }
        __syncthreads();
        const int npair = (n - tile * tp < tp) ? (int)(n - tile * tp) : tp;
        if (out_gram) {                           // the pre-mlp features, for a backward pass (training)
            for (int i = tid; i < npair * MF; i += MB) out_gram[tile * tp * MF + i] = feat[(i / MF) * TS + (i % MF)];
        }
        // ---- layer 1, hidden tile `wave`: H^T = W1 . X^T; lane (r, h) holds X[pair r][16 s + 8 h + j] as B operand
        f32x16 acc, y0, y1;
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc[q] = 0.0f; y0[q] = 0.0f; y1[q] = 0.0f; }
        if constexpr (F32) {
            float xs[32];                         // X[pair r][32 h + s], s = 0..31
#pragma unroll
            for (int q4 = 0; q4 < 8; ++q4) {
                const float4 x = *reinterpret_cast<const float4*>(feat + r * TS + 32 * h + 4 * q4);
                xs[4 * q4] = x.x; xs[4 * q4 + 1] = x.y; xs[4 * q4 + 2] = x.z; xs[4 * q4 + 3] = x.w;
            }
#pragma unroll
            for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f1[s], xs[s], acc, 0, 0, 0);
            // register q = hidden row 32 wave + (q&3) + 8 (q>>2) + 4 h of pair r: bias, ReLU; it IS the B operand of k-step q
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float x = acc[q] + bias1[q];
                x = x > 0.0f ? x : 0.0f;
                y0 = __builtin_amdgcn_mfma_f32_32x32x2f32(f2[0][q], x, y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f2[1][q], x, y1, 0, 0, 0);
            }
        } else if constexpr (X3) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 bxh, bxl;
                split8(feat + r * TS + 16 * s + 8 * h, bxh, bxl);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l[s], bxh, acc, 0, 0, 0);     // the small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[s], bxl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[s], bxh, acc, 0, 0, 0);
            }
            bf16x8 bhh[2], bhl[2];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float x = acc[q] + bias1[q];
                x = x > 0.0f ? x : 0.0f;
                const __bf16 hi = (__bf16)x;
                bhh[q >> 3][q & 7] = hi;
                bhl[q >> 3][q & 7] = (__bf16)(x - (float)hi);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2l[s2][0], bhh[s2], y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2l[s2][1], bhh[s2], y1, 0, 0, 0);
                y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[s2][0], bhl[s2], y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[s2][1], bhl[s2], y1, 0, 0, 0);
                y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[s2][0], bhh[s2], y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[s2][1], bhh[s2], y1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float* xr = feat + r * TS + 16 * s + 8 * h;
                const float4 lo = *reinterpret_cast<const float4*>(xr);
                const float4 hi = *reinterpret_cast<const float4*>(xr + 4);
                bf16x8 bx;
                bx[0] = (__bf16)lo.x; bx[1] = (__bf16)lo.y; bx[2] = (__bf16)lo.z; bx[3] = (__bf16)lo.w;
                bx[4] = (__bf16)hi.x; bx[5] = (__bf16)hi.y; bx[6] = (__bf16)hi.z; bx[7] = (__bf16)hi.w;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[s], bx, acc, 0, 0, 0);
            }
            // register q = hidden row 32 wave + (q&3) + 8 (q>>2) + 4 h, column = pair r  ->  bias, ReLU, B operand of layer 2
            bf16x8 bh[2];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float x = acc[q] + bias1[q];
                x = x > 0.0f ? x : 0.0f;
                bh[q >> 3][q & 7] = (__bf16)x;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[s2][0], bh[s2], y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[s2][1], bh[s2], y1, 0, 0, 0);
            }
        }
        // ---- the eight partial tiles, added in a fixed order: waves 0..3 park theirs, waves 4..7 add theirs on top, then
        // every thread sums the four slabs for its outputs.  y0[4i..4i+3] = outputs 8i + 4h + (0..3) of pair r, y1: + 32
        float* sl = slab[wave & 3] + r * TS;
        if (wave < 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = 8 * i + 4 * h;
                *reinterpret_cast<float4*>(sl + o) = make_float4(y0[4 * i], y0[4 * i + 1], y0[4 * i + 2], y0[4 * i + 3]);
                *reinterpret_cast<float4*>(sl + 32 + o) = make_float4(y1[4 * i], y1[4 * i + 1], y1[4 * i + 2], y1[4 * i + 3]);
            }
        }
        __syncthreads();
        if (wave >= 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = 8 * i + 4 * h;
                float4 a = *reinterpret_cast<float4*>(sl + o), b = *reinterpret_cast<float4*>(sl + 32 + o);
                a.x += y0[4 * i]; a.y += y0[4 * i + 1]; a.z += y0[4 * i + 2]; a.w += y0[4 * i + 3];
                b.x += y1[4 * i]; b.y += y1[4 * i + 1]; b.z += y1[4 * i + 2]; b.w += y1[4 * i + 3];
                *reinterpret_cast<float4*>(sl + o) = a;
                *reinterpret_cast<float4*>(sl + 32 + o) = b;
            }
        }
        __syncthreads();
        {
            const int pair = tid >> 4, o = (tid & 15) * 4;      // 512 threads x 4 outputs = 32 pairs x 64
            if (pair < npair) {
                const float4 s0 = *reinterpret_cast<const float4*>(slab[0] + pair * TS + o);
                const float4 s1 = *reinterpret_cast<const float4*>(slab[1] + pair * TS + o);
                const float4 s2 = *reinterpret_cast<const float4*>(slab[2] + pair * TS + o);
                const float4 s3 = *reinterpret_cast<const float4*>(slab[3] + pair * TS + o);
                const float4 bb = *reinterpret_cast<const float4*>(b2 + o);
                float4 y;
                y.x = ((s0.x + s1.x) + (s2.x + s3.x)) + bb.x;
                y.y = ((s0.y + s1.y) + (s2.y + s3.y)) + bb.y;
                y.z = ((s0.z + s1.z) + (s2.z + s3.z)) + bb.z;
                y.w = ((s0.w + s1.w) + (s2.w + s3.w)) + bb.w;
                *reinterpret_cast<float4*>(out + (tile * tp + pair) * MF + o) = y;
            }
        }
        __syncthreads();                          // the tiles are reused by the next tile of this workgroup
    }
}

// the fp32-class dense layers: 2 = split bf16 operands (default), 1 = fp32 MFMA (developer builds: TPNET_DEV_MLP_F32_MODE=1)
int mlp_f32_mode() {
    static const int m = TPNET_DEV_INT(MLP_F32_MODE, 2);
    return m == 1 ? 1 : 2;
}

bool pair_feature_mfma_supported(const tpnet_state& st) {
    const Geom gm = pick_geom(st.d);
    return st.L == 3 && gm.w == 4 && gm.lpp >= 16;
}

int launch_pair_feature_bf16(const tpnet_state& st, const int64_t* u, const int64_t* v, int64_t n, double now, double lambda,
                             uint32_t flags, const void* w1, const float* b1, const void* w2p, const float* b2,
                             float* out_gram, float* out, hipStream_t s, int mode, const float* feat_in) {
    if (n == 0) return TPNET_OK;
    if (st.L != 3 || (flags & TPNET_FLAG_PACKED)) return TPNET_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(b2)) & 15) return TPNET_ERR_BAD_ARG;
    const Geom gm = pick_geom(st.d);
    if (gm.w != 4 || gm.lpp < 16) return TPNET_ERR_BAD_ARG;       // narrow / scalar rows: readout kernel + tpnet_mlp64_bf16
    const bool full = st.d == gm.lpp * gm.vpl * 4;
    // pairs per tile: one readout pass (GPB pairs) while the list has fewer than two 32-pair tiles per CU
    const int gpb = MB / gm.lpp;
    static const int tp_dev = TPNET_DEV_INT(FEATURE_TP, 0);
    const int tp = feat_in ? 32 : (tp_dev ? (tp_dev < gpb ? gpb : (tp_dev > 32 ? 32 : tp_dev)) : ((gpb < 32 && n <= 256 * 64) ? gpb : 32));
    const int64_t tiles = (n + tp - 1) / tp;
    // (a workgroup's first act is to load its 128 KB of weights: one workgroup per CU, each amortising them over many tiles --
    // 80 000 rows: 51 us with a workgroup per tile up to 2 048, 39 us with 768, 32.5 us with 256; 800 000 rows: 244 / 237)
    static const int grid_cap = TPNET_DEV_INT(MLP_GRID, 256);
    const int grid = (int)(tiles < grid_cap ? tiles : grid_cap);
#define TPNET_PF(LPP_, VPL_, FULL_)                                                                                          \
    do {                                                                                                                     \
        if (mode == 2)                                                                                                       \
            hipLaunchKernelGGL((k_pair_feature_bf16<LPP_, VPL_, 4, FULL_, 2>), dim3(grid), dim3(MB), 0, s, st, u, v, n, now,    \
                               lambda, flags, w1, b1, w2p, b2, out_gram, out, feat_in, tp);                                         \
        else if (mode == 1)                                                                                                  \
            hipLaunchKernelGGL((k_pair_feature_bf16<LPP_, VPL_, 4, FULL_, 1>), dim3(grid), dim3(MB), 0, s, st, u, v, n, now,    \
                               lambda, flags, w1, b1, w2p, b2, out_gram, out, feat_in, tp);                                         \
        else                                                                                                                 \
            hipLaunchKernelGGL((k_pair_feature_bf16<LPP_, VPL_, 4, FULL_, 0>), dim3(grid), dim3(MB), 0, s, st, u, v, n, now,    \
                               lambda, flags, w1, b1, w2p, b2, out_gram, out, feat_in, tp);                                         \
    } while (0)
    if (gm.lpp == 16 && gm.vpl == 1) { if (full) TPNET_PF(16, 1, true); else TPNET_PF(16, 1, false); }
    else if (gm.lpp == 16) { if (full) TPNET_PF(16, 2, true); else TPNET_PF(16, 2, false); }
    else if (gm.lpp == 32 && gm.vpl == 1) { if (full) TPNET_PF(32, 1, true); else TPNET_PF(32, 1, false); }
    else if (gm.lpp == 32) { if (full) TPNET_PF(32, 2, true); else TPNET_PF(32, 2, false); }
    else if (gm.vpl == 1) { if (full) TPNET_PF(64, 1, true); else TPNET_PF(64, 1, false); }
    else { if (full) TPNET_PF(64, 2, true); else TPNET_PF(64, 2, false); }
#undef TPNET_PF
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

// the layouts the fused kernels read, from the Parameters of self.mlp: one launch per optimizer step (the torch expressions --
// two transposes, a contiguous copy and an index gather -- were ~100 us of a training step)
__global__ void k_mlp_prepare(const float* __restrict__ w1, const float* __restrict__ w2, int F, int H, float* __restrict__ w1t,
                              float* __restrict__ w2t, float* __restrict__ w2f) {
    const int n = F * H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        // w1 [H][F] -> w1t [F][H];  w2 [F][H] -> w2t [H][F]   (coalesced writes)
        { const int f = i / H, h = i - f * H; w1t[i] = w1[h * F + f]; }
        { const int h = i / F, f = i - h * F; w2t[i] = w2[f * H + h]; }
        if (w2f) {            // F = 64, H = 256: i = ((w * 2 + t) * 64 + lane) * 16 + s   (include/tpnet_hip.h, tpnet_mlp::w2f)
            const int s_ = i & 15, lane = (i >> 4) & 63, t = (i >> 10) & 1, w = i >> 11;
            w2f[i] = w2[(32 * t + (lane & 31)) * H + 32 * w + (s_ & 3) + 8 * (s_ >> 2) + 4 * (lane >> 5)];
        }
    }
}

}  // namespace tpnet

using namespace tpnet;

extern "C" int tpnet_mlp_prepare(const float* w1, const float* w2, int32_t F, int32_t H, float* w1t, float* w2t, float* w2f,
                                 void* stream) {
    if (!w1 || !w2 || !w1t || !w2t || F < 1 || H < 1 || (int64_t)F * H > (1 << 24)) return TPNET_ERR_BAD_ARG;
    if (w2f && (F != 64 || H != 256)) return TPNET_ERR_BAD_ARG;
    const int n = F * H;
    hipLaunchKernelGGL(k_mlp_prepare, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w1, w2, (int)F, (int)H, w1t, w2t, w2f);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

extern "C" int tpnet_pair_feature_bf16(const tpnet_state* st, const int64_t* u, const int64_t* v, int64_t n, double now_time,
                                       double lambda, uint32_t flags, const void* w1_bf16, const float* b1,
                                       const void* w2p_bf16, const float* b2, float* out_gram, float* out, void* stream) {
    if (!st || !st->p0 || !st->q || !st->meta || !st->err || st->N < 1 || st->d < 1) return TPNET_ERR_BAD_ARG;
    if (n < 0 || (n > 0 && (!u || !v || !out || !w1_bf16 || !b1 || !w2p_bf16 || !b2))) return TPNET_ERR_BAD_ARG;
    return launch_pair_feature_bf16(*st, u, v, n, now_time, lambda, flags, w1_bf16, b1, w2p_bf16, b2, out_gram, out,
                                    (hipStream_t)stream, 0, nullptr);
}

extern "C" int tpnet_mlp64_f32(const float* x, int64_t n, const tpnet_mlp* mlp, float* y, void* stream) {
    if (n < 0 || !mlp || (n > 0 && (!x || !y))) return TPNET_ERR_BAD_ARG;
    if (n == 0) return TPNET_OK;
    if (mlp->F != 64 || mlp->H != 256 || !mlp->w1 || !mlp->w2f || !mlp->b1 || !mlp->b2) return TPNET_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return TPNET_ERR_BAD_ARG;
    // long lists: every wave its own tiles, split weights in LDS (mlp_x3.hip)
    if (n >= mlp_x3_from() && mlp_f32_mode() == 2 && mlp_x3_available() &&
        launch_mlp_rows_x3(x, n, reinterpret_cast<const float*>(mlp->w1), mlp->b1, reinterpret_cast<const float*>(mlp->w2f),
                           mlp->b2, y, (hipStream_t)stream) == TPNET_OK)
        return TPNET_OK;
    tpnet_state st{};                      // (not dereferenced when the tile comes from `x`; geometry of d = 128 picks the 32-lane kernel)
    st.N = 1; st.d = 128; st.L = 3;
    return launch_pair_feature_bf16(st, nullptr, nullptr, n, 0.0, 0.0, 0, mlp->w1, mlp->b1, mlp->w2f, mlp->b2, nullptr, y,
                                    (hipStream_t)stream, mlp_f32_mode(), x);
}
