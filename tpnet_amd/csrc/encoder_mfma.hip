// The encoder's readout (models/TPNet.py:311-324: every sampled neighbour w of a row paired with the row's two anchors, the
// edge's src and dst) on the matrix cores, fp32 class.  Round 4: the vector-ALU readouts of an 80 000-pair call at d = 128 are
// VALU-bound (profiles/r04_encoder.md: 63 % VALU busy, ~130 wave instructions per pair, most of them the cross-lane reduction of
// the 64 inner products), not traffic-bound.  A matrix instruction accumulates over d inside the pipe, so the reduction vanishes:
//
//   tile of one wave = 4 consecutive neighbour slots of the flat [n_rows * K] list = 16 rows (4 neighbours x layers 0..3) x d
//   A = those 16 rows; B1 = the same 16 rows (-> the neighbours' own 4 x 4 blocks on the diagonal of A A^T);
//   B2 = 16 anchor rows: the two anchors x 4 layers of the node the tile starts in and of the next node (K >= 4: a tile spans
//   at most two nodes); the anchors' own blocks come from B2 B2^T, formed once per anchor set.
//   v_mfma_f32_16x16x32_bf16 on split operands: every f32 value x = h + m + l (three bf16 pieces, 24 significand bits) and every
//   product as l*h + h*l + m*m + m*h + h*m + h*h with fp32 accumulation (small terms first): what is dropped is 2^-24 relative,
//   the class of an fp32 fused multiply-add chain (SPLIT = 2: h + l and three products, 2^-16 -- kept for measurements).
//
// Operand layout (gfx950): lane (c = lane & 15, g = lane >> 4) holds row c, k-positions 8 g .. 8 g + 7 of a 32-deep step; any
// fixed assignment of a row's floats to k-positions is a valid contraction as long as A and B use the same one, so a step takes
// floats [32 s, 32 s + 32) with lane (c, g) holding {32 s + 4 g .. + 3} and {32 s + 16 + 4 g .. + 3}.  Rows are LOADED in
// another layout -- lane 4 r + p reads the 16-byte piece p of a 64-byte segment of row r, so every quad of lanes reads 64
// contiguous bytes -- and reach the operand layout through ds_bpermute_b32 (the LDS crossbar, no LDS memory).
// The 4 x 4 blocks leave the accumulators through an LDS tile of 8 feature rows per wave (the [w | anchor]^2 layout of
// get_pair_wise_feature, mirrored), are clamped / log-scaled there (models/TPNet.py:127-128) and stored as whole 256-byte rows.
#include "device_common.hpp"

namespace tpnet {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

static constexpr int EMB = 256;           // threads per workgroup: 4 waves, every wave on its own tiles
static constexpr int EM_RS = 68;          // floats per staged feature row (64 + 4: rows stay 16-byte aligned, b128 reads conflict-free)

template <int SPLIT>
struct SplitOp {
    bf16x8 p[SPLIT];                      // p[0] = the leading bf16 piece of 8 values, p[1], p[2] = the pieces below
};

template <int SPLIT>
__device__ __forceinline__ void split8(const float* v, SplitOp<SPLIT>& o) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = v[j];
#pragma unroll
        for (int t = 0; t < SPLIT; ++t) {
            const __bf16 b = (__bf16)x;
            o.p[t][j] = b;
            if (t + 1 < SPLIT) x = x - (float)b;
        }
    }
}

// c += A B^T over one 32-deep step of split operands, the small terms first
template <int SPLIT>
__device__ __forceinline__ f32x4 mm_step(const SplitOp<SPLIT>& a, const SplitOp<SPLIT>& b, f32x4 c) {
    if constexpr (SPLIT == 3) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[2], b.p[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b.p[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b.p[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[0], c, 0, 0, 0);
    } else {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b.p[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[0], c, 0, 0, 0);
    }
    return c;
}

// 16 rows of 32 KS floats, one row pointer per load-layout lane (already offset by the lane's 16-byte piece), scaled by the
// row's pending decay: raw[s][0..3] = floats 32 s + 4 p .., raw[s][4..7] = floats 32 s + 16 + 4 p ..
template <int KS>
__device__ __forceinline__ void load_rows(const float* __restrict__ rp, float (&raw)[KS][8]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const float4 x = *reinterpret_cast<const float4*>(rp + 32 * s);
        const float4 y = *reinterpret_cast<const float4*>(rp + 32 * s + 16);
        raw[s][0] = x.x; raw[s][1] = x.y; raw[s][2] = x.z; raw[s][3] = x.w;
        raw[s][4] = y.x; raw[s][5] = y.y; raw[s][6] = y.z; raw[s][7] = y.w;
    }
}

// load layout -> operand layout (lane (c, g) takes what lane 4 c + g loaded), decay applied on the way, then the split
template <int KS, int SPLIT>
__device__ __forceinline__ void to_operands(const float (&raw)[KS][8], float rs, int pull, SplitOp<SPLIT> (&op)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            v[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(pull, __float_as_int(raw[s][k] * rs)));
        split8<SPLIT>(v, op[s]);
    }
}

template <int KS, int SPLIT>
__global__ __launch_bounds__(EMB) void k_encoder_gram_mfma(tpnet_state S, const int64_t* __restrict__ neigh,
                                                           const int64_t* __restrict__ a1, const int64_t* __restrict__ a2,
                                                           int n_rows, int K, int T, int tpw, double now, double lambda,
                                                           uint32_t flags, float* __restrict__ out1, float* __restrict__ out2) {
    constexpr int L = 3;
    __shared__ __attribute__((aligned(16))) float stg_all[(EMB / 64) * 8 * EM_RS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* stg = stg_all + wave * (8 * EM_RS);
    const int c = lane & 15, g = lane >> 4;        // operand layout: row c of the tile, k-group g
    const int lr = lane >> 2, lp = lane & 3;       // load layout: row lr, 16-byte piece lp of a 64-byte segment
    const int pull = (4 * c + g) * 4;              // ds_bpermute address: operand lane (c, g) <- load lane 4 c + g
    const int l_layer = lr & 3;                    // load layout: the row's layer; its neighbour (or anchor) index is lr >> 2
    const int d = 32 * KS;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(S.meta);
    const bool do_scale = !(flags & TPNET_FLAG_NOT_SCALE);
    // a wave walks tpw consecutive tiles (4 slots each), 16 slots at a time: their ids and meta records are fetched lane-parallel
    const int wid = blockIdx.x * (EMB / 64) + wave;
    const int slot_begin = wid * tpw * 4;
    const int slot_end = (slot_begin + tpw * 4 < T) ? slot_begin + tpw * 4 : T;

    int cur_n0 = -1;
    SplitOp<SPLIT> anch[KS];
    f32x4 daa = {0.0f, 0.0f, 0.0f, 0.0f};
    bool aok0 = true, aok1 = true;                 // (wave-uniform) both anchors of node n0 / n0 + 1 are valid ids

    auto row_ptr = [&](int id, int copy, int layer) -> const float* {
        const float* p = (layer == 0) ? S.p0 + (int64_t)id * d
                                      : S.q + (((int64_t)copy * S.N + id) * L + (layer - 1)) * (int64_t)d;
        return p + 4 * lp;
    };
    auto decay_pow = [](float gd, int layer) -> float {       // g, g*g, (g*g)*g: the association of the vector-ALU readouts
        const float g2 = gd * gd;
        return layer == 0 ? 1.0f : layer == 1 ? gd : layer == 2 ? g2 : g2 * gd;
    };

    for (int base = slot_begin; base < slot_end; base += 16) {
        // ---- the next 16 slots: neighbour id, its node (row of the call), its meta record -- every lane for slot lane & 15
        const int j = base + c;
        const bool in = j < slot_end;
        const int64_t w64 = in ? neigh[j] : 0;
        const int node = (j < T ? j : T - 1) / K;
        const bool wok = in && (uint64_t)w64 < (uint64_t)S.N;
        if (in && !wok && g == 0) atomicAdd(S.err, 1u);
        const int w = wok ? (int)w64 : 0;
        const MetaView mv = read_meta(meta, w, READER_BID, now, lambda);

        float raw[KS][8];
        auto issue_tile = [&](int st) {            // rows of tile st of the group (load layout: neighbour lane >> 4 of the tile)
            const int sl = 4 * st + (lane >> 4);
            const int wv = __shfl(w, sl), cp = __shfl(mv.copy, sl);
            load_rows<KS>(row_ptr(wv, cp, l_layer), raw);
        };
        issue_tile(0);
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if (base + 4 * st >= slot_end) break;  // (wave-uniform: the wave's share of the list ended)
            const int n0 = __builtin_amdgcn_readlane(node, 4 * st);
            // ---- the anchors' 16 rows (nodes n0 and n0 + 1) and their own blocks, when the tile starts in a new node
            if (n0 != cur_n0) {
                cur_n0 = n0;
                const int ai = lr >> 2;                                   // anchor index: node (ai >> 1), side (ai & 1)
                const int an = (n0 + (ai >> 1) < n_rows) ? n0 + (ai >> 1) : n_rows - 1;
                const int64_t id64 = (ai & 1) ? a2[an] : a1[an];
                const bool bad = (uint64_t)id64 >= (uint64_t)S.N;
                if (bad && lp == 0 && l_layer == 0 && n0 + (ai >> 1) < n_rows) atomicAdd(S.err, 1u);
                const int id = bad ? 0 : (int)id64;
                const MetaView am = read_meta(meta, id, READER_BID, now, lambda);
                float ra[KS][8];
                load_rows<KS>(row_ptr(id, am.copy, l_layer), ra);
                const unsigned long long bb = __ballot(bad);
                aok0 = (bb & 0xFFFFFFFFull) == 0;
                aok1 = (bb >> 32) == 0;
                to_operands<KS, SPLIT>(ra, decay_pow(am.g, l_layer), pull, anch);
                daa = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int s = 0; s < KS; ++s) daa = mm_step<SPLIT>(anch[s], anch[s], daa);
            }
            // ---- this tile's rows -> operands; the next tile's rows ride under the matrix work
            SplitOp<SPLIT> aop[KS];
            {
                const float gd = __shfl(mv.g, 4 * st + (lane >> 4));
                to_operands<KS, SPLIT>(raw, decay_pow(gd, l_layer), pull, aop);
            }
            if (st < 3 && base + 4 * (st + 1) < slot_end) issue_tile(st + 1);
            f32x4 dww = {0.0f, 0.0f, 0.0f, 0.0f}, dwa = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                dww = mm_step<SPLIT>(aop[s], aop[s], dww);
                dwa = mm_step<SPLIT>(aop[s], anch[s], dwa);
            }
            // ---- accumulators -> the wave's LDS tile of 8 feature rows (pair p = side * 4 + neighbour; element 8 a + b of the
            // [w rows | anchor rows]^2 Gram).  Lane (c, g) holds D[4 g + q][c], q = 0..3.
            int ni[4];                                                    // node of neighbour nb relative to n0 (0 / 1)
            bool pok[4], pin[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                ni[nb] = __builtin_amdgcn_readlane(node, 4 * st + nb) - n0;
                pin[nb] = base + 4 * st + nb < slot_end;
                pok[nb] = (__builtin_amdgcn_readlane((int)wok, 4 * st + nb) != 0) && (ni[nb] == 0 ? aok0 : aok1);
            }
            const int ni_g = g == 0 ? ni[0] : g == 1 ? ni[1] : g == 2 ? ni[2] : ni[3];
            const int cb = c & 3, cn = c >> 2;
            if (cn == g) {                                                // w.w block of neighbour g: (a = q, b = cb), both sides
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    stg[g * EM_RS + 8 * q + cb] = dww[q];
                    stg[(4 + g) * EM_RS + 8 * q + cb] = dww[q];
                }
            }
            if ((c >> 3) == ni_g) {                                       // w.anchor block: column c = anchor (c >> 3, side (c >> 2) & 1), layer cb
                float* row = stg + (((c >> 2) & 1) * 4 + g) * EM_RS;
#pragma unroll
                for (int q = 0; q < 4; ++q) row[8 * q + 4 + cb] = dwa[q];
                *reinterpret_cast<f32x4*>(row + 32 + 8 * cb) = dwa;       // mirrored: (a = 4 + cb, b = 0..3)
            }
            if (cn == g) {                                                // anchor g's own block -> every neighbour of its node
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    if (ni[nb] == (g >> 1)) {
                        float* row = stg + ((g & 1) * 4 + nb) * EM_RS;
#pragma unroll
                        for (int q = 0; q < 4; ++q) row[8 * (4 + q) + 4 + cb] = daa[q];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();       // one wave: LDS executes in issue order
            // ---- feature rows out: clamp, log(x + 1) (models/TPNet.py:127-128), 16 lanes per 256-byte row
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int idx4 = it * 64 + lane;
                const int p = idx4 >> 4, col4 = idx4 & 15;
                const int nb = p & 3;
                f32x4 v = *reinterpret_cast<const f32x4*>(stg + p * EM_RS + 4 * col4);
                const bool ok = nb == 0 ? pok[0] : nb == 1 ? pok[1] : nb == 2 ? pok[2] : pok[3];
                const bool inr = nb == 0 ? pin[0] : nb == 1 ? pin[1] : nb == 2 ? pin[2] : pin[3];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float x = v[k];
                    if (do_scale) {
                        x = (x < 0.0f) ? 0.0f : x;     // NaN < 0 is false: NaN passes through, as in the reference (:127)
                        x = logf(x + 1.0f);             // log(x + 1), not log1p (:128)
                    }
                    v[k] = ok ? x : __builtin_nanf("");
                }
                if (inr) {
                    float* o = ((p >> 2) ? out2 : out1) + (int64_t)(base + 4 * st + nb) * 64 + 4 * col4;
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(o));
                }
            }
            __builtin_amdgcn_wave_barrier();       // the tile is rewritten by the next one
        }
    }
}

bool encoder_mfma_supported(const tpnet_state& st, int64_t n_rows, int K) {
    static const int off = TPNET_DEV_INT(NO_ENCODER_MFMA, 0);
    return !off && st.L == 3 && (st.d == 64 || st.d == 128) && K >= 4 && st.N < (int64_t)1 << 31 && n_rows > 0 &&
           n_rows * (int64_t)K < ((int64_t)1 << 31) / 64;
}

int launch_encoder_gram_mfma(const tpnet_state& st, const int64_t* neigh, const int64_t* a1, const int64_t* a2, int64_t n_rows,
                             int K, double now, double lambda, uint32_t flags, float* out1, float* out2, hipStream_t s) {
    if (n_rows == 0 || K == 0) return TPNET_OK;
    if (!encoder_mfma_supported(st, n_rows, K) || (flags & TPNET_FLAG_PACKED)) return TPNET_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(out1) | reinterpret_cast<uintptr_t>(out2)) & 15) return TPNET_ERR_BAD_ARG;
    const int T = (int)(n_rows * K);
    const int ntiles = (T + 3) / 4;
    static const int split_dev = TPNET_DEV_INT(ENCODER_SPLIT, 3);
    // tiles per wave: ONE round of the waves the chip holds (2 per SIMD at 172 VGPRs: 8 per CU), at least two tiles each
    static const int waves_dev = TPNET_DEV_INT(ENCODER_WAVES, 256 * 8);
    int tpw = (ntiles + waves_dev - 1) / waves_dev;
    if (tpw < 2) tpw = 2;
    const int nwaves = (ntiles + tpw - 1) / tpw;
    const int grid = (nwaves + EMB / 64 - 1) / (EMB / 64);
#define TPNET_EM_LAUNCH(KS_, SP_)                                                                                              \
    hipLaunchKernelGGL((k_encoder_gram_mfma<KS_, SP_>), dim3(grid), dim3(EMB), 0, s, st, neigh, a1, a2, (int)n_rows, K, T, tpw, \
                       now, lambda, flags, out1, out2)
    if (st.d == 128) { if (split_dev == 2) TPNET_EM_LAUNCH(4, 2); else TPNET_EM_LAUNCH(4, 3); }
    else { if (split_dev == 2) TPNET_EM_LAUNCH(2, 2); else TPNET_EM_LAUNCH(2, 3); }
#undef TPNET_EM_LAUNCH
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

}  // namespace tpnet
