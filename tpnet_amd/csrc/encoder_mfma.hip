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

// ---- the fused kernel (k_encoder_fused): self.mlp = Linear(64, 256) -> ReLU -> Linear(256, 64) (models/TPNet.py:63-65, 129)
// behind the Gram tiles in the same launch, fp32 class (two bf16 pieces per value, three products: the arithmetic of mlp_x3.hip on
// v_mfma_f32_16x16x32_bf16).  A workgroup of 8 waves, one per CU, is split by ROLE: waves 0..3 walk Gram tiles (latency- and
// vector-ALU-bound: gathers, splits, logs), waves 4..7 run the dense layers (matrix-pipe- and LDS-bound), wave 4 + p on the rows
// wave p produces -- the two share a SIMD, so one's matrix work runs under the other's loads and vector work (a first version in
// which every wave did both in turn took as long as the two kernels one after the other: 45 us per 80 000 pairs -- all eight
// waves were in the same phase at the same time).  Hand-off through LDS: the producer leaves the clamped / log-scaled features of a
// tile (8 rows) in one of two buffers and publishes a tile count; its consumer takes them as one half of the B operand of layer 1
// (lane (r, g): features 32 s + 8 g .. of row r), publishes that it did, and runs the layers on every second tile (16 rows):
// layer 1's accumulators of two hidden slices are, after bias + ReLU + split, the B operand of one 32-deep step of layer 2 (the
// weight image lists layer 2's k-positions in that order); a row's 64 outputs never leave the registers before they are stored.
// The split weights (hi / lo of W1 and W2, 128 KB, in exactly the per-lane operand order) + the biases come as ONE image
// (tpnet_mlp::wimg, written by tpnet_mlp_prepare_image when a Parameter changed) that a workgroup copies to its LDS once.  The
// pre-mlp features are stored only if the caller asks (a backward pass needs them): 20 MB of writes and 20 MB of reads less per
// 80 000-pair call, and one launch instead of two.
static constexpr int EMF_B = 512;                                           // threads per workgroup of the fused variant
static constexpr int IMG_W1H = 0, IMG_W1L = 32768, IMG_W2H = 65536, IMG_W2L = 98304, IMG_B1 = 131072, IMG_B2 = IMG_B1 + 1024;
static constexpr int IMG_BYTES = IMG_B2 + 256;                              // 132 352
static constexpr int EMF_NP = 4;                                            // producer waves (= consumer waves) per workgroup
static constexpr int EMF_TILE = 8 * EM_RS * 4;                              // bytes of a producer's own tile of 8 raw feature rows
static constexpr int EMF_HAND = 2048;                                       // bytes of a hand-off buffer: 8 rows as layer 1's B operand,
                                                                            // hi plane [(s * 4 + g) * 8 + p] x 16 bytes, lo plane + 1024
static constexpr int EMF_PP = EMF_TILE + 2 * EMF_HAND;                      // per producer: its tile + two hand-off buffers
static constexpr int EMF_STG = IMG_BYTES;
static constexpr int EMF_SYNC = EMF_STG + EMF_NP * EMF_PP;                  // per producer: ready, freed, 2 x {first slot, validity}
static constexpr int EMF_LDS = EMF_SYNC + EMF_NP * 32;                      // 157 568 bytes

#ifdef TPNET_STAMPS
// diagnostic build only (make STAMPS=1): shader-clock stamps of the fused kernel's waves, 64 per wave (tools/encoder_stamps.py)
__device__ unsigned long long g_em_stamps[4096 * 64];
#define EM_STAMP(slot)                                                                                          \
    do {                                                                                                        \
        if ((threadIdx.x & 63) == 0 && (slot) < 64)                                                             \
            g_em_stamps[(size_t)((blockIdx.x * 8 + (threadIdx.x >> 6)) & 4095) * 64 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define EM_STAMP(slot) do { } while (0)
#endif

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

// load layout -> operand layout (lane (c, g) takes what lane 4 c + g loaded); SCALE: the row's pending decay applied on the way
template <int KS, bool SCALE = true>
__device__ __forceinline__ void to_lanes(const float (&raw)[KS][8], float rs, int pull, float (&v)[KS][8]) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int k = 0; k < 8; ++k)
            v[s][k] = __int_as_float(__builtin_amdgcn_ds_bpermute(pull, __float_as_int(SCALE ? raw[s][k] * rs : raw[s][k])));
}
// ... and the split on top
template <int KS, int SPLIT>
__device__ __forceinline__ void to_operands(const float (&raw)[KS][8], float rs, int pull, SplitOp<SPLIT> (&op)[KS]) {
    float v[KS][8];
    to_lanes<KS>(raw, rs, pull, v);
#pragma unroll
    for (int s = 0; s < KS; ++s) split8<SPLIT>(v[s], op[s]);
}

// cw += A A^T, ca += A B^T over one 32-deep step: the two accumulation chains take turns in the matrix pipe (a chain's next
// product waits for its previous one), the small terms first
template <int SPLIT>
__device__ __forceinline__ void mm_step2(const SplitOp<SPLIT>& a, const SplitOp<SPLIT>& b, f32x4& cw, f32x4& ca) {
    if constexpr (SPLIT == 3) {
        cw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[2], a.p[0], cw, 0, 0, 0);
        ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[2], b.p[0], ca, 0, 0, 0);
        cw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], a.p[2], cw, 0, 0, 0);
        ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[2], ca, 0, 0, 0);
        cw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], a.p[1], cw, 0, 0, 0);
        ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b.p[1], ca, 0, 0, 0);
        cw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], a.p[0], cw, 0, 0, 0);
        ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b.p[0], ca, 0, 0, 0);
        cw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], a.p[1], cw, 0, 0, 0);
        ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[1], ca, 0, 0, 0);
        cw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], a.p[0], cw, 0, 0, 0);
        ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[0], ca, 0, 0, 0);
    } else {
        cw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], a.p[0], cw, 0, 0, 0);
        ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b.p[0], ca, 0, 0, 0);
        cw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], a.p[1], cw, 0, 0, 0);
        ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[1], ca, 0, 0, 0);
        cw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], a.p[0], cw, 0, 0, 0);
        ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[0], ca, 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// One wave walks the tiles of slots [slot_begin, slot_end) of the flat neighbour list.  Per group of 64 slots the neighbour ids,
// their nodes and meta records AND the anchors (ids + meta records of the up to 32 rows the group touches) are fetched
// lane-parallel, two dependent round trips for the whole group; a tile's 16 rows are in flight while the previous tile is
// computed, and so are the 16 anchor rows of the next tile when it starts in another node.
// HANDOFF = false: the finished feature rows are stored (out1 / out2).  HANDOFF = true (producer of the fused kernel): they stay
// in one of two LDS tiles for the consumer wave (sync[0] = tiles published, sync[1] = tiles taken, sync[2 + 2 b] / [3 + 2 b] =
// first slot / validity of the tile in buffer b) and are stored only if out1 is given.
// ---------------------------------------------------------------------------------------------------------------------------
template <int KS, int SPLIT, bool HANDOFF>
__device__ __forceinline__ int walk_tiles(const tpnet_state& S, const int64_t* __restrict__ neigh, const int64_t* __restrict__ a1,
                                           const int64_t* __restrict__ a2, int n_rows, int K, int T, int slot_begin, int slot_end,
                                           double now, double lambda, uint32_t flags, float* __restrict__ out1,
                                           float* __restrict__ out2, float* stg, int* sync) {
    constexpr int L = 3;
    const int lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;        // operand layout: row c of the tile, k-group g
    const int lr = lane >> 2, lp = lane & 3;       // load layout: row lr, 16-byte piece lp of a 64-byte segment
    const int pull = (4 * c + g) * 4;              // ds_bpermute address: operand lane (c, g) <- load lane 4 c + g
    const int l_layer = lr & 3;                    // load layout: the row's layer; its neighbour (or anchor) index is lr >> 2 = g
    const int d = 32 * KS;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(S.meta);
    const bool do_scale = !(flags & TPNET_FLAG_NOT_SCALE);

    const float* const tab_p0 = S.p0;
    const float* const tab_q = S.q;
    const int64_t tab_n = S.N;
    auto row_ptr = [&](int id, int copy, int layer) -> const float* {
        // (offsets selected, then ONE base select: a select between the two struct members became an indexed load from a stack copy)
        const int64_t off0 = (int64_t)id * d;
        const int64_t off1 = (((int64_t)copy * tab_n + id) * L + (layer - 1)) * (int64_t)d;
        const float* b = layer == 0 ? tab_p0 : tab_q;
        return b + (layer == 0 ? off0 : off1) + 4 * lp;
    };
    auto decay_pow = [](float gd, int layer) -> float {       // g, g*g, (g*g)*g: the association of the vector-ALU readouts
        const float g2 = gd * gd;
        return layer == 0 ? 1.0f : layer == 1 ? gd : layer == 2 ? g2 : g2 * gd;
    };

    int cur_n0 = -1, pre_n0 = -1;                  // node the anchor operands stand for; node whose anchor rows are in flight
    SplitOp<SPLIT> anch[KS];
    f32x4 daa = {0.0f, 0.0f, 0.0f, 0.0f};
    bool aok0 = true, aok1 = true;                 // (wave-uniform) both anchors of node n0 / n0 + 1 are valid ids
    float pre_g = 1.0f;
    int tile_no = 0;                               // tiles finished by this wave (HANDOFF: the published count)

    for (int base = slot_begin; base < slot_end; base += 64) {
        // ---- the group's slots: neighbour id, its node (row of the call), its meta record -- lane l for slot base + l
        const int j = base + lane;
        const bool in = j < slot_end;
        const int64_t w64 = in ? neigh[j] : 0;
        const int node = (j < T ? j : T - 1) / K;
        const bool wok = in && (uint64_t)w64 < (uint64_t)S.N;
        if (in && !wok) atomicAdd(S.err, 1u);
        const int w = wok ? (int)w64 : 0;
        // ---- the group's anchors: lane l for anchor (row nfirst + (l >> 1), side l & 1)
        const int nfirst = __builtin_amdgcn_readfirstlane(node);
        const int nslots = (slot_end - base < 64) ? slot_end - base : 64;
        const int nlast = __builtin_amdgcn_readlane(node, nslots - 1);
        const int an = (nfirst + (lane >> 1) < n_rows) ? nfirst + (lane >> 1) : n_rows - 1;
        const int64_t aid64 = (lane & 1) ? a2[an] : a1[an];
        const bool abad = (uint64_t)aid64 >= (uint64_t)S.N;
        if (abad && nfirst + (lane >> 1) <= nlast) atomicAdd(S.err, 1u);
        const int aid = abad ? 0 : (int)aid64;
        const unsigned long long abadmask = __ballot(abad);
        const MetaView mv = read_meta(meta, w, READER_BID, now, lambda);
        const MetaView am = read_meta(meta, aid, READER_BID, now, lambda);
        const int ntile = (nslots + 3) >> 2;
        if constexpr (HANDOFF) EM_STAMP(2);

        float raw[KS][8], ra[KS][8];
        auto issue_tile = [&](int t) {             // rows of tile t of the group (load layout: neighbour g of the tile)
            const int sl = 4 * t + g;
            const int wv = __shfl(w, sl), cp = __shfl(mv.copy, sl);
            load_rows<KS>(row_ptr(wv, cp, l_layer), raw);
        };
        auto issue_anchors = [&](int n0) {         // the 16 anchor rows of nodes n0, n0 + 1 (load layout: anchor index g)
            int al = 2 * (n0 - nfirst) + g;        // lane that holds anchor (n0 + (g >> 1), side g & 1)
            al = al > 63 ? 63 : al;
            const int id = __shfl(aid, al), cp = __shfl(am.copy, al);
            pre_g = __shfl(am.g, al);
            load_rows<KS>(row_ptr(id, cp, l_layer), ra);
            pre_n0 = n0;
        };
        issue_tile(0);
        for (int t = 0; t < ntile; ++t) {
            const int n0 = __builtin_amdgcn_readlane(node, 4 * t);
            // ---- the anchors' 16 rows (nodes n0 and n0 + 1) and their own blocks, when the tile starts in a new node
            if (n0 != cur_n0) {
                if (pre_n0 != n0) issue_anchors(n0);
                cur_n0 = n0;
                const int sh = 2 * (n0 - nfirst);
                aok0 = ((abadmask >> sh) & 3ull) == 0;
                aok1 = sh + 2 < 64 ? ((abadmask >> (sh + 2)) & 3ull) == 0 : true;
                to_operands<KS, SPLIT>(ra, decay_pow(pre_g, l_layer), pull, anch);
                daa = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int s = 0; s < KS; ++s) daa = mm_step<SPLIT>(anch[s], anch[s], daa);
            }
            // ---- this tile's rows -> operand lanes; the next tile's rows (and anchors) ride under the matrix work; then step by
            // step: split of step s + 1 under the products of step s
            if constexpr (HANDOFF) EM_STAMP(4 + 4 * tile_no);
            // (the neighbours' pending decay g^layer scales the 4 x 4 blocks below, 10 multiplications per lane, instead of the 32
            // loaded values: layer a of neighbour g against layer b of anything carries g^a, and the anchors' operands are scaled
            // when they are formed, once per anchor set)
            float av[KS][8];
            const float gd = __shfl(mv.g, 4 * t + g);
            to_lanes<KS, false>(raw, 1.0f, pull, av);
            if (t + 1 < ntile) {
                issue_tile(t + 1);
                const int n0n = __builtin_amdgcn_readlane(node, 4 * (t + 1));
                if (n0n != cur_n0) issue_anchors(n0n);
            }
            if constexpr (HANDOFF) EM_STAMP(5 + 4 * tile_no);
            f32x4 dww = {0.0f, 0.0f, 0.0f, 0.0f}, dwa = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                SplitOp<SPLIT> aop;
                split8<SPLIT>(av[s], aop);
                mm_step2<SPLIT>(aop, anch[s], dww, dwa);
            }
            {
                const float p2 = gd * gd, p3 = p2 * gd;            // g, g*g, (g*g)*g: decay_pow's association
                const int cbq = c & 3;
                const float scol = cbq == 0 ? 1.0f : cbq == 1 ? gd : cbq == 2 ? p2 : p3;   // the w.w block's column: layer c & 3 of the same neighbour
                dwa[1] *= gd; dwa[2] *= p2; dwa[3] *= p3;
                dww[0] *= scol; dww[1] *= gd * scol; dww[2] *= p2 * scol; dww[3] *= p3 * scol;
            }
            // ---- accumulators -> an LDS tile of 8 feature rows (pair p = side * 4 + neighbour; element 8 a + b of the
            // [w rows | anchor rows]^2 Gram).  Lane (c, g) holds D[4 g + q][c], q = 0..3.
            uint32_t nimask = 0, okmask = 0, inmask = 0;                  // (wave-uniform) bit nb: neighbour nb sits in node n0 + 1 /
#pragma unroll                                                            // is a valid pair / is inside this wave's share
            for (int nb = 0; nb < 4; ++nb) {
                const uint32_t ni = (uint32_t)(__builtin_amdgcn_readlane(node, 4 * t + nb) - n0) & 1u;
                const bool inn = base + 4 * t + nb < slot_end;
                const bool okk = (__builtin_amdgcn_readlane((int)wok, 4 * t + nb) != 0) && (ni ? aok1 : aok0);
                nimask |= ni << nb;
                inmask |= (inn ? 1u : 0u) << nb;
                okmask |= (okk ? 1u : 0u) << nb;
            }
            const int ni_g = (int)((nimask >> g) & 1u);
            const bool pok_g = (okmask >> g) & 1u;
            const bool pin_g = (inmask >> g) & 1u;
            float* tb = stg;
            if constexpr (HANDOFF) EM_STAMP(6 + 4 * tile_no);
            const int cb = c & 3, cn = c >> 2;
            if (cn == g) {                                                // w.w block of neighbour g: (a = q, b = cb), both sides
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    tb[g * EM_RS + 8 * q + cb] = dww[q];
                    tb[(4 + g) * EM_RS + 8 * q + cb] = dww[q];
                }
            }
            if ((c >> 3) == ni_g) {                                       // w.anchor block: column c = anchor (c >> 3, side (c >> 2) & 1), layer cb
                float* row = tb + (((c >> 2) & 1) * 4 + g) * EM_RS;
#pragma unroll
                for (int q = 0; q < 4; ++q) row[8 * q + 4 + cb] = dwa[q];
                *reinterpret_cast<f32x4*>(row + 32 + 8 * cb) = dwa;       // mirrored: (a = 4 + cb, b = 0..3)
            }
            if (cn == g) {                                                // anchor g's own block -> every neighbour of its node
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    if ((int)((nimask >> nb) & 1u) == (g >> 1)) {
                        float* row = tb + ((g & 1) * 4 + nb) * EM_RS;
#pragma unroll
                        for (int q = 0; q < 4; ++q) row[8 * (4 + q) + 4 + cb] = daa[q];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();       // one wave: LDS executes in issue order
            char* hb = nullptr;
            if constexpr (HANDOFF) {
                hb = reinterpret_cast<char*>(stg) + EMF_TILE + (tile_no & 1) * EMF_HAND;
                // the hand-off buffer's previous tile (two tiles ago) must have been taken
                if (tile_no >= 2)
                    while (__hip_atomic_load(sync + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < tile_no - 1)
                        __builtin_amdgcn_s_sleep(1);
            }
            // ---- feature rows out: clamp, log(x + 1) (models/TPNet.py:127-128), 16 lanes per 256-byte row
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int idx4 = it * 64 + lane;
                const int p = idx4 >> 4, col4 = idx4 & 15;         // p = 4 it + g: side it, neighbour g
                f32x4 v = *reinterpret_cast<const f32x4*>(tb + p * EM_RS + 4 * col4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float x = v[k];
                    if (do_scale) {
                        x = (x < 0.0f) ? 0.0f : x;     // NaN < 0 is false: NaN passes through, as in the reference (:127)
                        // log(x + 1), not log1p (:128).  The argument is >= 1 (or NaN): none of logf's range handling is needed, and
                        // v_log_f32 (1 ulp) times ln 2 is within 2 ulp of it -- 4 instructions instead of 14 in a kernel whose vector
                        // and matrix instructions do not overlap (tools/probes/mfma_valu_overlap.hip)
                        x = __builtin_amdgcn_logf(x + 1.0f) * 0.693147180559945309f;
                    }
                    v[k] = x;
                }
                if (!pok_g) v = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
                if constexpr (HANDOFF) {
                    // the dense layers' B operand, already in two pieces (here every lane holds 4 features: the split is dense):
                    // feature f = 4 col4 + k of pair p sits in step s = f >> 5, k-group (f >> 3) & 3, element f & 7
                    typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
                    bf16x4 hi, lo;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const __bf16 hh = (__bf16)v[k];
                        hi[k] = hh;
                        lo[k] = (__bf16)(v[k] - (float)hh);
                    }
                    char* e = hb + (((col4 >> 3) * 4 + ((col4 >> 1) & 3)) * 8 + p) * 16 + (col4 & 1) * 8;
                    *reinterpret_cast<bf16x4*>(e) = hi;
                    *reinterpret_cast<bf16x4*>(e + 1024) = lo;
                }
                if (pin_g && (!HANDOFF || out1 != nullptr)) {
                    float* o = (it ? out2 : out1) + (int64_t)(base + 4 * t + g) * 64 + 4 * col4;
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(o));
                }
            }
            if constexpr (HANDOFF) {
                if (tile_no == 0) __syncthreads();     // the workgroup's one barrier: the hand-off words are zero, the weights in LDS
                const uint32_t okb = (okmask & inmask) * 0x11u;           // pair p = side * 4 + neighbour
                if (lane == 0) {
                    sync[2 + 2 * (tile_no & 1)] = base + 4 * t;
                    sync[3 + 2 * (tile_no & 1)] = (int)okb;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(sync, tile_no + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                EM_STAMP(7 + 4 * tile_no);
            }
            ++tile_no;
            __builtin_amdgcn_wave_barrier();       // the tile is rewritten by a later one
        }
    }
    return tile_no;
}

// out1 / out2: the pre-mlp features of the two anchor sides, rows [slot][64]
template <int KS, int SPLIT>
__global__ __launch_bounds__(EMB) void k_encoder_gram_mfma(tpnet_state S, const int64_t* __restrict__ neigh,
                                                           const int64_t* __restrict__ a1, const int64_t* __restrict__ a2,
                                                           int n_rows, int K, int T, int tpw, double now, double lambda,
                                                           uint32_t flags, float* __restrict__ out1, float* __restrict__ out2) {
    __shared__ __attribute__((aligned(16))) float stg_all[(EMB / 64) * 8 * EM_RS];
    const int wave = threadIdx.x >> 6;
    const int wid = blockIdx.x * (EMB / 64) + wave;
    const int slot_begin = wid * tpw * 4;
    const int slot_end = (slot_begin + tpw * 4 < T) ? slot_begin + tpw * 4 : T;
    (void)walk_tiles<KS, SPLIT, false>(S, neigh, a1, a2, n_rows, K, T, slot_begin, slot_end, now, lambda, flags, out1, out2,
                                       stg_all + wave * (8 * EM_RS), nullptr);
}

// ReLU in ONE instruction: the signed-integer maximum of the bits and 0 (fmaxf / med3 cost a canonicalising v_max first; inline
// assembly would hide the matrix-result hazard from the compiler's wait-state insertion -- it did, with garbage features).
// Negative floats (-0 included) have the sign bit set = negative integers; a positive NaN stays a NaN, as in torch's relu.
__device__ __forceinline__ float relu1(float x) {
    const int b = __float_as_int(x);
    return __int_as_float(b > 0 ? b : 0);
}

// the dense layers of one consumer wave: takes the tiles its producer publishes, 16 rows per pass
__device__ __forceinline__ void dense_consumer(const char* smem, const float* stg, int* sync, int n_tiles, int T, int slot_end,
                                               float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const bf16x8* W1H = reinterpret_cast<const bf16x8*>(smem + IMG_W1H) + lane;
    const bf16x8* W1L = reinterpret_cast<const bf16x8*>(smem + IMG_W1L) + lane;
    const bf16x8* W2H = reinterpret_cast<const bf16x8*>(smem + IMG_W2H) + lane;
    const bf16x8* W2L = reinterpret_cast<const bf16x8*>(smem + IMG_W2L) + lane;
    const f32x4* B1 = reinterpret_cast<const f32x4*>(smem + IMG_B1) + g;
    const f32x4* B2 = reinterpret_cast<const f32x4*>(smem + IMG_B2) + g;
    SplitOp<2> bx[2];                              // B operand of layer 1: rows 0..7 = the even tile's pairs, 8..15 the odd one's
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int k = 0; k < 8; ++k) bx[s].p[t][k] = (__bf16)0.0f;
    int slot0 = 0;                                 // this lane's row: first slot and validity bits of its tile
    uint32_t okbits = 0;
    bool have = false;
    for (int k = 0; k < n_tiles; ++k) {
        const int e = k & 1;
        while (__hip_atomic_load(sync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < k + 1) __builtin_amdgcn_s_sleep(1);
        EM_STAMP(4 + 4 * k);
        const char* hb = reinterpret_cast<const char*>(stg) + EMF_TILE + e * EMF_HAND;
        if ((c >> 3) == e) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bx[s].p[0] = *reinterpret_cast<const bf16x8*>(hb + ((s * 4 + g) * 8 + (c & 7)) * 16);
                bx[s].p[1] = *reinterpret_cast<const bf16x8*>(hb + 1024 + ((s * 4 + g) * 8 + (c & 7)) * 16);
            }
            slot0 = sync[2 + 2 * e];
            okbits = (uint32_t)sync[3 + 2 * e];
            have = true;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // (the reads above are done before the buffer is given back)
        if (lane == 0) __hip_atomic_store(sync + 1, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (e == 0 && k + 1 < n_tiles) continue;   // wait for the odd tile of the pair
        // ---- the dense layers on the 16 rows
        EM_STAMP(5 + 4 * k);
        f32x4 yo[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) yo[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        // layer 1 of hidden slices 2 k2, 2 k2 + 1 (two accumulation chains taking turns, started from the bias: accumulator q of
        // lane (r, g) = hidden unit 16 w + 4 g + q of row r)
        auto layer1 = [&](int k2, f32x4& a0, f32x4& a1) {
            a0 = B1[(2 * k2) * 4];
            a1 = B1[(2 * k2 + 1) * 4];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 h0 = W1H[((2 * k2) * 2 + s) * 64], l0 = W1L[((2 * k2) * 2 + s) * 64];
                const bf16x8 h1 = W1H[((2 * k2 + 1) * 2 + s) * 64], l1 = W1L[((2 * k2 + 1) * 2 + s) * 64];
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0, bx[s].p[0], a0, 0, 0, 0);       // the small terms first
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1, bx[s].p[0], a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, bx[s].p[1], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, bx[s].p[1], a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, bx[s].p[0], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, bx[s].p[0], a1, 0, 0, 0);
            }
        };
        f32x4 a0, a1;
        layer1(0, a0, a1);
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            // the next slices' layer 1 enters the matrix pipe before this pair's ReLU + split: the vector work runs in its shadow
            f32x4 n0 = a0, n1 = a1;
            if (k2 + 1 < 8) layer1(k2 + 1, n0, n1);
            SplitOp<2> bh;                         // layer 2's B operand of this 32-deep step: hidden slices 2 k2 and 2 k2 + 1
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x0 = relu1(a0[q]), x1 = relu1(a1[q]);
                const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;
                bh.p[0][q] = h0;
                bh.p[0][4 + q] = h1;
                bh.p[1][q] = (__bf16)(x0 - (float)h0);
                bh.p[1][4 + q] = (__bf16)(x1 - (float)h1);
            }
            bf16x8 wh[4], wl[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { wh[t] = W2H[(k2 * 4 + t) * 64]; wl[t] = W2L[(k2 * 4 + t) * 64]; }
#pragma unroll
            for (int t = 0; t < 4; ++t) yo[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[t], bh.p[0], yo[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) yo[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[t], bh.p[1], yo[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) yo[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[t], bh.p[0], yo[t], 0, 0, 0);
            a0 = n0;
            a1 = n1;
        }
        // lane (r = c, g) holds outputs 16 t + 4 g + q of row r: pair p = r & 7 (side p >> 2, neighbour p & 3) of tile r >> 3
        const int p = c & 7;
        const int slot = slot0 + (p & 3);
        if (have && slot < slot_end) {
            const bool ok = (okbits >> p) & 1u;
            float* o = y + ((int64_t)(p >> 2) * T + slot) * 64 + 4 * g;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 bb = B2[t * 4];
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = ok ? yo[t][q] + bb[q] : __builtin_nanf("");
                *reinterpret_cast<f32x4*>(o + 16 * t) = v;
            }
        }
        have = false;
        EM_STAMP(6 + 4 * k);
    }
}

// gram (may be null): the pre-mlp features [2][T][64]; y = self.mlp(features), rows [side * T + slot]
template <int KS, int SPLIT>
__global__ __launch_bounds__(EMF_B) void k_encoder_fused(tpnet_state S, const int64_t* __restrict__ neigh, const int64_t* __restrict__ a1,
                                                         const int64_t* __restrict__ a2, int n_rows, int K, int T, int tpw, double now,
                                                         double lambda, uint32_t flags, float* __restrict__ gram,
                                                         const float4* __restrict__ wimg, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6;
    EM_STAMP(0);
    // ONE workgroup barrier: the consumer waves reach it when they have copied the weight image to LDS (only they read it), the
    // producer waves right before they publish their first tile -- ids, meta records, rows and the first tile's products are
    // under way meanwhile
    if (wave >= EMF_NP) {
        const int ct = threadIdx.x - EMF_NP * 64;
        constexpr int NV = IMG_BYTES / 16, CT = (EMF_B - EMF_NP * 64);
#pragma unroll 1
        for (int i0 = 0; i0 < NV; i0 += CT * 11) {
            float4 tmp[11];
#pragma unroll
            for (int u = 0; u < 11; ++u) {
                const int i = i0 + u * CT + ct;
                tmp[u] = i < NV ? wimg[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 11; ++u) {
                const int i = i0 + u * CT + ct;
                if (i < NV) reinterpret_cast<float4*>(smem)[i] = tmp[u];
            }
        }
        if (ct < EMF_NP * 8) reinterpret_cast<int*>(smem + EMF_SYNC)[ct] = 0;      // the hand-off words start at zero
        __syncthreads();
        EM_STAMP(1);
    }
    const int p = wave & (EMF_NP - 1);             // producer p = wave p, its consumer = wave 4 + p: the same SIMD
    const int wid = blockIdx.x * EMF_NP + p;
    const int slot_begin = (wid * tpw * 4 < T) ? wid * tpw * 4 : T;
    const int slot_end = (slot_begin + tpw * 4 < T) ? slot_begin + tpw * 4 : T;
    float* stg = reinterpret_cast<float*>(smem + EMF_STG + p * EMF_PP);
    int* sync = reinterpret_cast<int*>(smem + EMF_SYNC) + p * 8;
    if (wave < EMF_NP) {
        const int done = walk_tiles<KS, SPLIT, true>(S, neigh, a1, a2, n_rows, K, T, slot_begin, slot_end, now, lambda, flags, gram,
                                                     gram ? gram + (int64_t)T * 64 : nullptr, stg, sync);
        if (done == 0) __syncthreads();            // (a producer without tiles still owes the workgroup its barrier)
    } else
        dense_consumer(smem, stg, sync, (slot_end - slot_begin + 3) >> 2, T, slot_end, y);
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

// ---- the fused variant: readout + self.mlp in one launch -------------------------------------------------------------------
// 0: not decided; 1: available; -1: this device / runtime does not give a workgroup 150 KB of LDS
static int encoder_fused_state = 0;

static bool encoder_fused_available() {
    if (encoder_fused_state == 0) {
        static const int off = TPNET_DEV_INT(NO_ENCODER_FUSED, 0);
        bool ok = !off;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(k_encoder_fused<4, 3>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, EMF_LDS) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(k_encoder_fused<2, 3>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, EMF_LDS) == hipSuccess;
        (void)hipGetLastError();
        encoder_fused_state = ok ? 1 : -1;
    }
    return encoder_fused_state == 1;
}

bool encoder_fused_supported(const tpnet_state& st, int64_t n_rows, int K, const tpnet_mlp* mlp) {
    return mlp && mlp->wimg && mlp->F == 64 && mlp->H == 256 && encoder_mfma_supported(st, n_rows, K) && encoder_fused_available();
}

int launch_encoder_fused(const tpnet_state& st, const int64_t* neigh, const int64_t* a1, const int64_t* a2, int64_t n_rows, int K,
                         double now, double lambda, uint32_t flags, const tpnet_mlp* mlp, float* gram, float* out, hipStream_t s) {
    if (n_rows == 0 || K == 0) return TPNET_OK;
    if (!encoder_fused_supported(st, n_rows, K, mlp) || (flags & TPNET_FLAG_PACKED)) return TPNET_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(gram) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(mlp->wimg)) & 15)
        return TPNET_ERR_BAD_ARG;
    const int T = (int)(n_rows * K);
    const int ntiles = (T + 3) / 4;
    // ONE workgroup per CU (its LDS holds the weight image), four producer waves each: tiles per producer for one round
    static const int wg_dev = TPNET_DEV_INT(ENCODER_FUSED_WGS, 256);
    int tpw = (ntiles + wg_dev * EMF_NP - 1) / (wg_dev * EMF_NP);
    if (tpw < 2) tpw = 2;
    const int nprod = (ntiles + tpw - 1) / tpw;
    const int grid = (nprod + EMF_NP - 1) / EMF_NP;
    if (st.d == 128)
        hipLaunchKernelGGL((k_encoder_fused<4, 3>), dim3(grid), dim3(EMF_B), EMF_LDS, s, st, neigh, a1, a2, (int)n_rows, K, T, tpw, now,
                           lambda, flags, gram, reinterpret_cast<const float4*>(mlp->wimg), out);
    else
        hipLaunchKernelGGL((k_encoder_fused<2, 3>), dim3(grid), dim3(EMF_B), EMF_LDS, s, st, neigh, a1, a2, (int)n_rows, K, T, tpw, now,
                           lambda, flags, gram, reinterpret_cast<const float4*>(mlp->wimg), out);
    if (hipGetLastError() != hipSuccess) {             // (a runtime that refuses the launch: the callers fall back for good)
        encoder_fused_state = -1;
        return TPNET_ERR_BAD_ARG;
    }
    return TPNET_OK;
}

// ---- the weight image of the fused variant: [W1 hi | W1 lo | W2 hi | W2 lo | b1 | b2], every 16-byte element the operand of one
// lane of one matrix instruction.  W1 element ((w * 2 + s) * 64 + lane): W1[16 w + (lane & 15)][32 s + 8 (lane >> 4) + j];
// W2 element ((k2 * 4 + t) * 64 + lane): W2[16 t + (lane & 15)][16 (2 k2 + (j >> 2)) + 4 (lane >> 4) + (j & 3)], j = 0..7.
__global__ __launch_bounds__(256) void k_mlp_image(const float* __restrict__ w1, const float* __restrict__ b1,
                                                   const float* __restrict__ w2, const float* __restrict__ b2,
                                                   char* __restrict__ img) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 2048) {
        const int lane = i & 63, s = (i >> 6) & 1, w = i >> 7;
        const float* p = w1 + (16 * w + (lane & 15)) * 64 + 32 * s + 8 * (lane >> 4);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[j];
        SplitOp<2> o;
        split8<2>(v, o);
        reinterpret_cast<bf16x8*>(img + IMG_W1H)[i] = o.p[0];
        reinterpret_cast<bf16x8*>(img + IMG_W1L)[i] = o.p[1];
    } else if (i < 4096) {
        const int e = i - 2048;
        const int lane = e & 63, t = (e >> 6) & 3, k2 = e >> 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w2[(16 * t + (lane & 15)) * 256 + 16 * (2 * k2 + (j >> 2)) + 4 * (lane >> 4) + (j & 3)];
        SplitOp<2> o;
        split8<2>(v, o);
        reinterpret_cast<bf16x8*>(img + IMG_W2H)[e] = o.p[0];
        reinterpret_cast<bf16x8*>(img + IMG_W2L)[e] = o.p[1];
    } else if (i < 4096 + 256) {
        reinterpret_cast<float*>(img + IMG_B1)[i - 4096] = b1[i - 4096];
    } else if (i < 4096 + 256 + 64) {
        reinterpret_cast<float*>(img + IMG_B2)[i - 4352] = b2[i - 4352];
    }
}

}  // namespace tpnet

#ifdef TPNET_STAMPS
extern "C" int tpnet_dev_encoder_stamps(void* host_out, size_t bytes) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tpnet::g_em_stamps), bytes) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t tpnet_mlp_image_bytes(void) { return (size_t)tpnet::IMG_BYTES; }

extern "C" int tpnet_mlp_prepare_image(const float* w1, const float* b1, const float* w2, const float* b2, void* img, void* stream) {
    if (!w1 || !b1 || !w2 || !b2 || !img || (reinterpret_cast<uintptr_t>(img) & 15)) return TPNET_ERR_BAD_ARG;
    hipLaunchKernelGGL(tpnet::k_mlp_image, dim3((4096 + 320 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w1, b1, w2, b2, (char*)img);
    if (hipGetLastError() != hipSuccess) return TPNET_ERR_HIP;
    return TPNET_OK;
}
