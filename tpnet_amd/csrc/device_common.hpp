// Device-side building blocks shared by the kernels of the temporal-walk-matrix hot path (readout.hip, step.hip,
// tables.hip): geometry constants, vector loads / stores, the per-node meta record, the cross-lane reduction, the
// (geometry, L, exact-fit) dispatch.
//
// Geometry: LPP lanes of a 64-wide wavefront cooperate on one row (one pair, or one target node); each lane owns
// VPL vectors of W floats per column chunk, so a wave-level load instruction fetches whole rows with 16-byte lanes:
//   d = 64 -> 4 rows per instruction, d = 128 -> 2, d = 256 -> 1, d = 512 -> one row per two instructions.
// Compiled with -ffp-contract=off: the update path must not fuse (row*w)+acc, so that the eager/sequential mode
// reproduces the reference's rounding; the Gram uses explicit fmaf.
#pragma once
#include "tpnet_common.h"

#include <cstdlib>

namespace tpnet {

#ifdef TPNET_STAMPS
// diagnostic build only: lane 0 of every wave records (shader clock, 100 MHz wall clock) at phase boundaries
#define STAMP(slot)                                                                                        \
    do {                                                                                                   \
        if (dbg && (threadIdx.x & 63) == 0) {                                                              \
            const size_t wv = ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));                    \
            if (wv < 4000) {                                                                               \
                dbg[(wv * 8 + (slot)) * 2 + 0] = __builtin_amdgcn_s_memtime();                             \
                dbg[(wv * 8 + (slot)) * 2 + 1] = __builtin_amdgcn_s_memrealtime();                         \
            }                                                                                              \
        }                                                                                                  \
    } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

static constexpr uint32_t READER_BID = 0xFFFFFFFFu;  // never equals a stored (ver >> 1)
#ifndef TPNET_BLOCK
#define TPNET_BLOCK 512
#endif
static constexpr int BLOCK = TPNET_BLOCK;   // threads per workgroup of the step kernel for one-pass (small) batches
// Workgroups of 256 threads for everything that is many passes deep (large batches, long pair lists): with 146-175 VGPRs a
// CU holds 8-12 waves, i.e. ONE 512-thread workgroup, whose slots only free up when its slowest wave is done (measured mean
// occupancy at C3: 4.2 of 8 waves per CU); two 256-thread workgroups refill independently (C3 49 -> 37 us per batch,
// C5 -6 %, C1 -9 %).  A batch that fits ONE pass of 512-thread workgroups (400..1024 edges) keeps them: fewer
// workgroups to dispatch, 6 % faster at C2.
#ifndef TPNET_BLOCK_SMALL
#define TPNET_BLOCK_SMALL 256
#endif
static constexpr int BLOCK_SMALL = TPNET_BLOCK_SMALL;
// workgroups of a step launch reserved for heavy (item, layer) units: measured optima on MI355X (C2: 36-42, the
// one-pass grid leaves the rest to pairs and light items; B = 10 000: >= 128, the grid is multi-pass anyway)
// __launch_bounds__'s second argument in HIP-Clang = minimum waves per SIMD the kernel must fit (4 -> at most 128 VGPRs ->
// two 512-thread workgroups per CU; 2 -> no constraint below 256 VGPRs).
#ifndef TPNET_MINW16
#define TPNET_MINW16 2
#endif
#ifndef TPNET_MINW32
#define TPNET_MINW32 2
#endif
constexpr int min_waves_per_simd(int lpp, int vpl, int w) {
    return (w != 4) ? 2 : (lpp < 16) ? 4 : (lpp == 16 && vpl == 1) ? TPNET_MINW16 : (lpp == 32 && vpl == 1) ? TPNET_MINW32 : 2;
}
// the 256-thread step kernel on rows of 256 floats (32 lanes x 2 vectors: C3, C4) compiles to 173 VGPRs -- five registers above
// what three waves per SIMD allow (168), so a CU held 8 waves = 16 units in flight; C4 is bound by (units in flight) x (bytes
// per unit) / (latency of a unit's three dependent round trips), see profiles/r04_C4.md
#ifndef TPNET_MINW_WIDE256
#define TPNET_MINW_WIDE256 3
#endif
constexpr int step_min_waves(int lpp, int vpl, int w, int bs) {
    return (w == 4 && lpp == 32 && vpl == 2 && bs == BLOCK_SMALL) ? TPNET_MINW_WIDE256 : min_waves_per_simd(lpp, vpl, w);
}
static constexpr int HEAVY_BLOCKS_SMALL = 48, HEAVY_BLOCKS_LARGE = 128;

// ---------------------------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------------------------
static inline int grid_for(int64_t work_items, int per_block, int cap) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

template <int W>
__device__ __forceinline__ void ldv(const float* __restrict__ row, int vi, float* dst) {
    if constexpr (W == 4) {
        const float4 x = reinterpret_cast<const float4*>(row)[vi];
        dst[0] = x.x; dst[1] = x.y; dst[2] = x.z; dst[3] = x.w;
    } else {
        dst[0] = row[vi];
    }
}
// predicated load without a branch: out-of-range lanes read vector 0 of the (always valid) row and are zeroed
template <int W>
__device__ __forceinline__ void ldv_pred(const float* __restrict__ row, int vi, bool ok, float* dst) {
    ldv<W>(row, ok ? vi : 0, dst);
#pragma unroll
    for (int k = 0; k < W; ++k) dst[k] = ok ? dst[k] : 0.0f;
}
// NT: stream the new row out instead of leaving it dirty in this XCD's L2.  The end of a kernel has to write the dirty
// lines back anyway (the next launch's readers sit on all 8 XCDs), so for a mid-size batch doing it while the kernel
// runs shortens the kernel boundary (C2: 7.06 -> 6.81 us per step, d=64 B=2000: -3 %; slower at C1 and C3, where it is
// off).  Compile-time: LLVM sinks a nontemporal and a plain store of one value into ONE plain store.
template <int W, bool NT = false>
__device__ __forceinline__ void stv(float* __restrict__ row, int vi, const float* src) {
    if constexpr (W == 4) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f q = {src[0], src[1], src[2], src[3]};
        if constexpr (NT) __builtin_nontemporal_store(q, reinterpret_cast<v4f*>(row) + vi);
        else reinterpret_cast<v4f*>(row)[vi] = q;
    } else {
        row[vi] = src[0];
    }
}

// the writer's last act: the new copy's reference time, then the version word that makes the copy current
__device__ __forceinline__ void publish_meta(NodeMeta* m, int copy, double t_last, uint32_t bid) {
    m->tref[copy] = t_last;
    m->ver = (bid << 1) | (uint32_t)copy;
}

struct MetaView {
    int copy;  // copy holding the pre-launch bundle
    float g;   // exp(-lambda * (now - tref[copy])): pending decay of layer 1 (layer i: g^i)
};

// (a, b) = the two 16-byte halves of a node's meta record
__device__ __forceinline__ MetaView meta_view(const uint4 a, const uint4 b, uint32_t bid, double now, double lambda) {
    int c = (int)(a.x & 1u);
    if ((a.x >> 1) == bid) c ^= 1;  // rewritten by a concurrent group of THIS launch: the pre-batch copy is the other one
    const double t0 = __hiloint2double((int)a.w, (int)a.z);
    const double t1 = __hiloint2double((int)b.y, (int)b.x);
    const double tr = c ? t1 : t0;
    const float x = (float)(-lambda * (now - tr));
    MetaView m;
    m.copy = c;
    m.g = (x == 0.0f) ? 1.0f : expf(x);
    return m;
}
__device__ __forceinline__ MetaView read_meta(const NodeMeta* __restrict__ meta, int64_t n, uint32_t bid, double now,
                                              double lambda) {
    const uint4* p = reinterpret_cast<const uint4*>(meta + n);
    return meta_view(p[0], p[1], bid, now, lambda);
}

// Recursive halving: MP partial sums per lane over a group of 2*M lanes -> lane gl ends with the MP/(2M) complete sums
// of indices [gl*MP/(2M), ...).  63 exchanges for 64 values over 64 lanes instead of 6*64 for a butterfly all-reduce.
// Exchanges never touch LDS: v_permlane32_swap / v_permlane16_swap (gfx950) trade the two halves in one instruction
// (then one add, no select); inside a 16-lane row the partner comes through DPP (row_ror:8, row_half_mirror,
// quad_perm) fused into v_add_f32_dpp.  Any involution works as the pairing as long as the partners differ in the
// lane bit M that decides which half a lane keeps.
template <int M>
__device__ __forceinline__ float pair_sum(float a) {
    static_assert(M == 8 || M == 4 || M == 2 || M == 1, "row-level DPP pairing");
    constexpr int ctrl = (M == 8) ? 0x128 /* row_ror:8 */ : (M == 4) ? 0x141 /* row_half_mirror */
                       : (M == 2) ? 0x4E /* quad_perm [2,3,0,1] */ : 0xB1 /* quad_perm [1,0,3,2] */;
    const float p = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), ctrl, 0xF, 0xF, false));
    return a + p;
}

template <int C, int M>
struct Halve {
    static __device__ __forceinline__ void run(float* v, int gl) {
        if constexpr (M == 32) {
#pragma unroll
            for (int i = 0; i < C / 2; ++i) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + C / 2]),
                                                                false, false);
                v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
        } else if constexpr (M == 16) {
#pragma unroll
            for (int i = 0; i < C / 2; ++i) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + C / 2]),
                                                                false, false);
                v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
        } else {
            const bool upper = (gl & M) != 0;
#pragma unroll
            for (int i = 0; i < C / 2; ++i) {
                const float x = pair_sum<M>(v[i]);
                const float y = pair_sum<M>(v[i + C / 2]);
                v[i] = upper ? y : x;
            }
        }
        Halve<C / 2, M / 2>::run(v, gl);
    }
};
template <int C>
struct Halve<C, 0> {
    static __device__ __forceinline__ void run(float*, int) {}
};

// <x, y> over the F floats a lane holds of two rows.  F >= 8: even and odd elements are summed separately so that each
// step is ONE packed v_pk_fma_f32 on a register pair (half the issue slots of scalar FMAs: the readout of long pair
// lists is VALU-bound); F = 4 would pay the two extra adds with the two saved FMAs, so it keeps the scalar chain.
// (TPNET_NO_PKFMA: scalar everywhere, for A/B runs.)  PK = false for the shared-first-node kernel: its 62 slots plus
// register pairs cost it a wave of occupancy (128 -> 166 VGPRs, 800 000 pairs at d=256: 351 -> 382 us, measured), so its
// sums equal the generic kernel's bit for bit only where that one is scalar too (F = 4: d <= 128).
template <int F, bool PK = true>
__device__ __forceinline__ float dot_chunk(const float* x, const float* y) {
#ifndef TPNET_NO_PKFMA
    if constexpr (PK && F >= 8 && F % 2 == 0) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f s2 = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < F; k += 2) s2 = __builtin_elementwise_fma(v2f{x[k], x[k + 1]}, v2f{y[k], y[k + 1]}, s2);
        return s2.x + s2.y;
    } else
#endif
    {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < F; ++k) s = fmaf(x[k], y[k], s);
        return s;
    }
}

template <int LPP, int L>
struct GramCfg {
    static constexpr int NR = L + 1;
    static constexpr int NN = 2 * NR;
    static constexpr int NG = NN * NN;
    static constexpr int NT = NN * (NN + 1) / 2;   // distinct entries of the symmetric Gram (TPNET_FLAG_PACKED rows)
    static constexpr int MP = ((NG + LPP - 1) / LPP) * LPP;
    static constexpr int PER = MP / LPP;
    // Cross-lane reduction THROUGH LDS (LPP >= 16, 256-thread workgroups, L <= 3): every lane parks its NT distinct
    // partial sums in a [NT][LPP+4] tile of its group, lane v then adds row v (LPP values, float4 reads) and writes the
    // result to both mirrored positions of an [NG] output tile, from which every lane takes its PER consecutive outputs:
    // ~100 instructions and two LDS round trips instead of ~150 dependent DPP / permlane operations (measured on the
    // VALU-bound readout of long pair lists).  Wider workgroups would need more than 64 KB of LDS: they keep Halve.
#ifndef TPNET_LDSRED_MAX_KB
#define TPNET_LDSRED_MAX_KB 64
#endif
    static constexpr int RSTRIDE = LPP + 4;
    static constexpr int RED = NT * RSTRIDE + NG;                      // floats per group
    template <int BS>
    static constexpr bool lds_reduce() {
#ifdef TPNET_NO_LDSRED
        return false;
#else
        return LPP >= 16 && (BS / LPP) * RED * 4 <= TPNET_LDSRED_MAX_KB * 1024;
#endif
    }
    // LDS floats a workgroup of BS threads needs for the readout: the staged store of the narrow rows, or the reduction
    template <int BS>
    static constexpr int stage_floats() { return (LPP < 16) ? NG * (BS / LPP) : lds_reduce<BS>() ? RED * (BS / LPP) : 1; }
};

// FULL = the row is exactly one chunk (d == LPP*VPL*W): no tail predicate, no chunk loop; callers guarantee that an
// inactive group points at row 0, so the load itself needs no guard either
template <int W, bool FULL>
__device__ __forceinline__ void ldv_maybe(const float* __restrict__ row, int vi, bool ok, float* dst) {
    if constexpr (FULL) ldv<W>(row, vi, dst); else ldv_pred<W>(row, vi, ok, dst);
}


// dispatch over (geometry, L, exact-fit)
#define TPNET_DISPATCH_L(LPP_, VPL_, W_, FULL_, CALL)                           \
    switch (st.L) {                                                             \
        case 1: { constexpr int LPP = LPP_, VPL = VPL_, W = W_, L = 1; constexpr bool FULL = FULL_; CALL; } break; \
        case 2: { constexpr int LPP = LPP_, VPL = VPL_, W = W_, L = 2; constexpr bool FULL = FULL_; CALL; } break; \
        case 3: { constexpr int LPP = LPP_, VPL = VPL_, W = W_, L = 3; constexpr bool FULL = FULL_; CALL; } break; \
        case 4: { constexpr int LPP = LPP_, VPL = VPL_, W = W_, L = 4; constexpr bool FULL = FULL_; CALL; } break; \
        default: return TPNET_ERR_BAD_ARG;                                      \
    }
#define TPNET_DISPATCH(CALL) TPNET_DISPATCH_G(pick_geom(st.d), CALL)
#define TPNET_DISPATCH_G(GEOM, CALL)                                            \
    do {                                                                        \
        const Geom gm = (GEOM);                                                 \
        const bool full = gm.w == 4 && st.d == gm.lpp * gm.vpl * 4;             \
        if (gm.w == 1) { TPNET_DISPATCH_L(64, 1, 1, false, CALL) }              \
        else if (gm.lpp == 4) { if (full) { TPNET_DISPATCH_L(4, 1, 4, true, CALL) } else { TPNET_DISPATCH_L(4, 1, 4, false, CALL) } } \
        else if (gm.lpp == 8) { if (full) { TPNET_DISPATCH_L(8, 1, 4, true, CALL) } else { TPNET_DISPATCH_L(8, 1, 4, false, CALL) } } \
        else if (gm.lpp == 16 && gm.vpl == 2) { if (full) { TPNET_DISPATCH_L(16, 2, 4, true, CALL) } else { TPNET_DISPATCH_L(16, 2, 4, false, CALL) } } \
        else if (gm.lpp == 16) { if (full) { TPNET_DISPATCH_L(16, 1, 4, true, CALL) } else { TPNET_DISPATCH_L(16, 1, 4, false, CALL) } } \
        else if (gm.lpp == 32 && gm.vpl == 1) { if (full) { TPNET_DISPATCH_L(32, 1, 4, true, CALL) } else { TPNET_DISPATCH_L(32, 1, 4, false, CALL) } } \
        else if (gm.lpp == 32) { if (full) { TPNET_DISPATCH_L(32, 2, 4, true, CALL) } else { TPNET_DISPATCH_L(32, 2, 4, false, CALL) } } \
        else if (gm.vpl == 1) { if (full) { TPNET_DISPATCH_L(64, 1, 4, true, CALL) } else { TPNET_DISPATCH_L(64, 1, 4, false, CALL) } }  \
        else { if (full) { TPNET_DISPATCH_L(64, 2, 4, true, CALL) } else { TPNET_DISPATCH_L(64, 2, 4, false, CALL) } }                   \
    } while (0)

}  // namespace tpnet
