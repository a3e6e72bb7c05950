// gfx950 kernels of the temporal-walk-matrix hot path: pairwise Gram readout, batched scatter-add update, the
// fused per-batch "step", and the dense state passes (init / import / export / decay / row gather).
//
// Reference semantics restated by these kernels: models/TPNet.py:67-99 (update), :101-110 (row gather),
// :112-128 (pairwise readout before self.mlp), :131-157 (reset / backup / reload).
//
// Geometry: LPP lanes of a 64-wide wavefront cooperate on one row (one pair, or one target node); each lane owns
// VPL vectors of W floats per column chunk, so a wave-level load instruction fetches whole rows with 16-byte lanes:
//   d = 64 -> 4 rows per instruction, d = 128 -> 2, d = 256 -> 1, d = 512 -> one row per two instructions.
// Compiled with -ffp-contract=off: the update path must not fuse (row*w)+acc, so that the eager/sequential mode
// reproduces the reference's rounding; the Gram uses explicit fmaf.
#include "tpnet_common.h"

#include <cstdlib>

namespace tpnet {

#ifdef TPNET_STAMPS
// diagnostic build only: lane 0 of every wave records (shader clock, 100 MHz wall clock) at phase boundaries
#define STAMP(slot)                                                                                        \
    do {                                                                                                   \
        if (dbg && (threadIdx.x & 63) == 0) {                                                              \
            const size_t wv = ((size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6));                    \
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
static constexpr int BLOCK = TPNET_BLOCK;   // threads per workgroup of the pair/step kernels
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
static constexpr int HEAVY_BLOCKS_SMALL = 40, HEAVY_BLOCKS_MEDIUM = 64, HEAVY_BLOCKS_LARGE = 128;

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

__device__ __forceinline__ MetaView read_meta(const NodeMeta* __restrict__ meta, int64_t n, uint32_t bid, double now,
                                              double lambda) {
    const uint4* p = reinterpret_cast<const uint4*>(meta + n);
    const uint4 a = p[0];
    const uint4 b = p[1];
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
    static constexpr int STAGE = (LPP < 16) ? NG * (BLOCK / LPP) : 1;   // LDS floats of the staged store (narrow rows)
};

// FULL = the row is exactly one chunk (d == LPP*VPL*W): no tail predicate, no chunk loop; callers guarantee that an
// inactive group points at row 0, so the load itself needs no guard either
template <int W, bool FULL>
__device__ __forceinline__ void ldv_maybe(const float* __restrict__ row, int vi, bool ok, float* dst) {
    if constexpr (FULL) ldv<W>(row, vi, dst); else ldv_pred<W>(row, vi, ok, dst);
}

// ---------------------------------------------------------------------------------------------------------------
// pairwise readout of ONE pair by one group of LPP lanes (models/TPNet.py:119-128)
// ---------------------------------------------------------------------------------------------------------------
// PACKED (TPNET_FLAG_PACKED): only the NT distinct entries a <= b are written, raw, row-major upper triangle.
// `stage`: LDS, GramCfg::STAGE floats per workgroup, used by the narrow geometries (LPP < 16) only: there a lane ends up
// with 8 or 16 consecutive outputs, so a direct store touches one 64-byte line per lane and instruction; the values go
// through LDS instead and leave as whole lines (measured at d=16, B=8000: the store phase of a readout wave 5.6 -> 0.7 us).
template <int LPP, int VPL, int W, int L, bool FULL, bool PACKED = false>
__device__ __forceinline__ void gram_pair(const tpnet_state& S, int64_t u, int64_t v, bool valid, uint32_t bid,
                                          double now, double lambda, bool do_scale, float* __restrict__ out, int gl,
                                          unsigned long long* dbg = nullptr, float* __restrict__ stage = nullptr) {
    using C = GramCfg<LPP, L>;
    constexpr int NR = C::NR, NN = C::NN, F = VPL * W;
    const int d = S.d;
    const int nvec = d / W;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(S.meta);

    bool idok = valid && (uint64_t)u < (uint64_t)S.N && (uint64_t)v < (uint64_t)S.N;
    if (valid && !idok && gl == 0) atomicAdd(S.err, 1u);
    if (!idok) { u = 0; v = 0; }

    STAMP(1);
    const float* rowp[NN];
    float rs[NN];
    {
        const int64_t ids[2] = {u, v};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const MetaView m = read_meta(meta, ids[s], bid, now, lambda);
            rowp[s * NR] = S.p0 + ids[s] * (int64_t)d;
            rs[s * NR] = 1.0f;
            const float* qb = S.q + ((int64_t)m.copy * S.N + ids[s]) * ((int64_t)L * d);
            float g = 1.0f;
#pragma unroll
            for (int i = 1; i <= L; ++i) {
                g *= m.g;
                rowp[s * NR + i] = qb + (int64_t)(i - 1) * d;
                rs[s * NR + i] = g;
            }
        }
    }

    float acc[C::MP];
#pragma unroll
    for (int i = 0; i < C::MP; ++i) acc[i] = 0.0f;
    STAMP(2);

    for (int c0 = 0; c0 < (FULL ? 1 : nvec); c0 += LPP * VPL) {
        float f[NN][F];
#pragma unroll
        for (int a = 0; a < NN; ++a) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const int vi = c0 + j * LPP + gl;
                ldv_maybe<W, FULL>(rowp[a], vi, vi < nvec, &f[a][j * W]);
            }
        }
#pragma unroll
        for (int a = 0; a < NN; ++a) {
            if (a % NR != 0) {
#pragma unroll
                for (int k = 0; k < F; ++k) f[a][k] *= rs[a];
            }
        }
#pragma unroll
        for (int a = 0; a < NN; ++a) {
#pragma unroll
            for (int b = a; b < NN; ++b) acc[a * NN + b] = acc[a * NN + b] + dot_chunk<F>(f[a], f[b]);
        }
    }
#pragma unroll
    for (int a = 1; a < NN; ++a) {
#pragma unroll
        for (int b = 0; b < a; ++b) acc[a * NN + b] = acc[b * NN + a];
    }

    STAMP(3);
    Halve<C::MP, LPP / 2>::run(acc, gl);
    STAMP(4);

    if constexpr (LPP < 16) {
        // finished values -> this group's LDS row (full or packed layout) -> whole-line stores
        constexpr int NOUT = PACKED ? C::NT : C::NG;
        float* sg = stage + (threadIdx.x / LPP) * C::NG;
#pragma unroll
        for (int k = 0; k < C::PER; ++k) {
            const int idx = gl * C::PER + k;
            if (idx < C::NG) {
                float x = acc[k];
                if (do_scale && !PACKED) {
                    x = (x < 0.0f) ? 0.0f : x;  // NaN < 0 is false: NaN passes through, as in the reference (:127)
                    x = logf(x + 1.0f);          // log(x + 1), not log1p (:128)
                }
                if (!idok) x = __builtin_nanf("");
                if constexpr (PACKED) {
                    const int a = idx / NN, b = idx - a * NN;
                    if (a <= b) sg[a * NN - (a * (a - 1)) / 2 + (b - a)] = x;
                } else {
                    sg[idx] = x;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();            // the group's lanes are in one wave: LDS executes in issue order
        if constexpr (NOUT % 4 == 0) {
#pragma unroll
            for (int j = 0; j * LPP < NOUT / 4; ++j) {
                const int c = j * LPP + gl;
                if (valid && c < NOUT / 4) {
                    typedef float v4f __attribute__((ext_vector_type(4)));   // rows are 16-byte aligned (checked by the launchers)
                    const v4f q = reinterpret_cast<const v4f*>(sg)[c];
                    __builtin_nontemporal_store(q, reinterpret_cast<v4f*>(out) + c);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j * LPP < NOUT; ++j) {
                const int c = j * LPP + gl;
                if (valid && c < NOUT) __builtin_nontemporal_store(sg[c], out + c);
            }
        }
        __builtin_amdgcn_wave_barrier();            // the row is reused by the next pair of this group
    } else if (valid) {
#pragma unroll
        for (int k = 0; k < C::PER; ++k) {
            const int idx = gl * C::PER + k;
            if (idx < C::NG) {
                float x = acc[k];
                if (do_scale && !PACKED) {
                    x = (x < 0.0f) ? 0.0f : x;  // NaN < 0 is false: NaN passes through, as in the reference (:127)
                    x = logf(x + 1.0f);          // log(x + 1), not log1p (:128)
                }
                if (!idok) x = __builtin_nanf("");
                if constexpr (PACKED) {
                    const int a = idx / NN, b = idx - a * NN;
                    if (a <= b) __builtin_nontemporal_store(x, out + (a * NN - (a * (a - 1)) / 2 + (b - a)));
                } else {
                    __builtin_nontemporal_store(x, out + idx);   // features are consumed by another kernel: stream them out
                }
            }
        }
    }
}

template <int LPP, int VPL, int W, int L, bool FULL>
__global__ __launch_bounds__(BLOCK) void k_pair_gram(tpnet_state S, const int64_t* __restrict__ u,
                                                     const int64_t* __restrict__ v, int64_t n, double now,
                                                     double lambda, uint32_t flags, float* __restrict__ out) {
    constexpr int GPB = BLOCK / LPP;
    constexpr int NG = GramCfg<LPP, L>::NG;
    const int gl = threadIdx.x % LPP;
    const int g = threadIdx.x / LPP;
    const bool packed = (flags & TPNET_FLAG_PACKED) != 0;
    const bool do_scale = !(flags & TPNET_FLAG_NOT_SCALE) && !packed;
    const int stride = packed ? GramCfg<LPP, L>::NT : NG;
    __shared__ float stage[GramCfg<LPP, L>::STAGE];
    for (int64_t base = (int64_t)blockIdx.x * GPB; base < n; base += (int64_t)gridDim.x * GPB) {
        const int64_t p = base + g;
        const bool valid = p < n;
        const int64_t uu = valid ? u[p] : 0, vv = valid ? v[p] : 0;
        if (packed)
            gram_pair<LPP, VPL, W, L, FULL, true>(S, uu, vv, valid, READER_BID, now, lambda, do_scale, out + p * stride,
                                                  gl, nullptr, stage);
        else
            gram_pair<LPP, VPL, W, L, FULL, false>(S, uu, vv, valid, READER_BID, now, lambda, do_scale, out + p * stride,
                                                   gl, nullptr, stage);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// pairwise readout of TWO pairs that share their first node: out1 = G(u, v1), out2 = G(u, v2).  u's rows are loaded
// once.  This is the shape of both callers of the readout: the decoder's (src,dst) / (src,neg) pairs
// (models/modules.py:112, train_link_prediction.py:359-368) and the encoder's relative encodings, where every
// neighbour w is paired with the edge's src AND dst (models/TPNet.py:311-316: first half of the pair list =
// G(w, src), second half = G(w, dst)).
// ---------------------------------------------------------------------------------------------------------------
// Only the distinct inner products are formed and reduced, in ONE recursive-halving pass for both outputs:
//   slots = [ u.u (tri) | u.v1 (R*R) | u.v2 (R*R) | v1.v1 (tri) | v2.v2 (tri) ],  R = L+1, tri = R(R+1)/2
// = 62 slots at L = 3 (one 64-value reduction instead of two, 62 dot products instead of 72, 12 row loads instead of
// 16); afterwards every lane picks the (at most two per output) slots its output elements mirror from by shuffles.
template <int LPP, int L>
struct SharedCfg {
    static constexpr int R = L + 1;
    static constexpr int TRI = R * (R + 1) / 2;
    static constexpr int O_UU = 0, O_UV1 = TRI, O_UV2 = TRI + R * R, O_V1 = TRI + 2 * R * R, O_V2 = 2 * TRI + 2 * R * R;
    static constexpr int NS = 3 * TRI + 2 * R * R;
    static constexpr int MPS = ((NS + LPP - 1) / LPP) * LPP;
    static constexpr int PERS = MPS / LPP;
    static constexpr __host__ __device__ int tri(int i, int j) { return i * R - i * (i - 1) / 2 + (j - i); }   // i <= j
};

template <int LPP, int VPL, int W, int L, bool FULL>
__device__ __forceinline__ void gram_shared(const tpnet_state& S, int64_t u, int64_t v1, int64_t v2, bool valid,
                                            uint32_t bid, double now, double lambda, bool do_scale,
                                            float* __restrict__ out1, float* __restrict__ out2, int gl) {
    using C = GramCfg<LPP, L>;
    using SC = SharedCfg<LPP, L>;
    constexpr int NR = C::NR, NN = C::NN, F = VPL * W;
    const int d = S.d;
    const int nvec = d / W;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(S.meta);

    bool idok = valid && (uint64_t)u < (uint64_t)S.N && (uint64_t)v1 < (uint64_t)S.N && (uint64_t)v2 < (uint64_t)S.N;
    if (valid && !idok && gl == 0) atomicAdd(S.err, 1u);
    if (!idok) { u = 0; v1 = 0; v2 = 0; }

    const float* rowp[3][NR];
    float rs[3][NR];
    {
        const int64_t ids[3] = {u, v1, v2};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const MetaView m = read_meta(meta, ids[s], bid, now, lambda);
            rowp[s][0] = S.p0 + ids[s] * (int64_t)d;
            rs[s][0] = 1.0f;
            const float* qb = S.q + ((int64_t)m.copy * S.N + ids[s]) * ((int64_t)L * d);
            float g = 1.0f;
#pragma unroll
            for (int i = 1; i <= L; ++i) {
                g *= m.g;
                rowp[s][i] = qb + (int64_t)(i - 1) * d;
                rs[s][i] = g;
            }
        }
    }
    float acc[SC::MPS];
#pragma unroll
    for (int i = 0; i < SC::MPS; ++i) acc[i] = 0.0f;

    for (int c0 = 0; c0 < (FULL ? 1 : nvec); c0 += LPP * VPL) {
        float f[3][NR][F];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int a = 0; a < NR; ++a) {
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    const int vi = c0 + j * LPP + gl;
                    ldv_maybe<W, FULL>(rowp[s][a], vi, vi < nvec, &f[s][a][j * W]);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int a = 1; a < NR; ++a) {
#pragma unroll
                for (int k = 0; k < F; ++k) f[s][a][k] *= rs[s][a];
            }
        }
#pragma unroll
        for (int a = 0; a < NR; ++a) {
#pragma unroll
            for (int b = a; b < NR; ++b) {
                acc[SC::O_UU + SC::tri(a, b)] = acc[SC::O_UU + SC::tri(a, b)] + dot_chunk<F, false>(f[0][a], f[0][b]);
                acc[SC::O_V1 + SC::tri(a, b)] = acc[SC::O_V1 + SC::tri(a, b)] + dot_chunk<F, false>(f[1][a], f[1][b]);
                acc[SC::O_V2 + SC::tri(a, b)] = acc[SC::O_V2 + SC::tri(a, b)] + dot_chunk<F, false>(f[2][a], f[2][b]);
            }
#pragma unroll
            for (int b = 0; b < NR; ++b) {
                acc[SC::O_UV1 + a * NR + b] = acc[SC::O_UV1 + a * NR + b] + dot_chunk<F, false>(f[0][a], f[1][b]);
                acc[SC::O_UV2 + a * NR + b] = acc[SC::O_UV2 + a * NR + b] + dot_chunk<F, false>(f[0][a], f[2][b]);
            }
        }
    }
    Halve<SC::MPS, LPP / 2>::run(acc, gl);        // lane gl now holds the complete sums of slots [gl*PERS, ...)

    // output element idx = a*NN + b of the two Gram matrices -> the slot it comes from
#pragma unroll
    for (int k = 0; k < C::PER; ++k) {
        int idx = gl * C::PER + k;
        const bool in = idx < C::NG;
        idx = in ? idx : 0;
        const int a = idx / NN, b = idx - a * NN;
        int s1, s2;
        if (a < NR && b < NR) {
            const int i = a < b ? a : b, j = a < b ? b : a;
            s1 = s2 = SC::O_UU + SC::tri(i, j);
        } else if (a < NR) {                       // (u row a, v row b-NR)
            s1 = SC::O_UV1 + a * NR + (b - NR);
            s2 = SC::O_UV2 + a * NR + (b - NR);
        } else if (b < NR) {                       // mirrored
            s1 = SC::O_UV1 + b * NR + (a - NR);
            s2 = SC::O_UV2 + b * NR + (a - NR);
        } else {
            const int x = a - NR, y = b - NR;
            const int i = x < y ? x : y, j = x < y ? y : x;
            s1 = SC::O_V1 + SC::tri(i, j);
            s2 = SC::O_V2 + SC::tri(i, j);
        }
        float x1 = 0.0f, x2 = 0.0f;
#pragma unroll
        for (int j = 0; j < SC::PERS; ++j) {
            const float t1 = __shfl(acc[j], s1 / SC::PERS, LPP);
            const float t2 = __shfl(acc[j], s2 / SC::PERS, LPP);
            x1 = (s1 % SC::PERS == j) ? t1 : x1;
            x2 = (s2 % SC::PERS == j) ? t2 : x2;
        }
        if (do_scale) {
            x1 = (x1 < 0.0f) ? 0.0f : x1;          // NaN < 0 is false: NaN passes through (:127)
            x2 = (x2 < 0.0f) ? 0.0f : x2;
            x1 = logf(x1 + 1.0f);                  // log(x + 1), not log1p (:128)
            x2 = logf(x2 + 1.0f);
        }
        if (!idok) { x1 = __builtin_nanf(""); x2 = x1; }
        if (valid && in) {
            out1[idx] = x1;
            out2[idx] = x2;
        }
    }
}

template <int LPP, int VPL, int W, int L, bool FULL>
__global__ __launch_bounds__(BLOCK) void k_pair_gram_shared(tpnet_state S, const int64_t* __restrict__ u,
                                                            const int64_t* __restrict__ v1,
                                                            const int64_t* __restrict__ v2, int64_t n, double now,
                                                            double lambda, uint32_t flags, float* __restrict__ out1,
                                                            float* __restrict__ out2) {
    constexpr int GPB = BLOCK / LPP;
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

// ---------------------------------------------------------------------------------------------------------------
// update (models/TPNet.py:90-96, all layers at once) of ONE target node u:
//   new[i][u] = old[i][u] * g_u^i  +  sum_j  w_j * ( P[i-1][partner_j] * g_partner^(i-1) ),   i = 1..L
// Contributions are summed in sorted order = the reference's index order (src-side edges, then dst-side edges).
// All reads hit pre-launch copies; the result goes to the other copy.
//
// accumulate_range: one group of LPP lanes adds the contributions [jb, je) into acc.  Dependent memory round trips
// are what bounds a small batch, so: (1) the (partner, weight) pairs and the partners' meta records of up to LPP
// contributions are fetched lane-parallel (one per lane, two round trips in all) and handed round by shuffles;
// (2) U contributions' rows (U*L vectors per lane) are in flight at once.  All loops are wave-uniform (__any), the
// per-group bounds only predicate the loads, because the groups of one wave may own different targets.
// ---------------------------------------------------------------------------------------------------------------
template <int LPP, int VPL, int W, int L, int U, bool FULL>
__device__ __forceinline__ void accumulate_range(const tpnet_state& S, const int32_t* __restrict__ s_partner,
                                                 const float* __restrict__ s_coef, uint32_t jb, uint32_t je,
                                                 uint32_t bid, double t_last, double lambda, int gl, int c0,
                                                 float (&acc)[L][VPL * W]) {
    constexpr int F = VPL * W;
    const int d = S.d;
    const int nvec = d / W;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(S.meta);
    for (uint32_t r0 = jb; __any(r0 < je); r0 += LPP) {
        const uint32_t jm = r0 + (uint32_t)gl;
        const bool mine = jm < je;
        const int32_t my_pv = mine ? s_partner[jm] : 0;
        const float my_w = mine ? s_coef[jm] : 0.0f;
        const MetaView my_m = read_meta(meta, my_pv, bid, t_last, lambda);
        const uint32_t n_here = (r0 < je) ? ((je - r0 < (uint32_t)LPP) ? je - r0 : (uint32_t)LPP) : 0u;
        for (uint32_t k0 = 0; __any(k0 < n_here); k0 += U) {
            int32_t pv[U];
            float w[U], g[U];
            int cp[U];
            bool ok[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int sl = (int)k0 + k;
                ok[k] = (uint32_t)sl < n_here;
                pv[k] = __shfl(my_pv, sl, LPP);
                w[k] = __shfl(my_w, sl, LPP);
                g[k] = __shfl(my_m.g, sl, LPP);
                cp[k] = __shfl(my_m.copy, sl, LPP);
            }
            float r[U][L][F];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float* r0p = S.p0 + (int64_t)pv[k] * d;
                const float* rq = S.q + ((int64_t)cp[k] * S.N + pv[k]) * ((int64_t)L * d);
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    const float* rp = (i == 0) ? r0p : rq + (int64_t)(i - 1) * d;
#pragma unroll
                    for (int j = 0; j < VPL; ++j) {
                        const int vi = c0 + j * LPP + gl;
                        ldv_maybe<W, FULL>(rp, vi, ok[k] && vi < nvec, &r[k][i][j * W]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                if (ok[k]) {
                    float gi = 1.0f;
#pragma unroll
                    for (int i = 0; i < L; ++i) {
#pragma unroll
                        for (int x = 0; x < F; ++x) {
                            const float m = (r[k][i][x] * gi) * w[k];  // (P[i-1][partner], decayed) * time weight (:91-92)
                            acc[i][x] = acc[i][x] + m;                 // scatter-add, in index order (:93-96)
                        }
                        gi *= g[k];
                    }
                }
            }
        }
    }
}

// light item: one group owns the target (at most 2 + LPP contributions: the plan's heavy threshold is far below).
// The chain of dependent memory round trips is what a small batch pays for, so the loads are staged explicitly:
//   item record (carries the first two contributions)
//   -> { meta of target, partner 0, partner 1 }  and  { (partner, weight) of the remaining contributions, one per lane }
//   -> { rows of target, partner 0, partner 1 }   and  { meta of the remaining partners, one per lane }
//   -> rows of the remaining partners, U at a time
template <int LPP, int VPL, int W, int L, bool FULL, bool NT>
__device__ __forceinline__ void update_item(const tpnet_state& S, const int32_t* __restrict__ s_partner,
                                            const float* __restrict__ s_coef, Item I, bool valid, uint32_t bid,
                                            double t_last, double lambda, int gl) {
    constexpr int F = VPL * W;
    constexpr int U = (VPL * L <= 3) ? 6 : 4;   // one round of rows covers the whole tail of a light item (cnt <= 8)
    const int d = S.d;
    const int nvec = d / W;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    if (!valid) { I.target = 0; I.cnt = 0; I.p0 = 0; I.p1 = 0; I.j0 = 0; }
    const int32_t u = I.target;
    const bool has0 = I.cnt >= 1, has1 = I.cnt >= 2;
    const int32_t pv01[2] = {I.p0, has1 ? I.p1 : 0};
    const float pw01[2] = {I.w0, I.w1};
    const uint32_t n_tail = I.cnt > 2 ? I.cnt - 2 : 0;           // contributions j0+2 ..

    // stage 1: metas of the first three nodes + the tail's (partner, weight), lane-parallel
    const MetaView mu = read_meta(meta, u, bid, t_last, lambda);
    const MetaView mp[2] = {read_meta(meta, pv01[0], bid, t_last, lambda), read_meta(meta, pv01[1], bid, t_last, lambda)};
    const bool mine = (uint32_t)gl < n_tail;
    const int32_t my_pv = mine ? s_partner[I.j0 + 2 + gl] : 0;
    const float my_w = mine ? s_coef[I.j0 + 2 + gl] : 0.0f;

    const float* qold = S.q + ((int64_t)mu.copy * S.N + u) * ((int64_t)L * d);
    float* qnew = S.q + ((int64_t)(mu.copy ^ 1) * S.N + u) * ((int64_t)L * d);
    float gu[L];
    gu[0] = mu.g;
#pragma unroll
    for (int i = 1; i < L; ++i) gu[i] = gu[i - 1] * mu.g;

    for (int c0 = 0; c0 < (FULL ? 1 : nvec); c0 += LPP * VPL) {
        // stage 2: rows of the target and of the first two partners ...
        float acc[L][F];
        float r[2][L][F];
#pragma unroll
        for (int i = 0; i < L; ++i) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const int vi = c0 + j * LPP + gl;
                ldv_maybe<W, FULL>(qold + (int64_t)i * d, vi, valid && vi < nvec, &acc[i][j * W]);
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool okk = (k == 0) ? has0 : has1;
            const float* r0p = S.p0 + (int64_t)pv01[k] * d;
            const float* rq = S.q + ((int64_t)mp[k].copy * S.N + pv01[k]) * ((int64_t)L * d);
#pragma unroll
            for (int i = 0; i < L; ++i) {
                const float* rp = (i == 0) ? r0p : rq + (int64_t)(i - 1) * d;
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    const int vi = c0 + j * LPP + gl;
                    ldv_maybe<W, FULL>(rp, vi, okk && vi < nvec, &r[k][i][j * W]);
                }
            }
        }
        // ... and, in flight with them, the meta records of the remaining partners (one per lane)
        const MetaView my_m = read_meta(meta, my_pv, bid, t_last, lambda);

#pragma unroll
        for (int i = 0; i < L; ++i) {
#pragma unroll
            for (int k = 0; k < F; ++k) acc[i][k] *= gu[i];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool okk = (k == 0) ? has0 : has1;
            if (okk) {
                float gi = 1.0f;
#pragma unroll
                for (int i = 0; i < L; ++i) {
#pragma unroll
                    for (int x = 0; x < F; ++x) {
                        const float m = (r[k][i][x] * gi) * pw01[k];  // (P[i-1][partner], decayed) * time weight (:91-92)
                        acc[i][x] = acc[i][x] + m;                    // scatter-add, in index order (:93-96)
                    }
                    gi *= mp[k].g;
                }
            }
        }
        // stage 3: the remaining contributions, U rows-sets at a time, in order
        for (uint32_t k0 = 0; __any(k0 < n_tail); k0 += U) {
            int32_t pv[U];
            float w[U], g[U];
            int cp[U];
            bool ok[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int sl = (int)k0 + k;
                ok[k] = (uint32_t)sl < n_tail && sl < LPP;
                pv[k] = __shfl(my_pv, sl, LPP);
                w[k] = __shfl(my_w, sl, LPP);
                g[k] = __shfl(my_m.g, sl, LPP);
                cp[k] = __shfl(my_m.copy, sl, LPP);
            }
            float rt[U][L][F];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float* r0p = S.p0 + (int64_t)pv[k] * d;
                const float* rq = S.q + ((int64_t)cp[k] * S.N + pv[k]) * ((int64_t)L * d);
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    const float* rp = (i == 0) ? r0p : rq + (int64_t)(i - 1) * d;
#pragma unroll
                    for (int j = 0; j < VPL; ++j) {
                        const int vi = c0 + j * LPP + gl;
                        ldv_maybe<W, FULL>(rp, vi, ok[k] && vi < nvec, &rt[k][i][j * W]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                if (ok[k]) {
                    float gi = 1.0f;
#pragma unroll
                    for (int i = 0; i < L; ++i) {
#pragma unroll
                        for (int x = 0; x < F; ++x) {
                            const float m = (rt[k][i][x] * gi) * w[k];
                            acc[i][x] = acc[i][x] + m;
                        }
                        gi *= g[k];
                    }
                }
            }
        }
        // more than 2 + LPP contributions on a light item only happens with TPNET_FLAG_SEQUENTIAL (no heavy items)
        if (__any(n_tail > (uint32_t)LPP)) {
            const uint32_t jb = I.j0 + 2 + LPP, je = (n_tail > (uint32_t)LPP) ? I.j0 + I.cnt : jb;
            accumulate_range<LPP, VPL, W, L, 4, FULL>(S, s_partner, s_coef, jb, je, bid, t_last, lambda, gl, c0, acc);
        }
#pragma unroll
        for (int i = 0; i < L; ++i) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const int vi = c0 + j * LPP + gl;
                if (valid && (FULL || vi < nvec)) stv<W, NT>(qnew + (int64_t)i * d, vi, &acc[i][j * W]);
            }
        }
    }
    if (valid && gl == 0) {
        publish_meta(meta + u, mu.copy ^ 1, t_last, bid);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// light item on NARROW rows (LPP = 4 or 8 lanes per row: d <= 32, the column slices of a dim-sharded table).
// One row of such a table is a single 64- or 128-byte vector load, so a light item is nothing but a chain of dependent
// round trips -- and with LPP lanes per item only LPP contributions' (partner, weight, meta) could be fetched per
// round.  Here an item owns IL = 16 lanes = IG sub-groups of LPP lanes instead: the 16 lanes fetch 16 contributions'
// (partner, weight) and meta records at once, sub-group s accumulates contributions s, s+IG, s+2IG, ... and the IG
// partial sums are added through DPP row rotations at the end.  The association of the sum differs from the index
// order (like the heavy path's), so TPNET_FLAG_SEQUENTIAL launches use the 16-lane geometry instead (launch_step).
// ---------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_add(float a) {
    return a + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), CTRL, 0xF, 0xF, false));
}

template <int LPP, int L>
__device__ __forceinline__ void update_item_narrow(const tpnet_state& S, const int32_t* __restrict__ s_partner,
                                                   const float* __restrict__ s_coef, Item I, bool valid, uint32_t bid,
                                                   double t_last, double lambda, int il /* lane of the item, 0..15 */) {
    static_assert(LPP == 4 || LPP == 8, "narrow geometry");
    constexpr int IL = 16, IG = IL / LPP;
    constexpr int KS = 2;                           // steps (contributions per sub-group) whose rows are in flight at once
    const int d = S.d;
    const int nvec = d / 4;
    const int gl = il % LPP, sub = il / LPP;
    const bool vok = gl < nvec;                     // d < 4*LPP: the last lanes of a row have no vector
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    if (!valid) { I.target = 0; I.cnt = 0; I.p0 = 0; I.p1 = 0; I.j0 = 0; }
    const int32_t u = I.target;
    const uint32_t cnt = I.cnt;

    const MetaView mu = read_meta(meta, u, bid, t_last, lambda);
    const float* qold = S.q + ((int64_t)mu.copy * S.N + u) * ((int64_t)L * d);
    float* qnew = S.q + ((int64_t)(mu.copy ^ 1) * S.N + u) * ((int64_t)L * d);

    float acc[L][4];
    // the target's pre-batch rows (sub-group 0 carries them; the others start from zero)
#pragma unroll
    for (int i = 0; i < L; ++i) ldv_pred<4>(qold + (int64_t)i * d, gl, valid && sub == 0 && vok, acc[i]);
    {
        float g = 1.0f;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            g *= mu.g;
#pragma unroll
            for (int x = 0; x < 4; ++x) acc[i][x] *= g;
        }
    }

    for (uint32_t r0 = 0; __any(r0 < cnt); r0 += IL) {
        // (partner, weight) of contribution r0 + il, lane-parallel: the first two ride in the item record
        const uint32_t c = r0 + (uint32_t)il;
        const bool mine = c < cnt;
        int32_t my_pv = 0;
        float my_w = 0.0f;
        if (c == 0) { my_pv = I.p0; my_w = I.w0; }
        else if (c == 1) { my_pv = I.p1; my_w = I.w1; }
        else if (mine) { my_pv = s_partner[I.j0 + c]; my_w = s_coef[I.j0 + c]; }
        if (!mine) my_pv = 0;
        const MetaView my_m = read_meta(meta, my_pv, bid, t_last, lambda);
        const uint32_t n_here = (r0 < cnt) ? ((cnt - r0 < (uint32_t)IL) ? cnt - r0 : (uint32_t)IL) : 0u;
        for (uint32_t k0 = 0; __any(k0 * IG < n_here); k0 += KS) {
            float r[KS][L][4], w[KS], g[KS];
            bool ok[KS];
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const int sl = ((int)k0 + k) * IG + sub;           // contribution (of this round) of my sub-group
                ok[k] = (uint32_t)sl < n_here;
                const int32_t pv = __shfl(my_pv, sl, IL);
                const int cp = __shfl(my_m.copy, sl, IL);
                w[k] = __shfl(my_w, sl, IL);
                g[k] = __shfl(my_m.g, sl, IL);
                const float* r0p = S.p0 + (int64_t)pv * d;
                const float* rq = S.q + ((int64_t)cp * S.N + pv) * ((int64_t)L * d);
#pragma unroll
                for (int i = 0; i < L; ++i)
                    ldv_pred<4>((i == 0) ? r0p : rq + (int64_t)(i - 1) * d, gl, ok[k] && vok, r[k][i]);
            }
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                if (ok[k]) {
                    float gi = 1.0f;
#pragma unroll
                    for (int i = 0; i < L; ++i) {
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const float m = (r[k][i][x] * gi) * w[k];   // (P[i-1][partner], decayed) * time weight (:91-92)
                            acc[i][x] = acc[i][x] + m;
                        }
                        gi *= g[k];
                    }
                }
            }
        }
    }
    // add the sub-groups' partial sums: rotations inside the 16-lane row leave the total in every lane
#pragma unroll
    for (int i = 0; i < L; ++i) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            float v = dpp_add<0x128 /* row_ror:8 */>(acc[i][x]);
            if constexpr (IG == 4) v = dpp_add<0x124 /* row_ror:4 */>(v);
            acc[i][x] = v;
        }
    }
    if (valid && sub == 0 && vok) {
#pragma unroll
        for (int i = 0; i < L; ++i) stv<4>(qnew + (int64_t)i * d, gl, acc[i]);
    }
    if (valid && il == 0) {
        publish_meta(meta + u, mu.copy ^ 1, t_last, bid);
    }
}

// heavy item: a whole workgroup computes ONE LAYER of the target's new bundle.  The layers of the update are
// independent sums (new[i][u] needs only layer i-1 of the partners), so a hub is spread over L workgroups -- L CUs'
// worth of load bandwidth -- without any reduction across workgroups.  Inside the workgroup, group g sums the g-th
// slice of the contribution list; the slices' partial sums are added in slice order through LDS (fixed order:
// reproducible run to run, but the association differs from the strictly sequential sum -- TPNET_FLAG_SEQUENTIAL
// turns heavy items off).  `layer` = i-1 in 0..L-1; the layer-0 workgroup publishes the new version.
template <int LPP, int VPL, int W, int L, bool FULL, bool NT>
__device__ __forceinline__ void update_item_block(const tpnet_state& S, const int32_t* __restrict__ s_partner,
                                                  const float* __restrict__ s_coef, int32_t u, uint32_t j0,
                                                  uint32_t cnt, int layer, uint32_t bid, double t_last, double lambda,
                                                  float* __restrict__ part /* LDS [F][BLOCK] */) {
    constexpr int F = VPL * W;
    constexpr int G = BLOCK / LPP;
    constexpr int U = 8;
    const int d = S.d;
    const int nvec = d / W;
    const int gl = threadIdx.x % LPP;
    const int g = threadIdx.x / LPP;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    const uint32_t per = (cnt + G - 1) / G;
    const uint32_t lo = (uint32_t)g * per < cnt ? (uint32_t)g * per : cnt;
    const uint32_t hi = lo + per < cnt ? lo + per : cnt;
    const uint32_t jb = j0 + lo, je = j0 + hi;

    const MetaView mu = read_meta(meta, u, bid, t_last, lambda);
    const float* qold = S.q + ((int64_t)mu.copy * S.N + u) * ((int64_t)L * d) + (int64_t)layer * d;
    float* qnew = S.q + ((int64_t)(mu.copy ^ 1) * S.N + u) * ((int64_t)L * d) + (int64_t)layer * d;
    float gu = mu.g;                                    // target decay of layer (layer+1): g^(layer+1)
    for (int z = 0; z < layer; ++z) gu *= mu.g;

    for (int c0 = 0; c0 < (FULL ? 1 : nvec); c0 += LPP * VPL) {
        float acc[F], old[F];
#pragma unroll
        for (int k = 0; k < F; ++k) acc[k] = 0.0f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {                 // the target's pre-batch row: fetched up front by group 0
            const int vi = c0 + j * LPP + gl;
            ldv_pred<W>(qold, vi, g == 0 && vi < nvec, &old[j * W]);
        }
        for (uint32_t r0 = jb; __any(r0 < je); r0 += LPP) {
            // (partner, weight) and the partners' meta records lane-parallel, then U rows in flight per group
            const uint32_t jm = r0 + (uint32_t)gl;
            const bool mine = jm < je;
            const int32_t my_pv = mine ? s_partner[jm] : 0;
            const float my_w = mine ? s_coef[jm] : 0.0f;
            const MetaView my_m = read_meta(meta, my_pv, bid, t_last, lambda);
            float my_g = 1.0f;                          // partner decay of its layer `layer`: g^layer
            for (int z = 0; z < layer; ++z) my_g *= my_m.g;
            const uint32_t n_here = (r0 < je) ? ((je - r0 < (uint32_t)LPP) ? je - r0 : (uint32_t)LPP) : 0u;
            for (uint32_t k0 = 0; __any(k0 < n_here); k0 += U) {
                float r[U][F], w[U], gg[U];
                bool ok[U];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const int sl = (int)k0 + k;
                    ok[k] = (uint32_t)sl < n_here;
                    const int32_t pv = __shfl(my_pv, sl, LPP);
                    const int cp = __shfl(my_m.copy, sl, LPP);
                    w[k] = __shfl(my_w, sl, LPP);
                    gg[k] = __shfl(my_g, sl, LPP);
                    const float* rp = (layer == 0) ? S.p0 + (int64_t)pv * d
                                                   : S.q + ((int64_t)cp * S.N + pv) * ((int64_t)L * d) + (int64_t)(layer - 1) * d;
#pragma unroll
                    for (int j = 0; j < VPL; ++j) {
                        const int vi = c0 + j * LPP + gl;
                        ldv_maybe<W, FULL>(rp, vi, ok[k] && vi < nvec, &r[k][j * W]);
                    }
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    if (ok[k]) {
#pragma unroll
                        for (int x = 0; x < F; ++x) {
                            const float m = (r[k][x] * gg[k]) * w[k];
                            acc[x] = acc[x] + m;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < F; ++k) part[k * BLOCK + threadIdx.x] = acc[k];
        __syncthreads();
        if (g == 0) {
            float tot[F];
#pragma unroll
            for (int k = 0; k < F; ++k) tot[k] = old[k] * gu;
            for (int q = 0; q < G; ++q) {
#pragma unroll
                for (int k = 0; k < F; ++k) tot[k] = tot[k] + part[k * BLOCK + q * LPP + gl];
            }
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const int vi = c0 + j * LPP + gl;
                if (FULL || vi < nvec) stv<W, NT>(qnew, vi, &tot[j * W]);
            }
        }
        __syncthreads();
    }
    if (layer == 0 && threadIdx.x == 0) {
        publish_meta(meta + u, mu.copy ^ 1, t_last, bid);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// fused per-batch step: readout (src,dst) and (src,neg) on the pre-batch state + update, ONE launch.
// Blocks [0, HEAVY_BLOCKS) take the heavy update items (one workgroup per target and layer); the others walk a work index
// space: [0, RP) readout pairs (RP rounded up to whole waves so that a wave has one role), then the light items.
// ---------------------------------------------------------------------------------------------------------------
template <int LPP, int VPL, int W, int L, bool FULL, bool NT>
// The arguments every wave needs FIRST (the id arrays, e0/ne, the role flags) lead the signature as plain scalars: the
// first 16 SGPRs of kernel arguments are preloaded by the command processor at wave launch (-mllvm
// -amdgpu-kernarg-preload-count=16), so the id loads can be issued without waiting for a kernarg fetch.
__global__ __launch_bounds__(BLOCK, min_waves_per_simd(LPP, VPL, W)) void k_step(const int64_t* __restrict__ a_src, const int64_t* __restrict__ a_dst,
                                                const int64_t* __restrict__ a_neg, int64_t e0, int32_t ne_,
                                                uint32_t flags, uint32_t bid, int HEAVY_BLOCKS, double lambda,
                                                tpnet_state S, StreamArgs a, Plan p, int64_t b) {
    constexpr int GPB = BLOCK / LPP;
    constexpr int GPW = 64 / LPP;
    constexpr int NG = GramCfg<LPP, L>::NG;
    constexpr int STG = GramCfg<LPP, L>::STAGE;
    __shared__ float part[(VPL * W * BLOCK > STG) ? VPL * W * BLOCK : STG];   // heavy items' partial sums / readout staging
    unsigned long long* dbg = p.dbg;
    (void)dbg;
    STAMP(0);
    // e0/ne come with the launch and the item records are fetched speculatively (their slots always exist), so neither
    // the id loads nor the item loads wait for the batch descriptor (clocks, item counts).  The descriptor is read
    // AFTER those vector loads have been issued: it is a scalar load whose wait (lgkmcnt) would otherwise sit in front
    // of them and put one more memory round trip on every wave's critical path.
    const BatchDesc* __restrict__ Dp = p.desc + b;
    if ((int)blockIdx.x < HEAVY_BLOCKS) {
        if (flags & ROLE_UPDATE) {
            // heavy work unit = (item, layer): L consecutive workgroups share an item
            const Item* heavy = p.heavy + 2 * e0;
            const uint32_t cap = 2u * (uint32_t)ne_ * (uint32_t)L;
            for (uint32_t h = blockIdx.x; h < cap; h += HEAVY_BLOCKS) {
                const Item I = heavy[h / L];
                const uint32_t n_heavy = Dp->n_heavy;
                const double t_last = Dp->t_last;
                if (h / L >= n_heavy) break;
                if (a.own_mod > 1 && (I.target % a.own_mod) != a.own_rem) continue;
                update_item_block<LPP, VPL, W, L, FULL, NT>(S, p.s_partner, p.s_coef, I.target, I.j0, I.cnt, (int)(h % L),
                                                        bid, t_last, lambda, part);
                STAMP(7);
            }
        }
        return;
    }
    const int gl = threadIdx.x % LPP;
    const int g = threadIdx.x / LPP;
    const bool packed = (flags & TPNET_FLAG_PACKED) != 0;
    const bool do_scale = !(flags & TPNET_FLAG_NOT_SCALE) && !packed;
    const int ostride = packed ? GramCfg<LPP, L>::NT : NG;
    const int64_t ne = ne_;
    const int64_t npos = ((flags & ROLE_READOUT) && a.out_pos) ? ne : 0;
    const int64_t nneg = ((flags & ROLE_READOUT) && a.out_neg) ? ne : 0;
    // (a shared-src unit per edge -- gram_shared -- was measured here: it halves the readout waves but doubles each
    // wave's VALU chain: C1 +30 %, C2 +5 % slower, C3/C5 +1 % faster; the pair stays the unit)
    const int64_t npairs = npos + nneg;
    const int64_t RP = (npairs + GPW - 1) / GPW * GPW;
    const int64_t cap_items = (flags & ROLE_UPDATE) ? 2 * ne : 0;   // upper bound of the light items (slots exist)
    // narrow rows: an item takes 16 lanes = ISL group slots of the work index space (update_item_narrow)
    constexpr int ISL = (LPP < 16 && W == 4) ? 16 / LPP : 1;
    const int64_t total = RP + cap_items * ISL;
    const Item* items = p.light + 2 * e0;
    const int64_t nblk = (int64_t)gridDim.x - HEAVY_BLOCKS;

    for (int64_t base = ((int64_t)blockIdx.x - HEAVY_BLOCKS) * GPB; base < total; base += nblk * GPB) {
        const int64_t w = base + g;
        const int64_t wave0 = base + (g / GPW) * GPW;  // first work index of this wave: decides the wave's role
        if (wave0 < RP) {
            const bool valid = w < npairs;
            int64_t e = 0, u = 0, v = 0;
            float* out = nullptr;
            if (valid) {
                // (src,dst) and (src,neg) of one edge sit in ADJACENT lane groups (work index 2e, 2e+1): the two groups
                // load the src node's meta record and rows with the same instructions and the same addresses, which the
                // load unit coalesces -- one fetch from the memory side instead of two in different workgroups/XCDs
#ifndef TPNET_NO_INTERLEAVE
                const bool both = npos != 0 && nneg != 0;
#else
                const bool both = false;
#endif
                const int64_t idx = both ? (w >> 1) : (w < npos ? w : w - npos);
                const bool isneg = both ? (w & 1) != 0 : (w >= npos);
                e = e0 + idx;
                v = isneg ? a_neg[e] : a_dst[e];
                out = (isneg ? a.out_neg : a.out_pos) + e * ostride;
                u = a_src[e];
            }
            const double now = Dp->now;
            // row sharding: a pair is read out by the owner of its src node (the other ranks leave the output row alone)
            const bool mine = valid && (a.own_mod <= 1 || (int32_t)((uint64_t)u % (uint32_t)a.own_mod) == a.own_rem);
            if (!__any(mine)) continue;
            if (packed)
                gram_pair<LPP, VPL, W, L, FULL, true>(S, u, v, mine, bid, now, lambda, do_scale, out, gl, p.dbg, part);
            else
                gram_pair<LPP, VPL, W, L, FULL, false>(S, u, v, mine, bid, now, lambda, do_scale, out, gl, p.dbg, part);
            STAMP(5);
        } else {
            const int64_t it = (w - RP) / ISL;
            Item I = items[it < cap_items ? it : 0];
            const int64_t n_light = (int64_t)Dp->n_light;
            const double t_last = Dp->t_last;
            if ((wave0 - RP) / ISL >= n_light) break;   // wave-uniform: no item of this wave exists (and none later)
            const bool valid = it < n_light &&
                               (a.own_mod <= 1 || (I.target % a.own_mod) == a.own_rem);   // targets belong to their owner
            if constexpr (ISL > 1)
                update_item_narrow<LPP, L>(S, p.s_partner, p.s_coef, I, valid, bid, t_last, lambda, (int)(threadIdx.x % 16));
            else
                update_item<LPP, VPL, W, L, FULL, NT>(S, p.s_partner, p.s_coef, I, valid, bid, t_last, lambda, gl);
            STAMP(6);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dense passes
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_state_init(tpnet_state S, double t0) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t nq = 2 * S.N * (int64_t)S.L * S.d;
    for (int64_t i = tid; i < nq; i += stride) S.q[i] = 0.0f;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    for (int64_t n = tid; n < S.N; n += stride) {
        NodeMeta m;
        m.ver = 0; m.pad0 = 0; m.tref[0] = t0; m.tref[1] = t0; m.pad1 = 0;
        meta[n] = m;
    }
    if (tid < 4) S.err[tid] = 0;
}

struct LayerPtrs {
    float* p[TPNET_MAX_LAYERS];
};

// layers (row-major [N][d] each) -> copy 0 bundles; meta = {0, now}
__global__ void k_import(tpnet_state S, LayerPtrs lp, double now) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t d = S.d, L = S.L;
    const int64_t tot = S.N * L * d;
    for (int64_t x = tid; x < tot; x += stride) {
        const int64_t n = x / (L * d);
        const int64_t r = x - n * (L * d);
        const int64_t i = r / d, k = r - i * d;
        S.q[x] = lp.p[i][n * d + k];
    }
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    for (int64_t n = tid; n < S.N; n += stride) {
        NodeMeta m;
        m.ver = 0; m.pad0 = 0; m.tref[0] = now; m.tref[1] = now; m.pad1 = 0;
        meta[n] = m;
    }
}

// current bundles with the pending decay applied -> layers
__global__ void k_export(tpnet_state S, LayerPtrs lp, double now, double lambda) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t d = S.d, L = S.L;
    const int64_t tot = S.N * L * d;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(S.meta);
    for (int64_t x = tid; x < tot; x += stride) {
        const int64_t n = x / (L * d);
        const int64_t r = x - n * (L * d);
        const int64_t i = r / d, k = r - i * d;
        const MetaView m = read_meta(meta, n, READER_BID, now, lambda);
        float g = m.g;
        for (int64_t z = 0; z < i; ++z) g *= m.g;
        lp.p[i][n * d + k] = S.q[((int64_t)m.copy * S.N) * (L * d) + x] * g;
    }
}

struct DecayFactors {
    float f[TPNET_MAX_LAYERS];
};

// eager dense decay (models/TPNet.py:83-85): current copy of every row *= f[i]; tref = t_new.
// One block-stride loop over nodes; LPP-agnostic (one thread per element of the bundle).
__global__ void k_decay(tpnet_state S, DecayFactors df, double t_new) {
    const int64_t d = S.d, L = S.L;
    const int64_t per = L * d;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    for (int64_t n = blockIdx.x; n < S.N; n += gridDim.x) {
        const uint32_t ver = meta[n].ver;
        const int c = ver & 1;
        float* qb = S.q + ((int64_t)c * S.N + n) * per;
        for (int64_t r = threadIdx.x; r < per; r += blockDim.x) {
            const int64_t i = r / d;
            qb[r] = qb[r] * df.f[i];
        }
        __syncthreads();
        if (threadIdx.x == 0) meta[n].tref[c] = t_new;
    }
}

__global__ void k_decay_desc(tpnet_state S, const BatchDesc* __restrict__ desc, int64_t b) {
    const int64_t d = S.d, L = S.L;
    const int64_t per = L * d;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    const float* __restrict__ decay = desc[b].decay;  // indexed from memory (a private copy would be demoted to LDS)
    const double t_last = desc[b].t_last;
    for (int64_t n = blockIdx.x; n < S.N; n += gridDim.x) {
        const uint32_t ver = meta[n].ver;
        const int c = ver & 1;
        float* qb = S.q + ((int64_t)c * S.N + n) * per;
        for (int64_t r = threadIdx.x; r < per; r += blockDim.x) {
            const int64_t i = r / d;
            qb[r] = qb[r] * decay[i];
        }
        __syncthreads();
        if (threadIdx.x == 0) meta[n].tref[c] = t_last;
    }
}

// get_random_projections (models/TPNet.py:101-110): out[(i*n + k)*d + :] = P[i][ids[k]]
__global__ void k_gather_rows(tpnet_state S, const int64_t* __restrict__ ids, int64_t n, double now, double lambda,
                              float* __restrict__ out) {
    const int64_t d = S.d, L = S.L;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(S.meta);
    for (int64_t k = blockIdx.x; k < n; k += gridDim.x) {
        int64_t id = ids[k];
        const bool ok = (uint64_t)id < (uint64_t)S.N;
        if (!ok) {
            if (threadIdx.x == 0) atomicAdd(S.err, 1u);
            id = 0;
        }
        const MetaView m = read_meta(meta, id, READER_BID, now, lambda);
        const float* qb = S.q + ((int64_t)m.copy * S.N + id) * (L * d);
        for (int64_t r = threadIdx.x; r < (L + 1) * d; r += blockDim.x) {
            const int64_t i = r / d, c = r - i * d;
            float x;
            if (i == 0) {
                x = S.p0[id * d + c];
            } else {
                float g = m.g;
                for (int64_t z = 1; z < i; ++z) g *= m.g;
                x = qb[(i - 1) * d + c] * g;
            }
            out[(i * n + k) * d + c] = ok ? x : __builtin_nanf("");
        }
    }
}

// single elements of the layers: out[k][i] = P[i][rows[k]][cols[k]] at `now`, i = 0..L (the walk-matrix readout of
// models/MemoryModel.py:396-405, `matrix[src, dst]`, when the table is square: tpnet_amd/matrix_memory.py)
__global__ void k_gather_elems(tpnet_state S, const int64_t* __restrict__ rows, const int64_t* __restrict__ cols,
                               int64_t n, double now, double lambda, float* __restrict__ out) {
    const int64_t d = S.d, L = S.L;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(S.meta);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        int64_t r = rows[k], c = cols[k];
        const bool ok = (uint64_t)r < (uint64_t)S.N && (uint64_t)c < (uint64_t)d;
        if (!ok) {
            atomicAdd(S.err, 1u);
            r = 0; c = 0;
        }
        const MetaView m = read_meta(meta, r, READER_BID, now, lambda);
        const float* qb = S.q + ((int64_t)m.copy * S.N + r) * (L * d);
        float g = 1.0f;
        out[k * (L + 1)] = ok ? S.p0[r * d + c] : __builtin_nanf("");
        for (int64_t i = 1; i <= L; ++i) {
            g *= m.g;
            out[k * (L + 1) + i] = ok ? qb[(i - 1) * d + c] * g : __builtin_nanf("");
        }
    }
}

int launch_gather_elems(const tpnet_state& st, const int64_t* rows, const int64_t* cols, int64_t n, double now,
                        double lambda, float* out, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    hipLaunchKernelGGL(k_gather_elems, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, st, rows, cols, n, now, lambda, out);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

// row exchange of the sharded state: out[k][i][:] = P[i+1][ids[k]] at `now` (decay applied), i = 0..L-1
__global__ void k_pack_rows(tpnet_state S, const int64_t* __restrict__ ids, int64_t n, double now, double lambda,
                            float* __restrict__ out) {
    const int64_t d = S.d, L = S.L;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(S.meta);
    for (int64_t k = blockIdx.x; k < n; k += gridDim.x) {
        int64_t id = ids[k];
        if ((uint64_t)id >= (uint64_t)S.N) {
            if (threadIdx.x == 0) atomicAdd(S.err, 1u);
            id = 0;
        }
        const MetaView m = read_meta(meta, id, READER_BID, now, lambda);
        const float* qb = S.q + ((int64_t)m.copy * S.N + id) * (L * d);
        for (int64_t r = threadIdx.x; r < L * d; r += blockDim.x) {
            const int64_t i = r / d;
            float g = m.g;
            for (int64_t z = 0; z < i; ++z) g *= m.g;
            out[k * L * d + r] = qb[r] * g;
        }
    }
}

// the inverse: the current copy of row ids[k] <- in[k], expressed at `now` (rows owned by another rank: never the
// target of a local update, so no launch of this rank rewrites them concurrently)
__global__ void k_unpack_rows(tpnet_state S, const int64_t* __restrict__ ids, int64_t n, double now,
                              const float* __restrict__ in) {
    const int64_t d = S.d, L = S.L;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    for (int64_t k = blockIdx.x; k < n; k += gridDim.x) {
        const int64_t id = ids[k];
        if ((uint64_t)id >= (uint64_t)S.N) {
            if (threadIdx.x == 0) atomicAdd(S.err, 1u);
            continue;
        }
        const int c = (int)(meta[id].ver & 1u);
        float* qb = S.q + ((int64_t)c * S.N + id) * (L * d);
        for (int64_t r = threadIdx.x; r < L * d; r += blockDim.x) qb[r] = in[k * L * d + r];
        if (threadIdx.x == 0) meta[id].tref[c] = now;
    }
}

// all peers in one launch: ids = the batch's touched nodes ordered by (owner, node); recv = [G][maxc][L*d] as the
// all-gather delivered it; offs[r] = start of owner r's run inside ids.  Rows owned by `me` are skipped.
__global__ void k_unpack_gathered(tpnet_state S, const int64_t* __restrict__ ids, int64_t n, double now,
                                  const float* __restrict__ recv, int64_t maxc, const int64_t* __restrict__ offs, int G,
                                  int me) {
    const int64_t d = S.d, L = S.L;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    for (int64_t k = blockIdx.x; k < n; k += gridDim.x) {
        const int64_t id = ids[k];
        if ((uint64_t)id >= (uint64_t)S.N) {
            if (threadIdx.x == 0) atomicAdd(S.err, 1u);
            continue;
        }
        const int r = (int)(id % G);
        if (r == me) continue;
        const int64_t slot = k - offs[r];
        const float* in = recv + ((int64_t)r * maxc + slot) * (L * d);
        const int c = (int)(meta[id].ver & 1u);
        float* qb = S.q + ((int64_t)c * S.N + id) * (L * d);
        for (int64_t x = threadIdx.x; x < L * d; x += blockDim.x) qb[x] = in[x];
        if (threadIdx.x == 0) meta[id].tref[c] = now;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------------------
int launch_unpack_gathered(const tpnet_state& st, const int64_t* ids, int64_t n, double now, const float* recv,
                           int64_t maxc, const int64_t* offs, int G, int me, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    hipLaunchKernelGGL(k_unpack_gathered, dim3(grid_for(n, 1, 8192)), dim3(256), 0, s, st, ids, n, now, recv, maxc, offs,
                       G, me);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_pack_rows(const tpnet_state& st, const int64_t* ids, int64_t n, double now, double lambda, float* out,
                     hipStream_t s) {
    if (n == 0) return TPNET_OK;
    hipLaunchKernelGGL(k_pack_rows, dim3(grid_for(n, 1, 8192)), dim3(256), 0, s, st, ids, n, now, lambda, out);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_unpack_rows(const tpnet_state& st, const int64_t* ids, int64_t n, double now, const float* in, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    hipLaunchKernelGGL(k_unpack_rows, dim3(grid_for(n, 1, 8192)), dim3(256), 0, s, st, ids, n, now, in);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_state_init(const tpnet_state& st, double t0, hipStream_t s) {
    const int64_t nq = 2 * st.N * (int64_t)st.L * st.d;
    hipLaunchKernelGGL(k_state_init, dim3(grid_for(nq, 256 * 4, 4096)), dim3(256), 0, s, st, t0);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_import(const tpnet_state& st, const float* const* layers, double now, hipStream_t s) {
    LayerPtrs lp{};
    for (int i = 0; i < st.L; ++i) lp.p[i] = const_cast<float*>(layers[i]);
    hipLaunchKernelGGL(k_import, dim3(grid_for(st.N * (int64_t)st.L * st.d, 256 * 4, 4096)), dim3(256), 0, s, st, lp,
                       now);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_export(const tpnet_state& st, float* const* layers, double now, double lambda, hipStream_t s) {
    LayerPtrs lp{};
    for (int i = 0; i < st.L; ++i) lp.p[i] = layers[i];
    hipLaunchKernelGGL(k_export, dim3(grid_for(st.N * (int64_t)st.L * st.d, 256 * 4, 4096)), dim3(256), 0, s, st, lp,
                       now, lambda);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_decay(const tpnet_state& st, const float* factors_host, double t_new, hipStream_t s) {
    DecayFactors df{};
    for (int i = 0; i < st.L; ++i) df.f[i] = factors_host[i];
    hipLaunchKernelGGL(k_decay, dim3(grid_for(st.N, 1, 8192)), dim3(256), 0, s, st, df, t_new);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_decay_desc(const tpnet_state& st, const Plan& p, int64_t b, hipStream_t s) {
    hipLaunchKernelGGL(k_decay_desc, dim3(grid_for(st.N, 1, 8192)), dim3(256), 0, s, st, p.desc, b);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_gather_rows(const tpnet_state& st, const int64_t* ids, int64_t n, double now, double lambda, float* out,
                       hipStream_t s) {
    if (n == 0) return TPNET_OK;
    hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(n, 1, 8192)), dim3(256), 0, s, st, ids, n, now, lambda, out);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

// element-wise tail of the readout on a buffer of raw Gram entries (same two operations, in the same order, as the
// fused store of gram_pair)
__global__ void k_gram_finish(float* __restrict__ x, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = x[i];
        v = (v < 0.0f) ? 0.0f : v;
        x[i] = logf(v + 1.0f);
    }
}

// packed rows (TPNET_FLAG_PACKED: NN(NN+1)/2 raw entries a <= b) -> full [NN*NN] feature rows, with the element-wise tail
__global__ void k_gram_unpack(const float* __restrict__ packed, int64_t n, int NN, int do_scale, float* __restrict__ out) {
    const int NG = NN * NN, NT = NN * (NN + 1) / 2;
    const int64_t total = n * NG;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t p = i / NG;
        const int idx = (int)(i - p * NG);
        int a = idx / NN, b = idx - a * NN;
        if (a > b) { const int z = a; a = b; b = z; }
        float v = packed[p * NT + a * NN - (a * (a - 1)) / 2 + (b - a)];
        if (do_scale) {
            v = (v < 0.0f) ? 0.0f : v;
            v = logf(v + 1.0f);
        }
        out[i] = v;
    }
}

int launch_gram_unpack(const float* packed, int64_t n, int L, uint32_t flags, float* out, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    const int NN = 2 * L + 2;
    hipLaunchKernelGGL(k_gram_unpack, dim3(grid_for(n * NN * NN, 256, 256 * 16)), dim3(256), 0, s, packed, n, NN,
                       (flags & TPNET_FLAG_NOT_SCALE) ? 0 : 1, out);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_gram_finish(float* x, int64_t n, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    hipLaunchKernelGGL(k_gram_finish, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, x, n);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
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

int launch_pair_gram(const tpnet_state& st, const int64_t* u, const int64_t* v, int64_t n, double now, double lambda,
                     uint32_t flags, float* out, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    if (reinterpret_cast<uintptr_t>(out) & 15) return TPNET_ERR_BAD_ARG;   // feature rows leave as 16-byte vectors
    TPNET_DISPATCH_G(pick_geom(st.d, n >= 16384), ({
        const int grid = grid_for(n, BLOCK / LPP, 256 * 16);
        hipLaunchKernelGGL((k_pair_gram<LPP, VPL, W, L, FULL>), dim3(grid), dim3(BLOCK), 0, s, st, u, v, n, now, lambda,
                           flags, out);
    }));
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

// resident workgroups of a kernel on this device: occupancy (per CU, from the runtime) x CU count
template <typename K>
static int resident_blocks(K kernel) {
    int dev = 0, cus = 256, per_cu = 1;
    if (hipGetDevice(&dev) == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    return cus * per_cu;
}

int launch_pair_gram_shared(const tpnet_state& st, const int64_t* u, const int64_t* v1, const int64_t* v2, int64_t n,
                            double now, double lambda, uint32_t flags, float* out1, float* out2, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    TPNET_DISPATCH(({
        const int grid = grid_for(n, BLOCK / LPP, 256 * 16);
        hipLaunchKernelGGL((k_pair_gram_shared<LPP, VPL, W, L, FULL>), dim3(grid), dim3(BLOCK), 0, s, st, u, v1, v2, n,
                           now, lambda, flags, out1, out2);
    }));
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_step(const tpnet_state& st, const StreamArgs& a, const Plan& p, int64_t b, int64_t batch, int32_t ne,
                double lambda, uint32_t launch_id, uint32_t flags, hipStream_t s) {
    // narrow rows (d <= 32): 4 / 8 lanes per row with 16-lane items, whose sums are not in index order -- a
    // TPNET_FLAG_SEQUENTIAL launch keeps the 16-lane geometry
    if ((reinterpret_cast<uintptr_t>(a.out_pos) | reinterpret_cast<uintptr_t>(a.out_neg)) & 15) return TPNET_ERR_BAD_ARG;
    Geom geom = pick_geom(st.d);
    if ((flags & TPNET_FLAG_SEQUENTIAL) && geom.w == 4 && geom.lpp < 16) geom = Geom{16, 1, 4};
    TPNET_DISPATCH_G(geom, ({
        constexpr int GPB = BLOCK / LPP;
        constexpr int ISL = (LPP < 16 && W == 4) ? 16 / LPP : 1;
        static const int resident = resident_blocks(k_step<LPP, VPL, W, L, FULL, false>);
        static const int hb_env = getenv("TPNET_DEV_HEAVY_BLOCKS") ? atoi(getenv("TPNET_DEV_HEAVY_BLOCKS")) : 0;
        const int HEAVY_BLOCKS = hb_env > 0 ? hb_env : (ne <= 1024 ? HEAVY_BLOCKS_SMALL : ne <= 2048 ? HEAVY_BLOCKS_MEDIUM : HEAVY_BLOCKS_LARGE);
        // One pass when it fits: 2*ne readout pairs + up to 2*ne distinct targets.  A small batch is bound by its
        // chain of dependent memory round trips, so every workgroup should be resident at once (a workgroup that
        // starts after another one has finished doubles the chain) -- the item slots give way first (a batch
        // rarely has more than ne distinct light targets; the grid-stride loop covers the rest).
        const int pair_blocks = grid_for(2 * (int64_t)ne, GPB, 1 << 20);
        int item_blocks = grid_for(2 * (int64_t)ne * ISL, GPB, 1 << 20);
        const int room = resident - HEAVY_BLOCKS - pair_blocks;
        const int item_min = grid_for((((int64_t)ne * 3) / 4 + 1) * ISL, GPB, 1 << 20);
        if (item_blocks > room) item_blocks = room > item_min ? room : item_min;
        int grid = HEAVY_BLOCKS + pair_blocks + item_blocks;
        if (grid > HEAVY_BLOCKS + 256 * 8) grid = HEAVY_BLOCKS + 256 * 8;
        // streamed state stores for mid-size batches on the two geometries that serve them (see stv)
        static const char* nt_env = getenv("TPNET_DEV_NT_STATE");          // developer override: "0" / "1"
        constexpr bool NT_GEOM = W == 4 && VPL == 1 && (LPP == 16 || LPP == 32);
        const bool nt_state = NT_GEOM && (nt_env ? nt_env[0] == '1' : (ne >= 512 && ne <= 2048));
        if (nt_state)
            hipLaunchKernelGGL((k_step<LPP, VPL, W, L, FULL, NT_GEOM>), dim3(grid), dim3(BLOCK), 0, s, a.src, a.dst,
                               a.neg, b * batch, ne, flags, launch_id, HEAVY_BLOCKS, lambda, st, a, p, b);
        else
            hipLaunchKernelGGL((k_step<LPP, VPL, W, L, FULL, false>), dim3(grid), dim3(BLOCK), 0, s, a.src, a.dst,
                               a.neg, b * batch, ne, flags, launch_id, HEAVY_BLOCKS, lambda, st, a, p, b);
    }));
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

}  // namespace tpnet
