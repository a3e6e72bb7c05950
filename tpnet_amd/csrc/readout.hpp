// Pairwise readout (models/TPNet.py:112-128 before self.mlp) as device functions: one pair per group of LPP lanes
// (gram_pair) and two pairs that share their first node (gram_shared).  Used by readout.hip and by the fused step.
#pragma once
#include "device_common.hpp"

namespace tpnet {

// ---------------------------------------------------------------------------------------------------------------
// pairwise readout of ONE pair by one group of LPP lanes (models/TPNet.py:119-128)
// ---------------------------------------------------------------------------------------------------------------
// PACKED (TPNET_FLAG_PACKED): only the NT distinct entries a <= b are written, raw, row-major upper triangle.
// `stage`: LDS, GramCfg::STAGE floats per workgroup, used by the narrow geometries (LPP < 16) only: there a lane ends up
// with 8 or 16 consecutive outputs, so a direct store touches one 64-byte line per lane and instruction; the values go
// through LDS instead and leave as whole lines (measured at d=16, B=8000: the store phase of a readout wave 5.6 -> 0.7 us).
template <int LPP, int VPL, int W, int L, bool FULL, bool PACKED = false, bool FUSE = false, bool LDSRED = false>
__device__ __forceinline__ void gram_pair(const tpnet_state& S, int64_t u, int64_t v, bool valid, uint32_t bid,
                                          double now, double lambda, bool do_scale, float* __restrict__ out, int gl,
                                          unsigned long long* dbg = nullptr, float* __restrict__ stage = nullptr,
                                          uint32_t fuse = 0, float fuse_w = 0.0f, double t_last = 0.0) {
    // FUSE / fuse (bit 0: u, bit 1: v; only the fused step sets it, and only for an edge's (src,dst) pair): this group also
    // writes the new bundle of that endpoint -- it is the target's ONLY contribution in the batch (Plan::fuse_*), and
    // both operands, old[i][target] and P[i-1][partner], are among the rows loaded for the Gram:
    //   new[i][target] = old[i][target] * g_t^i + (P[i-1][partner] * g_p^(i-1)) * w,   g = decay to the batch's LAST time
    // (the same operations in the same order as update_item, so the result does not depend on which path ran)
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
    float gl_last[2] = {1.0f, 1.0f};     // decay of layer 1 to t_last (fused update only)
    int cur[2] = {0, 0};
    const int64_t ids[2] = {u, v};
    if (!FUSE || !idok) fuse = 0;
    // (the fused step's variant -- batches of thousands of edges, bound by memory: its layer-0 rows do not depend on the nodes' records,
    // so their loads go out WITH the records' instead of behind them: a quarter of a pair's bytes one round trip earlier)
    constexpr bool EARLY0 = FUSE && FULL;
    float f[NN][F];
    if constexpr (EARLY0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) ldv_maybe<W, true>(S.p0 + ids[s] * (int64_t)d, j * LPP + gl, true, &f[s * NR][j * W]);
        }
    }
    {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint4* mp = reinterpret_cast<const uint4*>(meta + ids[s]);
            const uint4 ma = mp[0], mb = mp[1];
            const MetaView m = meta_view(ma, mb, bid, now, lambda);
            if constexpr (FUSE) {
                if (fuse) gl_last[s] = meta_view(ma, mb, bid, t_last, lambda).g;
            }
            cur[s] = m.copy;
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
#pragma unroll
        for (int a = 0; a < NN; ++a) {
            if (EARLY0 && a % NR == 0) continue;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const int vi = c0 + j * LPP + gl;
                ldv_maybe<W, FULL>(rowp[a], vi, vi < nvec, &f[a][j * W]);
            }
        }
        if constexpr (FUSE) if (fuse) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (fuse & (1u << s)) {
                    float* qnew = S.q + ((int64_t)(cur[s] ^ 1) * S.N + ids[s]) * ((int64_t)L * d);
                    float gt = 1.0f, gp = 1.0f;
#pragma unroll
                    for (int i = 1; i <= L; ++i) {
                        gt *= gl_last[s];                            // g_t^i
                        float nw[F];
#pragma unroll
                        for (int k = 0; k < F; ++k) {
                            const float m = (f[(1 - s) * NR + i - 1][k] * gp) * fuse_w;
                            nw[k] = f[s * NR + i][k] * gt + m;
                        }
                        gp *= gl_last[1 - s];                        // g_p^(i-1) for the next layer
#pragma unroll
                        for (int j = 0; j < VPL; ++j) {
                            const int vi = c0 + j * LPP + gl;
                            if (FULL || vi < nvec) stv<W>(qnew + (int64_t)(i - 1) * d, vi, &nw[j * W]);
                        }
                    }
                }
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
    if constexpr (FUSE) if (fuse && gl == 0) {
        NodeMeta* wm = reinterpret_cast<NodeMeta*>(S.meta);
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (fuse & (1u << s)) publish_meta(wm + ids[s], cur[s] ^ 1, t_last, bid);
    }
    if constexpr (LDSRED) {
        // reduction through LDS (GramCfg::lds_reduce): park the NT distinct partials, sum row v, mirror into the output tile
        STAMP(3);
        constexpr int RS = C::RSTRIDE;
        float* red = stage + (threadIdx.x / LPP) * C::RED;
        float* so = red + C::NT * RS;
        {
            int tix = 0;
#pragma unroll
            for (int a = 0; a < NN; ++a) {
#pragma unroll
                for (int b = a; b < NN; ++b) {
                    red[tix * RS + gl] = acc[a * NN + b];
                    ++tix;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();            // a group's lanes are in one wave: LDS executes in issue order
#pragma unroll
        for (int it = 0; it * LPP < C::NT; ++it) {
            const int v = it * LPP + gl;
            if (v < C::NT) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f* row = reinterpret_cast<const v4f*>(red + v * RS);
                float sum = 0.0f;
#pragma unroll
                for (int l = 0; l < LPP / 4; ++l) {
                    const v4f q = row[l];
                    sum = (((sum + q.x) + q.y) + q.z) + q.w;
                }
                if constexpr (PACKED) {
                    so[v] = sum;
                } else {
                    int a = 0, off = 0;                          // v -> (a, b): row a of the upper triangle starts at off
#pragma unroll
                    for (int r = 1; r < NN; ++r) {
                        const int o = r * NN - (r * (r - 1)) / 2;
                        if (v >= o) { a = r; off = o; }
                    }
                    const int b = a + (v - off);
                    so[a * NN + b] = sum;
                    so[b * NN + a] = sum;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        STAMP(4);
        constexpr int NOUT = PACKED ? C::NT : C::NG;
#pragma unroll
        for (int j = 0; j * LPP < NOUT; ++j) {                    // lane gl takes outputs gl, gl+LPP, ...: whole lines
            const int c = j * LPP + gl;
            if (valid && c < NOUT) {
                float x = so[c];
                if (do_scale && !PACKED) {
                    x = (x < 0.0f) ? 0.0f : x;  // NaN < 0 is false: NaN passes through, as in the reference (:127)
                    x = logf(x + 1.0f);          // log(x + 1), not log1p (:128)
                }
                if (!idok) x = __builtin_nanf("");
                __builtin_nontemporal_store(x, out + c);
            }
        }
        __builtin_amdgcn_wave_barrier();            // the tiles are reused by the next pair of this group
        return;
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


// ---------------------------------------------------------------------------------------------------------------
// pairwise readout of ONE pair whose 2(L+1) rows have already been located (windowed stream path, wstep.hip: a row
// version is either a table bundle or a slot of the window's version log): the Gram, its reduction over the group and
// the store, exactly as in gram_pair (same sums in the same order).
// ---------------------------------------------------------------------------------------------------------------
template <int LPP, int VPL, int W, int L, bool FULL, bool PACKED, bool LDSRED>
__device__ __forceinline__ void gram_rows(const float* const (&rowp)[2 * (L + 1)], const float (&rs)[2 * (L + 1)], int d,
                                          bool valid, bool idok, bool do_scale, float* __restrict__ out, int gl,
                                          float* __restrict__ stage) {
    using C = GramCfg<LPP, L>;
    constexpr int NR = C::NR, NN = C::NN, F = VPL * W;
    const int nvec = d / W;
    float acc[C::MP];
#pragma unroll
    for (int i = 0; i < C::MP; ++i) acc[i] = 0.0f;
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
    if constexpr (LDSRED) {
        constexpr int RS = C::RSTRIDE;
        float* red = stage + (threadIdx.x / LPP) * C::RED;
        float* so = red + C::NT * RS;
        {
            int tix = 0;
#pragma unroll
            for (int a = 0; a < NN; ++a) {
#pragma unroll
                for (int b = a; b < NN; ++b) {
                    red[tix * RS + gl] = acc[a * NN + b];
                    ++tix;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it * LPP < C::NT; ++it) {
            const int v = it * LPP + gl;
            if (v < C::NT) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f* row = reinterpret_cast<const v4f*>(red + v * RS);
                float sum = 0.0f;
#pragma unroll
                for (int l = 0; l < LPP / 4; ++l) {
                    const v4f q = row[l];
                    sum = (((sum + q.x) + q.y) + q.z) + q.w;
                }
                if constexpr (PACKED) {
                    so[v] = sum;
                } else {
                    int a = 0, off = 0;
#pragma unroll
                    for (int r = 1; r < NN; ++r) {
                        const int o = r * NN - (r * (r - 1)) / 2;
                        if (v >= o) { a = r; off = o; }
                    }
                    const int b = a + (v - off);
                    so[a * NN + b] = sum;
                    so[b * NN + a] = sum;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        constexpr int NOUT = PACKED ? C::NT : C::NG;
#pragma unroll
        for (int j = 0; j * LPP < NOUT; ++j) {
            const int c = j * LPP + gl;
            if (valid && c < NOUT) {
                float x = so[c];
                if (do_scale && !PACKED) {
                    x = (x < 0.0f) ? 0.0f : x;  // NaN < 0 is false: NaN passes through, as in the reference (TPNet.py:127)
                    x = logf(x + 1.0f);          // log(x + 1), not log1p (TPNet.py:128)
                }
                if (!idok) x = __builtin_nanf("");
                __builtin_nontemporal_store(x, out + c);
            }
        }
        __builtin_amdgcn_wave_barrier();
        return;
    }
#pragma unroll
    for (int a = 1; a < NN; ++a) {
#pragma unroll
        for (int b = 0; b < a; ++b) acc[a * NN + b] = acc[b * NN + a];
    }
    Halve<C::MP, LPP / 2>::run(acc, gl);
    if constexpr (LPP < 16) {
        constexpr int NOUT = PACKED ? C::NT : C::NG;
        float* sg = stage + (threadIdx.x / LPP) * C::NG;
#pragma unroll
        for (int k = 0; k < C::PER; ++k) {
            const int idx = gl * C::PER + k;
            if (idx < C::NG) {
                float x = acc[k];
                if (do_scale && !PACKED) {
                    x = (x < 0.0f) ? 0.0f : x;
                    x = logf(x + 1.0f);
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
        __builtin_amdgcn_wave_barrier();
        if constexpr (NOUT % 4 == 0) {
#pragma unroll
            for (int j = 0; j * LPP < NOUT / 4; ++j) {
                const int c = j * LPP + gl;
                if (valid && c < NOUT / 4) {
                    typedef float v4f __attribute__((ext_vector_type(4)));
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
        __builtin_amdgcn_wave_barrier();
    } else if (valid) {
#pragma unroll
        for (int k = 0; k < C::PER; ++k) {
            const int idx = gl * C::PER + k;
            if (idx < C::NG) {
                float x = acc[k];
                if (do_scale && !PACKED) {
                    x = (x < 0.0f) ? 0.0f : x;
                    x = logf(x + 1.0f);
                }
                if (!idok) x = __builtin_nanf("");
                if constexpr (PACKED) {
                    const int a = idx / NN, b = idx - a * NN;
                    if (a <= b) __builtin_nontemporal_store(x, out + (a * NN - (a * (a - 1)) / 2 + (b - a)));
                } else {
                    __builtin_nontemporal_store(x, out + idx);
                }
            }
        }
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
    // (a reduction through LDS like gram_pair's was measured here: 62 slots x (LPP+4) floats per unit cost the kernel its
    // occupancy -- 800 000 units at d=256: 348 -> 414 us)
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


// ---------------------------------------------------------------------------------------------------------------
// The encoder's readout (models/TPNet.py:311-324): row i of the call has TWO anchors (the edge's src and dst) and K
// sampled neighbours w_0..w_{K-1}; the pair list is [G(w_k, src_i) for all i, k] followed by [G(w_k, dst_i)].  ONE lane
// group walks a row: the anchors' 2(L+1) rows are loaded once and stay in registers for all K neighbours, their own
// blocks <a,a> are reduced once per row, and per neighbour only its L+1 rows are fetched (instead of 3(L+1) per
// (w, src, dst) unit) and only the neighbour's blocks are formed and reduced:
//   slots = [ w.w (tri) | w.a1 (R*R) | w.a2 (R*R) ],  R = L+1, tri = R(R+1)/2    (42 slots at L = 3 against 62)
// ---------------------------------------------------------------------------------------------------------------
template <int LPP, int L>
struct AnchorCfg {
    static constexpr int R = L + 1;
    static constexpr int TRI = R * (R + 1) / 2;
    static constexpr int O_WW = 0, O_WA1 = TRI, O_WA2 = TRI + R * R;
    static constexpr int NS = TRI + 2 * R * R;
    static constexpr int MPS = ((NS + LPP - 1) / LPP) * LPP;
    static constexpr int PERS = MPS / LPP;
    static constexpr __host__ __device__ int tri(int i, int j) { return i * R - i * (i - 1) / 2 + (j - i); }   // i <= j
};

// all-reduce of one value over the LPP lanes of a group (butterfly; used once per row for the anchors' own blocks)
template <int LPP>
__device__ __forceinline__ float group_allreduce(float v) {
#pragma unroll
    for (int o = LPP / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, LPP);
    return v;
}

template <int LPP, int VPL, int W, int L, bool FULL>
__device__ __forceinline__ void gram_anchored(const tpnet_state& S, const int64_t* __restrict__ neigh, int64_t a1, int64_t a2,
                                              int k_begin, int k_end, bool valid, double now, double lambda, bool do_scale,
                                              float* __restrict__ out1, float* __restrict__ out2, int gl) {
    using C = GramCfg<LPP, L>;
    using AC = AnchorCfg<LPP, L>;
    constexpr int NR = C::NR, NN = C::NN, F = VPL * W;
    static_assert(FULL, "gram_anchored: rows of exactly one chunk");
    const int d = S.d;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(S.meta);
    bool aok = valid && (uint64_t)a1 < (uint64_t)S.N && (uint64_t)a2 < (uint64_t)S.N;
    if (valid && !aok && gl == 0) atomicAdd(S.err, 1u);
    if (!aok) { a1 = 0; a2 = 0; }
    // ---- the anchors: rows (decay applied) and their own blocks, once per row
    float fa[2][NR][F];
    {
        const int64_t ids[2] = {a1, a2};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const MetaView m = read_meta(meta, ids[s], READER_BID, now, lambda);
            const float* qb = S.q + ((int64_t)m.copy * S.N + ids[s]) * ((int64_t)L * d);
#pragma unroll
            for (int j = 0; j < VPL; ++j) ldv<W>(S.p0 + ids[s] * (int64_t)d, j * LPP + gl, &fa[s][0][j * W]);
            float g = 1.0f;
#pragma unroll
            for (int i = 1; i <= L; ++i) {
                g *= m.g;
#pragma unroll
                for (int j = 0; j < VPL; ++j) ldv<W>(qb + (int64_t)(i - 1) * d, j * LPP + gl, &fa[s][i][j * W]);
#pragma unroll
                for (int k = 0; k < F; ++k) fa[s][i][k] *= g;
            }
        }
    }
    float aa[2][AC::TRI];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int a = 0; a < NR; ++a) {
#pragma unroll
            for (int b = a; b < NR; ++b) aa[s][AC::tri(a, b)] = group_allreduce<LPP>(dot_chunk<F, false>(fa[s][a], fa[s][b]));
        }
    }
    // per-lane output plan (loop-invariant): where each of this lane's PER output elements comes from; the anchors' own
    // block is finished (clamp, log) once per row
    int o_s1[C::PER], o_s2[C::PER];
    bool o_aa[C::PER], o_in[C::PER];
    float o_y1[C::PER], o_y2[C::PER];
#pragma unroll
    for (int kk = 0; kk < C::PER; ++kk) {
        int idx = gl * C::PER + kk;
        o_in[kk] = idx < C::NG;
        idx = o_in[kk] ? idx : 0;
        const int a = idx / NN, b = idx - a * NN;
        o_aa[kk] = false;
        o_y1[kk] = 0.0f;
        o_y2[kk] = 0.0f;
        if (a < NR && b < NR) {
            const int i = a < b ? a : b, j = a < b ? b : a;
            o_s1[kk] = o_s2[kk] = AC::O_WW + AC::tri(i, j);
        } else if (a < NR) {                       // (w row a, anchor row b-NR)
            o_s1[kk] = AC::O_WA1 + a * NR + (b - NR);
            o_s2[kk] = AC::O_WA2 + a * NR + (b - NR);
        } else if (b < NR) {                       // mirrored
            o_s1[kk] = AC::O_WA1 + b * NR + (a - NR);
            o_s2[kk] = AC::O_WA2 + b * NR + (a - NR);
        } else {
            const int x = a - NR, y = b - NR;
            const int ta = AC::tri(x < y ? x : y, x < y ? y : x);
            o_aa[kk] = true;
            o_s1[kk] = o_s2[kk] = 0;
            float y1 = 0.0f, y2 = 0.0f;
#pragma unroll
            for (int q = 0; q < AC::TRI; ++q) {     // (register arrays: a select chain, no dynamic indexing)
                y1 = (ta == q) ? aa[0][q] : y1;
                y2 = (ta == q) ? aa[1][q] : y2;
            }
            if (do_scale) {
                y1 = (y1 < 0.0f) ? 0.0f : y1;
                y2 = (y2 < 0.0f) ? 0.0f : y2;
                y1 = logf(y1 + 1.0f);
                y2 = logf(y2 + 1.0f);
            }
            o_y1[kk] = y1;
            o_y2[kk] = y2;
        }
    }
    // ---- the neighbours [k_begin, k_end) of the row.  Software pipeline: ids and meta records of up to LPP neighbours are
    // fetched lane-parallel (two round trips for the whole chunk), and the rows of neighbour k+1 are in flight while
    // neighbour k's 42 inner products are formed and reduced
    for (int kc = k_begin; kc < k_end; kc += LPP) {
        const int nk = (k_end - kc < LPP) ? k_end - kc : LPP;
        int64_t my_w = (valid && gl < nk) ? neigh[kc + gl] : 0;
        const bool my_ok = aok && (uint64_t)my_w < (uint64_t)S.N;
        if (valid && aok && gl < nk && !my_ok) atomicAdd(S.err, 1u);
        if (!my_ok) my_w = 0;
        const MetaView my_m = read_meta(meta, my_w, READER_BID, now, lambda);
        float fn[NR][F];                                   // rows of the NEXT neighbour (raw)
        auto issue = [&](int j) {
            const int64_t w = __shfl(my_w, j, LPP);
            const int cp = __shfl(my_m.copy, j, LPP);
            const float* qb = S.q + ((int64_t)cp * S.N + w) * ((int64_t)L * d);
#pragma unroll
            for (int jj = 0; jj < VPL; ++jj) ldv<W>(S.p0 + w * (int64_t)d, jj * LPP + gl, &fn[0][jj * W]);
#pragma unroll
            for (int i = 1; i <= L; ++i) {
#pragma unroll
                for (int jj = 0; jj < VPL; ++jj) ldv<W>(qb + (int64_t)(i - 1) * d, jj * LPP + gl, &fn[i][jj * W]);
            }
        };
        issue(0);
        for (int j = 0; j < nk; ++j) {
        const int k = kc + j;
        const bool wok = __shfl((int)my_ok, j, LPP) != 0;
        const float mg = __shfl(my_m.g, j, LPP);
        float fw[NR][F];
#pragma unroll
        for (int x = 0; x < F; ++x) fw[0][x] = fn[0][x];
        {
            float g = 1.0f;
#pragma unroll
            for (int i = 1; i <= L; ++i) {
                g *= mg;
#pragma unroll
                for (int x = 0; x < F; ++x) fw[i][x] = fn[i][x] * g;
            }
        }
        if (j + 1 < nk) issue(j + 1);
        float acc[AC::MPS];
#pragma unroll
        for (int i = 0; i < AC::MPS; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int a = 0; a < NR; ++a) {
#pragma unroll
            for (int b = a; b < NR; ++b) acc[AC::O_WW + AC::tri(a, b)] = dot_chunk<F, false>(fw[a], fw[b]);
#pragma unroll
            for (int b = 0; b < NR; ++b) {
                acc[AC::O_WA1 + a * NR + b] = dot_chunk<F, false>(fw[a], fa[0][b]);
                acc[AC::O_WA2 + a * NR + b] = dot_chunk<F, false>(fw[a], fa[1][b]);
            }
        }
        Halve<AC::MPS, LPP / 2>::run(acc, gl);        // lane gl now holds the complete sums of slots [gl*PERS, ...)
        // output elements of the two Gram matrices [w rows | anchor rows]^2 from their slots (or the anchors' block)
#pragma unroll
        for (int kk = 0; kk < C::PER; ++kk) {
            const int idx = o_in[kk] ? gl * C::PER + kk : 0;
            const bool in = o_in[kk];
            float x1 = 0.0f, x2 = 0.0f;
#pragma unroll
            for (int j2 = 0; j2 < AC::PERS; ++j2) {
                const float t1 = __shfl(acc[j2], o_s1[kk] / AC::PERS, LPP);
                const float t2 = __shfl(acc[j2], o_s2[kk] / AC::PERS, LPP);
                x1 = (o_s1[kk] % AC::PERS == j2) ? t1 : x1;
                x2 = (o_s2[kk] % AC::PERS == j2) ? t2 : x2;
            }
            if (do_scale) {
                x1 = (x1 < 0.0f) ? 0.0f : x1;          // NaN < 0 is false: NaN passes through (:127)
                x2 = (x2 < 0.0f) ? 0.0f : x2;
                x1 = logf(x1 + 1.0f);                  // log(x + 1), not log1p (:128)
                x2 = logf(x2 + 1.0f);
            }
            if (o_aa[kk]) { x1 = o_y1[kk]; x2 = o_y2[kk]; }
            if (!wok) { x1 = __builtin_nanf(""); x2 = x1; }
            if (valid && in) {
                __builtin_nontemporal_store(x1, out1 + (int64_t)k * C::NG + idx);
                __builtin_nontemporal_store(x2, out2 + (int64_t)k * C::NG + idx);
            }
        }
        }
    }
}


}  // namespace tpnet
