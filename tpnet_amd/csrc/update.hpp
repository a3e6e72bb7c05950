// Update of one target node (models/TPNet.py:87-97) as device functions: light items (one group of lanes), light items
// on narrow rows (16 lanes over 4- or 8-lane row groups), heavy items (one workgroup per target and layer).
#pragma once
#include "device_common.hpp"

namespace tpnet {

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
template <int LPP, int VPL, int W, int L, bool FULL, bool NT, int BS>
__device__ __forceinline__ void update_item_block(const tpnet_state& S, const int32_t* __restrict__ s_partner,
                                                  const float* __restrict__ s_coef, int32_t u, uint32_t j0,
                                                  uint32_t cnt, int layer, uint32_t bid, double t_last, double lambda,
                                                  float* __restrict__ part /* LDS [F][BS] */, int vbase = 0) {
    // `vbase`: first 16-byte vector of the COLUMN PART this call covers.  The columns of a row are independent, so a hub at
    // wide rows is cut into parts of LPP*VPL vectors (LPP here = the lanes of one part, fewer than the kernel's row group):
    // more slices per workgroup share its contribution list and more workgroups share the hub, with nothing to combine.
    constexpr int F = VPL * W;
    constexpr int G = BS / LPP;
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
            const int vi = vbase + c0 + j * LPP + gl;
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
                        const int vi = vbase + c0 + j * LPP + gl;
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
        for (int k = 0; k < F; ++k) part[k * BS + threadIdx.x] = acc[k];
        __syncthreads();
        if (g == 0) {
            float tot[F];
#pragma unroll
            for (int k = 0; k < F; ++k) tot[k] = old[k] * gu;
            for (int q = 0; q < G; ++q) {
#pragma unroll
                for (int k = 0; k < F; ++k) tot[k] = tot[k] + part[k * BS + q * LPP + gl];
            }
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const int vi = vbase + c0 + j * LPP + gl;
                if (FULL || vi < nvec) stv<W, NT>(qnew, vi, &tot[j * W]);
            }
        }
        __syncthreads();
    }
    if (layer == 0 && vbase == 0 && threadIdx.x == 0) {
        publish_meta(meta + u, mu.copy ^ 1, t_last, bid);
    }
}


}  // namespace tpnet
