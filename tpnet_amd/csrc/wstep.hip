// gfx950 kernels of the WINDOWED stream path (tpnet_common.h, WPlan; plan: plan.hip): a window of K batches runs as
// L launches of k_wupdate (one per layer of models/TPNet.py:90-96, ascending) + one launch of k_wreadout (all the
// window's pairwise readouts, models/TPNet.py:112-128) instead of K fused per-batch launches.
//
// Why ascending layers are legal here: layer i of node u after batch b needs (1) its own layer i after u's previous
// batch and (2) layer i-1 of its partners BEFORE batch b.  (1) is a chain inside ONE unit (all runs of u in the window
// are walked in order by the same lanes); (2) was written by the previous launch -- every run's result goes to a slot of
// the window's version log, and the plan has resolved, for every read, which slot (or the table's pre-window row)
// holds "the row before batch b".  Layer 0 never changes.  So no unit of a launch reads what another unit of the same
// launch writes, and a whole layer of K batches is one dependency-free, bandwidth-bound launch.
//
// Arithmetic per run (node u, batch b), identical for the lane-group and the workgroup variant and independent of how
// the stream is cut into windows:   new = old * g_u^i ;  then for every block of WIN_BLOCK contributions in index order
// (src side, then dst side): s = c_0; s += c_1; ...; new += s,   c_j = (P[i-1][partner_j] * g_j^(i-1)) * w_j.
#include "readout.hpp"

namespace tpnet {

static constexpr int WB = 256;            // threads per workgroup of both kernels
static constexpr int WPART_FLOATS = 8192; // LDS floats for the partial sums of a workgroup-walked chain (32 KB)

// g^n by repeated multiplication, as the per-batch kernels form the layers' decay (update.hpp: gu[i] = gu[i-1] * g)
__device__ __forceinline__ float pow_rep(float g, int n) {
    float r = 1.0f;
    for (int z = 0; z < n; ++z) r *= g;
    return r;
}

// where a contribution's partner row of layer (layer-1) lives, and its pending decay g^(layer-1)
__device__ __forceinline__ const float* partner_row(const tpnet_state& S, const WPlan& P, int layer, int32_t pv,
                                                    uint32_t ref, float g_log, uint32_t bid, double lambda,
                                                    const BatchDesc* __restrict__ wdesc0, float& gp) {
    const int d = S.d, L = S.L;
    if (layer == 1) {
        gp = 1.0f;
        return S.p0 + (int64_t)pv * d;
    }
    if (ref & WREF_TABLE) {
        const double Tb = wdesc0[(ref >> WREF_BW_SHIFT) & 63u].t_last;
        const MetaView m = read_meta(reinterpret_cast<const NodeMeta*>(S.meta), pv, bid, Tb, lambda);
        gp = pow_rep(m.g, layer - 1);
        return S.q + (((int64_t)m.copy * S.N + pv) * L + (layer - 2)) * (int64_t)d;
    }
    gp = pow_rep(g_log, layer - 1);
    return P.log + ((int64_t)(ref & WREF_SLOT_MASK) * L + (layer - 2)) * (int64_t)d;
}

// ---------------------------------------------------------------------------------------------------------------
// a chain walked by ONE group of LPP lanes: all runs of node u in the window, in order, the row in registers
// ---------------------------------------------------------------------------------------------------------------
template <int LPP, int VPL, bool FULL>
__device__ __forceinline__ void chain_light(const tpnet_state& S, const WPlan& P, Chain c, bool valid, int layer,
                                            uint32_t bid, double lambda, int gl, int64_t wc0,
                                            const BatchDesc* __restrict__ wdesc0) {
    constexpr int W = 4, F = VPL * W, U = 8;
    const int d = S.d, L = S.L;
    const int nvec = d / W;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    if (!valid) { c.cnt = 0; c.target = 0; c.j0 = 0; }
    const int32_t u = c.target;
    const MetaView mu = read_meta(meta, u, bid, c.t_first, lambda);
    const float g0 = pow_rep(mu.g, layer);
    const float* qold = S.q + (((int64_t)mu.copy * S.N + u) * L + (layer - 1)) * (int64_t)d;
    float* qnew = S.q + (((int64_t)(mu.copy ^ 1) * S.N + u) * L + (layer - 1)) * (int64_t)d;
    const int32_t* __restrict__ s_partner = P.base.s_partner;
    const float* __restrict__ s_coef = P.base.s_coef;

    for (int c0 = 0; c0 < (FULL ? 1 : nvec); c0 += LPP * VPL) {
        float acc[F], sblk[F];
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int vi = c0 + j * LPP + gl;
            ldv_maybe<W, FULL>(qold, vi, valid && vi < nvec, &acc[j * W]);
        }
#pragma unroll
        for (int x = 0; x < F; ++x) sblk[x] = 0.0f;
        bool first = false;
        float cur_dec = 1.0f;
        for (uint32_t r0 = 0; __any(r0 < c.cnt); r0 += LPP) {
            const uint32_t jm = c.j0 + r0 + (uint32_t)gl;
            const bool mine = r0 + (uint32_t)gl < c.cnt;
            const int32_t my_pv = mine ? s_partner[jm] : 0;
            const float my_w = mine ? s_coef[jm] : 0.0f;
            const uint32_t my_ref = mine ? P.s_ref[jm] : 0u;
            const float my_glog = mine ? P.s_g[jm] : 1.0f;
            const float my_dec = mine ? pow_rep(P.s_dec[jm], layer) : 1.0f;
            float my_gp;
            const float* my_row = partner_row(S, P, layer, my_pv, my_ref, my_glog, bid, lambda, wdesc0, my_gp);
            const uint32_t n_here = (r0 < c.cnt) ? ((c.cnt - r0 < (uint32_t)LPP) ? c.cnt - r0 : (uint32_t)LPP) : 0u;
            for (uint32_t k0 = 0; __any(k0 < n_here); k0 += U) {
                const float* rp[U];
                float w[U], gp[U], dc[U];
                uint32_t fl[U];
                bool ok[U];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const int sl = (int)k0 + k;
                    ok[k] = (uint32_t)sl < n_here;
                    rp[k] = reinterpret_cast<const float*>(__shfl((long long)reinterpret_cast<uintptr_t>(my_row), sl, LPP));
                    w[k] = __shfl(my_w, sl, LPP);
                    gp[k] = __shfl(my_gp, sl, LPP);
                    dc[k] = __shfl(my_dec, sl, LPP);
                    fl[k] = (uint32_t)__shfl((int)my_ref, sl, LPP);
                    if (!ok[k]) rp[k] = S.p0;                 // row 0 of layer 0: always a valid address
                }
                float r[U][F];
#pragma unroll
                for (int k = 0; k < U; ++k) {
#pragma unroll
                    for (int j = 0; j < VPL; ++j) {
                        const int vi = c0 + j * LPP + gl;
                        ldv_maybe<W, FULL>(rp[k], vi, ok[k] && vi < nvec, &r[k][j * W]);
                    }
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    if (ok[k]) {
                        const uint32_t pos = c.j0 + r0 + k0 + (uint32_t)k;      // chunk-relative sorted position
                        const bool bh = (fl[k] & WREF_BLK_HEAD) != 0;
#pragma unroll
                        for (int x = 0; x < F; ++x) {
                            const float m = (r[k][x] * gp[k]) * w[k];   // (P[i-1][partner], decayed) * time weight (TPNet.py:91-92)
                            sblk[x] = bh ? m : sblk[x] + m;             // scatter-add in index order (TPNet.py:93-96)
                        }
                        if (fl[k] & WREF_RUN_HEAD) {
                            first = true;
                            cur_dec = (pos == c.j0) ? g0 : dc[k];
                        }
                        if (fl[k] & WREF_BLK_TAIL) {
                            if (first) {
#pragma unroll
                                for (int x = 0; x < F; ++x) acc[x] *= cur_dec;   // decay to the run's clock (TPNet.py:83-85)
                            }
#pragma unroll
                            for (int x = 0; x < F; ++x) acc[x] = acc[x] + sblk[x];
                            first = false;
                        }
                        if (fl[k] & WREF_RUN_TAIL) {
                            float* lrow = P.log + ((int64_t)(pos - (uint32_t)wc0) * L + (layer - 1)) * (int64_t)d;
#pragma unroll
                            for (int j = 0; j < VPL; ++j) {
                                const int vi = c0 + j * LPP + gl;
                                if (FULL || vi < nvec) {
                                    stv<W>(lrow, vi, &acc[j * W]);
                                    if (fl[k] & WREF_LAST_RUN) stv<W>(qnew, vi, &acc[j * W]);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    if (valid && gl == 0 && layer == 1) publish_meta(meta + u, mu.copy ^ 1, c.t_final, bid);
}

// ---------------------------------------------------------------------------------------------------------------
// a chain walked by a WORKGROUP, one column part (LPH lanes x 16 bytes) of the rows: hubs.  The chain's blocks of
// WIN_BLOCK contributions are the tasks, dealt round-robin to the groups of LPH lanes; a task's rows are loaded together
// and summed in order; the block sums wait in LDS and one group folds them into the row in task order (= run order,
// block order), storing every run's result.  Same association as chain_light.
// ---------------------------------------------------------------------------------------------------------------
template <int LPH>
__device__ __forceinline__ void chain_heavy(const tpnet_state& S, const WPlan& P, const Chain c, int part, int layer,
                                            uint32_t bid, double lambda, int64_t wc0,
                                            const BatchDesc* __restrict__ wdesc0, float* __restrict__ lds_part,
                                            uint32_t* __restrict__ lds_u) {
    constexpr int G = WB / LPH;
    constexpr int TMAX = WPART_FLOATS / (LPH * 4);
    constexpr int NQ = (WIN_BLOCK + LPH - 1) / LPH;
    const int tid = threadIdx.x, g = tid / LPH, gl = tid % LPH;
    const int d = S.d, L = S.L;
    const int nvec = d / 4;
    const int vi = part * LPH + gl;
    const bool vok = vi < nvec;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    uint32_t* run_start = lds_u;            // [64] chain-relative position of the run of batch bw (0xFFFFFFFF: none)
    uint32_t* run_end = lds_u + 64;         // [64]
    float* run_dec = reinterpret_cast<float*>(lds_u + 128);   // [64]
    uint32_t* task_off = lds_u + 192;       // [65]
    uint32_t* tflag = lds_u + 272;          // [TMAX]
    const int32_t* __restrict__ s_partner = P.base.s_partner;
    const float* __restrict__ s_coef = P.base.s_coef;

    __syncthreads();                        // the previous unit of this workgroup is done with the tables
    if (tid < 64) { run_start[tid] = 0xFFFFFFFFu; run_end[tid] = 0u; run_dec[tid] = 1.0f; }
    __syncthreads();
    for (uint32_t p = tid; p < c.cnt; p += WB) {
        const uint32_t ref = P.s_ref[c.j0 + p];
        const uint32_t bw = (ref >> WREF_BW_SHIFT) & 63u;
        if (ref & WREF_RUN_HEAD) { run_start[bw] = p; run_dec[bw] = P.s_dec[c.j0 + p]; }
        if (ref & WREF_RUN_TAIL) run_end[bw] = p;
    }
    __syncthreads();
    if (tid < 64) {
        const uint32_t st = run_start[tid];
        const uint32_t nt = (st == 0xFFFFFFFFu) ? 0u : (run_end[tid] - st + WIN_BLOCK) / WIN_BLOCK;
        uint32_t inc = nt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inc, o, 64);
            if (tid >= o) inc += v;
        }
        task_off[tid] = inc - nt;
        if (tid == 63) task_off[64] = inc;
    }
    __syncthreads();
    const uint32_t T = task_off[64];

    const int32_t u = c.target;
    const MetaView mu = read_meta(meta, u, bid, c.t_first, lambda);
    const float g0 = pow_rep(mu.g, layer);
    const float* qold = S.q + (((int64_t)mu.copy * S.N + u) * L + (layer - 1)) * (int64_t)d;
    float* qnew = S.q + (((int64_t)(mu.copy ^ 1) * S.N + u) * L + (layer - 1)) * (int64_t)d;
    float acc[4];
    ldv_pred<4>(qold, vi, g == 0 && vok, acc);
    bool firstrun = true;

    for (uint32_t T0 = 0; T0 < T; T0 += TMAX) {
        const uint32_t Tend = (T0 + TMAX < T) ? T0 + TMAX : T;
        for (uint32_t tb = T0; tb < Tend; tb += G) {
            const uint32_t tt = tb + (uint32_t)g;
            const bool active = tt < Tend;
            // the run of task tt: largest bw with task_off[bw] <= tt (runs without tasks share an offset with their successor)
            uint32_t lo = 0, hi = 64;                       // invariant: task_off[lo] <= tt < task_off[hi]
            const uint32_t ts = active ? tt : T0;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (task_off[mid] <= ts) lo = mid; else hi = mid;
            }
            const uint32_t bw = lo;
            const uint32_t kblk = ts - task_off[bw];
            const uint32_t pstart = run_start[bw] + WIN_BLOCK * kblk;
            const uint32_t rend = run_end[bw] + 1;
            const uint32_t pend = (pstart + WIN_BLOCK < rend) ? pstart + WIN_BLOCK : rend;
            const uint32_t n = active ? pend - pstart : 0u;
            const float* my_row[NQ];
            float my_w[NQ], my_gp[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const uint32_t idx = (uint32_t)(q * LPH + gl);
                const bool mine = idx < n;
                const uint32_t jm = c.j0 + pstart + idx;
                const int32_t pv = mine ? s_partner[jm] : 0;
                my_w[q] = mine ? s_coef[jm] : 0.0f;
                const uint32_t ref = mine ? P.s_ref[jm] : 0u;
                const float glog = mine ? P.s_g[jm] : 1.0f;
                my_row[q] = partner_row(S, P, layer, pv, ref, glog, bid, lambda, wdesc0, my_gp[q]);
            }
            float r[WIN_BLOCK][4], w[WIN_BLOCK], gp[WIN_BLOCK];
#pragma unroll
            for (int k = 0; k < WIN_BLOCK; ++k) {
                const int q = k / LPH, sl = k % LPH;
                const float* rp = reinterpret_cast<const float*>(__shfl((long long)reinterpret_cast<uintptr_t>(my_row[q]), sl, LPH));
                w[k] = __shfl(my_w[q], sl, LPH);
                gp[k] = __shfl(my_gp[q], sl, LPH);
                const bool ok = (uint32_t)k < n;
                ldv_pred<4>(ok ? rp : S.p0, vi, ok && vok, r[k]);
            }
            float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < WIN_BLOCK; ++k) {
                if ((uint32_t)k < n) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const float m = (r[k][x] * gp[k]) * w[k];
                        s[x] = (k == 0) ? m : s[x] + m;
                    }
                }
            }
            if (active) {
                float* dst = lds_part + (size_t)(tt - T0) * (LPH * 4) + gl * 4;
#pragma unroll
                for (int x = 0; x < 4; ++x) dst[x] = s[x];
                if (gl == 0) tflag[tt - T0] = (kblk == 0 ? 1u : 0u) | (pend == rend ? 2u : 0u) | (bw << 2);
            }
        }
        __syncthreads();
        if (g == 0) {
            for (uint32_t tt = T0; tt < Tend; ++tt) {
                const uint32_t tf = tflag[tt - T0];
                const uint32_t bw = tf >> 2;
                if (tf & 1u) {
                    const float dec = firstrun ? g0 : pow_rep(run_dec[bw], layer);
#pragma unroll
                    for (int x = 0; x < 4; ++x) acc[x] *= dec;
                    firstrun = false;
                }
                const float* src = lds_part + (size_t)(tt - T0) * (LPH * 4) + gl * 4;
#pragma unroll
                for (int x = 0; x < 4; ++x) acc[x] = acc[x] + src[x];
                if ((tf & 2u) && vok) {
                    const uint32_t pos = c.j0 + run_end[bw];
                    float* lrow = P.log + ((int64_t)(pos - (uint32_t)wc0) * L + (layer - 1)) * (int64_t)d;
                    stv<4>(lrow, vi, acc);
                    if (tt == T - 1) stv<4>(qnew, vi, acc);
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0 && part == 0 && layer == 1) publish_meta(meta + u, mu.copy ^ 1, c.t_final, bid);
}

template <int LPH>
static constexpr int heavy_lds_words() { return 272 + WPART_FLOATS / (LPH * 4); }

template <int LPP, int VPL, bool FULL, int LPH>
__global__ __launch_bounds__(WB) void k_wupdate(tpnet_state S, WPlan P, int64_t w, int layer, uint32_t bid,
                                                double lambda, int HB, int CP) {
    __shared__ float lds_part[WPART_FLOATS];
    __shared__ uint32_t lds_u[heavy_lds_words<LPH>()];
    const int64_t wc0 = 2 * w * P.Ew;
    const BatchDesc* __restrict__ wdesc0 = P.base.desc + w * P.K;
    const WinDesc wd = P.wdesc[w];
    if ((int)blockIdx.x < HB) {
        const Chain* __restrict__ heavy = reinterpret_cast<const Chain*>(P.base.heavy) + wc0;
        const uint32_t units = wd.n_heavy * (uint32_t)CP;
        for (uint32_t h = blockIdx.x; h < units; h += (uint32_t)HB)
            chain_heavy<LPH>(S, P, heavy[h / (uint32_t)CP], (int)(h % (uint32_t)CP), layer, bid, lambda, wc0, wdesc0,
                             lds_part, lds_u);
        return;
    }
    constexpr int GPB = WB / LPP;
    const int gl = threadIdx.x % LPP;
    const int g = threadIdx.x / LPP;
    const Chain* __restrict__ small = reinterpret_cast<const Chain*>(P.base.light) + wc0;
    const int64_t nblk = (int64_t)gridDim.x - HB;
    for (int64_t base = ((int64_t)blockIdx.x - HB) * GPB; base < (int64_t)wd.n_small; base += nblk * GPB) {
        const int64_t idx = base + g;
        const bool valid = idx < (int64_t)wd.n_small;
        const Chain c = small[valid ? idx : 0];
        chain_light<LPP, VPL, FULL>(S, P, c, valid, layer, bid, lambda, gl, wc0, wdesc0);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// all pairwise readouts of a window: (src,dst) and (src,neg) of every edge on the state BEFORE the edge's batch
// ---------------------------------------------------------------------------------------------------------------
template <int LPP, int VPL, int L, bool FULL>
__global__ __launch_bounds__(WB) void k_wreadout(tpnet_state S, WPlan P, StreamArgs a, int64_t w, int64_t Ec, int64_t B,
                                                 uint32_t bid, double lambda, uint32_t flags) {
    constexpr int W = 4;
    using C = GramCfg<LPP, L>;
    constexpr int NR = C::NR, NN = C::NN;
    constexpr int GPB = WB / LPP;
    constexpr bool LR = C::template lds_reduce<WB>();
    __shared__ float stage[C::template stage_floats<WB>()];
    const int gl = threadIdx.x % LPP;
    const int g = threadIdx.x / LPP;
    const int d = S.d;
    const bool packed = (flags & TPNET_FLAG_PACKED) != 0;
    const bool do_scale = !(flags & TPNET_FLAG_NOT_SCALE) && !packed;
    const int ostride = packed ? C::NT : C::NG;
    const int64_t e0 = w * P.Ew;
    const int64_t ne = (Ec - e0 < P.Ew) ? Ec - e0 : P.Ew;
    const int64_t npos = a.out_pos ? ne : 0;
    const int64_t nneg = a.out_neg ? ne : 0;
    const int64_t npairs = npos + nneg;
    const bool both = npos != 0 && nneg != 0;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(S.meta);
    for (int64_t base = (int64_t)blockIdx.x * GPB; base < npairs; base += (int64_t)gridDim.x * GPB) {
        const int64_t pw = base + g;
        const bool valid = pw < npairs;
        // (src,dst) and (src,neg) of one edge sit in adjacent lane groups: the src rows are fetched by the same instructions
        const int64_t idx = valid ? (both ? (pw >> 1) : (pw < npos ? pw : pw - npos)) : 0;
        const bool isneg = both ? (pw & 1) != 0 : (pw >= npos);
        const int64_t e = e0 + idx;
        int64_t ids[2] = {a.src[e], isneg ? a.neg[e] : a.dst[e]};
        const int which[2] = {0, isneg ? 2 : 1};
        float* out = (isneg ? a.out_neg : a.out_pos) + e * ostride;
        const bool idok = valid && (uint64_t)ids[0] < (uint64_t)S.N && (uint64_t)ids[1] < (uint64_t)S.N;
        if (valid && !idok && gl == 0) atomicAdd(S.err, 1u);
        if (!idok) { ids[0] = 0; ids[1] = 0; }
        const double now = P.base.desc[e / B].now;
        const float* rowp[NN];
        float rs[NN];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint32_t ref = idok ? P.e_ref[(int64_t)which[s] * Ec + e] : WREF_TABLE;
            const float* qb;
            float gs;
            if (ref & WREF_TABLE) {
                const MetaView m = read_meta(meta, ids[s], bid, now, lambda);
                qb = S.q + ((int64_t)m.copy * S.N + ids[s]) * ((int64_t)L * d);
                gs = m.g;
            } else {
                qb = P.log + (int64_t)(ref & WREF_SLOT_MASK) * ((int64_t)L * d);
                gs = P.e_g[(int64_t)which[s] * Ec + e];
            }
            rowp[s * NR] = S.p0 + ids[s] * (int64_t)d;
            rs[s * NR] = 1.0f;
            float gg = 1.0f;
#pragma unroll
            for (int i = 1; i <= L; ++i) {
                gg *= gs;
                rowp[s * NR + i] = qb + (int64_t)(i - 1) * d;
                rs[s * NR + i] = gg;
            }
        }
        if (packed)
            gram_rows<LPP, VPL, W, L, FULL, true, LR>(rowp, rs, d, valid, idok, do_scale, out, gl, stage);
        else
            gram_rows<LPP, VPL, W, L, FULL, false, LR>(rowp, rs, d, valid, idok, do_scale, out, gl, stage);
    }
}

// heavy-path geometry: lanes per column part
static inline int heavy_lph(int d) {
    const int nvec = d / 4;
    return nvec <= 32 ? 4 : (nvec <= 64 ? 8 : 16);
}

template <int LPP, int VPL, bool FULL>
static int launch_wupdate_g(const tpnet_state& st, const WPlan& p, int64_t w, int layer, uint32_t bid, double lambda,
                            int64_t nc_w, hipStream_t s) {
    static const int hb_env = getenv("TPNET_DEV_WIN_HB") ? atoi(getenv("TPNET_DEV_WIN_HB")) : 0;
    const int HB = hb_env > 0 ? hb_env : 512;
    const int lph = heavy_lph(st.d);
    const int CP = (st.d / 4 + lph - 1) / lph;
    constexpr int GPB = WB / LPP;
    int lb = (int)((nc_w + GPB - 1) / GPB);          // chains <= contributions
    if (lb > 2048) lb = 2048;
    if (lb < 1) lb = 1;
    const dim3 grid((unsigned)(HB + lb));
    if (lph == 4)
        hipLaunchKernelGGL((k_wupdate<LPP, VPL, FULL, 4>), grid, dim3(WB), 0, s, st, p, w, layer, bid, lambda, HB, CP);
    else if (lph == 8)
        hipLaunchKernelGGL((k_wupdate<LPP, VPL, FULL, 8>), grid, dim3(WB), 0, s, st, p, w, layer, bid, lambda, HB, CP);
    else
        hipLaunchKernelGGL((k_wupdate<LPP, VPL, FULL, 16>), grid, dim3(WB), 0, s, st, p, w, layer, bid, lambda, HB, CP);
    return TPNET_OK;
}

int launch_window(const tpnet_state& st, const StreamArgs& a, const WPlan& p, int64_t w, int64_t Ec, int64_t batch,
                  double lambda, uint32_t launch_id, uint32_t flags, hipStream_t s, hipEvent_t* ev) {
    if ((reinterpret_cast<uintptr_t>(a.out_pos) | reinterpret_cast<uintptr_t>(a.out_neg)) & 15) return TPNET_ERR_BAD_ARG;
    if (st.d % 4 != 0) return TPNET_ERR_BAD_ARG;
    const int64_t e0 = w * p.Ew;
    const int64_t ne = (Ec - e0 < p.Ew) ? Ec - e0 : p.Ew;
    if (ne < 1) return TPNET_ERR_BAD_ARG;
    const Geom geo = pick_geom(st.d, true);        // (TPNET_DISPATCH_G declares its own `gm`)
    const bool full = st.d == geo.lpp * geo.vpl * 4;
    if (ev) (void)hipEventRecord(ev[0], s);
    for (int layer = 1; layer <= st.L; ++layer) {
        int rc = TPNET_ERR_BAD_ARG;
#define TPNET_WUPD(LPP_, VPL_) \
        rc = full ? launch_wupdate_g<LPP_, VPL_, true>(st, p, w, layer, launch_id, lambda, 2 * ne, s) \
                  : launch_wupdate_g<LPP_, VPL_, false>(st, p, w, layer, launch_id, lambda, 2 * ne, s)
        if (geo.lpp == 4) { TPNET_WUPD(4, 1); }
        else if (geo.lpp == 8) { TPNET_WUPD(8, 1); }
        else if (geo.lpp == 16 && geo.vpl == 1) { TPNET_WUPD(16, 1); }
        else if (geo.lpp == 16) { TPNET_WUPD(16, 2); }
        else if (geo.lpp == 32 && geo.vpl == 1) { TPNET_WUPD(32, 1); }
        else if (geo.lpp == 32) { TPNET_WUPD(32, 2); }
        else if (geo.vpl == 1) { TPNET_WUPD(64, 1); }
        else { TPNET_WUPD(64, 2); }
#undef TPNET_WUPD
        if (rc) return rc;
        if (ev) (void)hipEventRecord(ev[layer], s);
    }
    if (a.out_pos || a.out_neg) {
        const int64_t npairs = (a.out_pos ? ne : 0) + (a.out_neg ? ne : 0);
        TPNET_DISPATCH_G(geo, ({
            if constexpr (W == 4) {
                const int grid = grid_for(npairs, WB / LPP, 256 * 16);
                hipLaunchKernelGGL((k_wreadout<LPP, VPL, L, FULL>), dim3(grid), dim3(WB), 0, s, st, p, a, w, Ec, batch,
                                   launch_id, lambda, flags);
            }
        }));
    }
    if (ev) (void)hipEventRecord(ev[st.L + 1], s);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

}  // namespace tpnet
