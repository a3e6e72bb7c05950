// gfx950 kernels of the WINDOWED stream path (tpnet_common.h, WPlan; plan: plan.hip).  A chunk of the stream is cut into
// windows of K batches; pipeline step j is ONE launch of k_wpipe that carries
//     update of layer 1 for window j, layer 2 for window j-1, ..., layer L for window j-L+1    (models/TPNet.py:90-96)
//     and all pairwise readouts of window j-L                                                   (models/TPNet.py:112-128)
// instead of one fused launch per batch.
//
// Why this is legal: layer i of node u after batch b needs (1) its own layer i after u's previous batch and (2) layer i-1
// of its partners BEFORE batch b.  Every (node, batch) run writes its result to its own slot of the chunk's version log,
// the table is frozen at its pre-chunk state until the write-back at the end of the chunk, and the plan has resolved
// every read "row of node n before batch b" to a log slot or the frozen table row.  (1) comes from a log slot written by
// the same role in an earlier step (or is carried in registers inside a chain), (2) from the role one layer down, which
// is one window ahead: written at step j-1 at the latest.  So no unit of a launch reads what another unit of the same
// launch writes, and the long dependent chain of a hub node's update hides behind the bandwidth-bound readout.
//
// Arithmetic per run (node u, batch b), identical for the lane-group and the workgroup variant and independent of how
// the stream is cut into windows:   new = old * g_u^i + S,   S = ((b_0 + b_1) + b_2) + ...,   b_k = the sum, in index order
// (src side, then dst side), of the k-th block of WIN_BLOCK contributions c_j = (P[i-1][partner_j] * g_j^(i-1)) * w_j.
#include "readout.hpp"

namespace tpnet {

static constexpr int WB = 256;            // threads per workgroup of both kernels
#ifndef TPNET_WPIPE_MINW
#define TPNET_WPIPE_MINW 4
#endif
// waves per SIMD the pipeline kernel is built for, per geometry: what the compiler reaches (a clean build has no -Wpass-failed).
// Narrow rows (<= 32 lanes x 1 vector) fit four; rows of 256 / 512 floats hold their 2 (L + 1) row fragments and the Gram's
// partial sums in 170-256 registers: three waves for exact-fit rows of up to 256 floats (L <= 3 on 32 x 2), two otherwise --
// those launches run at 0.57-0.58 of 8 TB/s on the memory side, where the plain copy is (profiles/r04_C3.md, r04_C5.md)
constexpr int wpipe_min_waves(int lpp, int vpl, int l, bool full) {
    return (lpp <= 32 && vpl == 1) ? TPNET_WPIPE_MINW
         : (lpp == 32 && vpl == 2) ? ((full && l <= 3) ? 3 : 2)
         : (lpp == 64 && vpl == 1) ? (full ? 3 : 2)
         : 2;
}

// g^n by repeated multiplication, as the per-batch kernels form the layers' decay (update.hpp: gu[i] = gu[i-1] * g)
__device__ __forceinline__ float pow_rep(float g, int n) {
    float r = 1.0f;
    for (int z = 0; z < n; ++z) r *= g;
    return r;
}

// where a contribution's partner row of layer (layer-1) lives, and its pending decay g^(layer-1).  The plan has resolved the
// version: a log slot, or a copy of the (frozen) table bundle; g = its decay to the run's clock.
__device__ __forceinline__ const float* partner_row(const tpnet_state& S, const WPlan& P, int layer, int32_t pv,
                                                    uint32_t ref, float g, float& gp) {
    const int d = S.d, L = S.L;
    if (layer == 1) {
        gp = 1.0f;
        return S.p0 + (int64_t)pv * d;
    }
    gp = pow_rep(g, layer - 1);
    if (ref & WREF_TABLE) return S.q + (((int64_t)(ref & 1u) * S.N + pv) * L + (layer - 2)) * (int64_t)d;
    return P.log + ((int64_t)(ref & WREF_SLOT_MASK) * L + (layer - 2)) * (int64_t)d;
}

// the row a chain starts from (layer `layer` of its node before the window) and, for a table row, its decay^layer to the
// first run's clock (a log row's decay comes with the run: WPlan::s_dec)
__device__ __forceinline__ const float* chain_start(const tpnet_state& S, const WPlan& P, const Chain& c, int layer,
                                                    float& g0) {
    const int d = S.d, L = S.L;
    g0 = pow_rep(c.g_first, layer);
    if (c.prev_ref & WREF_TABLE)
        return S.q + (((int64_t)(c.prev_ref & 1u) * S.N + c.target) * L + (layer - 1)) * (int64_t)d;
    return P.log + ((int64_t)(c.prev_ref & WREF_SLOT_MASK) * L + (layer - 1)) * (int64_t)d;
}

// ---------------------------------------------------------------------------------------------------------------
// a chain walked by ONE group of LPP lanes: all runs of node u in the window, in order, the row in registers
// ---------------------------------------------------------------------------------------------------------------
template <int LPP, int VPL, bool FULL>
__device__ __forceinline__ void chain_light(const tpnet_state& S, const WPlan& P, Chain c, bool valid, int layer,
                                            double lambda, int gl, unsigned long long* dbg) {
    (void)dbg;
    STAMP(5);
    constexpr int W = 4, F = VPL * W, U = (F == 4) ? 8 : 4;   // rows in flight per group: 32 VGPRs
    const int d = S.d, L = S.L;
    const int nvec = d / W;
    if (!valid) { c.cnt = 0; c.target = 0; c.j0 = 0; c.prev_ref = WREF_TABLE; c.g_first = 1.0f; }
    float g0;
    const float* qold = chain_start(S, P, c, layer, g0);
    const bool from_table = (c.prev_ref & WREF_TABLE) != 0;
    const int32_t* __restrict__ s_partner = P.base.s_partner;
    const float* __restrict__ s_coef = P.base.s_coef;

    for (int c0 = 0; c0 < (FULL ? 1 : nvec); c0 += LPP * VPL) {
        float acc[F], sblk[F], srun[F];
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int vi = c0 + j * LPP + gl;
            ldv_maybe<W, FULL>(qold, vi, valid && vi < nvec, &acc[j * W]);
        }
#pragma unroll
        for (int x = 0; x < F; ++x) { sblk[x] = 0.0f; srun[x] = 0.0f; }
        bool firstblk = false;
        float cur_dec = 1.0f;
        // a lane's contribution of a round of LPP: partner, weight, reference, decays -- the NEXT round's five loads are issued
        // before this round's rows are walked (one more dependent round trip per 32 contributions otherwise)
        int32_t nx_pv = 0;
        float nx_w = 0.0f, nx_glog = 1.0f, nx_decr = 1.0f;
        uint32_t nx_ref = 0u;
        auto fetch_meta = [&](uint32_t r0) {
            const uint32_t jm = c.j0 + r0 + (uint32_t)gl;
            const bool mine = r0 + (uint32_t)gl < c.cnt;
            nx_pv = mine ? s_partner[jm] : 0;
            nx_w = mine ? s_coef[jm] : 0.0f;
            nx_ref = mine ? P.s_ref[jm] : 0u;
            nx_glog = mine ? P.s_g[jm] : 1.0f;
            nx_decr = mine ? P.s_dec[jm] : 1.0f;
        };
        fetch_meta(0u);
        for (uint32_t r0 = 0; __any(r0 < c.cnt); r0 += LPP) {
            const int32_t my_pv = nx_pv;
            const float my_w = nx_w;
            const uint32_t my_ref = nx_ref;
            const float my_glog = nx_glog;
            const float my_dec = pow_rep(nx_decr, layer);
            if (__any(r0 + LPP < c.cnt)) fetch_meta(r0 + LPP);
            float my_gp;
            const float* my_row = partner_row(S, P, layer, my_pv, my_ref, my_glog, my_gp);
            const uint32_t n_here = (r0 < c.cnt) ? ((c.cnt - r0 < (uint32_t)LPP) ? c.cnt - r0 : (uint32_t)LPP) : 0u;
            // rows of U / 2 contributions at a time, TWO sets: a set's loads are issued before the other set is summed (with one
            // set of U, a chain of 64..96 contributions was 9..12 dependent row fetches of ~2 us each: the tail of every pipeline
            // step).  Measured and not kept: two sets of U, three sets of U / 2 -- 128 registers and spills.
            constexpr int US = U / 2;                          // rows per set; two sets: 4..8 rows in flight at any time
            float rA[US][F], rB[US][F];
            auto issue = [&](uint32_t k0, float (&r)[US][F]) {
#pragma unroll
                for (int k = 0; k < US; ++k) {
                    const int sl = (int)k0 + k;
                    const bool ok = (uint32_t)sl < n_here;
                    const float* rp = reinterpret_cast<const float*>(__shfl((long long)reinterpret_cast<uintptr_t>(my_row), sl, LPP));
                    if (!ok) rp = S.p0;                       // row 0 of layer 0: always a valid address
#pragma unroll
                    for (int j = 0; j < VPL; ++j) {
                        const int vi = c0 + j * LPP + gl;
                        ldv_maybe<W, FULL>(rp, vi, ok && vi < nvec, &r[k][j * W]);
                    }
                }
            };
            auto consume = [&](uint32_t k0, const float (&r)[US][F]) {
#pragma unroll
                for (int k = 0; k < US; ++k) {
                    const int sl = (int)k0 + k;
                    if ((uint32_t)sl < n_here) {
                        const float w = __shfl(my_w, sl, LPP);
                        const float gp = __shfl(my_gp, sl, LPP);
                        const float dc = __shfl(my_dec, sl, LPP);
                        const uint32_t fl = (uint32_t)__shfl((int)my_ref, sl, LPP);
                        const uint32_t pos = c.j0 + r0 + k0 + (uint32_t)k;      // chunk-relative sorted position
                        const bool bh = (fl & WREF_BLK_HEAD) != 0;
#pragma unroll
                        for (int x = 0; x < F; ++x) {
                            const float m = (r[k][x] * gp) * w;         // (P[i-1][partner], decayed) * time weight (TPNet.py:91-92)
                            sblk[x] = bh ? m : sblk[x] + m;             // scatter-add in index order (TPNet.py:93-96)
                        }
                        if (fl & WREF_RUN_HEAD) {
                            firstblk = true;
                            cur_dec = (pos == c.j0 && from_table) ? g0 : dc;
                        }
                        if (fl & WREF_BLK_TAIL) {
#pragma unroll
                            for (int x = 0; x < F; ++x) srun[x] = firstblk ? sblk[x] : srun[x] + sblk[x];
                            firstblk = false;
                        }
                        if (fl & WREF_RUN_TAIL) {
#pragma unroll
                            for (int x = 0; x < F; ++x) {
                                acc[x] *= cur_dec;                       // decay to the run's clock (TPNet.py:83-85)
                                acc[x] = acc[x] + srun[x];
                            }
                            float* lrow = P.log + ((int64_t)pos * L + (layer - 1)) * (int64_t)d;   // the run's slot of the log
#pragma unroll
                            for (int j = 0; j < VPL; ++j) {
                                const int vi = c0 + j * LPP + gl;
                                if (FULL || vi < nvec) stv<W>(lrow, vi, &acc[j * W]);
                            }
                        }
                    }
                }
            };
            if (__any(n_here > 0u)) issue(0u, rA);
            for (uint32_t k0 = 0; __any(k0 < n_here); k0 += 2 * US) {
                const bool more1 = __any(k0 + US < n_here);
                if (more1) issue(k0 + US, rB);
                consume(k0, rA);
                if (more1) {
                    if (__any(k0 + 2 * US < n_here)) issue(k0 + 2 * US, rA);
                    consume(k0 + US, rB);
                }
            }
        }
    }
    STAMP(6);
}

// ---------------------------------------------------------------------------------------------------------------
// a chain walked by a WORKGROUP, one column part (LPH lanes x 16 bytes) of the rows: hubs.
//   1. the runs of the chain (one per batch of the window the node is a target in) are tabled in LDS; the blocks of
//      WIN_BLOCK contributions are the TASKS, numbered in run order; the chain is walked in segments of <= TSEG tasks;
//   2. per segment: every thread resolves a few contributions (partner row address, weight, pending decay: the same
//      memory round trips whatever the chain's length) and parks them in LDS;
//   3. tasks are dealt round-robin to the groups of LPH lanes: rows loaded together, summed in order, block sum to LDS;
//   4. the block sums of a run are added in order by one group per run (runs in parallel);
//   5. one group walks the completed runs in order: row = row * decay + run sum, stores every run's result.
// Same association as chain_light.  Every loop bound is uniform over the workgroup (barriers inside).
// ---------------------------------------------------------------------------------------------------------------
struct HRec {             // a resolved contribution (12 bytes, LDS; 16 until round 5: the pointer)
    uint32_t row;         // partner's row of layer (layer-1), in rows of d floats of its array (0xFFFFFFFF: no contribution):
                          // layer 1 -> p0; else the sign bit of w selects the version log (set) or the table's q array (clear)
    float w, gp;          // time weight (>= 0: its sign bit is free), pending decay g^(layer-1)
};
static_assert(sizeof(HRec) == 12, "HRec must be 12 bytes");

// partner_row as (row index, array): what chain_heavy parks in LDS -- 12 bytes per contribution instead of 16 let a segment hold 152
// blocks of 8 in the LDS that held 128 (a C2 hub of ~1 050 contributions per 10-batch window is 135-140 blocks: ONE pass, not two)
__device__ __forceinline__ uint32_t partner_row_index(const tpnet_state& S, int layer, int32_t pv, uint32_t ref, float g, float& gp,
                                                      bool& in_log) {
    const int L = S.L;
    in_log = false;
    if (layer == 1) {
        gp = 1.0f;
        return (uint32_t)pv;
    }
    gp = pow_rep(g, layer - 1);
    if (ref & WREF_TABLE) return (uint32_t)((((int64_t)(ref & 1u) * S.N + pv) * L) + (layer - 2));
    in_log = true;
    return (uint32_t)((int64_t)(ref & WREF_SLOT_MASK) * L + (layer - 2));
}

template <int LPH>
struct HeavyCfg {
    static constexpr int G = WB / LPH;                       // groups per workgroup
    static constexpr int PW = LPH * 4;                       // floats of a row one part covers
    // tasks per segment: what fits the LDS that 128 tasks of 16-byte records took (per task: 8 records + PW partial sums + 2 words)
    static constexpr int TSEG = 128 * (8 * 4 + PW + 2) / (8 * 3 + PW + 2);
    static constexpr int HT = (TSEG + G - 1) / G < 4 ? (TSEG + G - 1) / G : 4;   // tasks a group has in flight
    // lds_u layout (32-bit words)
    static constexpr int O_START = 0, O_END = 64, O_DECP = 128 /* rank-indexed decay^layer */, O_TOFF = 192 /* [65] by bw */,
                         O_RLIST = 260 /* [64] */, O_NRUN = 324, O_RTOFF = 328 /* [65] by rank */, O_RSLOT = 396 /* [64] by rank */,
                         O_RSUM = 460 /* [64][PW] floats by rank */, O_REC = O_RSUM + 64 * PW /* HRec[TSEG*8] */,
                         O_PART = O_REC + TSEG * WIN_BLOCK * 3 /* [TSEG][PW] floats */, O_TPOS = O_PART + TSEG * PW /* [TSEG] */,
                         O_TN = O_TPOS + TSEG /* [TSEG] */, WORDS = O_TN + TSEG;
};

template <int LPH>
__device__ __forceinline__ void chain_heavy(const tpnet_state& S, const WPlan& P, const Chain c, int part, int layer,
                                            double lambda, uint32_t b0 /* first batch of the window */,
                                            uint32_t* __restrict__ lds_u, unsigned long long* dbg) {
    (void)dbg;
    STAMP(0);
    using H = HeavyCfg<LPH>;
    constexpr int G = H::G, PW = H::PW, TSEG = H::TSEG, HT = H::HT;
    const int tid = threadIdx.x, g = tid / LPH, gl = tid % LPH;
    const int d = S.d, L = S.L;
    const int nvec = d / 4;
    const int vi = part * LPH + gl;
    const bool vok = vi < nvec;
    uint32_t* run_start = lds_u + H::O_START;   // [64] by bw: chain-relative position of the run of batch bw (0xFFFFFFFF: none)
    uint32_t* run_end = lds_u + H::O_END;       // [64] by bw
    float* r_decp = reinterpret_cast<float*>(lds_u + H::O_DECP);   // [64] by rank: (decay from the node's previous run)^layer
    uint32_t* task_off = lds_u + H::O_TOFF;     // [65] by bw: first task of the run
    uint32_t* rlist = lds_u + H::O_RLIST;       // [64] the batches that have a run, ascending
    uint32_t* nrun_p = lds_u + H::O_NRUN;
    uint32_t* r_toff = lds_u + H::O_RTOFF;      // [65] by rank: first task of the run
    uint32_t* r_slot = lds_u + H::O_RSLOT;      // [64] by rank: window-relative log slot of the run
    float* r_sum = reinterpret_cast<float*>(lds_u + H::O_RSUM);    // [64][PW] by rank
    HRec* rec = reinterpret_cast<HRec*>(lds_u + H::O_REC);         // [TSEG * WIN_BLOCK]
    float* lds_part = reinterpret_cast<float*>(lds_u + H::O_PART); // [TSEG][PW]
    uint32_t* t_pos = lds_u + H::O_TPOS;        // [TSEG] chain-relative first position of the segment's tasks
    uint32_t* t_n = lds_u + H::O_TN;            // [TSEG] their lengths

    __syncthreads();                        // the previous unit of this workgroup is done with the tables
    if (tid < 64) { run_start[tid] = 0xFFFFFFFFu; run_end[tid] = 0u; r_decp[tid] = 1.0f; }
    __syncthreads();
    // (the flags and decays of 8 positions per thread are fetched together: one memory round trip per 2048 contributions)
    float* dec_bw = r_sum;                  // scratch until the runs are ranked: decay by bw
    for (uint32_t p0 = 0; p0 < c.cnt; p0 += 8 * WB) {
        uint32_t ref[8], bcq[8];
        float dec[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t p = p0 + (uint32_t)(q * WB + tid);
            const bool in = p < c.cnt;
            ref[q] = in ? P.s_ref[c.j0 + p] : 0u;
            dec[q] = in ? P.s_dec[c.j0 + p] : 1.0f;
            bcq[q] = in ? (uint32_t)P.s_bc[c.j0 + p] : b0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t p = p0 + (uint32_t)(q * WB + tid);
            const uint32_t bw = (bcq[q] - b0) & 63u;
            if (ref[q] & WREF_RUN_HEAD) { run_start[bw] = p; dec_bw[bw] = dec[q]; }
            if (ref[q] & WREF_RUN_TAIL) run_end[bw] = p;
        }
    }
    __syncthreads();
    if (tid < 64) {
        const uint32_t st = run_start[tid];
        const bool has = st != 0xFFFFFFFFu;
        const uint32_t nt = has ? (run_end[tid] - st + WIN_BLOCK) / WIN_BLOCK : 0u;
        uint32_t inc = nt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inc, o, 64);
            if (tid >= o) inc += v;
        }
        task_off[tid] = inc - nt;
        if (tid == 63) task_off[64] = inc;
        const unsigned long long hm = __ballot(has);
        const int rank = __popcll(hm & ((1ull << tid) - 1ull));
        const int nrun = __popcll(hm);
        const float dcp = has ? pow_rep(dec_bw[tid], layer) : 1.0f;     // (read before r_sum is reused: same wave, in order)
        if (has) {
            rlist[rank] = (uint32_t)tid;
            r_toff[rank] = inc - nt;
            r_slot[rank] = c.j0 + run_end[tid];               // chunk-relative sorted position of the run's tail
            r_decp[rank] = dcp;
        }
        if (tid == 0) { *nrun_p = (uint32_t)nrun; }
        if (tid == 63) r_toff[nrun] = inc;
    }
    __syncthreads();
    const uint32_t T = task_off[64];
    const uint32_t R = *nrun_p;
    STAMP(1);

    float g0;
    const float* qold = chain_start(S, P, c, layer, g0);
    const bool from_table = (c.prev_ref & WREF_TABLE) != 0;
    float acc[4];
    ldv_pred<4>(qold, vi, g == 0 && vok, acc);
    uint32_t next_run = 0;                  // rank of the first run the chain walk has not applied yet (wave 0)
    uint32_t next_run_all = 0;              // the same, known to every thread (through LDS)

    for (uint32_t T0 = 0; T0 < T; T0 += TSEG) {
        const uint32_t Tend = (T0 + TSEG < T) ? T0 + TSEG : T;
        // ---- 2. resolve the segment's contributions (task tt, element k) -> rec[(tt - T0) * 8 + k]
        for (uint32_t tt = T0 + (uint32_t)tid; tt < Tend; tt += WB) {      // first position / length of every task: ONE search each
            uint32_t lo = 0, hi = 64;                       // task_off[lo] <= tt < task_off[hi]: the run of task tt
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (task_off[mid] <= tt) lo = mid; else hi = mid;
            }
            const uint32_t ps = run_start[lo] + WIN_BLOCK * (tt - task_off[lo]);
            const uint32_t left = run_end[lo] + 1 - ps;
            t_pos[tt - T0] = ps;
            t_n[tt - T0] = left < (uint32_t)WIN_BLOCK ? left : (uint32_t)WIN_BLOCK;
        }
        __syncthreads();
        for (uint32_t x0 = 0; x0 < (Tend - T0) * WIN_BLOCK; x0 += 4 * WB) {
            int32_t pv[4];
            float w[4], glog[4];
            uint32_t ref[4];
            bool in[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t x = x0 + (uint32_t)(q * WB + tid);
                const uint32_t ti = x / WIN_BLOCK, k = x % WIN_BLOCK;
                const bool ok = ti < Tend - T0 && k < t_n[ti < Tend - T0 ? ti : 0];
                in[q] = ok;
                const uint32_t jm = c.j0 + (ok ? t_pos[ti] + k : 0u);
                pv[q] = ok ? P.base.s_partner[jm] : 0;
                w[q] = ok ? P.base.s_coef[jm] : 0.0f;
                ref[q] = ok ? P.s_ref[jm] : 0u;
                glog[q] = ok ? P.s_g[jm] : 1.0f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t x = x0 + (uint32_t)(q * WB + tid);
                HRec r;
                bool in_log;
                r.row = partner_row_index(S, layer, pv[q], ref[q], glog[q], r.gp, in_log);
                r.w = in_log ? __uint_as_float(__float_as_uint(w[q]) | 0x80000000u) : w[q];
                if (!in[q]) { r.row = 0xFFFFFFFFu; r.w = 0.0f; r.gp = 0.0f; }
                if (x < (Tend - T0) * WIN_BLOCK) rec[x] = r;
            }
        }
        __syncthreads();
        // ---- 3. block sums: HT tasks per group at once
        for (uint32_t tb = T0; tb < Tend; tb += G * HT) {
            float sb[HT][4];
            bool act[HT];
#pragma unroll
            for (int h = 0; h < HT; ++h) {
                const uint32_t tt = tb + (uint32_t)(h * G + g);
                act[h] = tt < Tend;
                const HRec* rr = rec + (size_t)((act[h] ? tt : T0) - T0) * WIN_BLOCK;
                float r[WIN_BLOCK][4], w[WIN_BLOCK], gp[WIN_BLOCK];
                bool ok[WIN_BLOCK];
#pragma unroll
                for (int k = 0; k < WIN_BLOCK; ++k) {
                    const HRec q = rr[k];
                    ok[k] = act[h] && q.row != 0xFFFFFFFFu;
                    w[k] = __uint_as_float(__float_as_uint(q.w) & 0x7FFFFFFFu);
                    gp[k] = q.gp;
                    const float* rbase = layer == 1 ? S.p0 : ((__float_as_uint(q.w) & 0x80000000u) ? P.log : S.q);
                    ldv_pred<4>(ok[k] ? rbase + (int64_t)q.row * d : S.p0, vi, ok[k] && vok, r[k]);
                }
#pragma unroll
                for (int x = 0; x < 4; ++x) sb[h][x] = 0.0f;
#pragma unroll
                for (int k = 0; k < WIN_BLOCK; ++k) {
                    if (ok[k]) {
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const float m = (r[k][x] * gp[k]) * w[k];
                            sb[h][x] = (k == 0) ? m : sb[h][x] + m;
                        }
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < HT; ++h) {
                const uint32_t tt = tb + (uint32_t)(h * G + g);
                if (act[h]) {
                    float* dst = lds_part + (size_t)(tt - T0) * PW + gl * 4;
#pragma unroll
                    for (int x = 0; x < 4; ++x) dst[x] = sb[h][x];
                }
            }
        }
        __syncthreads();
        STAMP(2);
        // ---- 4. run sums: group g adds the block sums of runs g, g+G, ... that have tasks in [T0, Tend)
        for (uint32_t rr = (uint32_t)g; rr < R; rr += G) {
            const uint32_t ta = r_toff[rr], tz = r_toff[rr + 1];              // tasks of the run
            const uint32_t a = ta > T0 ? ta : T0, z = tz < Tend ? tz : Tend;
            if (a < z) {
                float sr[4];
                float* rs_ = r_sum + (size_t)rr * PW + gl * 4;
                if (a == ta) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) sr[x] = lds_part[(size_t)(a - T0) * PW + gl * 4 + x];
                } else {
#pragma unroll
                    for (int x = 0; x < 4; ++x) sr[x] = rs_[x] + lds_part[(size_t)(a - T0) * PW + gl * 4 + x];
                }
                for (uint32_t tt = a + 1; tt < z; ++tt) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) sr[x] = sr[x] + lds_part[(size_t)(tt - T0) * PW + gl * 4 + x];
                }
#pragma unroll
                for (int x = 0; x < 4; ++x) rs_[x] = sr[x];
            }
        }
        __syncthreads();
        STAMP(3);
        // ---- 5. the chain: runs whose last task is in, in order.  Group 0 walks them in LDS (row = row * decay + run sum, left in
        // the run's slot of r_sum); then EVERY group stores results -- one store instruction per wave for up to 64 runs, where the
        // walking group's own stores were one instruction per run, each queued behind the CU's row gathers (3.8 of a hub's 10 us)
        uint32_t last = next_run;
        if (tid < 64) {
            const bool dn = (uint32_t)tid >= next_run && (uint32_t)tid < R && r_toff[tid + 1] <= Tend;   // (a prefix of the runs left)
            last = next_run + (uint32_t)__popcll(__ballot(dn));
        }
        if (g == 0) {
            for (uint32_t r0 = next_run; r0 < last; r0 += 4) {
                float dec[4], rs[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t rr = (r0 + q < last) ? r0 + q : last - 1;
                    dec[q] = (rr == 0 && from_table) ? g0 : r_decp[rr];
                    const float* rs_ = r_sum + (size_t)rr * PW + gl * 4;
#pragma unroll
                    for (int x = 0; x < 4; ++x) rs[q][x] = rs_[x];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t rr = r0 + q;
                    if (rr < last) {
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            acc[x] *= dec[q];                    // decay to the run's clock (TPNet.py:83-85)
                            acc[x] = acc[x] + rs[q][x];
                        }
                        float* rs_ = r_sum + (size_t)rr * PW + gl * 4;
#pragma unroll
                        for (int x = 0; x < 4; ++x) rs_[x] = acc[x];
                    }
                }
            }
            if (tid == 0) nrun_p[1] = last;
        }
        __syncthreads();
        {
            const uint32_t lastb_ = nrun_p[1];
            for (uint32_t rr = next_run_all + (uint32_t)g; rr < lastb_; rr += G) {
                float v[4];
                const float* rs_ = r_sum + (size_t)rr * PW + gl * 4;
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] = rs_[x];
                if (vok) {
                    float* lrow = P.log + ((int64_t)r_slot[rr] * L + (layer - 1)) * (int64_t)d;
                    stv<4>(lrow, vi, v);
                }
            }
            next_run_all = lastb_;
        }
        if (tid < 64) next_run = last;
        __syncthreads();
    }
    STAMP(4);
#ifdef TPNET_STAMPS
    if (dbg && (threadIdx.x & 63) == 0) {
        const size_t wv = ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
        if (wv < 4000) dbg[(wv * 8 + 7) * 2 + 0] = c.cnt;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// a chain walked by ONE workgroup, whole rows: the middle of the length distribution (WIN_MED_MIN contributions up to the heavy
// threshold).  One lane group walking such a chain is n / 8 dependent rounds of row fetches -- the tail of every pipeline step of
// a short stream; a workgroup per column part (chain_heavy) is eight workgroups per chain, mostly idle.  Here:
//   1. a thread per position of a segment (<= SEGP positions, <= BCAP blocks of WIN_BLOCK) resolves its contribution (partner row,
//      weight, pending decay, structure bits) and parks it in LDS; blocks are numbered by ballots;
//   2. lane group g owns the blocks whose head falls into positions [PG g, PG (g + 1)): it walks them in order, rows of U
//      contributions in flight (<= 3 rounds), and leaves every block's sum in LDS;
//   3. lane group 0 walks the blocks in order: block sums -> run sums -> row = row * decay + run sum, stores every run's result.
// Same association as chain_light / chain_heavy; rows that are exactly one chunk (FULL) only.
// ---------------------------------------------------------------------------------------------------------------
template <int LPP, int VPL, int LWORDS>
struct MedCfg {
    static constexpr int GPB = WB / LPP;                     // lane groups per workgroup
    static constexpr int PG = GPB * 16 <= WB ? 16 : WB / GPB;   // positions per group and segment
    static constexpr int SEGP = GPB * PG;                    // positions per segment
    static constexpr int RW = LPP * VPL * 4;                 // floats per row
    // LDS (32-bit words): per position {row pointer (2), weight, decay^(layer-1), flags, own decay^layer, block}; per block
    // {flags, decay, log slot}; per group its first position; the block sums
    static constexpr int O_ROW = 0, O_W = 2 * SEGP, O_GP = 3 * SEGP, O_FL = 4 * SEGP, O_DEC = 5 * SEGP, O_BLK = 6 * SEGP,
                         O_BFL = 7 * SEGP, BCAP_ = (LWORDS - 7 * SEGP - (GPB + 1) - 16 - 4) / (RW + 3),
                         BCAP = BCAP_ > SEGP ? SEGP : BCAP_, O_BDEC = O_BFL + BCAP, O_BSLOT = O_BDEC + BCAP,
                         O_OWN = O_BSLOT + BCAP, O_MISC = O_OWN + GPB + 1, O_SUM = (O_MISC + 16 + 3) / 4 * 4,
                         WORDS = O_SUM + BCAP * RW;
    static_assert(WORDS <= LWORDS, "chain_medium: LDS layout");
    static_assert(BCAP >= 8, "chain_medium: LDS for at least 8 block sums");
};

template <int LPP, int VPL, int LWORDS>
__device__ __forceinline__ void chain_medium(const tpnet_state& S, const WPlan& P, const Chain c, int layer,
                                             uint32_t* __restrict__ lds_u) {
    using M = MedCfg<LPP, VPL, LWORDS>;
    constexpr int W = 4, F = VPL * W, U = (F == 4) ? 8 : 4;
    constexpr int GPB = M::GPB, SEGP = M::SEGP, BCAP = M::BCAP, RW = M::RW;
    static_assert(SEGP <= WB, "a thread per position");
    const int tid = threadIdx.x, g = tid / LPP, gl = tid % LPP;
    const int d = S.d, L = S.L;
    const float** p_row = reinterpret_cast<const float**>(lds_u + M::O_ROW);
    float* p_w = reinterpret_cast<float*>(lds_u + M::O_W);
    float* p_gp = reinterpret_cast<float*>(lds_u + M::O_GP);
    uint32_t* p_fl = lds_u + M::O_FL;
    float* p_dec = reinterpret_cast<float*>(lds_u + M::O_DEC);
    uint32_t* p_blk = lds_u + M::O_BLK;
    uint32_t* b_fl = lds_u + M::O_BFL;                 // bit 0: the block heads a run, bit 1: it ends one
    float* b_dec = reinterpret_cast<float*>(lds_u + M::O_BDEC);
    uint32_t* b_slot = lds_u + M::O_BSLOT;             // chunk-relative position of the block's last contribution
    uint32_t* own = lds_u + M::O_OWN;                  // [GPB + 1] first position a group owns (SEGP: none)
    uint32_t* misc = lds_u + M::O_MISC;                // [2..5] block heads per wave, [6..9] one past a wave's last kept block tail
    float* bsum = reinterpret_cast<float*>(lds_u + M::O_SUM);

    float g0;
    const float* qold = chain_start(S, P, c, layer, g0);
    const bool from_table = (c.prev_ref & WREF_TABLE) != 0;
    float acc[F], srun[F];
#pragma unroll
    for (int j = 0; j < VPL; ++j) ldv<W>(g == 0 ? qold : S.p0, j * LPP + gl, &acc[j * W]);     // (only group 0 walks the chain)
#pragma unroll
    for (int x = 0; x < F; ++x) srun[x] = 0.0f;
    bool firstblk = false;
    float cur_dec = 1.0f;

    for (uint32_t s0 = 0; s0 < c.cnt;) {
        __syncthreads();                                   // the previous segment (or unit) is done with the tables
        // ---- 1. the segment's positions
        // (positions per group: the segment spread over all groups -- a chain of 24 contributions is 8 groups x 3 positions, one round
        // of row loads, not two groups x 16)
        const uint32_t left = c.cnt - s0 < (uint32_t)SEGP ? c.cnt - s0 : (uint32_t)SEGP;
        const uint32_t pg = (left + GPB - 1) / GPB < 2u ? 2u : (left + GPB - 1) / GPB;
        const uint32_t pos = s0 + (uint32_t)tid;
        const bool in = tid < SEGP && pos < c.cnt;
        const uint32_t jm = c.j0 + (in ? pos : 0u);
        const int32_t pv = in ? P.base.s_partner[jm] : 0;
        const float wgt = in ? P.base.s_coef[jm] : 0.0f;
        const uint32_t ref = in ? P.s_ref[jm] : 0u;
        const float glog = in ? P.s_g[jm] : 1.0f;
        const float decr = in ? P.s_dec[jm] : 1.0f;
        if (tid <= GPB) own[tid] = SEGP;
        // blocks are numbered in position order: heads counted by ballot over the segment's waves
        const bool head = in && (ref & WREF_BLK_HEAD) != 0;
        const unsigned long long hm = __ballot(head);
        const int lane = tid & 63, wave = tid >> 6;
        if (lane == 0 && wave < 4) misc[2 + wave] = (uint32_t)__popcll(hm);
        __syncthreads();
        uint32_t blk = (uint32_t)__popcll(hm & ((2ull << lane) - 1ull)) - 1u;      // heads at or before this position, minus one
        for (int wv = 0; wv < wave && wv < 4; ++wv) blk += misc[2 + wv];
        const bool keep = in && blk < (uint32_t)BCAP;      // (a segment ends after BCAP blocks)
        if (tid < SEGP) {
            float gp;
            const float* row = partner_row(S, P, layer, pv, ref, glog, gp);
            p_row[tid] = row;
            p_w[tid] = wgt;
            p_gp[tid] = gp;
            p_fl[tid] = keep ? ref : 0u;
            p_dec[tid] = pow_rep(decr, layer);
            p_blk[tid] = blk;
        }
        if (keep && head) {
            atomicMin(&own[(uint32_t)tid / pg], (uint32_t)tid);
            b_fl[blk] = (ref & WREF_RUN_HEAD) ? 1u : 0u;
            b_dec[blk] = (pos == 0u && from_table) ? g0 : pow_rep(decr, layer);
        }
        // the segment ends behind the last kept block that is whole (the SEGP positions may cut a block: it opens the next segment)
        const bool tail = keep && (ref & WREF_BLK_TAIL) != 0;
        const unsigned long long tm = __ballot(tail);
        if (lane == 0 && wave < 4) misc[6 + wave] = tm ? (uint32_t)(wave * 64 + 64 - __clzll((long long)tm)) : 0u;
        __syncthreads();
        if (tail) {
            if (ref & WREF_RUN_TAIL) atomicOr(&b_fl[blk], 2u);
            b_slot[blk] = c.j0 + pos;
        }
        uint32_t npos = misc[6];
#pragma unroll
        for (int wv = 1; wv < 4; ++wv) npos = misc[6 + wv] > npos ? misc[6 + wv] : npos;
        __syncthreads();
        const uint32_t nblk = p_blk[npos - 1u] + 1u;
        // ---- 2. block sums: group g walks positions [own[g], first position a later group owns)
        {
            uint32_t a0 = own[g], z0 = npos;
            for (int q = GPB - 1; q > g; --q)             // the first later group that owns a block bounds this group's walk
                if (own[q] < z0) z0 = own[q];
            if (a0 >= z0) { a0 = 0; z0 = 0; }
            float sblk[F];
#pragma unroll
            for (int x = 0; x < F; ++x) sblk[x] = 0.0f;
            for (uint32_t k0 = a0; k0 < z0; k0 += U) {
                float r[U][F];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const uint32_t q = k0 + k < z0 ? k0 + k : a0;
#pragma unroll
                    for (int j = 0; j < VPL; ++j) ldv<W>(p_row[q], j * LPP + gl, &r[k][j * W]);
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const uint32_t q = k0 + k;
                    if (q < z0) {
                        const float w = p_w[q], gp = p_gp[q];
                        const uint32_t fl = p_fl[q];
                        const bool bh = (fl & WREF_BLK_HEAD) != 0;
#pragma unroll
                        for (int x = 0; x < F; ++x) {
                            const float m = (r[k][x] * gp) * w;       // (P[i-1][partner], decayed) * time weight (TPNet.py:91-92)
                            sblk[x] = bh ? m : sblk[x] + m;           // scatter-add in index order (TPNet.py:93-96)
                        }
                        if (fl & WREF_BLK_TAIL) {
                            float* dst = bsum + (size_t)p_blk[q] * RW;
#pragma unroll
                            for (int j = 0; j < VPL; ++j) {
                                float4 v4 = make_float4(sblk[j * W], sblk[j * W + 1], sblk[j * W + 2], sblk[j * W + 3]);
                                reinterpret_cast<float4*>(dst)[j * LPP + gl] = v4;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        // ---- 3. the chain: blocks in order, in LDS (a run's result replaces its last block's sum); then every group stores rows --
        // the walking group's own stores would be one instruction per run, each queued behind the CU's row gathers
        if (g == 0) {
            for (uint32_t b = 0; b < nblk; ++b) {
                const uint32_t bf = b_fl[b];
                float sb[F];
                float4* slot = reinterpret_cast<float4*>(bsum + (size_t)b * RW);
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    const float4 v4 = slot[j * LPP + gl];
                    sb[j * W] = v4.x; sb[j * W + 1] = v4.y; sb[j * W + 2] = v4.z; sb[j * W + 3] = v4.w;
                }
                if (bf & 1u) { firstblk = true; cur_dec = b_dec[b]; }
#pragma unroll
                for (int x = 0; x < F; ++x) srun[x] = firstblk ? sb[x] : srun[x] + sb[x];
                firstblk = false;
                if (bf & 2u) {
#pragma unroll
                    for (int x = 0; x < F; ++x) {
                        acc[x] *= cur_dec;                   // decay to the run's clock (TPNet.py:83-85)
                        acc[x] = acc[x] + srun[x];
                    }
#pragma unroll
                    for (int j = 0; j < VPL; ++j)
                        slot[j * LPP + gl] = make_float4(acc[j * W], acc[j * W + 1], acc[j * W + 2], acc[j * W + 3]);
                }
            }
        }
        __syncthreads();
        for (uint32_t b = (uint32_t)g; b < nblk; b += GPB) {
            if (b_fl[b] & 2u) {
                const float4* slot = reinterpret_cast<const float4*>(bsum + (size_t)b * RW);
                float* lrow = P.log + ((int64_t)b_slot[b] * L + (layer - 1)) * (int64_t)d;
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    const float4 v4 = slot[j * LPP + gl];
                    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
                    stv<W>(lrow, j * LPP + gl, v);
                }
            }
        }
        s0 += npos;
    }
}

// lanes per column part of a workgroup-walked chain, by the geometry's row chunk (LPP * VPL vectors)
constexpr int heavy_lph(int lpp, int vpl) { return lpp * vpl <= 32 ? 4 : (lpp * vpl <= 64 ? 8 : 16); }

// what one pipeline step carries (host -> kernel, by value)
struct WStep {
    int64_t w_upd[TPNET_MAX_LAYERS];   // window of the update of layer i+1 at this step (-1: none)
    int64_t w_read;                    // window whose readouts run at this step (-1: none)
    uint32_t seg[2 * TPNET_MAX_LAYERS + 3];   // first block of: hub chains[layer 1..L], medium chains[layer 1..L], chains (all layers), readout; then the grid size
    int32_t CP;                        // column parts of a workgroup-walked chain
    int32_t dbg_layer;                 // diagnostic builds: the layer whose units are stamped (0: none)
    int32_t dbg_skip;                  // timing experiments: 1 = no hub chains, 2 = no chains, 4 = no readout
    uint32_t n_active;                 // layers that have a window at this step
    uint32_t hub_ext;                  // 2 / 1: this step walks the chains of length classes 1..2 / of class 1 by workgroups too ...
    uint32_t hub_blocks;               // ... where they fit ONE round of the layer's hub workgroups
    uint32_t med;                      // chains below the heavy threshold that get a workgroup each (chain_medium): 0 none, 1 / 2 / 3 =
                                       // from about half / a quarter of the threshold / from WIN_MED_MIN contributions
};

// chains of a window that are NOT walked by one lane group each: the hubs, and the medium ones where chain_medium serves
__device__ __forceinline__ uint32_t big_chains(const WStep& st, const WinDesc& wd, uint32_t n_hub) {
    const uint32_t n = st.med == 3u ? wd.n_med : st.med == 2u ? wd.n_ext : st.med == 1u ? wd.n_ext1 : 0u;
    return n > n_hub ? n : n_hub;
}

// the chains of a window this step walks by workgroups: the heavy ones, and -- in a step with workgroups to spare -- the next one
// or two length classes, as long as ONE round of the layer's hub workgroups takes them (a second round costs more than the lane
// groups' tail it removes; on streams with very heavy hubs the classes are too many and the step stays as it was)
__device__ __forceinline__ uint32_t hub_chains(const WStep& st, const WinDesc& wd) {
    if (st.hub_ext == 2 && wd.n_ext * (uint32_t)st.CP <= st.hub_blocks) return wd.n_ext;
    if (st.hub_ext >= 1 && wd.n_ext1 * (uint32_t)st.CP <= st.hub_blocks) return wd.n_ext1;
    return wd.n_heavy;
}

// ---------------------------------------------------------------------------------------------------------------
// one pipeline step: block ranges = [hub chains of layer 1 | ... | of layer L | chains of layer 1 | ... | L | readout pairs]
// (the long units first, the short uniform ones last: they fill the tail)
// ---------------------------------------------------------------------------------------------------------------
// RS: the readout takes the 16-lane x 2-vector geometry while the chains keep 32 lanes x 1 vector (rows of 17..32 vectors:
// a long list of independent pairs wants four pairs per wave, a chain wants a contribution per lane and round)
// SHARD: the launch of a row shard (wshard.hip): StreamArgs::own_mod = 0, local ids, only the pairs whose src node is this rank's
// (ids < own_rem) are read out -- a variant of its own: the C2 kernel sits exactly at the 128 registers four waves per SIMD allow, and
// the test cost it 28-44 bytes of scratch per lane (4 % per launch)
template <int LPP, int VPL, int L, bool FULL, bool RS, bool SHARD = false>
__global__ __launch_bounds__(WB, wpipe_min_waves(LPP, VPL, L, FULL)) void k_wpipe(tpnet_state S, WPlan P, StreamArgs a, WStep st, int64_t Ec, int64_t B,
                                              double lambda, uint32_t flags) {
    constexpr int W = 4;
    constexpr int LPH = heavy_lph(LPP, VPL);
    constexpr int RLPP = RS ? 16 : LPP, RVPL = RS ? 2 : VPL;
    static_assert(!RS || (LPP == 32 && VPL == 1), "readout split: chains on 32 x 1");
    using C = GramCfg<RLPP, L>;
    // (the readout adds its lanes' partial sums through DPP / permlane swaps here, not through LDS: the workgroup-walked
    // chains' tables are what the block's LDS is for, and 30 KB instead of 44-50 KB is one more resident workgroup per CU)
    constexpr bool LR = false;
    constexpr int RWORDS = (RLPP < 16) ? C::NG * (WB / RLPP) : 1;
    constexpr int LWORDS = HeavyCfg<LPH>::WORDS > RWORDS ? HeavyCfg<LPH>::WORDS : RWORDS;
    constexpr int SL = 2 * L;                                   // index of the lane-group chains' range
    __shared__ __attribute__((aligned(16))) uint32_t lds_u[LWORDS];
    const int gl = threadIdx.x % LPP;
    const int g = threadIdx.x / LPP;
    constexpr int GPB = WB / LPP;
    const uint32_t bx = blockIdx.x;
    // which range is this block in (compares against every bound: no dynamic indexing of the kernel arguments).
    // Ranges: 0..L-1 hub chains of layer 1..L; L..2L-1 medium chains of layer 1..L; 2L: the other chains of all active layers,
    // interleaved block by block (chain lists are sorted longest first: every layer's long chains start early); 2L+1: readout
    int sidx = 0;
    uint32_t seg_lo = 0, seg_hi = st.seg[1];
    int64_t w = st.w_upd[0];
#pragma unroll
    for (int i = 1; i <= SL + 1; ++i) {
        if (bx >= st.seg[i]) {
            sidx = i;
            seg_lo = st.seg[i];
            seg_hi = st.seg[i + 1];
            w = (i < L) ? st.w_upd[i] : (i < SL) ? st.w_upd[i - L] : st.w_read;
        }
    }
    uint32_t rb_ = bx - seg_lo, nblk_ = seg_hi - seg_lo;
    int layer = (sidx >= L && sidx < SL) ? sidx - L + 1 : sidx + 1;
    if (sidx == SL) {                                          // interleaved chains: block k -> the (k % nact)-th active layer
        const uint32_t nact = st.n_active;
        const uint32_t r = rb_ % nact;
        rb_ = rb_ / nact;
        nblk_ = nblk_ / nact;
        uint32_t seen = 0;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            if (st.w_upd[i] >= 0) {
                if (seen == r) { layer = i + 1; w = st.w_upd[i]; }
                ++seen;
            }
        }
    }
    const uint32_t rb = rb_;                                   // block inside its range (of its layer, for chains)
    const uint32_t nblk = nblk_;
#ifdef TPNET_STAMPS
    // diagnostic build: wall-clock start / end of every block of the stamped step (second half of the stamp buffer)
    unsigned long long* bst = (st.dbg_layer == -1 && P.base.dbg && bx < 16000) ? P.base.dbg + 65536 + (size_t)bx * 4 : nullptr;
    if (bst && threadIdx.x == 0) { bst[0] = __builtin_amdgcn_s_memrealtime(); bst[2] = (unsigned long long)(sidx == SL ? L + layer - 1 : (sidx > SL ? 2 * L : (sidx >= L ? 2 * L + 1 + sidx - L : sidx))); }
#define BEND() do { if (bst) { __syncthreads(); if (threadIdx.x == 0) bst[1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define BEND() do { } while (0)
#endif

    if (sidx == SL + 1) {
#ifdef TPNET_DEV
        if (st.dbg_skip & 4) return;
#endif
        // ---- readouts of window w_read: (src,dst) and (src,neg) of every edge on the state BEFORE the edge's batch
        constexpr int NR = C::NR, NN = C::NN;
        constexpr int RGPB = WB / RLPP;
        const int rgl = threadIdx.x % RLPP;
        const int rg = threadIdx.x / RLPP;
        float* stage = reinterpret_cast<float*>(lds_u);
        const int d = S.d;
        const bool packed = (flags & TPNET_FLAG_PACKED) != 0;
        const bool do_scale = !(flags & TPNET_FLAG_NOT_SCALE) && !packed;
        const int ostride = packed ? C::NT : C::NG;
        const int64_t e0 = w * P.Ew;
        // (a row shard walks the list of the window's edges whose src node is its own: a rank of G would otherwise walk G times its share)
        const int64_t l0 = SHARD ? (int64_t)a.own_start[w] : 0;
        const int64_t ne = SHARD ? (int64_t)a.own_start[w + 1] - l0 : ((Ec - e0 < P.Ew) ? Ec - e0 : P.Ew);
        const int64_t npos = a.out_pos ? ne : 0;
        const int64_t nneg = a.out_neg ? ne : 0;
        const int64_t npairs = npos + nneg;
        const bool both = npos != 0 && nneg != 0;
        for (int64_t base = (int64_t)rb * RGPB; base < npairs; base += (int64_t)nblk * RGPB) {
            const int64_t pw = base + rg;
            const bool valid = pw < npairs;
            // (src,dst) and (src,neg) of one edge sit in adjacent lane groups: the src rows are fetched by the same instructions
            const int64_t idx = valid ? (both ? (pw >> 1) : (pw < npos ? pw : pw - npos)) : 0;
            const bool isneg = valid && (both ? (pw & 1) != 0 : (pw >= npos));      // (a lane group past the list reads dst[e0]: `neg` may be null)
            int64_t e = e0 + idx;
            if constexpr (SHARD) e = valid ? (int64_t)a.own_list[l0 + idx] : e0;
            int64_t ids[2] = {a.src[e], isneg ? a.neg[e] : a.dst[e]};
            const int which[2] = {0, isneg ? 2 : 1};
            float* out = (isneg ? a.out_neg : a.out_pos) + e * ostride;
            const bool idok = valid && (uint64_t)ids[0] < (uint64_t)S.N && (uint64_t)ids[1] < (uint64_t)S.N;
            if (valid && !idok && rgl == 0) atomicAdd(S.err, 1u);
            if (!idok) { ids[0] = 0; ids[1] = 0; }
            const float* rowp[NN];
            float rs[NN];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // (indexed by the EDGE, not by the ids: requested together with the ids -- as `idok ? P.e_ref[..] : ..` the two loads
                // waited for the ids' round trip, a third dependent one in front of every pair's rows until round 5; e is a valid index
                // for every lane group, and the entries of an edge with a bad id are never used)
                const uint32_t ref_e = P.e_ref[(int64_t)which[s] * Ec + e];
                const float gs_e = P.e_g[(int64_t)which[s] * Ec + e];
                const uint32_t ref = idok ? ref_e : WREF_TABLE;
                const float gs = idok ? gs_e : 1.0f;
                const float* qb = (ref & WREF_TABLE) ? S.q + ((int64_t)(ref & 1u) * S.N + ids[s]) * ((int64_t)L * d)
                                                     : P.log + (int64_t)(ref & WREF_SLOT_MASK) * ((int64_t)L * d);
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
                gram_rows<RLPP, RVPL, W, L, FULL, true, LR>(rowp, rs, d, valid, idok, do_scale, out, rgl, stage);
            else
                gram_rows<RLPP, RVPL, W, L, FULL, false, LR>(rowp, rs, d, valid, idok, do_scale, out, rgl, stage);
        }
        BEND();
        return;
    }
    const WinDesc wd = P.wdesc[w];
    unsigned long long* dbg = (st.dbg_layer == layer) ? P.base.dbg : nullptr;   // stamps of diagnostic builds (-DTPNET_STAMPS)
    (void)dbg;
    if (sidx < L) {
#ifdef TPNET_DEV
        if (st.dbg_skip & 1) return;
#endif
        const Chain* __restrict__ heavy = P.chains + wd.start;         // the window's longest chains
        const uint32_t units = hub_chains(st, wd) * (uint32_t)st.CP;
        for (uint32_t h = rb; h < units; h += nblk)
            chain_heavy<LPH>(S, P, heavy[h / (uint32_t)st.CP], (int)(h % (uint32_t)st.CP), layer, lambda,
                             (uint32_t)(w * P.K), lds_u, dbg);
        BEND();
        return;
    }
    if (sidx < SL) {
        // ---- medium chains of one layer: a workgroup each
        if constexpr (FULL) {
            const uint32_t n_hub = hub_chains(st, wd);
            const uint32_t n_big = big_chains(st, wd, n_hub);
            const Chain* __restrict__ list = P.chains + wd.start;
            for (uint32_t m = n_hub + rb; m < n_big; m += nblk) chain_medium<LPP, VPL, LWORDS>(S, P, list[m], layer, lds_u);
        }
        BEND();
        return;
    }
#ifdef TPNET_DEV
    if (st.dbg_skip & 2) return;
#endif
    const uint32_t n_hub = big_chains(st, wd, hub_chains(st, wd));
    const Chain* __restrict__ small = P.chains + wd.start + n_hub;        // the others, longest first: a block's chains are alike
    const int64_t n_small = (int64_t)wd.n_chains - (int64_t)n_hub;
    for (int64_t base = (int64_t)rb * GPB; base < n_small; base += (int64_t)nblk * GPB) {
        const int64_t idx = base + g;
        const bool valid = idx < n_small;
        const Chain c = small[valid ? idx : 0];
        chain_light<LPP, VPL, FULL>(S, P, c, valid, layer, lambda, gl, dbg);
    }
    BEND();
}

// ---------------------------------------------------------------------------------------------------------------
// end of a chunk: the last version of every touched node -> the other copy of its table bundle, meta published
// ---------------------------------------------------------------------------------------------------------------
// (a workgroup scans WBS sorted positions, lists the last runs it finds (~WBS/6) in LDS and copies them with four loads per
// thread in flight at once: one load per iteration was a loop of ~15 dependent round trips)
static constexpr int WBS = 256;
__global__ __launch_bounds__(WB) void k_wwriteback(tpnet_state S, WPlan P, int64_t nc, uint32_t bid) {
    const int per = S.L * S.d;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    __shared__ uint32_t n_list;
    __shared__ uint32_t l_pos[WBS];
    __shared__ int32_t l_node[WBS];
    __shared__ int32_t l_copy[WBS];
    for (int64_t j0 = (int64_t)blockIdx.x * WBS; j0 < nc; j0 += (int64_t)gridDim.x * WBS) {
        __syncthreads();
        if (threadIdx.x == 0) n_list = 0;
        __syncthreads();
        const int64_t j = j0 + threadIdx.x;
        if (threadIdx.x < WBS && j < nc && (P.s_ref[j] & WREF_LAST_RUN)) {
            const int32_t u = P.base.s_target[j];
            const int c = (int)(meta[u].ver & 1u);
            const uint32_t k = atomicAdd(&n_list, 1u);
            l_pos[k] = (uint32_t)j;
            l_node[k] = u;
            l_copy[k] = c;
        }
        __syncthreads();
        const uint32_t n = n_list;
        if ((per & 3) == 0) {
            const uint32_t pv = (uint32_t)per / 4;
            const uint32_t tot = n * pv;
            for (uint32_t x0 = threadIdx.x; x0 < tot; x0 += 4 * WB) {           // four loads in flight per thread
                const uint32_t xa = x0, xb = x0 + WB, xc = x0 + 2 * WB, xd = x0 + 3 * WB;
                const uint32_t ka = xa / pv, kb = (xb < tot ? xb : xa) / pv, kc = (xc < tot ? xc : xa) / pv, kd = (xd < tot ? xd : xa) / pv;
                const uint32_t oa = xa - ka * pv, ob = (xb < tot ? xb : xa) - kb * pv, oc = (xc < tot ? xc : xa) - kc * pv,
                               od = (xd < tot ? xd : xa) - kd * pv;
                const float4 va = reinterpret_cast<const float4*>(P.log + (int64_t)l_pos[ka] * per)[oa];
                const float4 vb = reinterpret_cast<const float4*>(P.log + (int64_t)l_pos[kb] * per)[ob];
                const float4 vc = reinterpret_cast<const float4*>(P.log + (int64_t)l_pos[kc] * per)[oc];
                const float4 vd = reinterpret_cast<const float4*>(P.log + (int64_t)l_pos[kd] * per)[od];
                reinterpret_cast<float4*>(S.q + ((int64_t)(l_copy[ka] ^ 1) * S.N + l_node[ka]) * (int64_t)per)[oa] = va;
                if (xb < tot) reinterpret_cast<float4*>(S.q + ((int64_t)(l_copy[kb] ^ 1) * S.N + l_node[kb]) * (int64_t)per)[ob] = vb;
                if (xc < tot) reinterpret_cast<float4*>(S.q + ((int64_t)(l_copy[kc] ^ 1) * S.N + l_node[kc]) * (int64_t)per)[oc] = vc;
                if (xd < tot) reinterpret_cast<float4*>(S.q + ((int64_t)(l_copy[kd] ^ 1) * S.N + l_node[kd]) * (int64_t)per)[od] = vd;
            }
        } else {
            for (uint32_t x = threadIdx.x; x < n * (uint32_t)per; x += WB) {
                const uint32_t k = x / (uint32_t)per, o = x - k * (uint32_t)per;
                S.q[((int64_t)(l_copy[k] ^ 1) * S.N + l_node[k]) * (int64_t)per + o] = P.log[(int64_t)l_pos[k] * per + o];
            }
        }
        if (threadIdx.x < n) {
            const uint32_t k = threadIdx.x;
            publish_meta(meta + l_node[k], l_copy[k] ^ 1, P.base.desc[P.s_bc[l_pos[k]]].t_last, bid);
        }
    }
}

int launch_wwriteback(const tpnet_state& st, const WPlan& p, int64_t Ec, uint32_t launch_id, hipStream_t s) {
    const int64_t nc = 2 * Ec;
    int64_t grid = (nc + WBS - 1) / WBS;
    if (grid > 16384) grid = 16384;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_wwriteback, dim3((unsigned)grid), dim3(WB), 0, s, st, p, nc, launch_id);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_wstep(const tpnet_state& st, const StreamArgs& a, const WPlan& p, int64_t j, int64_t Ec, int64_t batch,
                 double lambda, uint32_t flags, hipStream_t s) {
    if ((reinterpret_cast<uintptr_t>(a.out_pos) | reinterpret_cast<uintptr_t>(a.out_neg)) & 15) return TPNET_ERR_BAD_ARG;
    if (st.d % 4 != 0) return TPNET_ERR_BAD_ARG;
    const int L = st.L;
    const int64_t nw = (Ec + p.Ew - 1) / p.Ew;
    static const int hb_env = TPNET_DEV_INT(WIN_HB, 0);
    static const int stamp_layer = TPNET_DEV_INT(STAMP_LAYER, 0);
    static const int geo_env = TPNET_DEV_INT(WIN_GEOM, 0);
    const Geom geo = pick_geom(st.d, geo_env == 1);           // (TPNET_DISPATCH_G declares its own `gm`)
    const int lph = heavy_lph(geo.lpp, geo.vpl);
    const int GPB = WB / geo.lpp;
    // (the readout of rows of 17..32 vectors on 16 lanes x 2 vectors: four pairs per wave, a reduction over 16 lanes.  No gain
    // with 16 K-edge windows; with 24 K-edge windows and 256 hub workgroups C2 goes 3.40 -> 3.18 us per batch: default on)
    static const int rs_env = TPNET_DEV_INT(WIN_RSPLIT, 1);
    const bool rsplit = rs_env == 1 && geo.lpp == 32 && geo.vpl == 1 && st.d / 4 > 16;   // rows of 17..32 vectors
    WStep ws;
    ws.CP = (st.d / 4 + lph - 1) / lph;
    static const int stamp_step = TPNET_DEV_INT(STAMP_STEP, -1);
    ws.dbg_layer = (stamp_layer == -1 && stamp_step >= 0 && j != stamp_step) ? 0 : stamp_layer;
    static const int skip_env = TPNET_DEV_INT(WIN_SKIP, 0);   // timing experiments only
    ws.dbg_skip = skip_env;
    const bool have_readout = a.out_pos || a.out_neg;
    auto win_edges = [&](int64_t w) { const int64_t e0 = w * p.Ew; return (Ec - e0 < p.Ew) ? Ec - e0 : p.Ew; };
    uint32_t nb = 0;
    for (int i = 0; i < TPNET_MAX_LAYERS; ++i) ws.w_upd[i] = -1;
    // A step that carries ONE update layer (the pipeline's first step, and its last one beside the readouts) is bound by its longest single
    // units, not by throughput -- a lane group walks a chain of 64..96 contributions in ~35 us, the step's other work is done in
    // ~12 (block stamps, tools/wblocks.py) -- and has workgroups to spare: it walks the chains of length classes 1..2 by
    // workgroups as well, on the hub workgroups the absent layers leave free.
    static const int ext_env = TPNET_DEV_INT(WIN_HUB_EXT, 2);
    int n_upd = 0;
    for (int i = 0; i < L; ++i) n_upd += (j - i >= 0 && j - i < nw) ? 1 : 0;
    const bool reads = have_readout && j - L >= 0 && j - L < nw;
    // (measured, C2: the first step 38 -> 30 us at 24 batches per window, 28 -> 19 at 15, 21 -> 18 at 10; the last update step beside
    // the readouts 44 -> 41, 31 -> 24, but 24 -> 27 at 10 batches per window; two update layers sharing the step: 41 -> 49, not taken)
    const bool med_on = wplan_medium_chains(st.d);             // (the medium chains have workgroups of their own then)
    ws.hub_ext = (!med_on && ext_env >= 1 && n_upd == 1 && (!reads || p.K >= 14)) ? 2u : 0u;
    if (!med_on && ext_env >= 2 && n_upd == 2 && !reads && p.K >= 14) ws.hub_ext = 1u;      // (two layers: the longest class only, 42 -> 33 us;
                                                                                 // in the full steps the same costs 55 -> 60 us: not taken)
    // (steps with three update layers, or two beside the readouts: no change with the longest class on workgroups -- measured)
    const uint32_t hub_blocks = ws.hub_ext ? (uint32_t)(hb_env > 0 ? hb_env : 384) * (uint32_t)L / (uint32_t)n_upd
                                           : (uint32_t)(hb_env > 0 ? hb_env : 384);
    ws.hub_blocks = hub_blocks;
    // hub chains first
    for (int i = 0; i < L; ++i) {
        const int64_t w = j - i;
        ws.seg[i] = nb;
        if (w >= 0 && w < nw) {
            ws.w_upd[i] = w;
            nb += hub_blocks;   // (four-law sweeps, tools/degree_sensitivity.py: 128 -> 256 -> 384, each step -5..-7 % on streams with heavy hubs, nothing lost elsewhere)
        }
    }
    // then the medium chains: a workgroup each (a window of 10 C2 batches has ~110 of them, one of 24 batches ~400)
    static const int mb_env = TPNET_DEV_INT(WIN_MB, 0);
    // (a step that carries every role is bound by throughput, not by its longest units: there a workgroup per chain of 16..31
    // contributions holds a slot of the CU with most of its lanes idle)
    static const int med_full = TPNET_DEV_INT(WIN_MED_FULL, 2);   // (C2 epoch 513 -> 498 us, 400 batches 1 203 -> 1 129 us against level 3; levels 0 / 1: 525 / 1 226)
    static const int med_part = TPNET_DEV_INT(WIN_MED_PART, 3);
    ws.med = wplan_medium_chains(st.d) ? (uint32_t)((n_upd == L && reads) ? med_full : med_part) : 0u;
    for (int i = 0; i < L; ++i) {
        ws.seg[L + i] = nb;
        if (ws.med && ws.w_upd[i] >= 0) nb += (uint32_t)(mb_env > 0 ? mb_env : 384);
    }
    ws.seg[2 * L] = nb;
    {
        uint32_t nact = 0, per = 0;
        for (int i = 0; i < L; ++i) {
            if (ws.w_upd[i] >= 0) {
                ++nact;
                // (chains <= contributions, typically a third of them; the lists are sorted longest first, so a block that
                // walks a second round of chains gets short ones)
                static const int lb_env = TPNET_DEV_INT(WIN_LB, 0);
                const uint32_t g_ = (uint32_t)grid_for(2 * win_edges(ws.w_upd[i]) / 3 + 1, GPB, lb_env > 0 ? lb_env : 768);
                per = g_ > per ? g_ : per;
            }
        }
        ws.n_active = nact ? nact : 1;
        nb += nact * per;
    }
    ws.seg[2 * L + 1] = nb;
    ws.w_read = -1;
    {
        const int64_t w = j - L;
        if (have_readout && w >= 0 && w < nw) {
            ws.w_read = w;
            const int64_t npairs = (a.out_pos ? win_edges(w) : 0) + (a.out_neg ? win_edges(w) : 0);
            nb += (uint32_t)grid_for(npairs, rsplit ? WB / 16 : GPB, 256 * 8);
        }
    }
    ws.seg[2 * L + 2] = nb;
    if (nb == 0) return TPNET_OK;
    const bool shard = a.own_mod == 0;
    TPNET_DISPATCH_G(geo, ({
        if constexpr (W == 4) {
            if constexpr (LPP == 32 && VPL == 1) {
                if (rsplit && shard)
                    hipLaunchKernelGGL((k_wpipe<LPP, VPL, L, FULL, true, true>), dim3(nb), dim3(WB), 0, s, st, p, a, ws, Ec, batch, lambda, flags);
                else if (rsplit)
                    hipLaunchKernelGGL((k_wpipe<LPP, VPL, L, FULL, true>), dim3(nb), dim3(WB), 0, s, st, p, a, ws, Ec, batch, lambda, flags);
                else if (shard)
                    hipLaunchKernelGGL((k_wpipe<LPP, VPL, L, FULL, false, true>), dim3(nb), dim3(WB), 0, s, st, p, a, ws, Ec, batch, lambda, flags);
                else
                    hipLaunchKernelGGL((k_wpipe<LPP, VPL, L, FULL, false>), dim3(nb), dim3(WB), 0, s, st, p, a, ws, Ec, batch, lambda, flags);
            } else {
                if (shard)
                    hipLaunchKernelGGL((k_wpipe<LPP, VPL, L, FULL, false, true>), dim3(nb), dim3(WB), 0, s, st, p, a, ws, Ec, batch, lambda, flags);
                else
                    hipLaunchKernelGGL((k_wpipe<LPP, VPL, L, FULL, false>), dim3(nb), dim3(WB), 0, s, st, p, a, ws, Ec, batch, lambda, flags);
            }
        } else {
            return TPNET_ERR_BAD_ARG;
        }
    }));
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

}  // namespace tpnet
