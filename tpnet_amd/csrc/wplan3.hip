// Plan of a windowed chunk in THREE kernels (tpnet_common.h: WPlan; kernels that consume it: wstep.hip).
//
// The chunk planner of plan.hip sorts all contributions of a chunk by (node, batch) with two device-wide radix sorts:
// ~20 launches, ~200 us before the first pipeline step of a 20-batch call can start.  What the pipeline kernels need is
// weaker: the contributions of one (node, window) CHAIN contiguous, its (node, batch) runs in batch order and each run in
// the reference's summation order, and every read "row of node n before batch b" resolved to a log slot or the frozen
// table.  For batches that fit one workgroup's LDS that takes no device-wide sort, and no search:
//   A  k_wsort    one workgroup per batch sorts the batch's 2B contributions by target node in LDS (stable: src-side
//                 scatter-adds first, then dst-side, models/TPNet.py:93-96) and enters every (node, batch) run into a hash
//                 table H1: (node, batch) -> (first batch-sorted position, length);
//   B  k_wchains  one thread per contribution probes H1 for its node in the K batches of its window (independent loads: one
//                 round trip, not a search): contributions of the node before / in / after its batch.  The first contribution
//                 of a node in a window claims the chain's range of positions with ONE atomic add on the window's cursor and
//                 enters it into H2: (node, window) -> (first position, length, last batch);
//   C  k_wrefs    places every contribution (chain base + contributions before its run + rank in the run) and resolves every
//                 version reference: inside the window from H2 + B's counts, across windows through a per-node bit mask of
//                 the windows the node is a target in + one probe of H2, else the table's meta record; chain records go to
//                 per-window lists bucketed by length; readout references of src / dst / neg.
// (+ one memset of the tables.)  Every kernel is a handful of dependent memory round trips deep, whatever the chunk's length.
// Same WPlan contents as wplan_build up to WHERE a chain sits inside its window's range of positions (claimed in arrival
// order here, node order there) and the order of the chains inside a length class -- the arithmetic of every run is
// identical, so both planners give bit-identical results.
#include "wplan_common.hpp"

namespace tpnet {

size_t wplan3_table_bytes(int64_t Ec, int64_t batch) {
    const size_t nb = (size_t)((Ec + batch - 1) / batch);
    return wplan3_h2_bytes(Ec) + 256 + nb * wplan3_seg(batch) * sizeof(HEnt) + 256;
}
size_t wplan3_blk_bytes(int64_t Ec, int64_t batch) {
    const size_t nblk = (size_t)((Ec + batch - 1) / batch) * wplan3_bpb(batch);
    return 2 * nblk * 9 * sizeof(uint32_t) + 256;
}

// ---------------------------------------------------------------------------------------------------------------
// A: per-batch sort.  One workgroup per batch (wplan_common.hpp: wsort_batch).
// ---------------------------------------------------------------------------------------------------------------
template <int BS, int IPT>
__global__ __launch_bounds__(BS) void k_wsort(WTmp q, BatchDesc* __restrict__ desc, const int64_t* __restrict__ src_c,
                                              const int64_t* __restrict__ dst_c, const double* __restrict__ t_c, int64_t Ec,
                                              int32_t Bfull, int64_t N, int node_bits, double now_time,
                                              const double* __restrict__ t_prev, double lambda, int L, uint32_t* err, int KW) {
    __shared__ WSortShared<BS, IPT> sh;
    wsort_batch<BS, IPT, false>(sh, q, desc, src_c, dst_c, t_c, Ec, Bfull, N, node_bits, now_time, t_prev, lambda, L, err, KW,
                                (int64_t)blockIdx.x, nullptr);
}


// ---------------------------------------------------------------------------------------------------------------
// B: chains.  One thread per batch-sorted contribution (a workgroup stays inside ONE batch, hence one window): the runs of
// its node in the batches of its window.  What the chain heads of a workgroup claim -- positions of their window, places in
// its per-class chain lists -- is summed in LDS and claimed with ONE global atomic per workgroup and counter: a counter per
// window is a single hot address for tens of thousands of chain heads otherwise (measured: 160 us instead of 10).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wchains(WPlan p, WTmp q, int64_t Ec, int64_t Bfull, int64_t nb, uint32_t bpb) {
    constexpr int G = 8;
    const int KW = p.K;
    const int64_t b = blockIdx.x / bpb;
    const int64_t e0b = b * Bfull;
    const uint32_t ne2 = 2u * (uint32_t)((Ec - e0b < Bfull) ? (Ec - e0b) : Bfull);
    const uint32_t jl = (blockIdx.x % bpb) * 256u + threadIdx.x;
    const bool active = jl < ne2;
    const int64_t x = 2 * e0b + jl;
    const int64_t w = b / KW;
    const int64_t bw0 = w * KW;
    __shared__ uint32_t l_cls[WCLS], l_pos;
    if (threadIdx.x < WCLS) l_cls[threadIdx.x] = 0;
    if (threadIdx.x == WCLS) l_pos = 0;
    __syncthreads();
    // a CHAIN HEAD -- the head of its node's first run in the window -- walks the node's runs of the window (the batches A marked
    // in the chain table; the probes of up to G runs are in flight together) and leaves with every run's head what the run's
    // contributions need: the chain's contributions before the run, the batch of the previous run.  (Every contribution used to
    // probe for all earlier runs of its node itself: 2.1 M probes per epoch of C2 against 0.2 M.)
    const bool run_head = active && q.bri[x] == 0u;
    const uint32_t node = run_head ? q.bkey[x] : 0u;
    CEnt* ce = run_head ? cfind(q, node, (uint32_t)w) : nullptr;
    const unsigned kb = (unsigned)(b - bw0);
    const unsigned long long mask = ce ? ce->mask : 0ull;
    const bool chain_head = run_head && (mask & ((1ull << kb) - 1ull)) == 0ull;
    uint32_t same_total = 0;
    if (chain_head) {
        unsigned long long todo = mask;
        uint32_t prefix = 0, last_tail = 0;
        int64_t prev_b = -1;
        while (todo) {
            uint32_t st[G], len[G];
            unsigned long long keys[G];
            bool valid[G];
            unsigned bit[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                valid[g] = todo != 0ull;
                bit[g] = valid[g] ? (unsigned)(__ffsll((long long)todo) - 1) : 0u;
                if (valid[g]) todo &= todo - 1ull;
                keys[g] = hkey(node, (uint32_t)(bw0 + bit[g]));
            }
            hfind_multi<G>(q.h1, q.seg, keys, valid, st, len);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (valid[g]) {
                    q.bsb[st[g]] = prefix;
                    if (prev_b >= 0) q.rprevb[st[g]] = (uint32_t)prev_b;
                    prefix += len[g];
                    prev_b = bw0 + bit[g];
                    last_tail = st[g] + len[g] - 1u;
                }
            }
        }
        same_total = prefix;
        q.bflags[last_tail] |= WT_WIN_LAST;              // (only this thread touches the chain's flags in this kernel)
    }
    const int cls = wchain_class(same_total, p.heavy_thr);
    uint32_t off = 0, rank = 0;
    if (chain_head) {
        off = atomicAdd(&l_pos, same_total);
        rank = atomicAdd(&l_cls[cls], 1u);
    }
    __syncthreads();
    if (threadIdx.x <= WCLS) q.blkcnt[(size_t)blockIdx.x * 9 + threadIdx.x] = threadIdx.x < WCLS ? l_cls[threadIdx.x] : l_pos;
    if (chain_head) {
        // the chain's range of positions starts at (window) + (earlier workgroups of the window: k_wscan) + off
        q.bflags[x] |= WT_CHAIN_HEAD;
        ce->xh = (uint32_t)x;
        ce->cnt = same_total;
        atomicOr(p.wmask + node, 1ull << (unsigned)w);
        q.boff[x] = off;
        q.rprevb[x] = rank;                              // (a chain head has no previous run in the window: the slot is free)
    }
}

// ---------------------------------------------------------------------------------------------------------------
// between B and C: a workgroup per window sums what the workgroups of k_wchains counted -- in workgroup order -- into every
// workgroup's bases and the window's totals (k_wrefs turns those into the lists' starts and the window descriptors).
// (No contended atomic anywhere: ten thousand chain heads adding to one counter per window cost 25..160 us, measured.)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(9 * 64) void k_wscan(WPlan p, WTmp q, int64_t nb) {
    const int KW = p.K;
    const int64_t w = blockIdx.x;
    const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;       // wave c sums counter c of the window, 64 workgroups a step
    const uint32_t* __restrict__ cnt = q.blkcnt;
    uint32_t* __restrict__ base = q.blkbase;
    const int64_t b0 = w * KW, b1 = (b0 + KW < nb) ? b0 + KW : nb;
    uint32_t run = 0;
    for (int64_t k0 = b0 * q.bpb; k0 < b1 * q.bpb; k0 += 64) {
        const int64_t k = k0 + lane;
        const uint32_t v = (k < b1 * q.bpb) ? cnt[k * 9 + c] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)inc, o, 64);
            if (lane >= o) inc += u;
        }
        if (k < b1 * q.bpb) base[k * 9 + c] = run + inc - v;
        run += (uint32_t)__shfl((int)inc, 63, 64);
    }
    if (lane == 0) q.wtot[w * 9 + c] = run;
}

// ---------------------------------------------------------------------------------------------------------------
struct VRef { uint32_t ref; double t_src; };

// the chain of `node` in window w (it has one): first position, contributions, last batch
__device__ __forceinline__ void chain_of(const WPlan& p, const WTmp& q, int64_t Bfull, const CEnt* ce, int64_t w, uint32_t& base,
                                         uint32_t& cnt, int64_t& last_b) {
    const uint4 v = *reinterpret_cast<const uint4*>(&ce->mask);          // mask (x, y), xh (z), cnt (w)
    const uint32_t xh = v.z;
    const unsigned long long mask = ((unsigned long long)v.y << 32) | v.x;
    const int64_t bh = (int64_t)xh / (2 * Bfull);
    const int64_t blk = bh * q.bpb + ((int64_t)xh - 2 * bh * Bfull) / 256;
    base = (uint32_t)(2 * w * p.Ew) + q.blkbase[blk * 9 + 8] + q.boff[xh];
    cnt = v.w;
    last_b = w * p.K + (63 - __clzll((long long)mask));
}
__device__ __forceinline__ void chain_of(const WPlan& p, const WTmp& q, int64_t Bfull, uint32_t node, int64_t w, uint32_t& base,
                                         uint32_t& cnt, int64_t& last_b) {
    chain_of(p, q, Bfull, cfind(q, node, (uint32_t)w), w, base, cnt, last_b);
}

__device__ __forceinline__ VRef version_before_window_m(const WPlan& p, const WTmp& q, int64_t Bfull, int64_t node, int64_t w,
                                                        unsigned long long wmask, const NodeMeta* __restrict__ meta) {
    VRef r;
    const unsigned long long m = w > 0 ? (wmask & ((1ull << (unsigned)w) - 1ull)) : 0ull;
    if (m) {
        const int64_t wp = 63 - __clzll((long long)m);
        uint32_t base, cnt;
        int64_t lb;
        chain_of(p, q, Bfull, (uint32_t)node, wp, base, cnt, lb);
        r.ref = base + cnt - 1u;
        r.t_src = p.base.desc[lb].t_last;
        return r;
    }
    const uint4 m0 = reinterpret_cast<const uint4*>(meta + node)[0];   // {ver, pad, tref[0]}
    const uint4 m1 = reinterpret_cast<const uint4*>(meta + node)[1];   // {tref[1], pad}
    const uint32_t c = m0.x & 1u;
    r.ref = WREF_TABLE | c;
    r.t_src = c ? __hiloint2double((int)m1.y, (int)m1.x) : __hiloint2double((int)m0.w, (int)m0.z);
    return r;
}

__device__ __forceinline__ VRef version_before_window(const WPlan& p, const WTmp& q, int64_t Bfull, int64_t node, int64_t w,
                                                      const NodeMeta* __restrict__ meta) {
    return version_before_window_m(p, q, Bfull, node, w, w > 0 ? p.wmask[node] : 0ull, meta);
}

// version of `node` before batch b when the node need not be a target in batch b (a negative, a bad edge's stand-in): the
// tail of its latest run in an earlier batch of the window, else version_before_window
__device__ __forceinline__ VRef version_before_batch(const WPlan& p, const WTmp& q, int64_t Bfull, int64_t node, int64_t b,
                                                     const NodeMeta* __restrict__ meta) {
    const int64_t w = b / p.K, bw0 = w * p.K;
    const CEnt* ce = cfind(q, (uint32_t)node, (uint32_t)w);
    if (ce) {
        const unsigned long long before = ce->mask & ((1ull << (unsigned)(b - bw0)) - 1ull);
        if (before) {
            const int64_t bp = bw0 + (63 - __clzll((long long)before));      // the latest earlier batch with a run of the node
            uint32_t s, l;
            hfind(q.h1, q.seg, hkey((uint32_t)node, (uint32_t)bp), s, l);
            // the run's tail sits at chain base + contributions before the run + (its length - 1)
            uint32_t base, cnt;
            int64_t lb;
            chain_of(p, q, Bfull, ce, w, base, cnt, lb);
            VRef r;
            r.ref = base + q.bsb[s] + l - 1u;
            r.t_src = p.base.desc[bp].t_last;
            return r;
        }
    }
    return version_before_window(p, q, Bfull, node, w, meta);
}

// ---------------------------------------------------------------------------------------------------------------
// C: positions and references.  Blocks [0, fgrid): one thread per EDGE -- the versions of its two endpoints before its batch
// serve four consumers: the readout of (src, dst), and the partner rows of the edge's two contributions (each endpoint is the
// other one's partner); the thread also places both contributions and, where one heads a chain, writes the chain record.
// The rest of the grid: the readout references of the negatives (all a replayed plan has to form again).
// ---------------------------------------------------------------------------------------------------------------
struct OwnSide {           // one endpoint of an edge as a target of its batch
    int64_t x;             // batch-sorted position of its contribution
    uint32_t node, sb, ri, fl, aux;
    int32_t partner;
    float coef;
    uint32_t base, cnt;    // its chain in the window
    unsigned long long wm; // the windows the node is a target in
    VRef ver;              // its version before the batch
};

__device__ __forceinline__ void side_load(const WTmp& q, int64_t x, OwnSide& o) {
    o.x = x;
    o.node = q.bkey[x];
    o.partner = q.bpart[x];
    o.ri = q.bri[x];
    o.fl = q.bflags[x];
    o.coef = q.bcoef[x];
    o.sb = q.bsb[x - o.ri];                               // (per run, kept with the run's head by k_wchains)
    o.aux = q.rprevb[x - o.ri];
}

__device__ __forceinline__ void side_version(const WPlan& p, const WTmp& q, int64_t Bfull, int64_t w, const CEnt* ce,
                                             const NodeMeta* __restrict__ meta, OwnSide& o) {
    int64_t lb;
    chain_of(p, q, Bfull, ce, w, o.base, o.cnt, lb);
    if (o.sb) {                                               // the node's previous run ends right in front of this run's head
        o.ver.ref = o.base + o.sb - 1u;
        o.ver.t_src = p.base.desc[o.aux].t_last;
    } else {
        o.ver = version_before_window_m(p, q, Bfull, o.node, w, o.wm, meta);
    }
}

__device__ __forceinline__ void side_store(const WPlan& p, const WTmp& q, int64_t Bfull, int64_t Ec, int64_t e, int sd, int64_t b,
                                           int64_t w, int64_t e0b, double Tb, double Tnow, double lambda,
                                           const NodeMeta* __restrict__ meta, const uint32_t* cabs, int want_edges,
                                           const OwnSide me, const OwnSide other) {
    const uint32_t f = me.base + me.sb + me.ri;
    p.base.s_partner[f] = me.partner;
    p.base.s_coef[f] = me.coef;
    p.base.s_target[f] = (int32_t)me.node;
    p.s_bc[f] = (uint16_t)b;
    // the partner is the edge's other endpoint, a target in this batch too: the row it reads is that one's version
    VRef pr = other.ver;
    if ((uint32_t)me.partner != other.node)                   // (a bad edge: weight 0 whatever the row)
        pr = version_before_batch(p, q, Bfull, (int64_t)me.partner, b, meta);
    uint32_t out = (me.fl & WT_STRUCT_MASK) | pr.ref;
    float dec = 1.0f;
    if (me.fl & WREF_RUN_HEAD) {
        if (me.fl & WT_CHAIN_HEAD) {
            float g_first = 1.0f;
            if (me.ver.ref & WREF_TABLE) g_first = decay3_f32(lambda, Tb - me.ver.t_src);
            else dec = decay3_f32(lambda, Tb - me.ver.t_src);        // (a log row's decay comes with the run: s_dec)
            // its place in the window's list of its length class: (class start) + (earlier workgroups) + (rank in its own)
            const int c = wchain_class(me.cnt, p.heavy_thr);
            const int64_t blk = b * q.bpb + (me.x - 2 * e0b) / 256;
            uint4* rec = reinterpret_cast<uint4*>(p.chains + (cabs[w * WCLS + c] + q.blkbase[blk * 9 + c] + me.aux));
            rec[0] = make_uint4(f, me.cnt, me.node, me.ver.ref);                 // Chain {j0, cnt, target, prev_ref,
            rec[1] = make_uint4(__float_as_uint(g_first), 0u, 0u, 0u);           //        g_first, pad}
        } else {
            dec = decay3_f32(lambda, Tb - me.ver.t_src);
        }
    }
    if ((me.fl & WT_WIN_LAST) && (w >= 63 || (me.wm >> (unsigned)(w + 1)) == 0ull)) out |= WREF_LAST_RUN;
    p.s_ref[f] = out;
    p.s_g[f] = decay3_f32(lambda, Tb - pr.t_src);
    p.s_dec[f] = dec;
    if (want_edges) {
        p.e_ref[(int64_t)sd * Ec + e] = me.ver.ref;
        p.e_g[(int64_t)sd * Ec + e] = decay3_f32(lambda, Tnow - me.ver.t_src);
    }
}

__global__ __launch_bounds__(256) void k_wrefs(WPlan p, WTmp q, const int64_t* __restrict__ neg, int64_t Ec, int64_t Bfull,
                                               int64_t nb, int64_t N, double lambda, const NodeMeta* __restrict__ meta,
                                               uint32_t fgrid, int want_edges) {
    const int KW = p.K;
    const BatchDesc* __restrict__ desc = p.base.desc;
    if (blockIdx.x >= fgrid) {
        for (int64_t e = (int64_t)(blockIdx.x - fgrid) * blockDim.x + threadIdx.x; e < Ec;
             e += (int64_t)(gridDim.x - fgrid) * blockDim.x) {
            int64_t node = neg[e];
            if ((uint64_t)node >= (uint64_t)N) node = 0;               // the readout reports the bad id itself
            const int64_t b = e / Bfull;
            const VRef r = version_before_batch(p, q, Bfull, node, b, meta);
            p.e_ref[2 * Ec + e] = r.ref;
            p.e_g[2 * Ec + e] = decay3_f32(lambda, desc[b].now - r.t_src);
        }
        return;
    }
    // every (window, class) list's start in WPlan::chains, from the window totals of k_wscan (<= 64 x 8 values)
    __shared__ uint32_t cabs[WIN_MAX_WINDOWS * WCLS];
    const int64_t nw = (nb + KW - 1) / KW;
    if (threadIdx.x < 64) {
        const int wv = threadIdx.x;
        uint32_t n = 0, t8[WCLS];
#pragma unroll
        for (int k = 0; k < WCLS; ++k) {
            t8[k] = wv < nw ? q.wtot[wv * 9 + k] : 0u;
            n += t8[k];
        }
        uint32_t inc = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inc, o, 64);
            if (wv >= o) inc += v;
        }
        uint32_t at = inc - n;
        if (blockIdx.x == 0 && wv < nw) {
            WinDesc D;
            D.start = at;
            D.n_heavy = t8[0];
            D.n_chains = n;
            D.n_ext = t8[0] + t8[1] + t8[2];
            D.n_ext1 = t8[0] + t8[1];
            D.n_med = 0;
            for (int k = 0; k <= wchain_med_classes(p.heavy_thr); ++k) D.n_med += t8[k];
            D.pad1 = D.pad2 = 0;
            p.wdesc[wv] = D;
        }
#pragma unroll
        for (int k = 0; k < WCLS; ++k) {
            cabs[wv * WCLS + k] = at;
            at += t8[k];
        }
    }
    __syncthreads();
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < Ec; e += (int64_t)fgrid * blockDim.x) {
        const int64_t b = e / Bfull;
        const int64_t w = b / KW;
        const int64_t e0b = b * Bfull;
        const int64_t neb = (Ec - e0b < Bfull) ? (Ec - e0b) : Bfull;
        const double Tb = desc[b].t_last, Tnow = desc[b].now;
        const int64_t x0 = q.binv[2 * e0b + (e - e0b)], x1 = q.binv[2 * e0b + neb + (e - e0b)];
        OwnSide s0, s1;
        side_load(q, x0, s0);
        side_load(q, x1, s1);
        const CEnt* c0 = cfind(q, s0.node, (uint32_t)w);
        const CEnt* c1 = cfind(q, s1.node, (uint32_t)w);
        s0.wm = p.wmask[s0.node];
        s1.wm = p.wmask[s1.node];
        side_version(p, q, Bfull, w, c0, meta, s0);
        side_version(p, q, Bfull, w, c1, meta, s1);
        side_store(p, q, Bfull, Ec, e, 0, b, w, e0b, Tb, Tnow, lambda, meta, cabs, want_edges, s0, s1);
        side_store(p, q, Bfull, Ec, e, 1, b, w, e0b, Tb, Tnow, lambda, meta, cabs, want_edges, s1, s0);
    }
}

// the planner's views of a chunk's workspace, as every stage (and the write-back) sees them
static WTmp wtmp_full(const tpnet_state& st, const WPlan& p, int64_t Ec, int64_t batch) {
    const int64_t nb = (Ec + batch - 1) / batch;
    WTmp q = wtmp_of(p, (size_t)(2 * Ec));
    q.bpb = wplan3_bpb(batch);
    q.seg = wplan3_seg(batch);
    const int64_t nw = (nb + p.K - 1) / p.K;
    q.direct2 = (nw * st.N <= (int64_t)q.hcap) ? 1u : 0u;
    q.n2 = (uint32_t)st.N;
    q.blkcnt = p.wblk;
    q.blkbase = p.wblk + (size_t)nb * q.bpb * 9;
    return q;
}

// ---------------------------------------------------------------------------------------------------------------
// end of a chunk planned here: the last version of every touched node -> the other copy of its table bundle, meta published.
// One lane group per NODE: the windows it is a target in (wmask) -> its chain in the last of them (chain table) -> the chain's last
// position = the tail of its last run = the log slot to copy.  Four dependent round trips and N lane groups, where the scan of
// every sorted position for the last-run flag (wstep.hip: k_wwriteback) is five and 2 E / 256 workgroups.
// ---------------------------------------------------------------------------------------------------------------
static constexpr int WBN_LANES = 32;
__global__ __launch_bounds__(256) void k_wwriteback_nodes(tpnet_state S, WPlan p, WTmp q, int64_t Bfull, uint32_t bid) {
    const int per = S.L * S.d;
    const int pv = per / 4;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    const int gl = threadIdx.x % WBN_LANES, g = threadIdx.x / WBN_LANES;
    constexpr int GPB = 256 / WBN_LANES;
    for (int64_t n = (int64_t)blockIdx.x * GPB + g; n < S.N; n += (int64_t)gridDim.x * GPB) {
        const unsigned long long wm = p.wmask[n];
        if (!wm) continue;
        const int64_t w = 63 - __clzll((long long)wm);
        uint32_t base, cnt;
        int64_t last_b;
        chain_of(p, q, Bfull, (uint32_t)n, w, base, cnt, last_b);
        const uint32_t pos = base + cnt - 1u;
        const int c = (int)(meta[n].ver & 1u);
        const float4* __restrict__ srow = reinterpret_cast<const float4*>(p.log + (int64_t)pos * per);
        float4* __restrict__ drow = reinterpret_cast<float4*>(S.q + ((int64_t)(c ^ 1) * S.N + n) * (int64_t)per);
        for (int x = gl; x < pv; x += WBN_LANES) drow[x] = srow[x];
        if (gl == 0) publish_meta(meta + n, c ^ 1, p.base.desc[last_b].t_last, bid);
    }
}

// (false: the caller takes the position scan -- rows that are not 16-byte vectors, or a table much larger than the chunk)
bool wplan3_writeback(const tpnet_state& st, const WPlan& p, int64_t Ec, int64_t batch, uint32_t launch_id, hipStream_t s) {
    static const int off = TPNET_DEV_INT(NO_WB_NODES, 0);
    if (off || (st.L * st.d) % 4 != 0 || st.N > 8 * Ec) return false;
    const WTmp q = wtmp_full(st, p, Ec, batch);
    int64_t grid = (st.N + 7) / 8;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(k_wwriteback_nodes, dim3((unsigned)grid), dim3(256), 0, s, st, p, q, batch, launch_id);
    return hipGetLastError() == hipSuccess;
}

bool wplan3_applies(const tpnet_state& st, int64_t Ec, int64_t batch, int K) {
    if (K < 1 || batch < 1 || batch > PLAN_ONE_MAX) return false;
    const int64_t nb = (Ec + batch - 1) / batch;
    const int64_t nw = (nb + K - 1) / K;
    int node_bits = 1;
    while (node_bits < 63 && (1ll << node_bits) < st.N) ++node_bits;
    return nw <= WIN_MAX_WINDOWS && nb < 65536 && node_bits <= 31 && (int64_t)K * batch * 2 < (1 << 24) &&
           2 * Ec < (int64_t)WREF_SLOT_MASK;
}

int wplan3_build(const tpnet_state& st, const WPlan& p, const int64_t* src, const int64_t* dst, const int64_t* neg,
                 const double* t, int64_t Ec, int64_t batch, double now_time, const double* t_prev_dev, double lambda,
                 bool want_readout, bool replay, hipStream_t s) {
    if (!wplan3_applies(st, Ec, batch, p.K) || !p.wmask || !p.wcls || !p.wtab || !p.wblk) return TPNET_ERR_BAD_ARG;
    const int64_t nb = (Ec + batch - 1) / batch;
    WTmp q = wtmp_full(st, p, Ec, batch);
    const int64_t nw = (nb + p.K - 1) / p.K;
    int node_bits = 1;
    while (node_bits < 31 && (1ll << node_bits) < st.N) ++node_bits;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(st.meta);
    const int nwhich = want_readout ? (neg ? 3 : 2) : 0;
    const int ngrid = (nwhich == 3) ? ((int)((Ec + 255) / 256) > 2048 ? 2048 : (int)((Ec + 255) / 256)) : 0;
    if (replay) {
        // a replayed plan (same stream, same table state): only the negatives change between epochs
        if (ngrid) {
            hipLaunchKernelGGL(k_wrefs, dim3(ngrid), dim3(256), 0, s, p, q, neg, Ec, batch, nb, st.N, lambda, meta, 0u, 0);
            TPNET_HIP_TRY(hipGetLastError());
        }
        return TPNET_OK;
    }
    // window mask, class counters, window cursors and what the chain table uses (contiguous in the workspace: ONE fill; the
    // runs' table is written whole by k_wsort)
    const size_t h2_used = (q.direct2 ? (size_t)nw * (size_t)st.N : (size_t)q.hcap) * sizeof(CEnt);
    TPNET_HIP_TRY(hipMemsetAsync(p.wmask, 0, p.wzero_bytes + h2_used, s));
    // A
    {
#define TPNET_WSORT(BS_, IPT_)                                                                                              \
    hipLaunchKernelGGL((k_wsort<BS_, IPT_>), dim3((unsigned)nb), dim3(BS_), 0, s, q, p.base.desc, src, dst, t, Ec,           \
                       (int32_t)batch, st.N, node_bits, now_time, t_prev_dev, lambda, (int)st.L, st.err, (int)p.K)
        const int64_t n2 = 2 * batch;
        if (n2 <= 512) TPNET_WSORT(256, 2);
        else if (n2 <= 1024) TPNET_WSORT(512, 2);
        else if (n2 <= 2048) TPNET_WSORT(1024, 2);
        else TPNET_WSORT(1024, 4);
#undef TPNET_WSORT
        TPNET_HIP_TRY(hipGetLastError());
    }
    // B
    hipLaunchKernelGGL(k_wchains, dim3((unsigned)(nb * q.bpb)), dim3(256), 0, s, p, q, Ec, batch, nb, q.bpb);
    hipLaunchKernelGGL(k_wscan, dim3((unsigned)((nb + p.K - 1) / p.K)), dim3(9 * 64), 0, s, p, q, nb);
    TPNET_HIP_TRY(hipGetLastError());
    // C
    int egrid = (int)((Ec + 255) / 256);
    if (egrid > 8192) egrid = 8192;
    hipLaunchKernelGGL(k_wrefs, dim3(egrid + ngrid), dim3(256), 0, s, p, q, neg, Ec, batch, nb, st.N, lambda, meta,
                       (uint32_t)egrid, nwhich >= 2 ? 1 : 0);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

}  // namespace tpnet
