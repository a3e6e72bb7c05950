// Plan of a windowed chunk through a DENSE (batch x node) matrix of run lengths (tpnet_common.h: WPlan; consumers: wstep.hip).
//
// The hashed planner (wplan3.hip) finds a node's runs through two hash tables: a fill and four kernels, each a handful of
// dependent probes deep.  Where the table is small against the stream (N * 12 bytes <= half the version-log bytes of one batch:
// every real dataset of the reference), the runs fit a dense matrix, one row per batch, and everything the pipeline needs
// becomes a direct lookup -- four launches, no fill, no hash table, any number of windows up to 256:
//   k_dense_sort   a workgroup per batch sorts the batch's 2B contributions by target in LDS (wplan_common.hpp: the same sort as
//                  the hashed planner's) and writes the batch's row len[b][n] = contributions to node n in batch b (every
//                  entry: nothing to zero);
//   k_dense_win    a thread per (node, window): the node's contributions in the window and its last run there; its chain in the
//                  window counted and ranked per (chunk of 256 nodes, window, length class);
//   k_dense_pre    a thread per (node, window): m[b][n] = {contributions of n in batches < b, len, batch of n's previous run}
//                  for the window's batches; node totals; chains of every (window, class) in the earlier chunks of nodes;
//   k_dense_place  a thread per contribution: position = first position of n (positions are node-major, a node's contributions
//                  in batch order: the sorted planner's order) + m[b][n].pre + rank in its run; the version of ANY node before
//                  batch b is the position base[v] + m[b][v].pre - 1 (the tail of its previous run) or, when pre = 0, the table
//                  row -- so the partner's version, the node's own previous run, the readout references of src / dst / neg are
//                  one lookup each; chain heads write their chain record into the window's list.
// (Built first as ONE persistent launch with grid barriers between the phases -- measured slower: a barrier with its release /
// acquire of the XCD's L2 cost 4-5 us against ~1.7 us for a kernel boundary, and every phase ran on the sort's few workgroups.)
// Same WPlan contents as the sorted planner up to the order of the chains inside a length class; the arithmetic of every run is
// identical, so all three planners give bit-identical results (tests/test_gpu_parity.py::test_three_launch_planner_equals_chunk_planner).
#include "wplan_dense.hpp"

namespace tpnet {

static constexpr int64_t DENSE_SHARD_MAX_BATCH = 8192;     // a row shard's global batch (wshard.hip): 16 contributions per thread of its sort
bool wplan_dense_eligible(int64_t N, int d, int L, int64_t batch) {
    static const int off = TPNET_DEV_INT(NO_PLAN_DENSE, 0);
    if (off || batch < 1 || batch > DENSE_SHARD_MAX_BATCH || N > (int64_t)DENSE_MAX_CHUNKS * 256) return false;
    return (int64_t)N * 12 <= batch * (int64_t)L * d * 4;     // half the version-log bytes of one batch
}

size_t wplan_dense_bytes(int64_t Ec, int64_t batch, int64_t N, int d, int L) {
    if (!wplan_dense_eligible(N, d, L, batch)) return 0;
    const size_t nb = (size_t)((Ec + batch - 1) / batch);
    const size_t Ns = (size_t)dense_ns(N);
    size_t nw = (nb + 1) / 2 + 1;
    if (nw > (size_t)DENSE_MAX_WINDOWS) nw = DENSE_MAX_WINDOWS;
    const size_t nchunks = (Ns + 255) / 256;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    return al(nb * Ns * 2) + al(nb * Ns * 8) + al(nw * Ns * 4) + al(nw * Ns * 8) + 4 * al(Ns * 4) + al(nchunks * 4) +
           2 * al(nchunks * nw * 8 * 4) + al(nw * 8 * 4) + 256;
}

struct DArgs {
    const int64_t* src;
    const int64_t* dst;
    const int64_t* neg;
    const double* t;
    const double* t_prev;
    const NodeMeta* meta;
    uint32_t* err;             // tpnet_state::err: [0] bad ids
    int64_t Ec, Bfull, N, nb, nw;
    double now_time, lambda;
    int32_t node_bits, L, nwhich /* 0: no readout, 2: src / dst, 3: + neg */, nchunks;
    int64_t own;               // row shard: local ids < own are this rank's rows (= N: everything)
    uint32_t* status;          // row shard: [0] batches whose owned contributions exceeded the sort's capacity
};

// phase C, contributions: batch-sorted position x
__device__ __forceinline__ void dense_place(const WPlan& p, const WTmp& q, const DView& D, const DArgs& a,
                                            const uint32_t* __restrict__ lstart, const uint32_t* __restrict__ cbase,
                                            const double* tl, int64_t x) {
    const BatchDesc* __restrict__ desc = p.base.desc;
    const int KW = p.K;
    const int64_t b = x / (2 * a.Bfull);
    const int64_t e0b = b * a.Bfull;
    const int64_t neb = (a.Ec - e0b < a.Bfull) ? (a.Ec - e0b) : a.Bfull;
    const uint32_t node = q.bkey[x];
    const int32_t partner = q.bpart[x];
    const float coef = q.bcoef[x];
    const uint32_t ri = q.bri[x];
    const uint32_t fl = q.bflags[x] & WT_STRUCT_MASK;
    const uint32_t val = q.bval[x];
    // everything that depends on the ids alone is issued together: one round trip
    const uint2 mo = D.m[b * D.Ns + node];
    const uint2 mp = D.m[b * D.Ns + (uint32_t)partner];
    const uint2 wsl_own = (fl & WREF_RUN_HEAD) ? D.wsl[(b / KW) * D.Ns + node] : make_uint2(0u, 0u);   // (chain heads: the chain's length)
    const uint32_t bo = D.base[node] + cbase[node / (uint32_t)DCH];
    const uint32_t bp = D.base[(uint32_t)partner] + cbase[(uint32_t)partner / (uint32_t)DCH];
    const uint4 o0 = reinterpret_cast<const uint4*>(a.meta + node)[0], o1 = reinterpret_cast<const uint4*>(a.meta + node)[1];
    const uint4 p0 = reinterpret_cast<const uint4*>(a.meta + partner)[0], p1 = reinterpret_cast<const uint4*>(a.meta + partner)[1];
    const double Tb = desc[b].t_last, Tnow = desc[b].now;
    const DVRef own = dense_version(desc, tl, mo, bo, o0, o1);
    const DVRef pr = dense_version(desc, tl, mp, bp, p0, p1);
    const uint32_t f = bo + mo.x + ri;
    p.base.s_partner[f] = partner;
    p.base.s_coef[f] = coef;
    p.base.s_target[f] = (int32_t)node;
    p.s_bc[f] = (uint16_t)b;
    uint32_t out = fl | pr.ref;
    float dec = 1.0f;
    if (fl & WREF_RUN_HEAD) {
        const int64_t w = b / KW;
        const uint32_t prevb = mo.y >> 16;
        const bool chain_head = mo.x == 0u || (int64_t)prevb < w * KW;
        if (chain_head) {
            float g_first = 1.0f;
            if (own.ref & WREF_TABLE) g_first = decay3_f32(a.lambda, Tb - own.t_src);
            else dec = decay3_f32(a.lambda, Tb - own.t_src);    // (a log row's decay comes with the run: s_dec)
            const uint32_t cnt = wsl_own.x;                      // the node's contributions in the window (k_dense_win)
            const int c = wchain_class(cnt, p.heavy_thr);
            const uint32_t pair = (uint32_t)w * 8u + (uint32_t)c;
            const uint32_t chunk = node / (uint32_t)DCH;
            const uint32_t at = lstart[pair] + D.cpre[(size_t)chunk * (size_t)(a.nw * 8) + pair] + D.crank[w * D.Ns + node];
            uint4* rec = reinterpret_cast<uint4*>(p.chains + at);
            rec[0] = make_uint4(f, cnt, node, own.ref);                          // Chain {j0, cnt, target, prev_ref,
            rec[1] = make_uint4(__float_as_uint(g_first), 0u, 0u, 0u);           //        g_first, pad}
        } else {
            dec = decay3_f32(a.lambda, Tb - own.t_src);
        }
    }
    if ((fl & WREF_RUN_TAIL) && D.lastb[node] == (uint32_t)b) out |= WREF_LAST_RUN;
    p.s_ref[f] = out;
    p.s_g[f] = decay3_f32(a.lambda, Tb - pr.t_src);
    p.s_dec[f] = dec;
    if (a.nwhich >= 2) {
        const bool side = val >= (uint32_t)neb;
        const int64_t e = e0b + (side ? (int64_t)val - neb : (int64_t)val);
        p.e_ref[(side ? a.Ec : 0) + e] = own.ref;
        p.e_g[(side ? a.Ec : 0) + e] = decay3_f32(a.lambda, Tnow - own.t_src);
    }
}

// phase C, negatives: the version of neg[e] before the batch of edge e
__device__ __forceinline__ void dense_neg(const WPlan& p, const DView& D, const DArgs& a, const uint32_t* __restrict__ basef,
                                          int64_t e) {
    const BatchDesc* __restrict__ desc = p.base.desc;
    int64_t node = a.neg[e];
    if ((uint64_t)node >= (uint64_t)a.N) node = 0;               // the readout reports the bad id itself
    const int64_t b = e / a.Bfull;
    const uint2 m = D.m[b * D.Ns + node];
    const uint32_t bs = basef[node];
    const uint4 m0 = reinterpret_cast<const uint4*>(a.meta + node)[0], m1 = reinterpret_cast<const uint4*>(a.meta + node)[1];
    const DVRef r = dense_version(desc, nullptr, m, bs, m0, m1);
    p.e_ref[2 * a.Ec + e] = r.ref;
    p.e_g[2 * a.Ec + e] = decay3_f32(a.lambda, desc[b].now - r.t_src);
}

// the same inside phase C: the node's first position = its chunk's base (LDS) + its place in the chunk
// (which = 2: the negative of edge e; which = 1, row shards only: its dst when another rank owns it -- no contribution of this rank
// has that node as its target, so dense_place left the reference of the (src, dst) readout unwritten)
__device__ __forceinline__ void dense_neg_c(const WPlan& p, const DView& D, const DArgs& a, const uint32_t* __restrict__ cbase,
                                            const double* tl, int64_t e, int which = 2) {
    const BatchDesc* __restrict__ desc = p.base.desc;
    int64_t node = which == 2 ? a.neg[e] : a.dst[e];
    if ((uint64_t)node >= (uint64_t)a.N) node = 0;
    if (which == 1 && node < a.own) return;
    const int64_t b = e / a.Bfull;
    const uint2 m = D.m[b * D.Ns + node];
    const uint32_t bs = D.base[node] + cbase[(uint32_t)node / (uint32_t)DCH];
    const uint4 m0 = reinterpret_cast<const uint4*>(a.meta + node)[0], m1 = reinterpret_cast<const uint4*>(a.meta + node)[1];
    const DVRef r = dense_version(desc, tl, m, bs, m0, m1);
    p.e_ref[(int64_t)which * a.Ec + e] = r.ref;
    p.e_g[(int64_t)which * a.Ec + e] = decay3_f32(a.lambda, desc[b].now - r.t_src);
}

// ---- A: a workgroup per batch (wplan_common.hpp: wsort_batch): the batch's contributions sorted by target, its row of len
template <int BS, int IPT>
__global__ __launch_bounds__(BS) void k_dense_sort(WPlan p, WTmp q, DView D, DArgs a) {
    __shared__ WSortShared<BS, IPT> sh;
    const int64_t bb = blockIdx.x;
    wsort_batch<BS, IPT, true>(sh, q, p.base.desc, a.src, a.dst, a.t, a.Ec, (int32_t)a.Bfull, a.N, a.node_bits, a.now_time,
                               a.t_prev, a.lambda, a.L, a.err, p.K, bb, D.len + bb * D.Ns);
}


// ---- A without a sort (batches of up to 1 024 edges over tables of up to DENSE_GROUP_MAX_N nodes: every Wikipedia-shaped config).
// The sort only serves to put every target's contributions side by side in the reference's order (ascending contribution number
// j); with one LDS word per NODE that is a counting pass: tab[n] = contributions to n (LDS atomics, any order) -> an exclusive
// scan in node order = the groups' spans (and the batch's whole row of the run-length matrix, written as it is scanned: nothing
// to zero, nothing to scatter) -> members listed by arrival -> a member's rank = the members of its group with a smaller j (groups
// of up to 16: counted; a hub: one wave, a bitmap over j, prefix pop-counts: plan.hip, k_plan_one_h) -> the arrays written span by
// span.  The spans are in node order and the ranks in j order: the arrays are the stable sort's, entry for entry.
static constexpr int DENSE_GROUP_MAX_N = 12288;
template <int BS, int IPT, int NT>
__global__ __launch_bounds__(BS) void k_dense_group(WPlan p, WTmp q, DView D, DArgs a) {
    constexpr int NC = BS * IPT;
    constexpr int SPT = NT / BS;                         // table entries per thread in the scan
    static_assert(SPT * BS == NT && SPT % 4 == 0, "k_dense_group: table size");
    constexpr int NW = BS / 64;
    constexpr int BMW = NC / 32;
    constexpr uint32_t SMALL = 16;
    static_assert(NC <= 2048 && BMW <= 64, "k_dense_group: j fits 11 bits, a hub's bitmap one word per lane");
    const int64_t bb = blockIdx.x;
    const int64_t e0 = bb * a.Bfull;
    const int32_t B = (int32_t)((a.Ec - e0 < a.Bfull) ? (a.Ec - e0) : a.Bfull);
    const int64_t* __restrict__ src = a.src + e0;
    const int64_t* __restrict__ dst = a.dst + e0;
    const double* __restrict__ t = a.t + e0;
    const int64_t g0 = 2 * e0;
    uint16_t* __restrict__ lenrow = D.len + bb * D.Ns;
    __shared__ __attribute__((aligned(16))) uint32_t tab[NT];   // contributions of node n; after the scan: span base << 16 | count
    __shared__ uint16_t mem[NC];                         // members of the groups, span by span, in arrival order
    __shared__ uint32_t sorted[NC];                      // position -> j | rank << 11 | (count - 1) << 22
    __shared__ uint32_t bm[NW][BMW];                     // a wave's bitmap over j for the hub it ranks
    __shared__ uint32_t e_src[NC / 2], e_dst[NC / 2];    // endpoint (0 if out of range) | bit 31: the EDGE has a bad endpoint
    __shared__ float e_w[NC / 2];
    __shared__ uint32_t n_big, wsum[NW];
    __shared__ uint32_t big[NC / SMALL];                 // the nodes of the groups larger than SMALL
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nc = 2 * B;
    constexpr int EPT = IPT / 2;
    int64_t rs[EPT], rd[EPT];
    double rt[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {                      // (the loads are in flight while the table is cleared)
        const int e = k * BS + tid;
        const int ec = e < B ? e : B - 1;
        rs[k] = src[ec];
        rd[k] = dst[ec];
        rt[k] = t[ec];
    }
    const double t_last = t[B - 1];                      // next_time = node_interact_times[-1]   (TPNet.py:76)
    // clock left by the previous batch (TPNet.py:99)
    const double t_now = (bb == 0) ? (a.t_prev ? *a.t_prev : a.now_time) : a.t[e0 - 1];
    if (tid == 0) n_big = 0;
#pragma unroll
    for (int k = 0; k < SPT / 4; ++k) reinterpret_cast<uint4*>(tab)[k * BS + tid] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = k * BS + tid;
        if (e < B) {
            const int64_t s = rs[k], dd = rd[k];
            const bool oks = (uint64_t)s < (uint64_t)a.N, okd = (uint64_t)dd < (uint64_t)a.N;
            const uint32_t bad = (oks && okd) ? 0u : 0x80000000u;
            if (bad) atomicAdd(a.err, 1u);               // once per bad edge
            e_src[e] = (oks ? (uint32_t)s : 0u) | bad;
            e_dst[e] = (okd ? (uint32_t)dd : 0u) | bad;
            const float x = (float)t_last - (float)rt[k];   // the reference's casts (models/TPNet.py:77-78), as wsort_batch
            e_w[e] = bad ? 0.0f : expf((float)(-a.lambda) * x);
        }
    }
    __syncthreads();
    auto target_of = [&](int j) -> uint32_t {            // first the src-side scatter-adds, then the dst-side ones (TPNet.py:93-96)
        return ((j >= B) ? e_dst[j - B] : e_src[j]) & 0x7FFFFFFFu;
    };
    uint32_t key[IPT], arr[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = k * BS + tid;
        key[k] = 0;
        arr[k] = 0;
        if (j < nc) {
            key[k] = target_of(j);
            arr[k] = atomicAdd(&tab[key[k]], 1u);
        }
    }
    __syncthreads();
    // spans: exclusive scan of the nodes' counts (thread `tid` holds nodes [tid * SPT, tid * SPT + SPT)); the row of run lengths
    uint32_t c[SPT], tsum = 0;
#pragma unroll
    for (int k = 0; k < SPT / 4; ++k) {
        const uint4 v = reinterpret_cast<const uint4*>(tab)[tid * (SPT / 4) + k];
        c[4 * k] = v.x; c[4 * k + 1] = v.y; c[4 * k + 2] = v.z; c[4 * k + 3] = v.w;
        tsum += v.x + v.y + v.z + v.w;
    }
    uint32_t inc = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const int64_t n = (int64_t)tid * SPT + k;
        if (n < a.N) lenrow[n] = (uint16_t)c[k];
    }
    __syncthreads();
    uint32_t base = inc - tsum;
#pragma unroll
    for (int i = 0; i < NW; ++i)
        if (i < wave) base += wsum[i];
    if (tsum) {
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
            tab[tid * SPT + k] = (base << 16) | c[k];
            if (c[k] > SMALL) big[atomicAdd(&n_big, 1u)] = (uint32_t)(tid * SPT + k);
            base += c[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = k * BS + tid;
        if (j < nc) mem[(tab[key[k]] >> 16) + arr[k]] = (uint16_t)j;
    }
    __syncthreads();
    auto emit = [&](uint32_t j, uint32_t gb, uint32_t rank, uint32_t cnt) {
        sorted[gb + rank] = j | (rank << 11) | ((cnt - 1u) << 22);
    };
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = k * BS + tid;
        if (j < nc) {
            const uint32_t cb = tab[key[k]];
            const uint32_t gb = cb >> 16, cnt = cb & 0xFFFFu;
            if (cnt <= SMALL) {
                uint32_t rank = 0;
                if (cnt > 1) {
                    for (uint32_t m = 0; m < cnt; m += 4) {          // (four independent LDS reads per round; a read past the
#pragma unroll                                                       //  span is masked)
                        for (uint32_t i = 0; i < 4; ++i) {
                            const uint32_t mm = m + i;
                            const uint32_t o = mem[(gb + mm) < (uint32_t)NC ? gb + mm : 0u];
                            rank += (mm < cnt && o < (uint32_t)j) ? 1u : 0u;
                        }
                    }
                }
                emit((uint32_t)j, gb, rank, cnt);
            }
        }
    }
    const uint32_t nbig = n_big;
    for (uint32_t k = wave; k < nbig; k += NW) {         // a hub: one wave, a bitmap over j, prefix pop-counts
        const uint32_t cb = tab[big[k]];
        const uint32_t gb = cb >> 16, cnt = cb & 0xFFFFu;
        if (lane < BMW) bm[wave][lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t m = lane; m < cnt; m += 64) {
            const uint32_t j = mem[gb + m];
            atomicOr(&bm[wave][j >> 5], 1u << (j & 31u));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t word = lane < BMW ? bm[wave][lane] : 0u;
        uint32_t pre = (uint32_t)__popc(word);
        const uint32_t own = pre;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)pre, o, 64);
            if (lane >= o) pre += v;
        }
        pre -= own;
        for (uint32_t m0 = 0; m0 < cnt; m0 += 64) {      // (uniform trip count: the shuffles are wave-wide)
            const uint32_t m = m0 + lane;
            const bool on = m < cnt;
            const uint32_t j = on ? mem[gb + m] : 0u;
            const uint32_t pw = (uint32_t)__shfl((int)pre, (int)(j >> 5), 64);
            const uint32_t ww = (uint32_t)__shfl((int)word, (int)(j >> 5), 64);
            const uint32_t rank = pw + (uint32_t)__popc(ww & ((1u << (j & 31u)) - 1u));
            if (on) emit(j, gb, rank, cnt);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // the batch's arrays, span by span (coalesced)
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int pos = k * BS + tid;
        if (pos < nc) {
            const uint32_t v = sorted[pos];
            const uint32_t j = v & 0x7FFu, ri = (v >> 11) & 0x7FFu, cnt = (v >> 22) + 1u;
            const bool side = j >= (uint32_t)B;
            const int e = side ? (int)j - B : (int)j;
            const uint32_t es = e_src[e], ed = e_dst[e];
            const bool ok = !(es & 0x80000000u);
            const bool tail = ri + 1u == cnt;
            uint32_t fl = 0;
            if (ri == 0) fl |= WREF_RUN_HEAD;
            if (tail) fl |= WREF_RUN_TAIL;
            if (ri % WIN_BLOCK == 0) fl |= WREF_BLK_HEAD;
            if (ri % WIN_BLOCK == WIN_BLOCK - 1 || tail) fl |= WREF_BLK_TAIL;
            q.bkey[g0 + pos] = (side ? ed : es) & 0x7FFFFFFFu;
            q.bpart[g0 + pos] = ok ? (int32_t)((side ? es : ed) & 0x7FFFFFFFu) : 0;
            q.bcoef[g0 + pos] = ok ? e_w[e] : 0.0f;
            q.bval[g0 + pos] = j;
            q.binv[g0 + j] = (uint32_t)(g0 + pos);
            q.bri[g0 + pos] = ri;
            q.bflags[g0 + pos] = fl;
        }
    }
    if (tid == 0) {
        BatchDesc Dn;
        Dn.e0 = e0;
        Dn.ne = B;
        Dn.pad = 0;
        Dn.t_last = t_last;
        Dn.now = t_now;
        Dn.n_light = 0;
        Dn.n_heavy = 0;
        for (int i = 0; i < TPNET_MAX_LAYERS; ++i) Dn.decay[i] = 1.0f;            // (as wsort_batch: read by the eager mode only)
        p.base.desc[bb] = Dn;
    }
}

// ---- A (row shard): the batch holds the edges of ALL ranks (local ids: rows < own are this rank's, the rows behind them halo rows
// of other ranks' nodes); only the contributions to OWNED targets are sorted -- ~2 B / G of the batch's 2 B, so a global batch of up
// to 8 192 edges fits the one-workgroup sort -- and a halo node that is a target in this batch gets len = 1: its run is ONE slot of
// the version log, which the owner's result is received into (wshard.hip).  Order: thread t takes contributions [t C, (t + 1) C) of
// the batch (src side first, then dst side, edge by edge: models/TPNet.py:93-96), the kept ones are compacted in that order (one
// block scan) and the stable sort keeps it inside a target's run.  desc[b].n_light = kept contributions (k_dense_place skips the rest).
template <int BS, int IPT>
__global__ __launch_bounds__(BS) void k_dense_sort_shard(WPlan p, WTmp q, DView D, DArgs a) {
    using Sort = rocprim::block_radix_sort<uint32_t, BS, IPT, uint32_t>;
    constexpr int NC = BS * IPT;
    constexpr int CMAX = 16;                               // contributions per thread before the compaction: 2 B <= 16 BS
    __shared__ union U {
        typename Sort::storage_type sort;
        struct { uint32_t key[NC]; uint32_t ustart[NC]; } s;
        struct { uint32_t ckey[NC]; uint32_t cval[NC]; } c;
    } u;
    __shared__ uint32_t wsum[BS / 64];
    // halo nodes that are targets in this batch: a bitmap in LDS, written out as whole rows of the length matrix afterwards (14 000
    // scattered 2-byte stores per batch otherwise); halo rows beyond HBITS * 32 (tables the dense planner does not reach anyway)
    // fall back to the direct store
    constexpr int HBITS = 2048;                            // 65 536 halo rows
    __shared__ uint32_t hbits[HBITS];
    static_assert(sizeof(U) + sizeof(uint32_t) * HBITS <= 62 * 1024, "k_dense_sort_shard: LDS budget");
    const int64_t bb = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t e0 = bb * a.Bfull;
    const int32_t B = (int32_t)((a.Ec - e0 < a.Bfull) ? (a.Ec - e0) : a.Bfull);
    const int64_t* __restrict__ src = a.src + e0;
    const int64_t* __restrict__ dst = a.dst + e0;
    const double* __restrict__ t = a.t + e0;
    const int64_t g0 = 2 * e0;
    uint16_t* __restrict__ lenrow = D.len + bb * D.Ns;
    for (int i = tid; i < HBITS; i += BS) hbits[i] = 0u;
    const bool hfit = a.N - a.own <= (int64_t)HBITS * 32;
    for (int64_t n = tid; n < (hfit ? a.own : a.N); n += BS) lenrow[n] = 0;
    __syncthreads();
    const double t_last = t[B - 1];
    const int nall = 2 * B;
    const int C = (nall + BS - 1) / BS;
    uint32_t rk[CMAX];
    uint32_t kept = 0;
#pragma unroll
    for (int k = 0; k < CMAX; ++k) {
        rk[k] = 0xFFFFFFFFu;
        const int j = tid * C + k;
        if (k < C && j < nall) {
            const bool side = j >= B;
            const int e = side ? j - B : j;
            const int64_t sv = src[e], dv = dst[e];
            const bool ok = (uint64_t)sv < (uint64_t)a.N && (uint64_t)dv < (uint64_t)a.N;
            if (!ok) {
                if (!side) atomicAdd(a.err, 1u);             // once per bad edge (the shard's relabelling has rejected such ids already)
            } else {
                const int64_t tg = side ? dv : sv;
                if (tg < a.own) { rk[k] = (uint32_t)tg; ++kept; }
                else if (hfit) atomicOr(&hbits[(tg - a.own) >> 5], 1u << ((tg - a.own) & 31));   // a halo node is a target here (LDS bitmap)
                else lenrow[tg] = 1;
            }
        }
    }
    uint32_t total;
    uint32_t off = block_excl_scan<BS>(kept, wsum, total);   // (its barriers also publish the bitmap)
    if (hfit)
        for (int64_t n = a.own + tid; n < a.N; n += BS) lenrow[n] = (uint16_t)((hbits[(n - a.own) >> 5] >> ((n - a.own) & 31)) & 1u);
    if (total > (uint32_t)NC) {                              // (uniform) more owned contributions than the sort holds: the caller falls back
        if (tid == 0 && a.status) atomicAdd(a.status, 1u);
        total = NC;
    }
#pragma unroll
    for (int k = 0; k < CMAX; ++k) {
        if (rk[k] != 0xFFFFFFFFu) {
            if (off < (uint32_t)NC) { u.c.ckey[off] = rk[k]; u.c.cval[off] = (uint32_t)(tid * C + k); }
            ++off;
        }
    }
    __syncthreads();
    const int nc = (int)total;
    uint32_t keys[IPT], vals[IPT];
    const uint32_t pad_key = 1u << a.node_bits;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = tid * IPT + k;
        keys[k] = j < nc ? u.c.ckey[j] : pad_key;
        vals[k] = j < nc ? u.c.cval[j] : 0u;
    }
    __syncthreads();
    Sort().sort(keys, vals, u.sort, 0u, (unsigned)(a.node_bits + 1));
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IPT; ++k) u.s.key[tid * IPT + k] = keys[k];
    __syncthreads();
    uint32_t hcount = 0;
    bool hd[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = tid * IPT + k;
        hd[k] = j < nc && (j == 0 || u.s.key[j - 1] != keys[k]);
        hcount += hd[k] ? 1u : 0u;
        if (j < nc) {
            const uint32_t val = vals[k];
            const bool side = val >= (uint32_t)B;
            const int e = side ? (int)val - B : (int)val;
            const int64_t sv = src[e], dv = dst[e];
            const float x = (float)t_last - (float)t[e];     // the reference's casts (models/TPNet.py:77-78), as wsort_batch
            q.bkey[g0 + j] = keys[k];
            q.bpart[g0 + j] = (int32_t)(side ? sv : dv);
            q.bcoef[g0 + j] = expf((float)(-a.lambda) * x);
            q.bval[g0 + j] = val;
            q.binv[g0 + val] = (uint32_t)(g0 + j);
        }
    }
    uint32_t htotal;
    uint32_t rank = block_excl_scan<BS>(hcount, wsum, htotal);
    uint32_t myrun[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        if (hd[k]) {
            u.s.ustart[rank] = (uint32_t)(tid * IPT + k);
            ++rank;
        }
        myrun[k] = rank - 1u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = tid * IPT + k;
        if (j < nc) {
            const uint32_t st = u.s.ustart[myrun[k]];
            const uint32_t en = (myrun[k] + 1 < htotal) ? u.s.ustart[myrun[k] + 1] : (uint32_t)nc;
            const uint32_t ri = (uint32_t)j - st;
            const bool tail = (uint32_t)j + 1u == en;
            uint32_t fl = 0;
            if (ri == 0) fl |= WREF_RUN_HEAD;
            if (tail) fl |= WREF_RUN_TAIL;
            if (ri % WIN_BLOCK == 0) fl |= WREF_BLK_HEAD;
            if (ri % WIN_BLOCK == WIN_BLOCK - 1 || tail) fl |= WREF_BLK_TAIL;
            q.bri[g0 + j] = ri;
            q.bflags[g0 + j] = fl;
        }
    }
    for (uint32_t r = tid; r < htotal; r += BS) {
        const uint32_t st = u.s.ustart[r];
        const uint32_t en = (r + 1 < htotal) ? u.s.ustart[r + 1] : (uint32_t)nc;
        lenrow[u.s.key[st]] = (uint16_t)(en - st);
    }
    if (tid == 0) {
        BatchDesc Dd;
        Dd.e0 = e0;
        Dd.ne = B;
        Dd.pad = 0;
        Dd.t_last = t_last;
        Dd.now = (bb == 0) ? (a.t_prev ? *a.t_prev : a.now_time) : a.t[e0 - 1];
        Dd.n_light = (uint32_t)nc;
        Dd.n_heavy = 0;
        for (int i = 0; i < TPNET_MAX_LAYERS; ++i) Dd.decay[i] = 1.0f;
        p.base.desc[bb] = Dd;
    }
}

// ---- B1: a workgroup per (chunk of DCH nodes, window), a thread per node: the node's contributions in the window and the
// batch of its last run there; its chain in the window (that many contributions) ranked among the chunk's chains of the same
// (window, class) in node order -- ballots inside a wave, the waves' counts through LDS.  (A thread per node that walked its
// whole column was one dependent round trip per 16 batches with the stores of m in between: 15 us for 20 batches, 57 us for a
// Wikipedia epoch.)
__global__ __launch_bounds__(DCH) void k_dense_win(WPlan p, DView D, DArgs a) {
    __shared__ uint32_t wcnt[(DCH / 64) * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KW = p.K;
    const int64_t c = blockIdx.x / a.nw, w = blockIdx.x - c * a.nw;
    const int64_t n = c * DCH + tid;
    const bool live = n < a.N;
    const int64_t nld = live ? n : 0;                            // (dead threads load node 0's column and store nothing)
    const int64_t bw0 = w * KW, bw1 = (bw0 + KW < a.nb) ? bw0 + KW : a.nb;
    uint32_t sum = 0, last = 0xFFFFu;
    constexpr int UB = 16;
    for (int64_t b0 = bw0; b0 < bw1; b0 += UB) {
        uint32_t l[UB];
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            const int64_t b = (b0 + k < bw1) ? b0 + k : bw1 - 1;          // no branch around a load: rows past the end are clamped
            l[k] = (uint32_t)D.len[b * D.Ns + nld];
        }
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            if (b0 + k < bw1) {
                sum += l[k];
                last = l[k] ? (uint32_t)(b0 + k) : last;
            }
        }
    }
    if (!live) sum = 0;
    if (live) D.wsl[w * D.Ns + n] = make_uint2(sum, last);
    const int cl = (sum && n < a.own) ? wchain_class(sum, p.heavy_thr) : -1;     // (a halo node's runs are filled by the exchange: no chain)
    uint32_t myrank = 0;
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        const unsigned long long mk = __ballot(cl == c8);
        if (cl == c8) myrank = (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[wave * 8 + c8] = (uint32_t)__popcll(mk);
    }
    __syncthreads();
    if (cl >= 0) {
        for (int wv = 0; wv < wave; ++wv) myrank += wcnt[wv * 8 + cl];
        D.crank[w * D.Ns + n] = myrank;
    }
    if (tid < 8) {
        uint32_t tsum = 0;
        for (int wv = 0; wv < DCH / 64; ++wv) tsum += wcnt[wv * 8 + tid];
        D.ccnt[(size_t)c * (size_t)(a.nw * 8) + w * 8 + tid] = tsum;
    }
}

// ---- B2: the same items: the node's contributions and last run BEFORE the window (from the windows' sums), then m for the
// window's batches; chains of every class in the earlier chunks of nodes; the chunk's last window: node totals
__global__ __launch_bounds__(DCH) void k_dense_pre(WPlan p, DView D, DArgs a) {
    __shared__ uint32_t wsum[DCH / 64];
    __shared__ uint32_t part[DCH];
    const int tid = threadIdx.x;
    const int KW = p.K;
    const int64_t c = blockIdx.x / a.nw, w = blockIdx.x - c * a.nw;
    const int64_t n = c * DCH + tid;
    const bool live = n < a.N;
    const int64_t nld = live ? n : 0;
    const int64_t bw0 = w * KW, bw1 = (bw0 + KW < a.nb) ? bw0 + KW : a.nb;
    const size_t npair = (size_t)(a.nw * 8);
    constexpr int UB = 16;
    // chains of (window, class tid % 8) in the chunks before this one: 32 strided partial sums per class, added in LDS
    {
        uint32_t ps = 0;
        for (int64_t cq = tid >> 3; cq < c; cq += DCH / 8) ps += D.ccnt[(size_t)cq * npair + w * 8 + (tid & 7)];
        part[tid] = ps;
    }
    uint32_t run = 0, prevb = 0xFFFFu;
    for (int64_t w0 = 0; w0 < w; w0 += UB) {
        uint2 v[UB];
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            const int64_t wq = (w0 + k < w) ? w0 + k : w - 1;
            v[k] = D.wsl[wq * D.Ns + nld];
        }
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            if (w0 + k < w) {
                run += v[k].x;
                prevb = v[k].x ? v[k].y : prevb;
            }
        }
    }
    for (int64_t b0 = bw0; b0 < bw1; b0 += UB) {
        uint32_t l[UB];
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            const int64_t b = (b0 + k < bw1) ? b0 + k : bw1 - 1;
            l[k] = (uint32_t)D.len[b * D.Ns + nld];
        }
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            const int64_t b = b0 + k;
            if (b < bw1) {                                       // (uniform)
                if (live) D.m[b * D.Ns + n] = make_uint2(run, l[k] | (prevb << 16));
                run += l[k];
                prevb = l[k] ? (uint32_t)b : prevb;
            }
        }
    }
    __syncthreads();
    if (tid < 8) {
        uint32_t before = 0;
        for (int k = 0; k < DCH / 8; ++k) before += part[k * 8 + tid];
        D.cpre[(size_t)c * npair + w * 8 + tid] = before;
        if (c == a.nchunks - 1) D.ptot[w * 8 + tid] = before + D.ccnt[(size_t)c * npair + w * 8 + tid];
    }
    if (w == a.nw - 1) {                                         // (uniform) the chunk's last window: node totals
        if (!live) run = 0;
        uint32_t total;
        const uint32_t ex = block_excl_scan<DCH>(run, wsum, total);
        if (live) { D.tot[n] = run; D.lastb[n] = prevb; D.base[n] = ex; }
        if (tid == 0) D.ctot[c] = total;
    }
}

// ---- C: positions and references.  Every workgroup scans the chunks' totals (first position of every chunk of nodes) and the
// (window, class) totals (the lists' starts) for itself: <= 2 048 values each
__global__ __launch_bounds__(DCH) void k_dense_place(WPlan p, WTmp q, DView D, DArgs a) {
    __shared__ uint32_t wsum[DCH / 64];
    __shared__ uint32_t cbase[DENSE_MAX_CHUNKS];
    __shared__ uint32_t lstart[DENSE_MAX_WINDOWS * 8];
    __shared__ double tlast[DENSE_TL];                   // the closing clocks of up to DENSE_TL batches (one dependent load less per lookup)
    const int tid = threadIdx.x;
    const int npair = (int)a.nw * 8;
    const double* tl = a.nb <= DENSE_TL ? tlast : nullptr;
    if (tl)
        for (int64_t b = tid; b < a.nb; b += DCH) tlast[b] = p.base.desc[b].t_last;
    {
        uint32_t carry = 0;
        for (int64_t c0 = 0; c0 < a.nchunks; c0 += DCH) {
            const int64_t c = c0 + tid;
            const uint32_t v = c < a.nchunks ? D.ctot[c] : 0u;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<DCH>(v, wsum, tot);
            if (c < a.nchunks) cbase[c] = carry + ex;
            carry += tot;
        }
        carry = 0;
        for (int i0 = 0; i0 < npair; i0 += DCH) {
            const int i = i0 + tid;
            const uint32_t v = i < npair ? D.ptot[i] : 0u;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<DCH>(v, wsum, tot);
            if (i < npair) lstart[i] = carry + ex;
            carry += tot;
        }
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        for (int64_t w = tid; w < a.nw; w += DCH) {
            uint32_t t8[8], n = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { t8[k] = D.ptot[w * 8 + k]; n += t8[k]; }
            WinDesc W;
            W.start = lstart[w * 8];
            W.n_heavy = t8[0];
            W.n_chains = n;
            W.n_ext = t8[0] + t8[1] + t8[2];
            W.n_ext1 = t8[0] + t8[1];
            W.n_med = 0;
            for (int k = 0; k <= wchain_med_classes(p.heavy_thr); ++k) W.n_med += t8[k];
            W.pad1 = W.pad2 = 0;
            p.wdesc[w] = W;
        }
    }
    const int64_t gid = (int64_t)blockIdx.x * DCH + tid, gsz = (int64_t)gridDim.x * DCH;
    for (int64_t n = gid; n < a.N; n += gsz) D.basef[n] = D.base[n] + cbase[n / DCH];   // (for the write-back and a replay)
    const int64_t nc = 2 * a.Ec;
    const bool shard = a.own < a.N;
    const int64_t n_neg = a.nwhich == 3 ? a.Ec : 0;
    const int64_t items = nc + n_neg + ((shard && a.nwhich >= 2) ? a.Ec : 0);
    for (int64_t x = gid; x < items; x += gsz) {
        if (x < nc) {
            if (shard) {                                  // (a shard sorted only the contributions to its own targets: the batch's first n_light positions)
                const int64_t b = x / (2 * a.Bfull);
                if (x - 2 * b * a.Bfull >= (int64_t)p.base.desc[b].n_light) continue;
            }
            dense_place(p, q, D, a, lstart, cbase, tl, x);
        } else if (x < nc + n_neg) {
            dense_neg_c(p, D, a, cbase, tl, x - nc);
        } else {
            dense_neg_c(p, D, a, cbase, tl, x - nc - n_neg, 1);
        }
    }
}

// a replayed plan (same stream, same table state): only the negatives change between epochs
__global__ __launch_bounds__(256) void k_wplan_dense_negs(WPlan p, DView D, DArgs a) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < a.Ec; e += (int64_t)gridDim.x * blockDim.x)
        dense_neg(p, D, a, D.basef, e);
}

// ---------------------------------------------------------------------------------------------------------------
// end of a chunk planned here: the last version of every touched node -> the other copy of its table bundle, meta published.
// One lane group per node: {tot, base, lastb} -> the log row -> the table (three dependent round trips).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wwriteback_dense(tpnet_state S, WPlan p, DView D, uint32_t bid, int64_t own) {
    constexpr int LANES = 32;
    const int per = S.L * S.d;
    const int pv = per / 4;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    const int gl = threadIdx.x % LANES, g = threadIdx.x / LANES;
    constexpr int GPB = 256 / LANES;
    for (int64_t n = (int64_t)blockIdx.x * GPB + g; n < own; n += (int64_t)gridDim.x * GPB) {
        const uint32_t tot = D.tot[n];
        const uint32_t bs = D.basef[n], lb = D.lastb[n];
        const uint32_t ver = meta[n].ver;
        if (!tot) continue;
        const uint32_t pos = bs + tot - 1u;
        const int c = (int)(ver & 1u);
        const float4* __restrict__ srow = reinterpret_cast<const float4*>(p.log + (int64_t)pos * per);
        float4* __restrict__ drow = reinterpret_cast<float4*>(S.q + ((int64_t)(c ^ 1) * S.N + n) * (int64_t)per);
        for (int x = gl; x < pv; x += LANES) drow[x] = srow[x];
        if (gl == 0) publish_meta(meta + n, c ^ 1, p.base.desc[lb].t_last, bid);
    }
}

bool wplan_dense_applies(const tpnet_state& st, const WPlan& p, int64_t Ec, int64_t batch, int K) {
    return batch <= PLAN_ONE_MAX && wplan_dense_applies_shard(st, p, Ec, batch, K);
}
bool wplan_dense_applies_shard(const tpnet_state& st, const WPlan& p, int64_t Ec, int64_t batch, int K) {
    if (!p.dense || K < 1 || !wplan_dense_eligible(st.N, st.d, st.L, batch)) return false;
    const int64_t nb = (Ec + batch - 1) / batch;
    const int64_t nw = (nb + K - 1) / K;
    int node_bits = 1;
    while (node_bits < 63 && (1ll << node_bits) < st.N) ++node_bits;
    return nw <= DENSE_MAX_WINDOWS && nw <= (nb + 1) / 2 + 1 && nb < 65535 && node_bits <= 31 && 2 * Ec < (int64_t)WREF_SLOT_MASK;
}

int wplan_dense_build(const tpnet_state& st, const WPlan& p, const int64_t* src, const int64_t* dst, const int64_t* neg,
                      const double* t, int64_t Ec, int64_t batch, double now_time, const double* t_prev_dev, double lambda,
                      bool want_readout, bool replay, hipStream_t s, int64_t own, uint32_t* status) {
    if (own < 0 || own > st.N) own = st.N;
    const bool shard = own < st.N;
    if (!(shard ? wplan_dense_applies_shard(st, p, Ec, batch, p.K) : wplan_dense_applies(st, p, Ec, batch, p.K))) return TPNET_ERR_BAD_ARG;
    if (shard && replay) return TPNET_ERR_BAD_ARG;                      // (a shard's plan is kept by its runner, not replayed here)
    const int64_t nb = (Ec + batch - 1) / batch;
    const WTmp q = wtmp_of(p, (size_t)(2 * Ec));
    const DView D = dview_of(p, Ec, batch, st.N);
    DArgs a;
    a.src = src; a.dst = dst; a.neg = neg; a.t = t; a.t_prev = t_prev_dev;
    a.meta = reinterpret_cast<const NodeMeta*>(st.meta);
    a.err = st.err;
    a.Ec = Ec; a.Bfull = batch; a.N = st.N; a.nb = nb; a.nw = (nb + p.K - 1) / p.K;
    a.now_time = now_time; a.lambda = lambda;
    int node_bits = 1;
    while (node_bits < 31 && (1ll << node_bits) < st.N) ++node_bits;
    a.node_bits = node_bits;
    a.L = st.L;
    a.nwhich = want_readout ? (neg ? 3 : 2) : 0;
    a.nchunks = (int32_t)((st.N + DCH - 1) / DCH);
    a.own = own;
    a.status = status;
    if (replay) {
        if (a.nwhich == 3) {
            int g = (int)((Ec + 255) / 256);
            if (g > 2048) g = 2048;
            hipLaunchKernelGGL(k_wplan_dense_negs, dim3(g), dim3(256), 0, s, p, D, a);
            TPNET_HIP_TRY(hipGetLastError());
        }
        return TPNET_OK;
    }
#define TPNET_WDENSE(BS_, IPT_) hipLaunchKernelGGL((k_dense_sort<BS_, IPT_>), dim3((unsigned)nb), dim3(BS_), 0, s, p, q, D, a)
    const int64_t n2 = 2 * batch;
    static const int no_group = TPNET_DEV_INT(DENSE_SORT, 0);             // developer override: the sorting phase A for every batch size
    if (shard) hipLaunchKernelGGL((k_dense_sort_shard<1024, 4>), dim3((unsigned)nb), dim3(1024), 0, s, p, q, D, a);
    else if (!no_group && n2 <= 2048 && st.N <= DENSE_GROUP_MAX_N)
        hipLaunchKernelGGL((k_dense_group<1024, 2, DENSE_GROUP_MAX_N>), dim3((unsigned)nb), dim3(1024), 0, s, p, q, D, a);
    else if (n2 <= 512) TPNET_WDENSE(256, 2);
    else if (n2 <= 1024) TPNET_WDENSE(512, 2);
    else if (n2 <= 2048) TPNET_WDENSE(1024, 2);
    else TPNET_WDENSE(1024, 4);
#undef TPNET_WDENSE
    const unsigned bgrid = (unsigned)((int64_t)a.nchunks * a.nw);
    hipLaunchKernelGGL(k_dense_win, dim3(bgrid), dim3(DCH), 0, s, p, D, a);
    hipLaunchKernelGGL(k_dense_pre, dim3(bgrid), dim3(DCH), 0, s, p, D, a);
    int64_t cgrid = (3 * Ec + DCH - 1) / DCH;
    if (cgrid > 4096) cgrid = 4096;
    hipLaunchKernelGGL(k_dense_place, dim3((unsigned)cgrid), dim3(DCH), 0, s, p, q, D, a);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

bool wplan_dense_writeback(const tpnet_state& st, const WPlan& p, int64_t Ec, int64_t batch, uint32_t launch_id, hipStream_t s,
                           int64_t own) {
    if ((st.L * st.d) % 4 != 0) return false;
    if (own < 0 || own > st.N) own = st.N;
    const DView D = dview_of(p, Ec, batch, st.N);
    int64_t grid = (own + 7) / 8;
    if (grid > 16384) grid = 16384;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_wwriteback_dense, dim3((unsigned)grid), dim3(256), 0, s, st, p, D, launch_id, own);
    return hipGetLastError() == hipSuccess;
}

}  // namespace tpnet
