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
#include "tpnet_common.h"
#include "device_common.hpp"

#include <rocprim/block/block_radix_sort.hpp>

namespace tpnet {

static constexpr uint32_t WT_CHAIN_HEAD = 1u;       // bflags: first contribution of its node in the window
static constexpr uint32_t WT_WIN_LAST = 2u;         // bflags: tail of its node's last run in the window
static constexpr uint32_t WT_STRUCT_MASK = WREF_RUN_HEAD | WREF_RUN_TAIL | WREF_BLK_HEAD | WREF_BLK_TAIL;
static constexpr int WCLS = 8;                      // length classes of a window's chain list (class 0: workgroup-walked)

// H1, runs: one open-addressing SEGMENT per batch (linear probing inside the segment, at most a third full), built by the
// batch's workgroup in LDS and written out whole with plain stores -- no global atomics, nothing to zero.  (One chunk-wide
// table claimed with atomicCAS cost k_wsort 25 of its 37 us: ~240 000 scattered returning atomics per epoch of C2.)
// H2, chains: indexed [window][node] where that fits the table's bytes (one non-returning atomic OR per run marks the batch);
// else open addressing over the whole table (key 0 = empty; zeroed before every plan).
struct HEnt {
    unsigned long long key;
    uint32_t a, b;
};
static_assert(sizeof(HEnt) == 16, "HEnt must be 16 bytes");
struct CEnt {                  // chain table: one entry per (node, window)
    unsigned long long key;
    unsigned long long mask;   // bit k: the node is a target in batch (first batch of the window) + k      (A: atomic or)
    uint32_t xh;               // batch-sorted position of the chain's head                                 (B: the chain head)
    uint32_t cnt;              // contributions of the chain                                                (B)
    uint32_t pad0, pad1;
};
static_assert(sizeof(CEnt) == 32, "CEnt must be 32 bytes");

#ifdef TPNET_STAMPS
#define WSTAMP(slot)                                                                          \
    do {                                                                                      \
        if (q.dbg && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) {                       \
            q.dbg[(slot) * 2 + 0] = __builtin_amdgcn_s_memtime();                             \
            q.dbg[(slot) * 2 + 1] = __builtin_amdgcn_s_memrealtime();                         \
        }                                                                                     \
    } while (0)
#else
#define WSTAMP(slot) do { } while (0)
#endif

struct WTmp {              // scratch of the stages, aliased onto arrays of the base plan that this planner does not use
    uint32_t* bkey;        // [nc] target node of the contribution at this batch-sorted position (batch b owns [2 e0(b), ...))
    int32_t* bpart;        // [nc] its partner
    float* bcoef;          // [nc] its time weight
    uint32_t* bval;        // [nc] its pre-sort index inside the batch (side * ne + edge)
    uint32_t* binv;        // [nc] chunk-relative pre-sort index -> chunk-relative batch-sorted position
    uint32_t* bsb;         // [nc] at a run's HEAD: contributions of the node in EARLIER batches of the window
    uint32_t* bri;         // [nc] rank inside its (node, batch) run (k_wsort)
    uint32_t* rprevb;      // [nc] at a run's HEAD: batch of the node's previous run inside the window (bsb > 0); chain heads: their place in the window's list of their length class
    uint32_t* bflags;      // [nc] WREF_* structure bits (k_wsort) | WT_* (k_wchains)
    HEnt* h1;              // (node, batch) -> a = chunk-relative batch-sorted position of the run's head, b = its length
    CEnt* h2;              // (node, window) -> the batches the node is a target in, its chain
    uint32_t hcap;         // entries of H2
    uint32_t seg;          // entries of a batch's segment of H1
    uint32_t direct2;      // H2 is indexed [window][node] (n2 nodes per window)
    uint32_t n2;
    uint32_t* blkcnt;      // [nblk][9] per workgroup of k_wchains: chain heads per length class, positions their chains need
    uint32_t* blkbase;     // [nblk][9] k_wscan: the same, summed over the earlier workgroups of the window
    uint32_t* boff;        // [nc] chain heads: what the chain heads before it in its workgroup claimed (positions)
    uint32_t* wtot;        // [WIN_MAX_WINDOWS][9] k_wscan: the window's totals of the same counters
    uint32_t bpb;          // workgroups of k_wchains per batch
    unsigned long long* dbg;   // diagnostic builds (-DTPNET_STAMPS): phase stamps of workgroup `stamp block`
};

static inline uint32_t wplan3_table_entries(size_t nc) { return (uint32_t)(2 * nc + 1021); }   // H2: at most half full
// contributions a workgroup of k_wsort sorts (its BS * IPT), and the segment of H1 it builds: the LDS of its three per-edge arrays
static inline uint32_t wplan3_nc_block(int64_t batch) {
    const int64_t n2 = 2 * batch;
    return n2 <= 512 ? 512u : n2 <= 1024 ? 1024u : n2 <= 2048 ? 2048u : 4096u;
}
static inline uint32_t wplan3_seg(int64_t batch) { return wplan3_nc_block(batch) * 3; }   // (16-bit slots in 3 * NC / 2 words)
static inline size_t wplan3_h2_bytes(int64_t Ec) { return (size_t)wplan3_table_entries(2 * (size_t)Ec) * sizeof(CEnt); }
size_t wplan3_table_bytes(int64_t Ec, int64_t batch) {
    const size_t nb = (size_t)((Ec + batch - 1) / batch);
    return wplan3_h2_bytes(Ec) + 256 + nb * wplan3_seg(batch) * sizeof(HEnt) + 256;
}
static inline uint32_t wplan3_bpb(int64_t batch) { return (uint32_t)((2 * batch + 255) / 256); }
size_t wplan3_blk_bytes(int64_t Ec, int64_t batch) {
    const size_t nblk = (size_t)((Ec + batch - 1) / batch) * wplan3_bpb(batch);
    return 2 * nblk * 9 * sizeof(uint32_t) + 256;
}

static WTmp wtmp_of(const WPlan& p, size_t nc) {
    WTmp q;
    q.dbg = p.base.dbg;
    uint32_t* ki = reinterpret_cast<uint32_t*>(p.base.keys_in);     // 8 nc bytes
    uint32_t* ko = reinterpret_cast<uint32_t*>(p.base.keys_out);    // 8 nc bytes
    uint32_t* hv = reinterpret_cast<uint32_t*>(p.base.heavy);       // 32 nc bytes (the chunk planner's chains_sparse)
    q.bkey = ki;
    q.bpart = reinterpret_cast<int32_t*>(ki + nc);
    q.bcoef = reinterpret_cast<float*>(ko);
    q.bval = p.base.vals_out;
    q.binv = p.inv;
    q.bsb = hv;
    q.bri = hv + nc;
    q.rprevb = hv + 2 * nc;
    q.bflags = hv + 3 * nc;
    const uint32_t cap = wplan3_table_entries(nc);
    q.h2 = reinterpret_cast<CEnt*>(p.wtab);                          // (first: the fill of a plan ends behind what H2 uses)
    q.h1 = reinterpret_cast<HEnt*>(reinterpret_cast<char*>(p.wtab) + (wplan3_h2_bytes((int64_t)(nc / 2)) + 255) / 256 * 256);
    q.hcap = cap;
    q.seg = 0;
    q.direct2 = 0;
    q.n2 = 0;
    q.boff = p.rhead;
    q.wtot = p.wcls;
    return q;
}

__device__ __forceinline__ float decay3_f32(double lambda, double dt) {   // as plan.hip decay_f32 / device_common.hpp meta_view
    const float x = (float)(-lambda * dt);
    return (x == 0.0f) ? 1.0f : expf(x);
}

__device__ __forceinline__ uint32_t hmix(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return (uint32_t)k;
}
__device__ __forceinline__ unsigned long long hkey(uint32_t node, uint32_t minor) {      // minor: batch (H1) / window (H2), < 2^16
    return ((unsigned long long)(node + 1u) << 16) | (unsigned long long)minor;
}
// the run of (node, batch): a = chunk-relative batch-sorted position of its head, b = its length (false: the node is no target there)
__device__ __forceinline__ bool hfind(const HEnt* __restrict__ T, uint32_t seg, unsigned long long key, uint32_t& a, uint32_t& b) {
    const HEnt* __restrict__ S = T + (size_t)(key & 0xFFFFull) * seg;
    uint32_t i = hmix(key) % seg;
    for (;;) {
        const uint4 e = *reinterpret_cast<const uint4*>(S + i);
        const unsigned long long k = ((unsigned long long)e.y << 32) | e.x;
        if (k == key) { a = e.z; b = e.w; return true; }
        if (k == 0ull) { a = 0; b = 0; return false; }
        i = (i + 1 == seg) ? 0u : i + 1;
    }
}

// G probes at once: all first loads are issued before any is looked at (a loop per probe would be G dependent round
// trips); a probe that lands on another key's slot walks on alone (rare: a segment is at most a third full)
template <int G>
__device__ __forceinline__ void hfind_multi(const HEnt* __restrict__ T, uint32_t seg, const unsigned long long* key,
                                            const bool* valid, uint32_t* a, uint32_t* b) {
    uint32_t idx[G];
    uint4 e[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        idx[g] = hmix(key[g]) % seg;
        e[g] = *reinterpret_cast<const uint4*>(T + (valid[g] ? (size_t)(key[g] & 0xFFFFull) * seg + idx[g] : (size_t)0));
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        a[g] = 0;
        b[g] = 0;
        if (!valid[g]) continue;
        const HEnt* __restrict__ S = T + (size_t)(key[g] & 0xFFFFull) * seg;
        uint32_t i = idx[g];
        uint4 x = e[g];
        for (;;) {
            const unsigned long long k = ((unsigned long long)x.y << 32) | x.x;
            if (k == key[g]) { a[g] = x.z; b[g] = x.w; break; }
            if (k == 0ull) break;
            i = (i + 1 == seg) ? 0u : i + 1;
            x = *reinterpret_cast<const uint4*>(S + i);
        }
    }
}

// chain table: A marks batch `bit` of window w in the node's entry (hashed: created by whoever comes first)
__device__ __forceinline__ void cmark(const WTmp& q, uint32_t node, uint32_t w, unsigned bit) {
    if (q.direct2) {
        (void)__hip_atomic_fetch_or(&q.h2[(size_t)w * q.n2 + node].mask, 1ull << bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const unsigned long long key = hkey(node, w);
    uint32_t i = hmix(key ^ 0x5bd1e995ull) % q.hcap;
    for (;;) {
        const unsigned long long old = atomicCAS(&q.h2[i].key, 0ull, key);
        if (old == 0ull || old == key) {
            (void)__hip_atomic_fetch_or(&q.h2[i].mask, 1ull << bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        i = (i + 1 == q.hcap) ? 0u : i + 1;
    }
}
// the entry of (node, window) (nullptr: the node is no target in the window)
__device__ __forceinline__ CEnt* cfind(const WTmp& q, uint32_t node, uint32_t w) {
    if (q.direct2) {
        CEnt* e = q.h2 + ((size_t)w * q.n2 + node);
        return e->mask ? e : nullptr;
    }
    const unsigned long long key = hkey(node, w);
    uint32_t i = hmix(key ^ 0x5bd1e995ull) % q.hcap;
    for (;;) {
        const unsigned long long k = q.h2[i].key;
        if (k == key) return q.h2 + i;
        if (k == 0ull) return nullptr;
        i = (i + 1 == q.hcap) ? 0u : i + 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// A: per-batch sort.  One workgroup per batch.
// ---------------------------------------------------------------------------------------------------------------
template <int BS, int IPT>
__global__ __launch_bounds__(BS) void k_wsort(WTmp q, BatchDesc* __restrict__ desc, const int64_t* __restrict__ src_c,
                                              const int64_t* __restrict__ dst_c, const double* __restrict__ t_c, int64_t Ec,
                                              int32_t Bfull, int64_t N, int node_bits, double now_time,
                                              const double* __restrict__ t_prev, double lambda, int L, uint32_t* err, int KW) {
    const int64_t bb = blockIdx.x;
    const int64_t e0 = bb * Bfull;
    const int32_t B = (int32_t)((Ec - e0 < Bfull) ? (Ec - e0) : Bfull);
    const int64_t* __restrict__ src = src_c + e0;
    const int64_t* __restrict__ dst = dst_c + e0;
    const double* __restrict__ t = t_c + e0;
    const int64_t g0 = 2 * e0;
    using Sort = rocprim::block_radix_sort<uint32_t, BS, IPT, uint32_t>;
    constexpr int NC = BS * IPT;
    __shared__ union U {
        typename Sort::storage_type sort;
        struct { uint32_t key[NC]; uint32_t ustart[NC]; } s;     // sorted keys; then the heads of the distinct targets' runs
    } u;
    // per edge: endpoints (0 if out of range) | bit 31: the EDGE has a bad endpoint; time weight.  Dead once the sorted arrays are
    // stored: the batch's segment of H1 is then built in the same 3 * NC / 2 words
    __shared__ uint32_t e_all[3 * (NC / 2)];
    uint32_t* const e_src = e_all;
    uint32_t* const e_dst = e_all + NC / 2;
    float* const e_w = reinterpret_cast<float*>(e_all + NC);
    constexpr uint32_t SEG = 3 * NC;                  // slots of 16 bits: at most a third full (a probe that walks is a dependent
                                                      // round trip for every reader of the table: short clusters matter more than bytes)
    __shared__ uint32_t wsum[BS / 64];
    static_assert(sizeof(U) + 3 * (NC / 2) * 4 + (BS / 64) * 4 <= 64 * 1024, "k_wsort: LDS budget");
    static_assert(NC <= 65535 / 2 * 2, "k_wsort: run indices");
    const int tid = threadIdx.x;
    const int nc = 2 * B;
    WSTAMP(0);
    constexpr int EPT = IPT / 2;
    int64_t rs[EPT], rd[EPT];
    double rt[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = k * BS + tid;
        const int ec = e < B ? e : B - 1;
        rs[k] = src[ec];
        rd[k] = dst[ec];
        rt[k] = t[ec];
    }
    const double t_last = t[B - 1];                      // next_time = node_interact_times[-1]   (TPNet.py:76)
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = k * BS + tid;
        if (e < B) {
            const int64_t s = rs[k], dd = rd[k];
            const bool oks = (uint64_t)s < (uint64_t)N, okd = (uint64_t)dd < (uint64_t)N;
            const uint32_t bad = (oks && okd) ? 0u : 0x80000000u;
            if (bad) atomicAdd(err, 1u);                 // once per bad edge
            e_src[e] = (oks ? (uint32_t)s : 0u) | bad;
            e_dst[e] = (okd ? (uint32_t)dd : 0u) | bad;
            // time weight with the reference's casts: absolute times rounded to f32 BEFORE the subtraction (models/TPNet.py:77-78)
            const float x = (float)t_last - (float)rt[k];
            e_w[e] = bad ? 0.0f : expf((float)(-lambda) * x);
        }
    }
    __syncthreads();
    WSTAMP(1);
    uint32_t keys[IPT], vals[IPT];
    const uint32_t pad_key = 1u << node_bits;            // above every node id: padding sorts last
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = tid * IPT + k;
        vals[k] = (uint32_t)j;
        if (j < nc) {
            const bool side = j >= B;                    // first the src-side scatter-adds, then the dst-side ones (TPNet.py:93-96)
            const int e = side ? j - B : j;
            keys[k] = (side ? e_dst[e] : e_src[e]) & 0x7FFFFFFFu;
        } else {
            keys[k] = pad_key;
        }
    }
    Sort().sort(keys, vals, u.sort, 0u, (unsigned)(node_bits + 1));
    __syncthreads();
    WSTAMP(2);
#pragma unroll
    for (int k = 0; k < IPT; ++k) u.s.key[tid * IPT + k] = keys[k];
    __syncthreads();
    // the sorted contributions; the heads of the runs are ranked by a block-wide exclusive scan (thread `tid` holds
    // positions [tid * IPT, tid * IPT + IPT))
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
            const uint32_t es = e_src[e], ed = e_dst[e];
            const bool ok = !(es & 0x80000000u);
            q.bkey[g0 + j] = keys[k];
            q.bpart[g0 + j] = ok ? (int32_t)((side ? es : ed) & 0x7FFFFFFFu) : 0;
            q.bcoef[g0 + j] = ok ? e_w[e] : 0.0f;
            q.bval[g0 + j] = val;
            q.binv[g0 + val] = (uint32_t)(g0 + j);
        }
    }
    uint32_t inc = hcount;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    WSTAMP(3);
    for (uint32_t i = tid; i < SEG / 2; i += BS) e_all[i] = 0u;      // (every thread has read its edges' entries: the barrier above)
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int i = 0; i < BS / 64; ++i) {
        const uint32_t v = wsum[i];
        if (i < wave) base += v;
        total += v;
    }
    uint32_t rank = base + inc - hcount;
    uint32_t myrun[IPT];                                 // the run every item of this thread belongs to
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        if (hd[k]) {
            u.s.ustart[rank] = (uint32_t)(tid * IPT + k);
            ++rank;
        }
        myrun[k] = rank - 1u;                            // (item 0 is a head: never underflows for j < nc)
    }
    __syncthreads();
    WSTAMP(4);
    // every contribution's rank inside its run and the structure bits that follow from it (k_wchains adds the chain's bits)
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = tid * IPT + k;
        if (j < nc) {
            const uint32_t st = u.s.ustart[myrun[k]];
            const uint32_t en = (myrun[k] + 1 < total) ? u.s.ustart[myrun[k] + 1] : (uint32_t)nc;
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
    // ---- the runs: the batch's segment of H1 in LDS (slot = run + 1; the runs' nodes are distinct: a claim needs no key
    // compare), the node's entry of the chain table marked with this batch
    for (uint32_t r = tid; r < total; r += BS) {
        const uint32_t st = u.s.ustart[r];
        const uint32_t node = u.s.key[st];
        uint32_t i = hmix(hkey(node, (uint32_t)bb)) % SEG;
        for (;;) {                                       // claim the first empty 16-bit slot from i on (two slots per LDS word)
            const uint32_t sh = (i & 1u) * 16u;
            uint32_t old = e_all[i >> 1];
            bool done = false;
            while (((old >> sh) & 0xFFFFu) == 0u) {
                const uint32_t seen = atomicCAS(&e_all[i >> 1], old, old | ((r + 1u) << sh));
                if (seen == old) { done = true; break; }
                old = seen;                              // (the word's other slot was claimed meanwhile, or this one)
            }
            if (done) break;
            i = (i + 1 == SEG) ? 0u : i + 1;
        }
        cmark(q, node, (uint32_t)(bb / KW), (unsigned)(bb % KW));
    }
    __syncthreads();
    {
        HEnt* __restrict__ S = q.h1 + (size_t)bb * SEG;
        for (uint32_t i = tid; i < SEG; i += BS) {
            const uint32_t v = (e_all[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu;
            uint4 e = make_uint4(0u, 0u, 0u, 0u);
            if (v) {
                const uint32_t r = v - 1u;
                const uint32_t st = u.s.ustart[r];
                const uint32_t en = (r + 1 < total) ? u.s.ustart[r + 1] : (uint32_t)nc;
                const unsigned long long k1 = hkey(u.s.key[st], (uint32_t)bb);
                e = make_uint4((uint32_t)k1, (uint32_t)(k1 >> 32), (uint32_t)(g0 + st), en - st);
            }
            *reinterpret_cast<uint4*>(S + i) = e;
        }
    }
    WSTAMP(5);
    if (tid == 0) {
        BatchDesc D;
        D.e0 = e0;
        D.ne = B;
        D.pad = 0;
        D.t_last = t_last;
        D.now = (bb == 0) ? (t_prev ? *t_prev : now_time) : t_c[e0 - 1];   // clock left by the previous batch (TPNet.py:99)
        D.n_light = 0;
        D.n_heavy = 0;
        // (the dense decay's factors are read by the eager mode only, which never takes the windowed schedule: no exp / pow here --
        // ~2 us of this one thread's time at the tail of the workgroup)
        for (int i = 0; i < TPNET_MAX_LAYERS; ++i) D.decay[i] = 1.0f;
        desc[bb] = D;
    }
    WSTAMP(6);
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
            D.pad0 = D.pad1 = D.pad2 = 0;
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
    const int64_t nc = 2 * Ec;
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
