// Shared by the planners of a windowed chunk that sort every batch in LDS (wplan3.hip: the hashed planner, wplan_dense.hip:
// the dense planner): scratch views of the workspace, the runs' / chains' hash tables of the hashed planner, and the per-batch
// sort itself (one workgroup per batch).
#pragma once
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
static inline uint32_t wplan3_bpb(int64_t batch) { return (uint32_t)((2 * batch + 255) / 256); }

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
struct WSortShared {
    using Sort = rocprim::block_radix_sort<uint32_t, BS, IPT, uint32_t>;
    static constexpr int NC = BS * IPT;
    union U {
        typename Sort::storage_type sort;
        struct { uint32_t key[NC]; uint32_t ustart[NC]; } s;     // sorted keys; then the heads of the distinct targets' runs
    } u;
    // per edge: endpoints (0 if out of range) | bit 31: the EDGE has a bad endpoint; time weight.  Dead once the sorted arrays are
    // stored: the batch's segment of H1 is then built in the same 3 * NC / 2 words
    uint32_t e_all[3 * (NC / 2)];
    uint32_t wsum[BS / 64];
};

// DENSE = the dense planner's variant: instead of the batch's segment of H1 and the marks in the chain table, the batch's ROW of
// the run-length matrix (lenrow[n] = contributions to node n in this batch; every entry written: nothing to zero beforehand)
template <int BS, int IPT, bool DENSE>
__device__ __forceinline__ void wsort_batch(WSortShared<BS, IPT>& sh, const WTmp& q, BatchDesc* __restrict__ desc,
                                            const int64_t* __restrict__ src_c, const int64_t* __restrict__ dst_c,
                                            const double* __restrict__ t_c, int64_t Ec, int32_t Bfull, int64_t N, int node_bits,
                                            double now_time, const double* __restrict__ t_prev, double lambda, int L, uint32_t* err,
                                            int KW, int64_t bb, uint16_t* __restrict__ lenrow) {
    const int64_t e0 = bb * Bfull;
    const int32_t B = (int32_t)((Ec - e0 < Bfull) ? (Ec - e0) : Bfull);
    const int64_t* __restrict__ src = src_c + e0;
    const int64_t* __restrict__ dst = dst_c + e0;
    const double* __restrict__ t = t_c + e0;
    const int64_t g0 = 2 * e0;
    using Sort = typename WSortShared<BS, IPT>::Sort;
    constexpr int NC = BS * IPT;
    auto& u = sh.u;
    uint32_t* const e_all = sh.e_all;
    uint32_t* const wsum = sh.wsum;
    uint32_t* const e_src = e_all;
    uint32_t* const e_dst = e_all + NC / 2;
    float* const e_w = reinterpret_cast<float*>(e_all + NC);
    constexpr uint32_t SEG = 3 * NC;                  // slots of 16 bits: at most a third full (a probe that walks is a dependent
                                                      // round trip for every reader of the table: short clusters matter more than bytes)
    static_assert(sizeof(WSortShared<BS, IPT>) <= 64 * 1024, "k_wsort: LDS budget");
    static_assert(NC <= 65535 / 2 * 2, "k_wsort: run indices");
    const int tid = threadIdx.x;
    const int nc = 2 * B;
    WSTAMP(0);
    if constexpr (DENSE) {
        // (16-bit stores, coalesced; the barriers between here and the scatter of the runs' lengths order the two)
        for (int64_t n = tid; n < N; n += BS) lenrow[n] = 0;
    }
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
    if constexpr (!DENSE)
        for (uint32_t i = tid; i < SEG / 2; i += BS) e_all[i] = 0u;  // (every thread has read its edges' entries: the barrier above)
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
    if constexpr (DENSE) {
        // ---- the runs: this batch's row of the run-length matrix (zeroed by this workgroup before its first barrier)
        for (uint32_t r = tid; r < total; r += BS) {
            const uint32_t st = u.s.ustart[r];
            const uint32_t en = (r + 1 < total) ? u.s.ustart[r + 1] : (uint32_t)nc;
            lenrow[u.s.key[st]] = (uint16_t)(en - st);
        }
    } else {
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

}  // namespace tpnet
