// Exchange plan of the row-sharded stream with the TARGETED exchange (tpnet_amd/sharded.py), on the device.
// Rank `me` of G reads, in a batch: dst and neg of the edges whose src it owns, and src of the edges whose dst it owns (owner(n) =
// n % G).  Every rank derives every list from the stream, so no request round is needed -- but the torch version of the plan is ~40
// small launches and three host synchronisations per run_stream call (~1 ms: most of a 20-step multi-GPU run).  Here:
//   k_xplan     one workgroup per batch, two passes over the batch's (reader, node) items, each a compaction into LDS, a block radix
//               sort and a distinct-scan:  pass 0 = what THIS rank receives (reader == me, owner != me), ordered (owner, node) -- the
//               order in which the peers' messages land in the halo rows;  pass 1 = what this rank SENDS (owner == me, reader != me),
//               ordered (reader, node) -- the pack order, a row once per reader.  Both orders are functions of the batch alone, so
//               sender and receiver agree element for element.
//   k_xrelabel  one thread per edge endpoint: owned -> n / G; received in the edge's batch -> its halo row (binary search in the
//               batch's receive list); anything else -> the first halo row (never dereferenced by a unit this rank computes).
// The host reads back 2 * G counts per batch (the message sizes of the grouped ncclSend / ncclRecv) and two status words.
#include "tpnet_common.h"

#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace tpnet {

static constexpr int XBS = 1024, XIPT = 8, XNC = XBS * XIPT;      // items of one class a batch may have (else: status[1], fallback)

struct XArgs {
    const int64_t* src;
    const int64_t* dst;
    const int64_t* neg;          // may be NULL
    int64_t E, B, N;
    int32_t G, me, nbits;        // nbits: bits of a node id inside a key (minor * 2^nbits + node < 2^31)
    int32_t sort_bits;           // bits of a key
    uint32_t* recv_keys;         // [nb][XNC] (owner << nbits | node), ascending: the batch's receive list
    int64_t* pack_ids;           // [nb][XNC] local rows (node / G) to pack, ordered (reader, node)
    int64_t* cnt;                // [nb][2][G]: rows received from each owner; rows sent to each reader
    uint32_t* status;            // [0] ids out of range, [1] batches whose lists did not fit
};

__global__ __launch_bounds__(XBS) void k_xplan(XArgs a) {
    using Sort = rocprim::block_radix_sort<uint32_t, XBS, XIPT>;
    __shared__ union U {
        typename Sort::storage_type sort;
        uint32_t key[XNC];
    } u;
    __shared__ uint32_t n_items, wsum[XBS / 64], lcnt[64];
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int64_t e0 = b * a.B;
    const int64_t ne = (a.E - e0 < a.B) ? (a.E - e0) : a.B;
    const int nk = a.neg ? 3 : 2;
    const uint32_t nmask = (a.nbits >= 32) ? 0xFFFFFFFFu : ((1u << a.nbits) - 1u);
    for (int pass = 0; pass < 2; ++pass) {
        if (tid == 0) n_items = 0;
        if (tid < 64) lcnt[tid] = 0;
        __syncthreads();
        for (int64_t i = tid; i < nk * ne; i += XBS) {
            const int kind = (int)(i / ne);
            const int64_t e = e0 + (i - (int64_t)kind * ne);
            const int64_t s = a.src[e], d = a.dst[e];
            const int64_t g = (kind == 2) ? a.neg[e] : 0;
            const bool ok = (uint64_t)s < (uint64_t)a.N && (uint64_t)d < (uint64_t)a.N && (uint64_t)g < (uint64_t)a.N;
            if (!ok) {
                if (pass == 0 && kind != 1) atomicAdd(a.status, 1u);     // (an edge's src / dst once, its negative once)
                continue;
            }
            const int64_t rn = (kind == 1) ? d : s;                      // the node whose owner reads ...
            const int64_t nd = (kind == 0) ? d : (kind == 1 ? s : g);    // ... this node
            const int32_t reader = (int32_t)(rn % a.G), owner = (int32_t)(nd % a.G);
            const bool keep = pass == 0 ? (reader == a.me && owner != a.me) : (owner == a.me && reader != a.me);
            if (keep) {
                const uint32_t k = atomicAdd(&n_items, 1u);
                if (k < (uint32_t)XNC) u.key[k] = ((uint32_t)(pass == 0 ? owner : reader) << a.nbits) | (uint32_t)nd;
            }
        }
        __syncthreads();
        const uint32_t nall = n_items;
        const uint32_t n = nall < (uint32_t)XNC ? nall : (uint32_t)XNC;
        if (tid == 0 && nall > (uint32_t)XNC) atomicAdd(a.status + 1, 1u);
        uint32_t keys[XIPT];
#pragma unroll
        for (int k = 0; k < XIPT; ++k) {
            const uint32_t j = (uint32_t)(tid * XIPT + k);
            keys[k] = j < n ? u.key[j] : 0xFFFFFFFFu;
        }
        __syncthreads();                                     // (the list is in registers: its LDS is the sort's now)
        Sort().sort(keys, u.sort, 0u, (unsigned)a.sort_bits);  // (stable: the padding stays behind an equal real key)
        __syncthreads();
#pragma unroll
        for (int k = 0; k < XIPT; ++k) u.key[tid * XIPT + k] = keys[k];
        __syncthreads();
        // distinct items, ranked by a block-wide exclusive scan of the head flags
        uint32_t hcount = 0;
        bool hd[XIPT];
#pragma unroll
        for (int k = 0; k < XIPT; ++k) {
            const uint32_t j = (uint32_t)(tid * XIPT + k);
            hd[k] = j < n && (j == 0 || u.key[j - 1] != keys[k]);
            hcount += hd[k] ? 1u : 0u;
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
        uint32_t rank = inc - hcount;
#pragma unroll
        for (int i = 0; i < XBS / 64; ++i)
            if (i < wave) rank += wsum[i];
#pragma unroll
        for (int k = 0; k < XIPT; ++k) {
            if (hd[k]) {
                const uint32_t minor = keys[k] >> a.nbits;   // owner (pass 0) / reader (pass 1): < G <= 64
                atomicAdd(&lcnt[minor & 63u], 1u);
                if (pass == 0) a.recv_keys[b * XNC + rank] = keys[k];
                else a.pack_ids[b * XNC + rank] = (int64_t)((keys[k] & nmask) / (uint32_t)a.G);
                ++rank;
            }
        }
        __syncthreads();
        if (tid < a.G) a.cnt[(b * 2 + pass) * a.G + tid] = (int64_t)lcnt[tid];
        __syncthreads();
    }
}

__global__ void k_xrelabel(XArgs a, int32_t n_cap, int64_t* __restrict__ lsrc, int64_t* __restrict__ ldst,
                           int64_t* __restrict__ lneg) {
    const int nk = a.neg ? 3 : 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nk * a.E; i += (int64_t)gridDim.x * blockDim.x) {
        const int kind = (int)(i / a.E);
        const int64_t e = i - (int64_t)kind * a.E;
        const int64_t nd = kind == 0 ? a.src[e] : (kind == 1 ? a.dst[e] : a.neg[e]);
        int64_t loc = n_cap;
        if ((uint64_t)nd < (uint64_t)a.N) {
            const int32_t owner = (int32_t)(nd % a.G);
            if (owner == a.me) {
                loc = nd / a.G;
            } else {
                const int64_t b = e / a.B;
                int64_t nr = 0;
                for (int o = 0; o < a.G; ++o) nr += a.cnt[(b * 2) * a.G + o];
                if (nr > XNC) nr = XNC;
                const uint32_t key = ((uint32_t)owner << a.nbits) | (uint32_t)nd;
                const uint32_t* __restrict__ lst = a.recv_keys + b * XNC;
                int64_t lo = 0, hi = nr;                      // first position with lst[pos] >= key
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (lst[mid] < key) lo = mid + 1; else hi = mid;
                }
                if (lo < nr && lst[lo] == key) loc = n_cap + lo;
            }
        }
        (kind == 0 ? lsrc : (kind == 1 ? ldst : lneg))[e] = loc;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same plan for batches whose lists do not fit one workgroup's LDS (C4's law over 8 ranks: 25 000 rows received per
// batch of 80 000 edges): the (list, batch, peer, node) keys of the whole call appended to one array (wave-aggregated
// atomics; the order is restored by the sort), ONE device-wide radix sort, head flags + scan + compaction = the distinct keys,
// counts and batch starts by binary search, relabelling by binary search in the batch's segment.  list 0 = what this rank
// receives, ordered (batch, owner, node); list 1 = what it sends, ordered (batch, reader, node): the same orders as k_xplan's.
// ---------------------------------------------------------------------------------------------------------------------
struct XLArgs {
    const int64_t* src;
    const int64_t* dst;
    const int64_t* neg;
    int64_t E, B, N, nb;
    int32_t G, me, nbits, gbits, bbits;
    unsigned long long* keys;        // [cap] appended keys
    unsigned long long* sorted;      // [cap]
    uint32_t* heads;                 // [cap] head flags, then (in place) their exclusive scan
    unsigned long long* uniq;        // [cap] distinct keys
    uint32_t* counter;               // [0] appended keys; [1] distinct keys
    uint32_t* off;                   // [2 * nb * G + 1] first distinct key of every (list, batch, peer)
    uint32_t* status;
};

__device__ __forceinline__ unsigned long long xl_key(const XLArgs& a, int list, int64_t b, int32_t minor, int64_t node) {
    return ((unsigned long long)list << (a.bbits + a.gbits + a.nbits)) | ((unsigned long long)b << (a.gbits + a.nbits)) |
           ((unsigned long long)minor << a.nbits) | (unsigned long long)node;
}

__global__ __launch_bounds__(256) void k_xl_keys(XLArgs a) {
    const int nk = a.neg ? 3 : 2;
    const int lane = threadIdx.x & 63;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < nk * a.E; i0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = i0 + threadIdx.x;
        unsigned long long k0 = 0, k1 = 0;
        bool h0 = false, h1 = false;
        if (i < nk * a.E) {
            const int kind = (int)(i / a.E);
            const int64_t e = i - (int64_t)kind * a.E;
            const int64_t s = a.src[e], d = a.dst[e];
            const int64_t g = (kind == 2) ? a.neg[e] : 0;
            const bool ok = (uint64_t)s < (uint64_t)a.N && (uint64_t)d < (uint64_t)a.N && (uint64_t)g < (uint64_t)a.N;
            if (!ok) {
                if (kind != 1) atomicAdd(a.status, 1u);                  // (an edge's src / dst once, its negative once)
            } else {
                const int64_t rn = (kind == 1) ? d : s;                  // the node whose owner reads ...
                const int64_t nd = (kind == 0) ? d : (kind == 1 ? s : g);    // ... this node
                const int32_t reader = (int32_t)(rn % a.G), owner = (int32_t)(nd % a.G);
                const int64_t b = e / a.B;
                h0 = reader == a.me && owner != a.me;
                h1 = owner == a.me && reader != a.me;
                if (h0) k0 = xl_key(a, 0, b, owner, nd);
                if (h1) k1 = xl_key(a, 1, b, reader, nd);
            }
        }
        // one atomic per wave and list
        const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
        const uint32_t n0 = (uint32_t)__popcll(m0), n1 = (uint32_t)__popcll(m1);
        uint32_t base = 0;
        if (lane == 0 && n0 + n1) base = atomicAdd(a.counter, n0 + n1);
        base = (uint32_t)__shfl((int)base, 0, 64);
        if (h0) a.keys[base + (uint32_t)__popcll(m0 & ((1ull << lane) - 1ull))] = k0;
        if (h1) a.keys[base + n0 + (uint32_t)__popcll(m1 & ((1ull << lane) - 1ull))] = k1;
    }
}

__global__ __launch_bounds__(256) void k_xl_heads(XLArgs a, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        a.heads[i] = (i == 0 || a.sorted[i - 1] != a.sorted[i]) ? 1u : 0u;
}

// heads[] holds the exclusive scan of the flags now: element i is distinct iff it is the last one or heads[i + 1] != heads[i]
__global__ __launch_bounds__(256) void k_xl_compact(XLArgs a, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const bool head = i == 0 || a.sorted[i - 1] != a.sorted[i];
        if (head) a.uniq[a.heads[i]] = a.sorted[i];
        if (i == n - 1) a.counter[1] = a.heads[i] + (head ? 1u : 0u);
    }
}

__device__ __forceinline__ uint32_t xl_lower(const unsigned long long* __restrict__ u, uint32_t lo, uint32_t hi, unsigned long long key) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (u[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_xl_offsets(XLArgs a) {
    const uint32_t nu = a.counter[1];
    const int64_t nq = 2 * a.nb * a.G;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q <= nq; q += (int64_t)gridDim.x * blockDim.x) {
        if (q == nq) { a.off[q] = nu; continue; }
        const int list = (int)(q / (a.nb * a.G));
        const int64_t r = q - (int64_t)list * a.nb * a.G;
        a.off[q] = xl_lower(a.uniq, 0u, nu, xl_key(a, list, r / a.G, (int32_t)(r % a.G), 0));
    }
}

// counts [nb][2][G], the rows to pack (list 1, in order), every endpoint relabelled
__global__ __launch_bounds__(256) void k_xl_finish(XLArgs a, int32_t n_cap, int64_t* __restrict__ cnt, int64_t* __restrict__ pack_ids,
                                                   int64_t* __restrict__ lsrc, int64_t* __restrict__ ldst, int64_t* __restrict__ lneg) {
    const int nk = a.neg ? 3 : 2;
    const int64_t nq = 2 * a.nb * a.G;
    const uint32_t send0 = a.off[a.nb * a.G], nu = a.off[nq];
    const unsigned long long nmask = (1ull << a.nbits) - 1ull;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = gid; q < nq; q += gsz) {
        const int list = (int)(q / (a.nb * a.G));
        const int64_t r = q - (int64_t)list * a.nb * a.G;
        cnt[((r / a.G) * 2 + list) * a.G + (r % a.G)] = (int64_t)(a.off[q + 1] - a.off[q]);
    }
    for (int64_t j = gid; j < (int64_t)(nu - send0); j += gsz) pack_ids[j] = (int64_t)((a.uniq[send0 + j] & nmask) / (unsigned)a.G);
    for (int64_t i = gid; i < nk * a.E; i += gsz) {
        const int kind = (int)(i / a.E);
        const int64_t e = i - (int64_t)kind * a.E;
        const int64_t nd = kind == 0 ? a.src[e] : (kind == 1 ? a.dst[e] : a.neg[e]);
        int64_t loc = n_cap;
        if ((uint64_t)nd < (uint64_t)a.N) {
            const int32_t owner = (int32_t)(nd % a.G);
            if (owner == a.me) {
                loc = nd / a.G;
            } else {
                const int64_t b = e / a.B;
                const uint32_t lo = a.off[b * a.G], hi = a.off[(b + 1) * a.G];      // the batch's receive list
                const unsigned long long key = xl_key(a, 0, b, owner, nd);
                const uint32_t pos = xl_lower(a.uniq, lo, hi, key);
                if (pos < hi && a.uniq[pos] == key) loc = n_cap + (int64_t)(pos - lo);
            }
        }
        (kind == 0 ? lsrc : (kind == 1 ? ldst : lneg))[e] = loc;
    }
}

static size_t xl_align(size_t x) { return (x + 255) / 256 * 256; }
static size_t xl_sort_tmp(size_t cap) {
    size_t bytes = 0;
    unsigned long long* k = nullptr;
    (void)rocprim::radix_sort_keys(nullptr, bytes, k, k, cap, 0u, 64u, (hipStream_t)0, false);
    size_t b2 = 0;
    uint32_t* h = nullptr;
    (void)rocprim::exclusive_scan(nullptr, b2, h, h, 0u, cap, rocprim::plus<uint32_t>(), (hipStream_t)0, false);
    return xl_align(bytes > b2 ? bytes : b2);
}

}  // namespace tpnet

using namespace tpnet;

extern "C" {

int64_t tpnet_xplan_capacity(void) { return XNC; }

int tpnet_xplan_targeted(const int64_t* src, const int64_t* dst, const int64_t* neg, int64_t E, int64_t batch, int64_t N, int32_t G,
                         int32_t me, int32_t n_owned, uint32_t* recv_keys, int64_t* pack_ids, int64_t* cnt, uint32_t* status,
                         int64_t* local_src, int64_t* local_dst, int64_t* local_neg, void* stream) {
    if (!src || !dst || E < 0 || batch < 1 || N < 1 || G < 1 || G > 64 || me < 0 || me >= G || n_owned < 0) return TPNET_ERR_BAD_ARG;
    if (!recv_keys || !pack_ids || !cnt || !status || !local_src || !local_dst || (neg && !local_neg)) return TPNET_ERR_BAD_ARG;
    if (E == 0) return TPNET_OK;
    int nbits = 1, gbits = 1;
    while (nbits < 63 && (1ll << nbits) < N) ++nbits;
    while ((1 << gbits) < G) ++gbits;
    if (nbits + gbits > 31) return TPNET_ERR_BAD_ARG;        // (keys below 2^31: the caller falls back to its own plan)
    const int64_t nb = (E + batch - 1) / batch;
    if (nb > 0x7FFFFFFF) return TPNET_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    TPNET_HIP_TRY(hipMemsetAsync(status, 0, 2 * sizeof(uint32_t), s));
    XArgs a{src, dst, neg, E, batch, N, G, me, nbits, nbits + gbits, recv_keys, pack_ids, cnt, status};
    hipLaunchKernelGGL(k_xplan, dim3((unsigned)nb), dim3(XBS), 0, s, a);
    const int64_t items = (neg ? 3 : 2) * E;
    int grid = (int)((items + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_xrelabel, dim3(grid), dim3(256), 0, s, a, n_owned, local_src, local_dst, local_neg);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

size_t tpnet_xplan_large_bytes(int64_t E, int64_t batch, int32_t G) {
    if (E < 1 || batch < 1 || G < 1) return 0;
    const size_t cap = 6 * (size_t)E;                           // (an item is at most one key of each list)
    const size_t nb = (size_t)((E + batch - 1) / batch);
    return 3 * xl_align(cap * 8) + xl_align(cap * 4) + xl_align((2 * nb * (size_t)G + 1) * 4) + 256 + xl_sort_tmp(cap) + 256;
}

int tpnet_xplan_targeted_large(const int64_t* src, const int64_t* dst, const int64_t* neg, int64_t E, int64_t batch, int64_t N,
                               int32_t G, int32_t me, int32_t n_owned, void* scratch, size_t scratch_bytes, int64_t* pack_ids,
                               int64_t* cnt, uint32_t* status, int64_t* local_src, int64_t* local_dst, int64_t* local_neg,
                               void* stream) {
    if (!src || !dst || E < 0 || batch < 1 || N < 1 || G < 1 || G > 64 || me < 0 || me >= G || n_owned < 0) return TPNET_ERR_BAD_ARG;
    if (!scratch || !pack_ids || !cnt || !status || !local_src || !local_dst || (neg && !local_neg)) return TPNET_ERR_BAD_ARG;
    if (E == 0) return TPNET_OK;
    if (6 * E >= ((int64_t)1 << 32)) return TPNET_ERR_BAD_ARG;                   // 32-bit positions
    if (scratch_bytes < tpnet_xplan_large_bytes(E, batch, G)) return TPNET_ERR_WORKSPACE;
    const int64_t nb = (E + batch - 1) / batch;
    int nbits = 1, gbits = 1, bbits = 1;
    while (nbits < 62 && (1ll << nbits) < N) ++nbits;
    while ((1 << gbits) < G) ++gbits;
    while (bbits < 62 && (1ll << bbits) < nb) ++bbits;
    if (nbits + gbits + bbits + 1 > 63) return TPNET_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t cap = 6 * (size_t)E;
    char* c = reinterpret_cast<char*>((reinterpret_cast<size_t>(scratch) + 255) / 256 * 256);
    auto take = [&](size_t bytes) { void* r = c; c += xl_align(bytes); return r; };
    XLArgs a{};
    a.src = src; a.dst = dst; a.neg = neg; a.E = E; a.B = batch; a.N = N; a.nb = nb;
    a.G = G; a.me = me; a.nbits = nbits; a.gbits = gbits; a.bbits = bbits;
    a.keys = (unsigned long long*)take(cap * 8);
    a.sorted = (unsigned long long*)take(cap * 8);
    a.uniq = (unsigned long long*)take(cap * 8);
    a.heads = (uint32_t*)take(cap * 4);
    a.off = (uint32_t*)take((2 * (size_t)nb * (size_t)G + 1) * 4);
    a.counter = (uint32_t*)take(8);
    a.status = status;
    void* tmp = c;
    size_t tmp_bytes = xl_sort_tmp(cap);
    TPNET_HIP_TRY(hipMemsetAsync(status, 0, 2 * sizeof(uint32_t), s));
    TPNET_HIP_TRY(hipMemsetAsync(a.counter, 0, 8, s));
    const int64_t items = (neg ? 3 : 2) * E;
    int grid = (int)((items + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_xl_keys, dim3(grid), dim3(256), 0, s, a);
    // how many keys there are decides the size of the sort: ONE small read-back (the caller reads the counts back afterwards anyway)
    uint32_t n_keys = 0;
    TPNET_HIP_TRY(hipMemcpyAsync(&n_keys, a.counter, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    TPNET_HIP_TRY(hipStreamSynchronize(s));
    if (n_keys > cap) return TPNET_ERR_WORKSPACE;
    if (n_keys > 0) {
        size_t tb = tmp_bytes;
        TPNET_HIP_TRY(rocprim::radix_sort_keys(tmp, tb, a.keys, a.sorted, (size_t)n_keys, 0u, (unsigned)(nbits + gbits + bbits + 1), s, false));
        int g2 = (int)((n_keys + 255) / 256);
        if (g2 > 8192) g2 = 8192;
        hipLaunchKernelGGL(k_xl_heads, dim3(g2), dim3(256), 0, s, a, n_keys);
        tb = tmp_bytes;
        TPNET_HIP_TRY(rocprim::exclusive_scan(tmp, tb, a.heads, a.heads, 0u, (size_t)n_keys, rocprim::plus<uint32_t>(), s, false));
        hipLaunchKernelGGL(k_xl_compact, dim3(g2), dim3(256), 0, s, a, n_keys);
    }
    const int64_t nq = 2 * nb * G + 1;
    int g3 = (int)((nq + 255) / 256);
    if (g3 > 4096) g3 = 4096;
    hipLaunchKernelGGL(k_xl_offsets, dim3(g3), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_xl_finish, dim3(grid), dim3(256), 0, s, a, n_owned, cnt, pack_ids, local_src, local_dst, local_neg);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

}  // extern "C"
