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

}  // extern "C"
