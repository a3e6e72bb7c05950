// Device-side 'recent' neighbour sampler: the step in front of the hot path.  The reference builds
// per-node time-sorted adjacency lists on the host and, per query, runs np.searchsorted + a last-K slice in a Python
// loop over the batch (utils/utils.py:140-152, 160-224, 293-312).  Here: one CSR of the undirected interaction graph
// (stable sort by (node, time), ties in the reference's append order) and one thread per (query, slot).
#include "tpnet_common.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace tpnet {

static inline size_t al(size_t x) { return (x + 255) / 256 * 256; }

// order-preserving map of an IEEE double onto uint64
__device__ __forceinline__ uint64_t time_key(double t) {
    const uint64_t b = (uint64_t)__double_as_longlong(t);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

// entry 2e = (node src[e] -> neighbour dst[e]), entry 2e+1 = (node dst[e] -> neighbour src[e]): the reference's
// append order (utils/utils.py:307-309)
__global__ void k_adj_keys(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                           const double* __restrict__ t, int64_t E, uint64_t* __restrict__ tkey,
                           uint32_t* __restrict__ idx) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < 2 * E; j += (int64_t)gridDim.x * blockDim.x) {
        tkey[j] = time_key(t[j >> 1]);
        idx[j] = (uint32_t)j;
    }
}

__global__ void k_adj_nodekeys(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E,
                               const uint32_t* __restrict__ idx, uint32_t* __restrict__ nkey) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < 2 * E; j += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t o = idx[j];
        nkey[j] = (uint32_t)((o & 1u) ? dst[o >> 1] : src[o >> 1]);
    }
}

// sorted order -> CSR payload (neighbour, edge id, time) and row starts
__global__ void k_adj_fill(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                           const double* __restrict__ t, const int64_t* __restrict__ eid, int64_t E,
                           const uint32_t* __restrict__ nkey_sorted, const uint32_t* __restrict__ idx_sorted,
                           int64_t num_nodes, int64_t* __restrict__ row_start, int32_t* __restrict__ nbr,
                           int64_t* __restrict__ nbr_eid, double* __restrict__ nbr_t) {
    const int64_t n2 = 2 * E;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n2; j += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t o = idx_sorted[j];
        const int64_t e = o >> 1;
        nbr[j] = (int32_t)((o & 1u) ? src[e] : dst[e]);
        nbr_eid[j] = eid ? eid[e] : e + 1;
        nbr_t[j] = t[e];
        const uint32_t k = nkey_sorted[j];
        const uint32_t kp = j ? nkey_sorted[j - 1] : 0xFFFFFFFFu;
        if (j == 0 || k != kp) {
            // nodes (kp, k] start here (empty lists in between)
            const int64_t first = (j == 0) ? 0 : (int64_t)kp + 1;
            for (int64_t q = first; q <= (int64_t)k && q <= num_nodes; ++q) row_start[q] = j;
        }
        if (j == n2 - 1)
            for (int64_t q = (int64_t)k + 1; q <= num_nodes; ++q) row_start[q] = n2;
    }
}

__global__ void k_adj_empty(int64_t num_nodes, int64_t* __restrict__ row_start) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q <= num_nodes; q += (int64_t)gridDim.x * blockDim.x)
        row_start[q] = 0;
}

// one thread per (query, slot): cut = first interaction of the node with time >= query time (searchsorted left,
// utils/utils.py:152); the K most recent before it go to the BACK of the row, the front is padded with 0 (:211-219)
__global__ void k_sample_recent(const int64_t* __restrict__ row_start, const int32_t* __restrict__ nbr,
                                const int64_t* __restrict__ nbr_eid, const double* __restrict__ nbr_t,
                                int64_t num_nodes, const int64_t* __restrict__ node_ids,
                                const double* __restrict__ times, int64_t n, int K, int64_t* __restrict__ out_ids,
                                int64_t* __restrict__ out_eids, double* __restrict__ out_t) {
    const int64_t tot = n * K;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < tot; x += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = x / K;
        const int k = (int)(x - i * K);
        const int64_t nid = node_ids[i];
        int64_t id = 0, ee = 0;
        double tt = 0.0;
        if ((uint64_t)nid < (uint64_t)num_nodes) {
            const int64_t lo0 = row_start[nid], hi0 = row_start[nid + 1];
            const double q = times[i];
            int64_t lo = lo0, hi = hi0;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (nbr_t[mid] < q) lo = mid + 1; else hi = mid;
            }
            const int64_t cut = lo;
            const int64_t pos = cut - K + k;          // slot k of the K-wide window ending at cut
            if (pos >= lo0) {
                id = nbr[pos];
                ee = nbr_eid[pos];
                tt = nbr_t[pos];
            }
        }
        out_ids[x] = id;
        if (out_eids) out_eids[x] = ee;
        if (out_t) out_t[x] = tt;
    }
}

// the encoder's rows for one (src, other) batch (models/TPNet.py:280-316): nodes = [src; other], times = tile(t, 2), and the
// two anchors of row i = the edge's endpoints (src[i % B], other[i % B])
__global__ void k_encoder_rows(const int64_t* __restrict__ src, const int64_t* __restrict__ other, const double* __restrict__ t,
                               int64_t B, int64_t* __restrict__ nodes, double* __restrict__ t2, int64_t* __restrict__ a1,
                               int64_t* __restrict__ a2) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * B; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i < B ? i : i - B;
        const int64_t s = src[e], o = other[e];
        nodes[i] = i < B ? s : o;
        t2[i] = t[e];
        a1[i] = s;
        a2[i] = o;
    }
}

// the two kernels above in ONE launch for the encoder's call (round 3): thread (row i, slot k) derives the row's node and time
// from the batch itself (three loads that a wave shares: its 64 threads span 3-4 rows), slot 0 also writes the row's record
// (node, time, anchors); then the sampler's cut and window as in k_sample_recent.  One launch less per encoder call (~5 us each).
__global__ void k_encoder_sample(const int64_t* __restrict__ row_start, const int32_t* __restrict__ nbr,
                                 const double* __restrict__ nbr_t, int64_t num_nodes, const int64_t* __restrict__ src,
                                 const int64_t* __restrict__ other, const double* __restrict__ t, int64_t B, int K,
                                 int64_t* __restrict__ nodes, double* __restrict__ t2, int64_t* __restrict__ a1,
                                 int64_t* __restrict__ a2, int64_t* __restrict__ out_ids, int64_t* __restrict__ pu,
                                 int64_t* __restrict__ pv) {
    const int64_t tot = 2 * B * K;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < tot; x += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = x / K;
        const int k = (int)(x - i * K);
        const int64_t e = i < B ? i : i - B;
        const int64_t sn = src[e], on = other[e];
        const double q = t[e];
        const int64_t nid = i < B ? sn : on;
        if (k == 0) {
            nodes[i] = nid;
            t2[i] = q;
            a1[i] = sn;
            a2[i] = on;
        }
        int64_t id = 0;
        if ((uint64_t)nid < (uint64_t)num_nodes) {
            const int64_t lo0 = row_start[nid], hi0 = row_start[nid + 1];
            int64_t lo = lo0, hi = hi0;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (nbr_t[mid] < q) lo = mid + 1; else hi = mid;
            }
            const int64_t pos = lo - K + k;           // slot k of the K-wide window ending at the cut
            if (pos >= lo0) id = nbr[pos];
        }
        out_ids[x] = id;
        if (pu) {
            // narrow rows: the call's pairs spelled out, (neighbour, src) then (neighbour, other) -- get_pair_wise_feature's own
            // order (models/TPNet.py:311-316) -- for the generic readout, which beats the anchored walk on rows of <= 128 floats
            pu[x] = id;
            pu[tot + x] = id;
            pv[x] = sn;
            pv[tot + x] = on;
        }
    }
}

}  // namespace tpnet

using namespace tpnet;

// rows of <= 128 floats: the generic pair kernel's 16-lane geometry reads an encoder-style list faster than the anchored walk
// (tools/encoder_readout.py, 80 000 pairs at d=128: 27.1 against 31.3 us; 16 000 at d=64: 7.1 against 13.2) -- the encoder's calls
// take it there, from pair lists the sampler kernel writes beside the neighbour ids
bool tpnet::encoder_generic_readout(const tpnet_state& st) {
    static const int off = TPNET_DEV_INT(ENCODER_ANCHORED_ONLY, 0);
    return !off && st.d <= 128;
}

extern "C" {

size_t tpnet_sampler_bytes(int64_t E, int64_t num_nodes) {
    const size_t n2 = 2 * (size_t)(E < 1 ? 1 : E);
    size_t tmp1 = 0, tmp2 = 0;
    uint64_t* k64 = nullptr;
    uint32_t* k32 = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, tmp1, k64, k64, k32, k32, n2, 0, 64, (hipStream_t)0, false);
    (void)rocprim::radix_sort_pairs(nullptr, tmp2, k32, k32, k32, k32, n2, 0, 32, (hipStream_t)0, false);
    const size_t tmp = tmp1 > tmp2 ? tmp1 : tmp2;
    return al((size_t)(num_nodes + 1) * 8) + al(n2 * 4) + al(n2 * 8) * 2          // CSR
           + al(n2 * 8) * 2 + al(n2 * 4) * 4 + al(tmp) + 256;                      // build scratch
}

// Layout of the sampler buffer (caller-owned device memory of tpnet_sampler_bytes): CSR first, scratch behind it.
struct SamplerView {
    int64_t* row_start; int32_t* nbr; int64_t* nbr_eid; double* nbr_t;
    uint64_t* tk_a; uint64_t* tk_b; uint32_t* ix_a; uint32_t* ix_b; uint32_t* nk_a; uint32_t* nk_b; void* tmp; size_t tmp_bytes;
};
static SamplerView carve(void* buf, int64_t E, int64_t num_nodes) {
    const size_t n2 = 2 * (size_t)(E < 1 ? 1 : E);
    char* p = (char*)(((size_t)buf + 255) / 256 * 256);
    SamplerView v;
    auto take = [&](size_t b) { void* r = p; p += al(b); return r; };
    v.row_start = (int64_t*)take((size_t)(num_nodes + 1) * 8);
    v.nbr = (int32_t*)take(n2 * 4);
    v.nbr_eid = (int64_t*)take(n2 * 8);
    v.nbr_t = (double*)take(n2 * 8);
    v.tk_a = (uint64_t*)take(n2 * 8);
    v.tk_b = (uint64_t*)take(n2 * 8);
    v.ix_a = (uint32_t*)take(n2 * 4);
    v.ix_b = (uint32_t*)take(n2 * 4);
    v.nk_a = (uint32_t*)take(n2 * 4);
    v.nk_b = (uint32_t*)take(n2 * 4);
    v.tmp = p;
    v.tmp_bytes = 0;
    return v;
}

int tpnet_sampler_build(void* sampler, size_t sampler_bytes, const int64_t* src, const int64_t* dst, const double* t,
                        const int64_t* edge_ids, int64_t E, int64_t num_nodes, void* stream) {
    if (!sampler || E < 0 || num_nodes < 1 || num_nodes >= (1ll << 31) || (E > 0 && (!src || !dst || !t)))
        return TPNET_ERR_BAD_ARG;
    if (E >= (1ll << 30)) return TPNET_ERR_BAD_ARG;
    if (sampler_bytes < tpnet_sampler_bytes(E, num_nodes)) return TPNET_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    SamplerView v = carve(sampler, E, num_nodes);
    if (E == 0) {
        hipLaunchKernelGGL(k_adj_empty, dim3(64), dim3(256), 0, s, num_nodes, v.row_start);
        TPNET_HIP_TRY(hipGetLastError());
        return TPNET_OK;
    }
    const size_t n2 = 2 * (size_t)E;
    int grid = (int)((n2 + 255) / 256);
    if (grid > 4096) grid = 4096;
    // stable sort by time, then stable sort by node  ==  sort by (node, time) with ties in append order
    hipLaunchKernelGGL(k_adj_keys, dim3(grid), dim3(256), 0, s, src, dst, t, E, v.tk_a, v.ix_a);
    size_t tmp = sampler_bytes - (size_t)((char*)v.tmp - (char*)sampler);
    TPNET_HIP_TRY(rocprim::radix_sort_pairs(v.tmp, tmp, v.tk_a, v.tk_b, v.ix_a, v.ix_b, n2, 0u, 64u, s, false));
    hipLaunchKernelGGL(k_adj_nodekeys, dim3(grid), dim3(256), 0, s, src, dst, E, v.ix_b, v.nk_a);
    int bits = 1;
    while ((1ll << bits) < num_nodes && bits < 32) ++bits;
    tmp = sampler_bytes - (size_t)((char*)v.tmp - (char*)sampler);
    TPNET_HIP_TRY(rocprim::radix_sort_pairs(v.tmp, tmp, v.nk_a, v.nk_b, v.ix_b, v.ix_a, n2, 0u, (unsigned)bits, s, false));
    hipLaunchKernelGGL(k_adj_fill, dim3(grid), dim3(256), 0, s, src, dst, t, edge_ids, E, v.nk_b, v.ix_a, num_nodes,
                       v.row_start, v.nbr, v.nbr_eid, v.nbr_t);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int tpnet_sample_recent(const void* sampler, int64_t E, int64_t num_nodes, const int64_t* node_ids, const double* times,
                        int64_t n, int32_t K, int64_t* out_ids, int64_t* out_eids, double* out_times, void* stream) {
    if (!sampler || n < 0 || K < 1 || num_nodes < 1 || (n > 0 && (!node_ids || !times || !out_ids)))
        return TPNET_ERR_BAD_ARG;
    if (n == 0) return TPNET_OK;
    SamplerView v = carve(const_cast<void*>(sampler), E, num_nodes);
    int64_t tot = n * (int64_t)K;
    int grid = (int)((tot + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_sample_recent, dim3(grid), dim3(256), 0, (hipStream_t)stream, v.row_start, v.nbr, v.nbr_eid,
                       v.nbr_t, num_nodes, node_ids, times, n, (int)K, out_ids, out_eids, out_times);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

size_t tpnet_encoder_scratch_bytes(int64_t B, int32_t K) {
    if (B < 0) B = 0;
    if (K < 1) K = 1;
    return (size_t)(8 * B + 10 * B * (int64_t)K) * 8 + 256;       // (+ the spelled-out pair lists of narrow rows: 2 x 4 B K ids)
}

int tpnet_encoder_rows(const tpnet_state* st, const void* sampler, int64_t E, int64_t num_nodes, const int64_t* src,
                       const int64_t* other, const double* t, int64_t B, int32_t K, void* scratch, size_t scratch_bytes,
                       void* stream) {
    if (!st || !sampler || B < 0 || K < 1 || num_nodes < 1 || (B > 0 && (!src || !other || !t || !scratch))) return TPNET_ERR_BAD_ARG;
    if (B == 0) return TPNET_OK;
    if (scratch_bytes < tpnet_encoder_scratch_bytes(B, K)) return TPNET_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int64_t* nodes = (int64_t*)(((size_t)scratch + 255) / 256 * 256);
    double* t2 = (double*)(nodes + 2 * B);
    int64_t* a1 = (int64_t*)(t2 + 2 * B);
    int64_t* a2 = a1 + 2 * B;
    int64_t* neigh = a2 + 2 * B;
    static const int two = TPNET_DEV_INT(ENCODER_TWO_LAUNCHES, 0);     // developer override: row set-up and sampler apart
    if (two) {
        int g0 = (int)((2 * B + 255) / 256);
        if (g0 > 1024) g0 = 1024;
        hipLaunchKernelGGL(k_encoder_rows, dim3(g0), dim3(256), 0, s, src, other, t, B, nodes, t2, a1, a2);
        return tpnet_sample_recent(sampler, E, num_nodes, nodes, t2, 2 * B, K, neigh, nullptr, nullptr, stream);
    }
    SamplerView v = carve(const_cast<void*>(sampler), E, num_nodes);
    const int64_t tot = 2 * B * (int64_t)K;
    int grid = (int)((tot + 255) / 256);
    if (grid > 8192) grid = 8192;
    int64_t* pu = (encoder_generic_readout(*st) && !encoder_mfma_supported(*st, 2 * B, K)) ? neigh + 2 * B * (int64_t)K : nullptr;
    hipLaunchKernelGGL(k_encoder_sample, dim3(grid), dim3(256), 0, s, v.row_start, v.nbr, v.nbr_t, num_nodes, src, other, t, B, (int)K,
                       nodes, t2, a1, a2, neigh, pu, pu ? pu + 4 * B * (int64_t)K : nullptr);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int tpnet_encoder_gram(const tpnet_state* st, const void* sampler, int64_t E, int64_t num_nodes, const int64_t* src,
                       const int64_t* other, const double* t, int64_t B, int32_t K, double now_time, double lambda,
                       uint32_t flags, void* scratch, size_t scratch_bytes, float* out, void* stream) {
    if (!st || !out) return TPNET_ERR_BAD_ARG;
    if (flags & TPNET_FLAG_PACKED) return TPNET_ERR_BAD_ARG;
    if (B == 0) return TPNET_OK;
    if (!tpnet_pair_gram_anchored_supported(st)) return TPNET_ERR_BAD_ARG;
    int rc = tpnet_encoder_rows(st, sampler, E, num_nodes, src, other, t, B, K, scratch, scratch_bytes, stream);
    if (rc) return rc;
    int64_t* nodes = (int64_t*)(((size_t)scratch + 255) / 256 * 256);
    int64_t* a1 = nodes + 4 * B;
    int64_t* a2 = a1 + 2 * B;
    int64_t* neigh = a2 + 2 * B;
    const int NN = 2 * st->L + 2;
    if (encoder_generic_readout(*st) && !encoder_mfma_supported(*st, 2 * B, K)) {
        const int64_t half = 2 * B * (int64_t)K;
        return tpnet_pair_gram(st, neigh + half, neigh + 3 * half, 2 * half, now_time, lambda, flags, out, stream);
    }
    return tpnet_pair_gram_anchored(st, neigh, a1, a2, 2 * B, K, now_time, lambda, flags, out,
                                    out + (size_t)(2 * B) * (size_t)K * (size_t)(NN * NN), stream);
}

}  // extern "C"
