// Batch planning for the update: group the 2*B (target <- partner) contributions of every batch by target node,
// keeping the reference's summation order inside a target (models/TPNet.py:93-96: all src-side scatter-adds in edge
// order, then all dst-side ones).  Done for a whole chunk of the stream at once (one stable device radix sort over
// (batch, target) keys), so the per-batch step kernels find ready-made item lists and the cost is amortised.
#include "tpnet_common.h"

#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace tpnet {

static constexpr uint32_t HEAVY_THRESHOLD = 8u;  // contributions per target above which an item is "heavy"

static inline int ceil_log2_u64(uint64_t x) {
    int b = 0;
    while (b < 63 && (1ull << b) < x) ++b;
    return b;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// A batch's 2*B keys are contiguous (k_make_keys), so grouping by target is a SEGMENTED sort on the node bits alone, one
// segment per batch: rocPRIM sorts a segment of a few thousand keys inside one workgroup's LDS, all batches in one
// launch, instead of several device-wide passes over (batch, node) keys.  Very long segments (one workgroup would walk
// them alone) keep the device-wide sort.
static constexpr int64_t SEGSORT_MAX_KEYS = 8192;   // (20 000-key segments on 24 node bits: 363 us per 1.2 M keys, the device-wide sort ~250)
struct SegOffset {   // offset of segment boundary (i + shift): begin iterator = shift 0, end iterator = shift 1
    unsigned per, total, shift;
    __host__ __device__ unsigned operator()(unsigned i) const {
        const unsigned long long e = (unsigned long long)(i + shift) * per;
        return e < total ? (unsigned)e : total;
    }
};
static bool use_segmented_sort(int64_t batch, size_t nc) {
    static const char* env = getenv("TPNET_DEV_SEGSORT");       // developer override: "0" / "1"
    if (nc >= (1ull << 32)) return false;
    if (env) return env[0] == '1';
    return 2 * batch <= SEGSORT_MAX_KEYS;
}

static size_t sort_tmp_bytes(size_t n, int64_t batch) {
    size_t bytes = 0, bytes32 = 0, bytes_seg = 0;
    uint64_t* kn64 = nullptr;
    uint32_t* kn = nullptr;
    uint32_t* vn = nullptr;
    // size queries only: no kernel is launched (keys are 32-bit whenever (batch, node) fits, else 64-bit: plan_build)
    (void)rocprim::radix_sort_pairs(nullptr, bytes, kn64, kn64, vn, vn, n, 0, 64, (hipStream_t)0, false);
    (void)rocprim::radix_sort_pairs(nullptr, bytes32, kn, kn, vn, vn, n, 0, 32, (hipStream_t)0, false);
    if (bytes32 > bytes) bytes = bytes32;
    if (n < (1ull << 32)) {
        const unsigned nseg = (unsigned)((n + 2 * (size_t)batch - 1) / (2 * (size_t)batch));
        auto cnt = rocprim::counting_iterator<unsigned>(0);
        auto b = rocprim::make_transform_iterator(cnt, SegOffset{(unsigned)(2 * batch), (unsigned)n, 0u});
        auto e = rocprim::make_transform_iterator(cnt, SegOffset{(unsigned)(2 * batch), (unsigned)n, 1u});
        (void)rocprim::segmented_radix_sort_pairs(nullptr, bytes_seg, kn, kn, vn, vn, (unsigned)n, nseg, b, e, 0u, 32u,
                                                  (hipStream_t)0, false);
    }
    return bytes > bytes_seg ? bytes : bytes_seg;
}

size_t plan_bytes(int64_t max_edges, int64_t batch) {
    if (max_edges < 1) max_edges = 1;
    if (batch < 1) batch = 1;
    const size_t nc = 2 * (size_t)max_edges;
    const size_t nb = (size_t)((max_edges + batch - 1) / batch);
    size_t tot = TPNET_DBG_BYTES;
    tot += align_up(nc * sizeof(uint64_t), 256) * 2;   // keys in/out
    tot += align_up(nc * sizeof(uint32_t), 256) * 2;   // vals in/out
    tot += align_up(nc * sizeof(int32_t), 256) * 2;    // s_partner, s_target
    tot += align_up(nc * sizeof(float), 256);          // s_coef
    tot += align_up(nc * sizeof(Item), 256) * 2;       // light, heavy
    tot += align_up(nb * sizeof(BatchDesc), 256);
    tot += align_up(sort_tmp_bytes(nc, batch), 256) + 256;
    return tot;
}

int plan_carve(void* ws, size_t ws_bytes, int64_t Ec, int64_t batch, Plan* out) {
    if (!ws) return TPNET_ERR_WORKSPACE;
    const size_t nc = 2 * (size_t)Ec;
    const size_t nb = (size_t)((Ec + batch - 1) / batch);
    char* p = reinterpret_cast<char*>(ws);
    char* const end = p + ws_bytes;
    p = reinterpret_cast<char*>(align_up(reinterpret_cast<size_t>(p), 256));
    auto take = [&](size_t bytes) -> void* {
        void* r = p;
        p += align_up(bytes, 256);
        return r;
    };
    out->dbg = (unsigned long long*)take(TPNET_DBG_BYTES);
    out->keys_in = (uint64_t*)take(nc * sizeof(uint64_t));
    out->keys_out = (uint64_t*)take(nc * sizeof(uint64_t));
    out->vals_in = (uint32_t*)take(nc * sizeof(uint32_t));
    out->vals_out = (uint32_t*)take(nc * sizeof(uint32_t));
    out->s_partner = (int32_t*)take(nc * sizeof(int32_t));
    out->s_target = (int32_t*)take(nc * sizeof(int32_t));
    out->s_coef = (float*)take(nc * sizeof(float));
    out->light = (Item*)take(nc * sizeof(Item));
    out->heavy = (Item*)take(nc * sizeof(Item));
    out->desc = (BatchDesc*)take(nb * sizeof(BatchDesc));
    out->sort_tmp_bytes = sort_tmp_bytes(nc, batch);
    out->sort_tmp = take(out->sort_tmp_bytes);
    // edge-fused update flags alias keys_in (dead after the sort): 2*Ec bytes + Ec floats <= 16*Ec bytes for Ec >= 26
    const size_t fw_off = align_up(2 * (size_t)Ec, 256);
    if (fw_off + 4 * (size_t)Ec <= nc * sizeof(uint64_t)) {
        out->fuse_src = reinterpret_cast<uint8_t*>(out->keys_in);
        out->fuse_dst = out->fuse_src + Ec;
        out->fuse_w = reinterpret_cast<float*>(reinterpret_cast<char*>(out->keys_in) + fw_off);
    } else {
        out->fuse_src = out->fuse_dst = nullptr;
        out->fuse_w = nullptr;
    }
    if (p > end) return TPNET_ERR_WORKSPACE;
    return TPNET_OK;
}

// ---- kernels ------------------------------------------------------------------------------------------------
__global__ void k_batch_desc(BatchDesc* __restrict__ desc, const double* __restrict__ t, int64_t Ec, int64_t B,
                             int64_t nb, double now_time, const double* __restrict__ t_prev, double lambda, int L) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    BatchDesc D;
    D.e0 = b * B;
    const int64_t ne = (Ec - D.e0 < B) ? (Ec - D.e0) : B;
    D.ne = (int32_t)ne;
    D.pad = 0;
    D.t_last = t[D.e0 + ne - 1];                       // next_time = node_interact_times[-1]   (TPNet.py:76)
    D.now = (b == 0) ? (t_prev ? *t_prev : now_time) : t[D.e0 - 1];         // clock left by the previous batch       (TPNet.py:99)
    D.n_light = 0;
    D.n_heavy = 0;
    const double g = exp(-lambda * (D.t_last - D.now));  // TPNet.py:84-85, f64 then rounded to f32 once
    double gi = 1.0;
    for (int i = 0; i < TPNET_MAX_LAYERS; ++i) {
        gi *= g;
        D.decay[i] = (i < L) ? (float)gi : 1.0f;
    }
    desc[b] = D;
}

// contribution j of the chunk: batch b = j / (2B); inside the batch the first ne are the src-side scatter-adds
// (target src[e] <- partner dst[e]), the next ne the dst-side ones.
__device__ __forceinline__ void decode(int64_t j, int64_t B, int64_t Ec, int64_t& b, int& side, int64_t& e) {
    b = j / (2 * B);
    const int64_t e0 = b * B;
    const int64_t ne = (Ec - e0 < B) ? (Ec - e0) : B;
    const int64_t r = j - 2 * e0;
    side = (r >= ne) ? 1 : 0;
    e = e0 + (side ? r - ne : r);
}

// K = uint32_t whenever the key fits: the node alone for the segmented sort (the batch is the segment), (batch, node) in
// 32 bits for the device-wide sort of short chunks; uint64_t otherwise.  Half the key bytes through the sort.
template <typename K>
__global__ void k_make_keys(K* __restrict__ keys, uint32_t* __restrict__ vals,
                            const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t Ec, int64_t B,
                            int64_t N, int node_bits, int with_batch, uint32_t* err) {
    const int64_t nc = 2 * Ec;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nc; j += (int64_t)gridDim.x * blockDim.x) {
        int64_t b, e;
        int side;
        decode(j, B, Ec, b, side, e);
        int64_t tgt = side ? dst[e] : src[e];
        if ((uint64_t)tgt >= (uint64_t)N) tgt = 0;  // counted (once per edge) in k_finish
        keys[j] = with_batch ? (K)(((uint64_t)b << node_bits) | (uint64_t)tgt) : (K)tgt;
        vals[j] = (uint32_t)j;
    }
}

// partner and time weight of the contribution whose pre-sort index is `val`
__device__ __forceinline__ void contribution(const Plan& p, const int64_t* __restrict__ src,
                                             const int64_t* __restrict__ dst, const double* __restrict__ t, int64_t Ec,
                                             int64_t B, int64_t N, double lambda, uint32_t val, int32_t& partner,
                                             float& w, uint32_t* err) {
    int64_t b, e;
    int side;
    decode((int64_t)val, B, Ec, b, side, e);
    const int64_t s = src[e], dd = dst[e];
    const bool ok = (uint64_t)s < (uint64_t)N && (uint64_t)dd < (uint64_t)N;
    if (err && !ok && side == 0) atomicAdd(err, 1u);   // once per bad edge
    const double t_last = p.desc[b].t_last;
    // time weight, with the reference's casts: absolute times rounded to f32 BEFORE the subtraction, f32 lambda
    // (models/TPNet.py:77-78)
    const float x = (float)t_last - (float)t[e];
    w = ok ? expf((float)(-lambda) * x) : 0.0f;
    partner = ok ? (int32_t)(side ? s : dd) : 0;
}

// Wave-aggregated append: the lanes with `pred` that share a batch take consecutive slots of that batch's list with
// ONE atomic per (wave, batch) instead of one per item (a batch's counter is a single hot address otherwise).
__device__ __forceinline__ uint32_t agg_append(BatchDesc* __restrict__ desc, int64_t b, bool pred, bool heavy) {
    const int lane = (int)(threadIdx.x & 63);
    uint32_t idx = 0;
    unsigned long long todo = __ballot(pred);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int64_t bsel = __shfl(b, leader, 64);
        const unsigned long long grp = __ballot(pred && b == bsel);
        uint32_t base = 0;
        if (lane == leader)
            base = atomicAdd(heavy ? &desc[bsel].n_heavy : &desc[bsel].n_light, (uint32_t)__popcll(grp));
        base = __shfl(base, leader, 64);
        if (pred && b == bsel) idx = base + (uint32_t)__popcll(grp & ((1ull << lane) - 1ull));
        todo &= ~grp;
    }
    return idx;
}

template <typename K>
__global__ void k_finish(Plan p, const K* __restrict__ keys_out, const int64_t* __restrict__ src,
                         const int64_t* __restrict__ dst, const double* __restrict__ t, int64_t Ec, int64_t B, int64_t N,
                         int node_bits, double lambda, uint32_t heavy_threshold, int fuse, uint32_t* err) {
    const int64_t nc = 2 * Ec;
    const uint64_t node_mask = (1ull << node_bits) - 1;
    for (int64_t jb = (int64_t)blockIdx.x * blockDim.x; jb < nc; jb += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = jb + threadIdx.x;
        const bool active = j < nc;
        bool head = false;
        int64_t b = 0, c0 = 0;
        Item it;
        it.j0 = 0; it.cnt = 0; it.target = 0; it.p0 = 0; it.w0 = 0.f; it.p1 = 0; it.w1 = 0.f; it.pad = 0;
        if (active) {
            const K key = keys_out[j];
            const uint32_t val = p.vals_out[j];
            b = j / (2 * B);                            // a batch's contributions stay in its range of positions
            int32_t partner;
            float w;
            contribution(p, src, dst, t, Ec, B, N, lambda, val, partner, w, err);
            p.s_partner[j] = partner;
            p.s_coef[j] = w;
            p.s_target[j] = (int32_t)(key & node_mask);

            c0 = 2 * b * B;
            const int64_t ne = (Ec - b * B < B) ? (Ec - b * B) : B;
            const int64_t cend = c0 + 2 * ne;
            head = (j == c0) || (keys_out[j - 1] != key);
            if (head) {
                // run length by galloping + binary search on the sorted keys
                int64_t lo = j, hi, step = 1;
                for (;;) {
                    const int64_t nx = lo + step;
                    if (nx >= cend) { hi = cend; break; }
                    if (keys_out[nx] != key) { hi = nx; break; }
                    lo = nx;
                    step <<= 1;
                }
                while (hi - lo > 1) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (keys_out[mid] == key) lo = mid; else hi = mid;
                }
                it.j0 = (uint32_t)j;
                it.cnt = (uint32_t)(hi - j);
                it.target = (int32_t)(key & node_mask);
                it.p0 = partner;
                it.w0 = w;
                if (it.cnt >= 2)
                    contribution(p, src, dst, t, Ec, B, N, lambda, p.vals_out[j + 1], it.p1, it.w1, nullptr);
            }
        }
        // a lone contribution rides with its edge's readout pair instead of becoming an item (see Plan::fuse_*)
        bool fused = false;
        if (fuse && head && it.cnt == 1) {
            int64_t fb, fe;
            int side;
            decode((int64_t)p.vals_out[j], B, Ec, fb, side, fe);
            (side ? p.fuse_dst : p.fuse_src)[fe] = 1;
            p.fuse_w[fe] = it.w0;
            fused = true;
        }
        const bool is_heavy = head && it.cnt > heavy_threshold;
        const bool is_light = head && !is_heavy && !fused;
        const uint32_t il = agg_append(p.desc, b, is_light, false);
        const uint32_t ih = agg_append(p.desc, b, is_heavy, true);
        if (is_light) p.light[c0 + il] = it;
        if (is_heavy) p.heavy[c0 + ih] = it;
    }
}

int plan_build(const tpnet_state& st, const Plan& p, const int64_t* src, const int64_t* dst, const double* t,
               int64_t Ec, int64_t batch, double now_time, const double* t_prev_dev, double lambda, uint32_t flags,
               hipStream_t s) {
    const int64_t nb = (Ec + batch - 1) / batch;
    const int64_t nc = 2 * Ec;
    const int node_bits = ceil_log2_u64((uint64_t)st.N) < 1 ? 1 : ceil_log2_u64((uint64_t)st.N);
    const int batch_bits = ceil_log2_u64((uint64_t)nb) < 1 ? 1 : ceil_log2_u64((uint64_t)nb);
    if (node_bits + batch_bits > 64) return TPNET_ERR_BAD_ARG;

    hipLaunchKernelGGL(k_batch_desc, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, p.desc, t, Ec, batch, nb,
                       now_time, t_prev_dev, lambda, (int)st.L);
    int grid = (int)((nc + 255) / 256);
    if (grid > 4096) grid = 4096;
    const bool seg = use_segmented_sort(batch, (size_t)nc);
    const bool narrow = seg || node_bits + batch_bits <= 32;
    uint32_t* k32_in = reinterpret_cast<uint32_t*>(p.keys_in);
    uint32_t* k32_out = reinterpret_cast<uint32_t*>(p.keys_out);
    if (narrow)
        hipLaunchKernelGGL(k_make_keys<uint32_t>, dim3(grid), dim3(256), 0, s, k32_in, p.vals_in, src, dst, Ec, batch,
                           st.N, node_bits, seg ? 0 : 1, st.err);
    else
        hipLaunchKernelGGL(k_make_keys<uint64_t>, dim3(grid), dim3(256), 0, s, p.keys_in, p.vals_in, src, dst, Ec, batch,
                           st.N, node_bits, 1, st.err);
    TPNET_HIP_TRY(hipGetLastError());
    size_t tmp = p.sort_tmp_bytes;
    if (seg) {
        const unsigned nseg = (unsigned)((nc + 2 * batch - 1) / (2 * batch));
        auto cnt = rocprim::counting_iterator<unsigned>(0);
        auto sb = rocprim::make_transform_iterator(cnt, SegOffset{(unsigned)(2 * batch), (unsigned)nc, 0u});
        auto se = rocprim::make_transform_iterator(cnt, SegOffset{(unsigned)(2 * batch), (unsigned)nc, 1u});
        TPNET_HIP_TRY(rocprim::segmented_radix_sort_pairs(p.sort_tmp, tmp, k32_in, k32_out, p.vals_in, p.vals_out,
                                                          (unsigned)nc, nseg, sb, se, 0u, (unsigned)node_bits, s, false));
    } else if (narrow) {
        TPNET_HIP_TRY(rocprim::radix_sort_pairs(p.sort_tmp, tmp, k32_in, k32_out, p.vals_in, p.vals_out, (size_t)nc, 0u,
                                                (unsigned)(node_bits + batch_bits), s, false));
    } else {
        TPNET_HIP_TRY(rocprim::radix_sort_pairs(p.sort_tmp, tmp, p.keys_in, p.keys_out, p.vals_in, p.vals_out,
                                                (size_t)nc, 0u, (unsigned)(node_bits + batch_bits), s, false));
    }
    // heavy threshold: a target with more contributions than this gets L workgroups (one per layer).  Small batches are
    // bound by the longest dependent chain, so the bar is low (8); big batches are bound by throughput and by the serial
    // loop of the reserved heavy workgroups, so only real hubs qualify (measured optimum ~B/300 on Reddit- and
    // LastFM-shaped streams at B = 10 000).
    static const char* thr_env = getenv("TPNET_DEV_HEAVY_THRESHOLD");
    uint32_t thr = (uint32_t)(batch / 300);
    thr = thr < HEAVY_THRESHOLD ? HEAVY_THRESHOLD : (thr > 128u ? 128u : thr);
    if (thr_env) thr = (uint32_t)atoi(thr_env);
    if (flags & TPNET_FLAG_SEQUENTIAL) thr = 0xFFFFFFFFu;
    const int fuse = ((flags & PLAN_FUSE) && p.fuse_src) ? 1 : 0;
    if (fuse) TPNET_HIP_TRY(hipMemsetAsync(p.fuse_src, 0, 2 * (size_t)Ec, s));   // after the sort: the bytes were keys_in
    if (narrow)
        hipLaunchKernelGGL(k_finish<uint32_t>, dim3(grid), dim3(256), 0, s, p, k32_out, src, dst, t, Ec, batch, st.N,
                           node_bits, lambda, thr, fuse, st.err);
    else
        hipLaunchKernelGGL(k_finish<uint64_t>, dim3(grid), dim3(256), 0, s, p, p.keys_out, src, dst, t, Ec, batch, st.N,
                           node_bits, lambda, thr, fuse, st.err);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

}  // namespace tpnet
