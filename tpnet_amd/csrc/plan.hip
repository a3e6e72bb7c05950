// Batch planning for the update: group the 2*B (target <- partner) contributions of every batch by target node,
// keeping the reference's summation order inside a target (models/TPNet.py:93-96: all src-side scatter-adds in edge
// order, then all dst-side ones).  Done for a whole chunk of the stream at once (one stable device radix sort over
// (batch, target) keys), so the per-batch step kernels find ready-made item lists and the cost is amortised.
#include "tpnet_common.h"

#include <cstdlib>
#include <cstring>
#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#ifdef TPNET_STAMPS
// diagnostic build only: wave 0 of workgroup 0 records (shader clock, 100 MHz wall clock) at the phase boundaries of k_plan_one
#define PSTAMP(slot)                                                                          \
    do {                                                                                      \
        if (dbg && blockIdx.x == 0 && threadIdx.x == 0) {                                     \
            dbg[(slot) * 2 + 0] = __builtin_amdgcn_s_memtime();                               \
            dbg[(slot) * 2 + 1] = __builtin_amdgcn_s_memrealtime();                           \
        }                                                                                     \
    } while (0)
#else
#define PSTAMP(slot) do { } while (0)
#endif

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
    static const char* env = TPNET_DEV_STR(SEGSORT);       // developer override: "0" / "1"
    if (nc >= (1ull << 32)) return false;
    if (env) return env[0] == '1';
    return 2 * batch <= SEGSORT_MAX_KEYS;
}

static size_t sort_tmp_bytes(size_t n, int64_t batch) {
    // (the size queries below cost microseconds each and the same sizes are asked for several times per call)
    static thread_local size_t memo_n[4] = {0, 0, 0, 0}, memo_v[4] = {0, 0, 0, 0};
    static thread_local int64_t memo_b[4] = {0, 0, 0, 0};
    static thread_local int memo_next = 0;
    for (int i = 0; i < 4; ++i)
        if (memo_v[i] && memo_n[i] == n && memo_b[i] == batch) return memo_v[i];
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
    const size_t res = bytes > bytes_seg ? bytes : bytes_seg;
    memo_n[memo_next] = n; memo_b[memo_next] = batch; memo_v[memo_next] = res;
    memo_next = (memo_next + 1) & 3;
    return res;
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
__device__ __forceinline__ void batch_desc(int64_t b, BatchDesc* __restrict__ desc, const double* __restrict__ t, int64_t Ec,
                                           int64_t B, int64_t nb, double now_time, const double* __restrict__ t_prev,
                                           double lambda, int L) {
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
    for (int i = 0; i < TPNET_MAX_LAYERS; ++i)
        D.decay[i] = (i < L) ? (float)pow(g, (double)(i + 1)) : 1.0f;     // np.power(g, i) like the reference (TPNet.py:84-85)
    desc[b] = D;
}

struct DescArgs {          // the batch descriptors ride in the last blocks of the key kernels (one launch less per plan)
    BatchDesc* desc;
    const double* t;
    int64_t nb;
    double now_time;
    const double* t_prev;
    double lambda;
    int L;
    uint32_t kblocks;      // blocks [0, kblocks) make keys, the rest descriptors
};

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
                            int64_t N, int node_bits, int with_batch, uint32_t* err, DescArgs da) {
    if (blockIdx.x >= da.kblocks) {
        for (int64_t b = (int64_t)(blockIdx.x - da.kblocks) * blockDim.x + threadIdx.x; b < da.nb;
             b += (int64_t)(gridDim.x - da.kblocks) * blockDim.x)
            batch_desc(b, da.desc, da.t, Ec, B, da.nb, da.now_time, da.t_prev, da.lambda, da.L);
        return;
    }
    const int64_t nc = 2 * Ec;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nc; j += (int64_t)da.kblocks * blockDim.x) {
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

static uint32_t heavy_threshold_for(int64_t batch, uint32_t flags) {
    static const char* thr_env = TPNET_DEV_STR(HEAVY_THRESHOLD);
    uint32_t thr = (uint32_t)(batch / 300);
    thr = thr < HEAVY_THRESHOLD ? HEAVY_THRESHOLD : (thr > 128u ? 128u : thr);
    if (thr_env) thr = (uint32_t)atoi(thr_env);
    if (flags & TPNET_FLAG_SEQUENTIAL) thr = 0xFFFFFFFFu;
    return thr;
}

int plan_blocks(const tpnet_state& st, const Plan& p, const int64_t* src, const int64_t* dst, const double* t, int64_t Ec,
                int64_t B, double now_time, const double* t_prev_dev, double lambda, uint32_t flags, hipStream_t s);

// ---------------------------------------------------------------------------------------------------------------
// The plan of ONE batch of up to PLAN_ONE_MAX edges in ONE workgroup (tpnet_update / tpnet_host_update: the per-batch call
// of the reference's loop, train_link_prediction.py:372): edges staged in LDS, the 2B (target, contribution) keys sorted
// by rocPRIM's block-level radix sort (stable: the reference's summation order inside a target survives), then exactly
// what k_make_keys + the device sort + k_finish produce for batch 0 of a chunk -- sorted partner / weight / target arrays,
// light and heavy item lists, the batch descriptor -- from one launch instead of four.  src / dst / t may live in
// device-mapped HOST memory (tpnet_host_update): each is read once, coalesced.
// ---------------------------------------------------------------------------------------------------------------
template <int BS, int IPT>
__global__ __launch_bounds__(BS) void k_plan_one(Plan p, const int64_t* __restrict__ src_c, const int64_t* __restrict__ dst_c,
                                                 const double* __restrict__ t_c, int64_t Ec, int32_t Bfull, int64_t N,
                                                 int node_bits, double now_time, const double* __restrict__ t_prev,
                                                 double lambda, int L, uint32_t heavy_threshold, uint32_t* err, int eager) {
    // workgroup b plans batch b of the chunk: edges [b * Bfull, ...), sorted positions [2 * b * Bfull, ...) (as plan_build lays
    // a chunk out); a single-batch call is the chunk of one batch
    const int64_t bb = blockIdx.x;
    const int64_t e0 = bb * Bfull;
    const int32_t B = (int32_t)((Ec - e0 < Bfull) ? (Ec - e0) : Bfull);
    const int64_t* __restrict__ src = src_c + e0;
    const int64_t* __restrict__ dst = dst_c + e0;
    const double* __restrict__ t = t_c + e0;
    p.s_partner += 2 * e0;
    p.s_coef += 2 * e0;
    p.s_target += 2 * e0;
    p.light += 2 * e0;
    p.heavy += 2 * e0;
    using Sort = rocprim::block_radix_sort<uint32_t, BS, IPT, uint32_t>;
    constexpr int NC = BS * IPT;
    __shared__ union U {
        typename Sort::storage_type sort;
        struct { uint32_t key[NC]; uint32_t val[NC]; } s;
    } u;
    __shared__ uint32_t e_src[NC / 2], e_dst[NC / 2];   // endpoint (0 if out of range) | bit 31: the EDGE has a bad endpoint
    __shared__ float e_w[NC / 2];
    __shared__ uint32_t n_light, n_heavy;
    static_assert(sizeof(U) + 3 * (NC / 2) * 4 + 8 <= 64 * 1024, "k_plan_one: LDS budget");
    const int tid = threadIdx.x;
    const int nc = 2 * B;
    [[maybe_unused]] unsigned long long* dbg = p.dbg;     // (phase stamps of diagnostic builds)
    PSTAMP(0);
    if (tid == 0) { n_light = 0; n_heavy = 0; }
    // every thread's edges in ONE burst of independent loads (src / dst / t may sit in host memory: a load is microseconds)
    constexpr int EPT = IPT / 2;
    int64_t rs[EPT], rd[EPT];
    double rt[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * BS + tid;
        const int ec = e < B ? e : B - 1;
        rs[q] = src[ec];
        rd[q] = dst[ec];
        rt[q] = t[ec];
    }
    const double t_last = t[B - 1];                      // next_time = node_interact_times[-1]   (TPNet.py:76)
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * BS + tid;
        if (e < B) {
            const int64_t s = rs[q], dd = rd[q];
            const bool oks = (uint64_t)s < (uint64_t)N, okd = (uint64_t)dd < (uint64_t)N;
            const uint32_t bad = (oks && okd) ? 0u : 0x80000000u;
            if (bad) atomicAdd(err, 1u);                 // once per bad edge
            e_src[e] = (oks ? (uint32_t)s : 0u) | bad;
            e_dst[e] = (okd ? (uint32_t)dd : 0u) | bad;
            // time weight with the reference's casts (models/TPNet.py:77-78), as contribution()
            const float x = (float)t_last - (float)rt[q];
            e_w[e] = bad ? 0.0f : expf((float)(-lambda) * x);
        }
    }
    __syncthreads();
    PSTAMP(1);
    uint32_t keys[IPT], vals[IPT];
    const uint32_t pad_key = 1u << node_bits;            // above every node id: padding sorts last
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
        const int j = tid * IPT + q;
        vals[q] = (uint32_t)j;
        if (j < nc) {
            const bool side = j >= B;                    // first the src-side scatter-adds, then the dst-side ones (TPNet.py:93-96)
            const int e = side ? j - B : j;
            keys[q] = (side ? e_dst[e] : e_src[e]) & 0x7FFFFFFFu;
        } else {
            keys[q] = pad_key;
        }
    }
    Sort().sort(keys, vals, u.sort, 0u, (unsigned)(node_bits + 1));
    __syncthreads();
    PSTAMP(2);
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
        u.s.key[tid * IPT + q] = keys[q];
        u.s.val[tid * IPT + q] = vals[q];
    }
    __syncthreads();
    PSTAMP(3);
    auto contrib = [&](uint32_t val, int32_t& partner, float& w) {
        const bool side = val >= (uint32_t)B;
        const int e = side ? (int)val - B : (int)val;
        const uint32_t es = e_src[e], ed = e_dst[e];
        const bool ok = !(es & 0x80000000u);
        partner = ok ? (int32_t)((side ? es : ed) & 0x7FFFFFFFu) : 0;
        w = ok ? e_w[e] : 0.0f;
    };
    const int lane = tid & 63;
    for (int jb = 0; jb < nc; jb += BS) {                // (uniform trip count: the list appends below are wave-wide)
        const int j = jb + tid;
        Item it;
        bool lead = false;
        if (j < nc) {
            const uint32_t key = u.s.key[j];
            int32_t partner;
            float w;
            contrib(u.s.val[j], partner, w);
            p.s_partner[j] = partner;
            p.s_coef[j] = w;
            p.s_target[j] = (int32_t)key;
            lead = j == 0 || u.s.key[j - 1] != key;
            if (lead) {
                // run length by galloping + binary search on the sorted keys (LDS)
                int lo = j, hi, step = 1;
                for (;;) {
                    const int nx = lo + step;
                    if (nx >= nc) { hi = nc; break; }
                    if (u.s.key[nx] != key) { hi = nx; break; }
                    lo = nx;
                    step <<= 1;
                }
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (u.s.key[mid] == key) lo = mid; else hi = mid;
                }
                it.j0 = (uint32_t)(2 * e0 + j);                    // chunk-relative position in the sorted arrays
                it.cnt = (uint32_t)(hi - j);
                it.target = (int32_t)key;
                it.p0 = partner;
                it.w0 = w;
                it.p1 = 0;
                it.w1 = 0.0f;
                it.pad = 0;
                if (it.cnt >= 2) contrib(u.s.val[j + 1], it.p1, it.w1);
            }
        }
        // list appends, one LDS atomic per wave and list (a typical batch has ~1500 leaders: as many same-address atomics)
        const bool hv = lead && it.cnt > heavy_threshold, lt = lead && !hv;
        const unsigned long long ml = __ballot(lt), mh = __ballot(hv);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (ml) {
            const int first = __ffsll((long long)ml) - 1;
            uint32_t base = 0;
            if (lane == first) base = atomicAdd(&n_light, (uint32_t)__popcll(ml));
            base = __shfl(base, first);
            if (lt) p.light[base + (uint32_t)__popcll(ml & below)] = it;
        }
        if (mh) {
            const int first = __ffsll((long long)mh) - 1;
            uint32_t base = 0;
            if (lane == first) base = atomicAdd(&n_heavy, (uint32_t)__popcll(mh));
            base = __shfl(base, first);
            if (hv) p.heavy[base + (uint32_t)__popcll(mh & below)] = it;
        }
    }
    __syncthreads();
    PSTAMP(4);
    if (tid == 0) {
        BatchDesc D;
        D.e0 = e0;
        D.ne = B;
        D.pad = 0;
        D.t_last = t_last;
        D.now = (bb == 0) ? (t_prev ? *t_prev : now_time) : t_c[e0 - 1];   // clock left by the previous batch (TPNet.py:99)
        D.n_light = n_light;
        D.n_heavy = n_heavy;
        // the dense decay's factors only where a dense decay will read them (k_decay_desc, eager mode): exp + pow in f64 are
        // ~2 us of this one thread's time behind the kernel's last barrier
        const double g = eager ? exp(-lambda * (t_last - D.now)) : 1.0;
        for (int i = 0; i < TPNET_MAX_LAYERS; ++i)
            D.decay[i] = (eager && i < L) ? (float)pow(g, (double)(i + 1)) : 1.0f;
        p.desc[bb] = D;
    }
    PSTAMP(5);
}

// ---------------------------------------------------------------------------------------------------------------
// The same plan WITHOUT a sort, for batches of up to 1 024 edges (round 3): nothing downstream needs the batch's targets in
// node order -- only every target's contributions side by side, in the reference's order (all src-side adds in edge order,
// then the dst-side ones = ascending contribution number j).  So: an LDS hash table groups the 2B contributions by target
// (compare-and-swap insert, one counter per slot), a scan of the counters gives every group its span, the members of a
// group are listed in arrival order and then RANKED by j -- a group of up to 32 by counting the smaller members, a larger
// one (a hub) by a wave with a bitmap over j and prefix pop-counts.  Six barriers, no radix pass: 14 -> ~8 us for a C2
// batch, the first link of every per-batch call's chain.  The position of a group inside the sorted arrays (and of an item
// inside its list) depends on the table's slot order, not the results: every sum has its terms in the same order as before.
// ---------------------------------------------------------------------------------------------------------------
template <int BS, int IPT>
__global__ __launch_bounds__(BS) void k_plan_one_h(Plan p, const int64_t* __restrict__ src_c, const int64_t* __restrict__ dst_c,
                                                   const double* __restrict__ t_c, int64_t Ec, int32_t Bfull, int64_t N,
                                                   double now_time, const double* __restrict__ t_prev, double lambda, int L,
                                                   uint32_t heavy_threshold, uint32_t* err, int eager) {
    constexpr int NC = BS * IPT;                         // contributions a workgroup can hold
    constexpr int SLOTS = 2 * NC;                        // hash table: load <= 1/2
    constexpr int SPT = SLOTS / BS;                      // slots per thread in the scan
    constexpr int LOG_SLOTS = (SLOTS == 4096) ? 12 : (SLOTS == 2048) ? 11 : 10;
    static_assert((1 << LOG_SLOTS) == SLOTS, "k_plan_one_h: table size");
    constexpr int NW = BS / 64;
    constexpr int BMW = NC / 32;                         // words of a hub's bitmap over j
    constexpr uint32_t EMPTY = 0xFFFFFFFFu;
    constexpr uint32_t SMALL = 16;                       // groups up to this size are ranked by their own members
    static_assert(NC <= 2048 && BMW <= 64, "k_plan_one_h: j fits 11 bits, a hub's bitmap one word per lane");
    const int64_t bb = blockIdx.x;
    const int64_t e0 = bb * Bfull;
    const int32_t B = (int32_t)((Ec - e0 < Bfull) ? (Ec - e0) : Bfull);
    const int64_t* __restrict__ src = src_c + e0;
    const int64_t* __restrict__ dst = dst_c + e0;
    const double* __restrict__ t = t_c + e0;
    p.s_partner += 2 * e0;
    p.s_coef += 2 * e0;
    p.s_target += 2 * e0;
    p.light += 2 * e0;
    p.heavy += 2 * e0;
    __shared__ uint32_t hkey[SLOTS];                     // the slot's node
    __shared__ uint32_t hcb[SLOTS];                      // members counted so far; after the scan: span base << 16 | count
    __shared__ uint16_t mem[NC];                         // members of the groups, span by span, in arrival order
    __shared__ uint16_t sorted[NC];                      // position -> j (bit 15: the first of its group)
    __shared__ uint16_t cnt_at[NC];                      // at a group's first position: its size
    __shared__ uint32_t bm[NW][BMW];                     // a wave's bitmap over j for the hub it ranks
    __shared__ uint32_t e_src[NC / 2], e_dst[NC / 2];    // endpoint (0 if out of range) | bit 31: the EDGE has a bad endpoint
    __shared__ float e_w[NC / 2];
    __shared__ uint32_t n_light, n_heavy, n_big, wsum[NW];
    __shared__ uint16_t big[NC / SMALL];                 // slots of the groups larger than SMALL
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nc = 2 * B;
    [[maybe_unused]] unsigned long long* dbg = p.dbg;
    PSTAMP(0);
    constexpr int EPT = IPT / 2;
    int64_t rs[EPT], rd[EPT];
    double rt[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {                      // (the loads are in flight while the table is cleared)
        const int e = q * BS + tid;
        const int ec = e < B ? e : B - 1;
        rs[q] = src[ec];
        rd[q] = dst[ec];
        rt[q] = t[ec];
    }
    const double t_last = t[B - 1];                      // next_time = node_interact_times[-1]   (TPNet.py:76)
    if (tid == 0) { n_light = 0; n_heavy = 0; n_big = 0; }
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        hkey[k * BS + tid] = EMPTY;
        hcb[k * BS + tid] = 0;
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * BS + tid;
        if (e < B) {
            const int64_t s = rs[q], dd = rd[q];
            const bool oks = (uint64_t)s < (uint64_t)N, okd = (uint64_t)dd < (uint64_t)N;
            const uint32_t bad = (oks && okd) ? 0u : 0x80000000u;
            if (bad) atomicAdd(err, 1u);                 // once per bad edge
            e_src[e] = (oks ? (uint32_t)s : 0u) | bad;
            e_dst[e] = (okd ? (uint32_t)dd : 0u) | bad;
            const float x = (float)t_last - (float)rt[q];   // the reference's casts (models/TPNet.py:77-78), as contribution()
            e_w[e] = bad ? 0.0f : expf((float)(-lambda) * x);
        }
    }
    __syncthreads();
    PSTAMP(1);
    auto target_of = [&](int j) -> uint32_t {            // first the src-side scatter-adds, then the dst-side ones (TPNet.py:93-96)
        return ((j >= B) ? e_dst[j - B] : e_src[j]) & 0x7FFFFFFFu;
    };
    auto contrib = [&](int j, int32_t& partner, float& w) {
        const bool side = j >= B;
        const int e = side ? j - B : j;
        const uint32_t es = e_src[e], ed = e_dst[e];
        const bool ok = !(es & 0x80000000u);
        partner = ok ? (int32_t)((side ? es : ed) & 0x7FFFFFFFu) : 0;
        w = ok ? e_w[e] : 0.0f;
    };
    // group by target: slot and arrival number of every contribution
    uint32_t slot[IPT], arr[IPT];
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
        const int j = q * BS + tid;
        slot[q] = 0;
        arr[q] = 0;
        if (j < nc) {
            const uint32_t key = target_of(j);
            uint32_t h = (key * 2654435761u) >> (32 - LOG_SLOTS);
            for (;;) {                                   // (the table is never more than half full: the walk ends)
                const uint32_t prev = atomicCAS(&hkey[h], EMPTY, key);
                if (prev == EMPTY || prev == key) break;
                h = (h + 1) & (uint32_t)(SLOTS - 1);
            }
            slot[q] = h;
            arr[q] = atomicAdd(&hcb[h], 1u);
        }
    }
    __syncthreads();
    PSTAMP(2);
    // spans: exclusive scan of the slots' counts
    uint32_t c[SPT], tsum = 0;
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        c[k] = hcb[tid * SPT + k];
        tsum += c[k];
    }
    uint32_t inc = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = inc - tsum;
#pragma unroll
    for (int i = 0; i < NW; ++i)
        if (i < wave) base += wsum[i];
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        hcb[tid * SPT + k] = (base << 16) | c[k];
        if (c[k] > SMALL) big[atomicAdd(&n_big, 1u)] = (uint16_t)(tid * SPT + k);
        base += c[k];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
        const int j = q * BS + tid;
        if (j < nc) mem[(hcb[slot[q]] >> 16) + arr[q]] = (uint16_t)j;
    }
    __syncthreads();
    PSTAMP(3);
    // ranks and the sorted arrays
    auto emit = [&](int j, uint32_t pos, bool first, uint32_t cnt) {    // (the arrays themselves are written by position below: coalesced)
        sorted[pos] = (uint16_t)(j | (first ? 0x8000 : 0));
        if (first) cnt_at[pos] = (uint16_t)cnt;
    };
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
        const int j = q * BS + tid;
        if (j < nc) {
            const uint32_t cb = hcb[slot[q]];
            const uint32_t gb = cb >> 16, cnt = cb & 0xFFFFu;
            if (cnt <= SMALL) {
                uint32_t rank = 0;
                if (cnt > 1) {
                    for (uint32_t m = 0; m < cnt; m += 4) {          // (four independent LDS reads per round; the span of the
#pragma unroll                                                       //  last group ends inside mem[], a read past a span is masked)
                        for (uint32_t k = 0; k < 4; ++k) {
                            const uint32_t mm = m + k;
                            const uint32_t o = mem[(gb + mm) < (uint32_t)NC ? gb + mm : 0u];
                            rank += (mm < cnt && o < (uint32_t)j) ? 1u : 0u;
                        }
                    }
                }
                emit(j, gb + rank, rank == 0, cnt);
            }
        }
    }
    const uint32_t nbig = n_big;
    for (uint32_t k = wave; k < nbig; k += NW) {         // a hub: one wave, a bitmap over j, prefix pop-counts
        const uint32_t cb = hcb[big[k]];
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
            if (on) emit((int)j, gb + rank, rank == 0, cnt);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    PSTAMP(4);
    // the items: one per group, at the group's first position
#pragma unroll
    for (int q = 0; q < IPT; ++q) {                      // (uniform trip count: the list appends are wave-wide)
        const int pos = q * BS + tid;
        Item it;
        bool lead = false;
        if (pos < nc) {
            const uint32_t v = sorted[pos];
            lead = (v & 0x8000u) != 0;
            const int j = (int)(v & 0x7FFFu);
            const int32_t tg = (int32_t)target_of(j);
            contrib(j, it.p0, it.w0);
            p.s_partner[pos] = it.p0;
            p.s_coef[pos] = it.w0;
            p.s_target[pos] = tg;
            if (lead) {
                it.j0 = (uint32_t)(2 * e0 + pos);                  // chunk-relative position in the sorted arrays
                it.cnt = cnt_at[pos];
                it.target = tg;
                it.p1 = 0;
                it.w1 = 0.0f;
                it.pad = 0;
                if (it.cnt >= 2) contrib((int)(sorted[pos + 1] & 0x7FFFu), it.p1, it.w1);
            }
        }
        const bool hv = lead && it.cnt > heavy_threshold, lt = lead && !hv;
        const unsigned long long ml = __ballot(lt), mh = __ballot(hv);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (ml) {
            const int first = __ffsll((long long)ml) - 1;
            uint32_t lb = 0;
            if (lane == first) lb = atomicAdd(&n_light, (uint32_t)__popcll(ml));
            lb = __shfl(lb, first);
            if (lt) p.light[lb + (uint32_t)__popcll(ml & below)] = it;
        }
        if (mh) {
            const int first = __ffsll((long long)mh) - 1;
            uint32_t hb = 0;
            if (lane == first) hb = atomicAdd(&n_heavy, (uint32_t)__popcll(mh));
            hb = __shfl(hb, first);
            if (hv) p.heavy[hb + (uint32_t)__popcll(mh & below)] = it;
        }
    }
    __syncthreads();
    PSTAMP(5);
    if (tid == 0) {
        BatchDesc D;
        D.e0 = e0;
        D.ne = B;
        D.pad = 0;
        D.t_last = t_last;
        D.now = (bb == 0) ? (t_prev ? *t_prev : now_time) : t_c[e0 - 1];   // clock left by the previous batch (TPNet.py:99)
        D.n_light = n_light;
        D.n_heavy = n_heavy;
        const double g = eager ? exp(-lambda * (t_last - D.now)) : 1.0;    // (as k_plan_one)
        for (int i = 0; i < TPNET_MAX_LAYERS; ++i)
            D.decay[i] = (eager && i < L) ? (float)pow(g, (double)(i + 1)) : 1.0f;
        p.desc[bb] = D;
    }
}

int64_t plan_one_max_batch() {
    static const int off = TPNET_DEV_STR(NO_PLAN_ONE) ? 1 : 0;     // developer override: always the chunk planner
    return off ? 0 : PLAN_ONE_MAX;
}

int plan_one(const tpnet_state& st, const Plan& p, const int64_t* src, const int64_t* dst, const double* t, int64_t B,
             double now_time, double lambda, uint32_t flags, hipStream_t s) {
    return plan_blocks(st, p, src, dst, t, B, B, now_time, nullptr, lambda, flags, s);
}

// every batch of a chunk by its own workgroup, ONE launch (batches of up to PLAN_ONE_MAX edges)
int plan_blocks(const tpnet_state& st, const Plan& p, const int64_t* src, const int64_t* dst, const double* t, int64_t Ec,
                int64_t B, double now_time, const double* t_prev_dev, double lambda, uint32_t flags, hipStream_t s) {
    if (B < 1 || B > PLAN_ONE_MAX || Ec < 1) return TPNET_ERR_BAD_ARG;
    const int64_t nbl = (Ec + B - 1) / B;
    if (nbl > 0x7FFFFFFF) return TPNET_ERR_BAD_ARG;
    const int node_bits = ceil_log2_u64((uint64_t)st.N) < 1 ? 1 : ceil_log2_u64((uint64_t)st.N);
    if (node_bits > 31) return TPNET_ERR_BAD_ARG;
    const uint32_t thr = heavy_threshold_for(B, flags);
    static const int bs_env = TPNET_DEV_INT(PLAN_ONE_BS, 0);   // developer override
#define TPNET_PLAN_ONE(BS_, IPT_)                                                                                         \
    hipLaunchKernelGGL((k_plan_one<BS_, IPT_>), dim3((unsigned)nbl), dim3(BS_), 0, s, p, src, dst, t, Ec, (int32_t)B, st.N,   \
                       node_bits, now_time, t_prev_dev, lambda, (int)st.L, thr, st.err, (flags & TPNET_FLAG_EAGER_DECAY) ? 1 : 0)
    const int64_t nc = 2 * B;
    static const int sort_env = TPNET_DEV_INT(PLAN_ONE_SORT, 0);   // developer override: the sorting planner for every size
#define TPNET_PLAN_ONE_H(BS_, IPT_)                                                                                         \
    hipLaunchKernelGGL((k_plan_one_h<BS_, IPT_>), dim3((unsigned)nbl), dim3(BS_), 0, s, p, src, dst, t, Ec, (int32_t)B, st.N, \
                       now_time, t_prev_dev, lambda, (int)st.L, thr, st.err, (flags & TPNET_FLAG_EAGER_DECAY) ? 1 : 0)
    if (nc <= 2048 && !sort_env) {
        if (nc <= 512) TPNET_PLAN_ONE_H(256, 2);
        else if (nc <= 1024) TPNET_PLAN_ONE_H(512, 2);
        else TPNET_PLAN_ONE_H(1024, 2);
        TPNET_HIP_TRY(hipGetLastError());
        return TPNET_OK;
    }
#undef TPNET_PLAN_ONE_H
    if (nc <= 512) {
        if (bs_env == 512) TPNET_PLAN_ONE(256, 2); else TPNET_PLAN_ONE(256, 2);
    } else if (nc <= 1024) {
        if (bs_env == 256) TPNET_PLAN_ONE(256, 4); else TPNET_PLAN_ONE(512, 2);
    } else if (nc <= 2048) {
        if (bs_env == 256) TPNET_PLAN_ONE(256, 8); else if (bs_env == 512) TPNET_PLAN_ONE(512, 4); else TPNET_PLAN_ONE(1024, 2);
    } else {
        if (bs_env == 256) TPNET_PLAN_ONE(256, 16); else if (bs_env == 512) TPNET_PLAN_ONE(512, 8); else TPNET_PLAN_ONE(1024, 4);
    }
#undef TPNET_PLAN_ONE
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int plan_build(const tpnet_state& st, const Plan& p, const int64_t* src, const int64_t* dst, const double* t,
               int64_t Ec, int64_t batch, double now_time, const double* t_prev_dev, double lambda, uint32_t flags,
               hipStream_t s) {
    const int64_t nb = (Ec + batch - 1) / batch;
    const int64_t nc = 2 * Ec;
    const int node_bits = ceil_log2_u64((uint64_t)st.N) < 1 ? 1 : ceil_log2_u64((uint64_t)st.N);
    const int batch_bits = ceil_log2_u64((uint64_t)nb) < 1 ? 1 : ceil_log2_u64((uint64_t)nb);
    if (node_bits + batch_bits > 64) return TPNET_ERR_BAD_ARG;
    // batches that fit one workgroup's LDS: every batch planned by its own workgroup in ONE launch (keys, sort, item lists and
    // descriptors; no device-wide sort, no second pass over the sorted keys) -- a 20-batch call at C2: 65 -> ~25 us of planning
    if (batch <= plan_one_max_batch() && !(flags & PLAN_FUSE) && node_bits <= 31)
        return plan_blocks(st, p, src, dst, t, Ec, batch, now_time, t_prev_dev, lambda, flags, s);

    int grid = (int)((nc + 255) / 256);
    if (grid > 4096) grid = 4096;
    const int dgrid = (int)((nb + 255) / 256) > 64 ? 64 : (int)((nb + 255) / 256);
    const DescArgs da{p.desc, t, nb, now_time, t_prev_dev, lambda, (int)st.L, (uint32_t)grid};
    const bool seg = use_segmented_sort(batch, (size_t)nc);
    const bool narrow = seg || node_bits + batch_bits <= 32;
    uint32_t* k32_in = reinterpret_cast<uint32_t*>(p.keys_in);
    uint32_t* k32_out = reinterpret_cast<uint32_t*>(p.keys_out);
    if (narrow)
        hipLaunchKernelGGL(k_make_keys<uint32_t>, dim3(grid + dgrid), dim3(256), 0, s, k32_in, p.vals_in, src, dst, Ec, batch,
                           st.N, node_bits, seg ? 0 : 1, st.err, da);
    else
        hipLaunchKernelGGL(k_make_keys<uint64_t>, dim3(grid + dgrid), dim3(256), 0, s, p.keys_in, p.vals_in, src, dst, Ec, batch,
                           st.N, node_bits, 1, st.err, da);
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
    const uint32_t thr = heavy_threshold_for(batch, flags);
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


// =====================================================================================================================
// Windowed stream path: plan of a chunk (see tpnet_common.h, WPlan).
// Sort key = (target node, batch in the chunk): ALL contributions of a node in the chunk are contiguous, ordered by
// batch (its RUNS), inside a run in the reference's summation order (stable sort on the payload order: src-side edges,
// then dst-side edges, models/TPNet.py:93-96).  The part of a node's range that falls into one window is a CHAIN (the
// unit of the update kernels).  Everything a kernel needs to find a row version without touching another unit's output
// of the same launch is resolved here, once per chunk:
//   * the version of a node "before batch b" = its run with the largest batch < b in the chunk (binary search over
//     the sorted keys) -> a slot of the version log (the run's last sorted position), or else the frozen table row;
//   * per run: the decay from the node's previous run; per contribution: block boundaries of the fixed-association sum;
//   * per window: the lists of chains (walked by a lane group / by a workgroup per column part);
//   * the list of every node's last run in the chunk (what the write-back copies to the table).
// =====================================================================================================================
int wplan_window_batches(int64_t batch, int d, int L) {
    static const char* env = TPNET_DEV_STR(WINDOW);       // developer override: 0 = off, n = batches per window
    (void)L;
    if (d % 4 != 0 || d < 4 || batch < 1) return 0;             // scalar-load rows (use_matrix, odd d) keep the per-batch path
    if (batch > 4096) return 0;                                 // such batches are bandwidth-bound one launch at a time already
    int64_t K = 24576 / batch;                                  // ~24 K edges per window (C2, four degree laws, tools/degree_sensitivity.py: 3.44-3.78 us per batch at 24 K, 3.55-3.90 at 16 K, 3.35-4.12 at 32 K)
    if (env) K = atoi(env);
    if (K > WIN_MAX_BATCHES) K = WIN_MAX_BATCHES;
    return K >= 2 ? (int)K : 0;
}

// Contributions per (node, window) above which a workgroup per column part walks the chain instead of one lane group: a lane
// group keeps 8 rows in flight, so a chain of n contributions is n / 8 dependent memory round trips -- the latency floor of a
// launch, which has to stay below the launch's bandwidth time, i.e. scale with the bytes of a window (batches x edges x row
// length).  Measured (tools/short_trace.sh, tools/sweep_c1c5.sh): C2 (d=128) 20 batches in 2 windows: 127 us at 96, 85 us at 32;
// C1 (d=64, B=200, windows of 64 batches): 198 M edges/s at 96, 281 M at 32; rows of 256 / 512 floats and the 24-batch
// windows of long C2 streams: 96 (128: -0..10 %).
bool wplan_medium_chains(int d) {
    static const int off = TPNET_DEV_INT(NO_MEDIUM, 0);
    if (off || d % 4 != 0) return false;
    const Geom g = pick_geom(d);
    return g.w == 4 && d == g.lpp * g.vpl * 4;
}

uint32_t wplan_heavy_threshold(int K, int64_t batch, int d) {
    static const int env = TPNET_DEV_INT(WIN_HEAVY, 0);
    if (env > 0) return (uint32_t)env;
    // chains of WIN_MED_MIN .. 128 contributions get a workgroup each (chain_medium: one segment of 128 positions at d = 128);
    // beyond that a workgroup per column part
    if (wplan_medium_chains(d)) return 128u;
    const int64_t t = (int64_t)K * batch * d / 24576;
    return (uint32_t)(t < 16 ? 16 : (t > 96 ? 96 : t));
}

int64_t wplan_max_chunk_edges(int64_t batch, int d, int L) {
    static const char* env = TPNET_DEV_STR(WIN_CHUNK_MB);
    // bytes of version log per chunk (C2: 1 GiB 4.6 us per batch, 4 GiB 4.1: fewer pipeline drains).  16 GiB since round 4 (4 until
    // then): a plan is replayed across epochs only where the stream is ONE chunk, and on a GPU with 288 GB the log of every dataset
    // of the reference fits one (C2 rows: 5.6 M edges per chunk; a caller short of memory caps it: tpnet_stream_workspace_bytes_capped)
    const int64_t log_cap = (env ? (int64_t)atoi(env) : 16384) << 20;
    int64_t e = log_cap / (2 * (int64_t)L * d * 4);
    const int64_t hard = (int64_t)(WREF_SLOT_MASK >> 1);                     // slots are 26-bit sorted positions
    if (e > hard) e = hard;
    if (e > 65535 * batch) e = 65535 * batch;                               // 16-bit batch numbers
    e = e / batch * batch;
    return e < batch ? batch : e;
}

static size_t wplan_extra_bytes(int64_t Ec, int64_t batch, int64_t N, int d, int L, int K) {
    const size_t nc = 2 * (size_t)Ec;
    (void)K;
    const size_t nw = (size_t)((Ec + batch - 1) / batch);       // (room for windows of ONE batch: the window length is chosen per chunk)
    size_t tot = 0;
    tot += align_up(nc * 4, 256) * 3;                           // s_ref, s_g, s_dec
    tot += align_up(nc * 2, 256);                               // s_bc
    tot += align_up(nw * sizeof(WinDesc), 256) + 256;           // wdesc, wb_count
    tot += align_up(3 * (size_t)Ec * 4, 256) * 2;               // e_ref, e_g
    tot += align_up(nc * (size_t)L * (size_t)d * 4, 256);       // version log
    tot += align_up((size_t)N * 4, 256) * 2;                    // node_lo, node_hi
    tot += align_up(nc * 4, 256) * 2;                           // inv, rhead
    tot += align_up((size_t)N * 8, 256) + align_up((2 * WIN_MAX_WINDOWS * 8 + WIN_MAX_WINDOWS) * 4, 256) +
           align_up(wplan3_table_bytes(Ec, batch), 256) + align_up(wplan3_blk_bytes(Ec, batch), 256);   // wmask, wcls, wtab, wblk (hashed planner, wplan3.hip)
    tot += align_up(wplan_dense_bytes(Ec, batch, N, d, L), 256);       // the dense planner's matrices (wplan_dense.hip; 0: not eligible)
    return tot + 256;
}

size_t wplan_bytes(int64_t max_edges, int64_t batch, int64_t N, int d, int L) {
    if (max_edges < 1) max_edges = 1;
    if (batch < 1) batch = 1;
    const int K = wplan_window_batches(batch, d, L);
    if (K == 0) return plan_bytes(max_edges, batch);
    return plan_bytes(max_edges, batch) + wplan_extra_bytes(max_edges, batch, N, d, L, K);
}

size_t wplan_log_bytes(int64_t max_edges, int d, int L) {
    if (max_edges < 1) max_edges = 1;
    return align_up(2 * (size_t)max_edges * (size_t)L * (size_t)d * 4, 256);
}

// `shared_log` (may be null): the version log lives there instead of inside [ws, ws + ws_bytes) -- the plans of the chunks of a
// multi-chunk stream each keep a region of their own (so that every one of them can be replayed, api.hip) and share ONE log
// (the plan of a row shard's chunk, wshard.hip: global batches beyond 4096 edges, of which a rank computes its share -- the window
// length is chosen there)
size_t wplan_bytes_shard(int64_t max_edges, int64_t batch, int64_t N, int d, int L) {
    if (max_edges < 1) max_edges = 1;
    if (batch < 1) batch = 1;
    return plan_bytes(max_edges, batch) + wplan_extra_bytes(max_edges, batch, N, d, L, 2);
}

int wplan_carve(void* ws, size_t ws_bytes, int64_t Ec, int64_t batch, int64_t N, int d, int L, int K, WPlan* out, float* shared_log,
                bool shard) {
    if (K < 1 || K > WIN_MAX_BATCHES || (!shard && K > wplan_window_batches(batch, d, L))) return TPNET_ERR_BAD_ARG;
    if (Ec > wplan_max_chunk_edges(batch, d, L)) return TPNET_ERR_BAD_ARG;
    int rc = plan_carve(ws, ws_bytes, Ec, batch, &out->base);
    if (rc) return rc;
    const size_t base_bytes = plan_bytes(Ec, batch);
    if (base_bytes + wplan_extra_bytes(Ec, batch, N, d, L, K) - (shared_log ? wplan_log_bytes(Ec, d, L) : 0) > ws_bytes)
        return TPNET_ERR_WORKSPACE;
    char* p = reinterpret_cast<char*>(align_up(reinterpret_cast<size_t>(ws), 256)) + align_up(base_bytes, 256);
    // (plan_carve aligned the base the same way; plan_bytes includes its slack)
    auto take = [&](size_t bytes) -> void* {
        void* r = p;
        p += align_up(bytes, 256);
        return r;
    };
    const size_t nc = 2 * (size_t)Ec;
    const int64_t Ew = (int64_t)K * batch;
    const size_t nw = (size_t)((Ec + batch - 1) / batch);
    out->s_ref = (uint32_t*)take(nc * 4);
    out->s_g = (float*)take(nc * 4);
    out->s_dec = (float*)take(nc * 4);
    out->s_bc = (uint16_t*)take(nc * 2);
    out->wdesc = (WinDesc*)take(nw * sizeof(WinDesc));
    (void)take(256);
    out->e_ref = (uint32_t*)take(3 * (size_t)Ec * 4);
    out->e_g = (float*)take(3 * (size_t)Ec * 4);
    out->log = shared_log ? shared_log : (float*)take(nc * (size_t)L * (size_t)d * 4);
    out->node_lo = (uint32_t*)take((size_t)N * 4);
    out->node_hi = (uint32_t*)take((size_t)N * 4);
    out->inv = (uint32_t*)take(nc * 4);
    out->rhead = (uint32_t*)take(nc * 4);
    out->wmask = (unsigned long long*)take((size_t)N * 8);
    out->wcls = (uint32_t*)take((2 * WIN_MAX_WINDOWS * 8 + WIN_MAX_WINDOWS) * 4);
    out->wzero_bytes = (size_t)(p - reinterpret_cast<char*>(out->wmask));      // (+ the chain table's share of wtab: wplan3_build)
    out->wtab = take(wplan3_table_bytes(Ec, batch));
    out->wblk = (uint32_t*)take(wplan3_blk_bytes(Ec, batch));
    {
        const size_t db = wplan_dense_bytes(Ec, batch, N, d, L);
        out->dense = db ? take(db) : nullptr;
    }
    out->heavy_thr = wplan_heavy_threshold(K, batch, d);
    out->chains = reinterpret_cast<Chain*>(out->base.light);
    out->chains_sparse = reinterpret_cast<Chain*>(out->base.heavy);
    // the second sort's keys live where the first sort's unsorted keys were (8 bytes per contribution, dead by then), its
    // payload where the first sort's payload was (dead once k_finish_w has read it)
    out->lk_in = reinterpret_cast<uint32_t*>(out->base.keys_in);
    out->lk_out = out->lk_in + nc;
    out->lv_in = out->base.vals_in;
    out->lv_out = out->base.vals_out;
    out->K = K;
    out->Ew = Ew;
    if (p > reinterpret_cast<char*>(ws) + ws_bytes) return TPNET_ERR_WORKSPACE;
    return TPNET_OK;
}

template <typename K>
__global__ void k_make_keys_w(K* __restrict__ keys, uint32_t* __restrict__ vals, const int64_t* __restrict__ src,
                              const int64_t* __restrict__ dst, int64_t Ec, int64_t B, int64_t N, int batch_bits, DescArgs da) {
    if (blockIdx.x >= da.kblocks) {
        for (int64_t b = (int64_t)(blockIdx.x - da.kblocks) * blockDim.x + threadIdx.x; b < da.nb;
             b += (int64_t)(gridDim.x - da.kblocks) * blockDim.x)
            batch_desc(b, da.desc, da.t, Ec, B, da.nb, da.now_time, da.t_prev, da.lambda, da.L);
        return;
    }
    const int64_t nc = 2 * Ec;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nc; j += (int64_t)da.kblocks * blockDim.x) {
        int64_t b, e;
        int side;
        decode(j, B, Ec, b, side, e);
        int64_t tgt = side ? dst[e] : src[e];
        if ((uint64_t)tgt >= (uint64_t)N) tgt = 0;  // counted (once per edge) in k_finish_w
        keys[j] = (K)(((uint64_t)tgt << batch_bits) | (uint64_t)b);
        vals[j] = (uint32_t)j;
    }
}

// first position in [lo, hi) whose key is >= key
template <typename K>
__device__ __forceinline__ int64_t lower_bound_keys(const K* __restrict__ keys, int64_t lo, int64_t hi, K key) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// exp(-lambda * dt) exactly as the kernels form a pending decay (device_common.hpp meta_view)
__device__ __forceinline__ float decay_f32(double lambda, double dt) {
    const float x = (float)(-lambda * dt);
    return (x == 0.0f) ? 1.0f : expf(x);
}

// version of `node` before batch b of the chunk: its last run with a smaller batch, if any, else the table row -- which is
// frozen while the chunk runs, so which copy is current and as of when is read from the node's meta record HERE, once:
// ref = log slot, or WREF_TABLE | copy; g = the version's decay to t_to
template <typename K>
__device__ __forceinline__ void resolve_version(const K* __restrict__ keys, int64_t nc, int batch_bits, int64_t node,
                                                int64_t b, const BatchDesc* __restrict__ desc,
                                                const NodeMeta* __restrict__ meta, double lambda, double t_to,
                                                uint32_t& ref, float& g, const uint32_t* __restrict__ node_lo,
                                                const uint32_t* __restrict__ node_hi) {
    const K skey = (K)(((uint64_t)node << batch_bits) | (uint64_t)b);
    // the node's own range of sorted positions bounds the search (k_node_ranges): a handful of steps for a typical node
    // instead of log2(2 * Ec) dependent loads over the whole chunk; [0, 0) for a node that is no target in this chunk
    const int64_t lo = node_lo[node], hi = node_hi[node];
    const int64_t pos = lower_bound_keys(keys, lo, hi, skey);
    (void)nc;
    if (pos > lo) {
        const K pk = keys[pos - 1];
        if ((pk >> batch_bits) == (skey >> batch_bits)) {
            ref = (uint32_t)(pos - 1);                              // the run's tail position = its log slot
            const int64_t bp = (int64_t)(pk & (((K)1 << batch_bits) - 1));
            g = decay_f32(lambda, t_to - desc[bp].t_last);
            return;
        }
    }
    const NodeMeta m = meta[node];
    const uint32_t c = m.ver & 1u;
    ref = WREF_TABLE | c;
    g = decay_f32(lambda, t_to - m.tref[c]);
}

// One pass over the sorted keys: [node_lo[n], node_hi[n]) = the sorted positions of node n's contributions in the chunk (both
// 0 for a node without any: the arrays are zeroed first); inv[v] = the sorted position of the contribution whose pre-sort
// index is v; rhead[j] = the first position of j's (node, batch) run.  With these, "the version of node p before batch b"
// needs NO search whenever p is itself a target in batch b -- which every partner of a contribution and both endpoints of
// an edge are (an edge contributes to both of its endpoints): it is the run that precedes p's run of batch b.
template <typename K>
__global__ void k_node_ranges(const K* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t nc, int batch_bits,
                              uint32_t* __restrict__ node_lo, uint32_t* __restrict__ node_hi, uint32_t* __restrict__ inv,
                              uint32_t* __restrict__ rhead) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nc; j += (int64_t)gridDim.x * blockDim.x) {
        const K key = keys[j];
        const K node = key >> batch_bits;
        if (j == 0 || (keys[j - 1] >> batch_bits) != node) node_lo[node] = (uint32_t)j;
        if (j == nc - 1 || (keys[j + 1] >> batch_bits) != node) node_hi[node] = (uint32_t)(j + 1);
        inv[vals[j]] = (uint32_t)j;
        // head of the run (backward gallop + binary search)
        int64_t h = j;
        if (j > 0 && keys[j - 1] == key) {
            int64_t hi = j, step = 1, lo;            // keys[hi] == key; find lo with keys[lo] != key (or -1)
            for (;;) {
                lo = hi - step;
                if (lo < 0) { lo = -1; break; }
                if (keys[lo] != key) break;
                hi = lo;
                step <<= 1;
            }
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if (keys[mid] == key) hi = mid; else lo = mid;
            }
            h = hi;
        }
        rhead[j] = (uint32_t)h;
    }
}

// version of `node` before batch b when the node is a target IN batch b and `pos` is one of its contributions of that batch:
// the run before the one `pos` is in, or the table.  false = `pos` is not such a position (bad ids): take the search.
template <typename K>
__device__ __forceinline__ bool version_before_own_run(const WPlan& p, const K* __restrict__ keys, int batch_bits, int64_t node,
                                                       int64_t b, uint32_t pos, const BatchDesc* __restrict__ desc,
                                                       const NodeMeta* __restrict__ meta, double lambda, double t_to,
                                                       uint32_t& ref, float& g) {
    const K bmask = ((K)1 << batch_bits) - 1;
    const K want = (K)(((uint64_t)node << batch_bits) | (uint64_t)b);
    if (keys[pos] != want) return false;
    const uint32_t h = p.rhead[pos];
    if (h > p.node_lo[node]) {                                 // an earlier run of the same node: its tail is its log slot
        ref = h - 1;
        g = decay_f32(lambda, t_to - desc[(int64_t)(keys[h - 1] & bmask)].t_last);
    } else {
        const NodeMeta m = meta[node];
        const uint32_t c = m.ver & 1u;
        ref = WREF_TABLE | c;
        g = decay_f32(lambda, t_to - m.tref[c]);
    }
    return true;
}

template <typename K>
__device__ __forceinline__ void finish_w(uint32_t bx, uint32_t nbx, WPlan p, const K* __restrict__ keys, const int64_t* __restrict__ src,
                           const int64_t* __restrict__ dst, const double* __restrict__ t, int64_t Ec, int64_t B, int64_t N,
                           int batch_bits, double lambda, uint32_t* __restrict__ lk, uint32_t* __restrict__ lv,
                           uint32_t lk_none /* key of a position that heads no chain: above every (window, length) */,
                           const NodeMeta* __restrict__ meta, uint32_t* err) {
    const int64_t nc = 2 * Ec;
    const int KW = p.K;
    const K bmask = ((K)1 << batch_bits) - 1;
    const BatchDesc* __restrict__ desc = p.base.desc;
    for (int64_t jb = (int64_t)bx * blockDim.x; jb < nc; jb += (int64_t)nbx * blockDim.x) {
        const int64_t j = jb + threadIdx.x;
        const bool active = j < nc;
        bool chain_head = false, last_run = false;
        int64_t w = 0;
        if (active) lv[j] = (uint32_t)j;
        Chain ch;
        ch.j0 = 0; ch.cnt = 0; ch.target = 0; ch.prev_ref = WREF_TABLE; ch.g_first = 1.0f; ch.pad0 = 0; ch.pad1 = 0; ch.pad2 = 0;
        if (active) {
            const K key = keys[j];
            const uint32_t val = p.base.vals_out[j];
            const int64_t b = (int64_t)(key & bmask);
            const int64_t node = (int64_t)(key >> batch_bits);
            w = b / KW;
            const double Tb = desc[b].t_last;
            int32_t partner;
            float wgt;
            contribution(p.base, src, dst, t, Ec, B, N, lambda, val, partner, wgt, err);
            p.base.s_partner[j] = partner;
            p.base.s_coef[j] = wgt;
            p.s_bc[j] = (uint16_t)b;
            p.base.s_target[j] = (int32_t)node;

            const bool has_prev = j > 0, has_next = j < nc - 1;
            const K kp = has_prev ? keys[j - 1] : key, kn = has_next ? keys[j + 1] : key;
            const bool run_head = !has_prev || kp != key;
            const bool run_tail = !has_next || kn != key;
            const bool same_node_prev = has_prev && (kp >> batch_bits) == (key >> batch_bits);
            const bool same_node_next = has_next && (kn >> batch_bits) == (key >> batch_bits);
            chain_head = !same_node_prev || (int64_t)(kp & bmask) / KW != w;
            last_run = run_tail && !same_node_next;
            const int64_t h = p.rhead[j];                // head of the run (k_node_ranges)
            const int64_t ri = j - h;
            uint32_t ref;
            float g;
            {
                // the partner is a target in this batch too (the other side of the same edge): its contribution's position
                int64_t cb, ce;
                int cside;
                decode((int64_t)val, B, Ec, cb, cside, ce);
                const int64_t e0b = cb * B;
                const int64_t neb = (Ec - e0b < B) ? (Ec - e0b) : B;
                const uint32_t sym = (uint32_t)(2 * e0b + (cside ? 0 : neb) + (ce - e0b));
                if (!version_before_own_run<K>(p, keys, batch_bits, (int64_t)partner, b, p.inv[sym], desc, meta, lambda, Tb, ref, g))
                    resolve_version(keys, nc, batch_bits, (int64_t)partner, b, desc, meta, lambda, Tb, ref, g, p.node_lo, p.node_hi);
            }
            uint32_t fl = 0;
            if (run_head) fl |= WREF_RUN_HEAD;
            if (run_tail) fl |= WREF_RUN_TAIL;
            if (ri % WIN_BLOCK == 0) fl |= WREF_BLK_HEAD;
            if (ri % WIN_BLOCK == WIN_BLOCK - 1 || run_tail) fl |= WREF_BLK_TAIL;
            if (last_run) fl |= WREF_LAST_RUN;
            p.s_ref[j] = ref | fl;
            p.s_g[j] = g;
            float dec = 1.0f;
            if (run_head && same_node_prev) dec = decay_f32(lambda, Tb - desc[(int64_t)(kp & bmask)].t_last);
            p.s_dec[j] = dec;

            if (chain_head) {
                // the chain ends where the node or the window does: first key >= (node, first batch of the next window)
                const K bound = (K)(((uint64_t)node << batch_bits) | (uint64_t)((w + 1) * KW));
                // (w+1)*KW may exceed the batch field only when it is past the chunk's last batch: then every key of
                // the node is smaller and the node's own range bounds the chain
                const K node_end = (K)(((uint64_t)(node + 1)) << batch_bits);
                const K stop = ((uint64_t)((w + 1) * KW) > (uint64_t)bmask) ? node_end : bound;
                int64_t lo = j, hi, step = 1;
                for (;;) {
                    const int64_t nx = lo + step;
                    if (nx >= nc) { hi = nc; break; }
                    if (keys[nx] >= stop) { hi = nx; break; }
                    lo = nx;
                    step <<= 1;
                }
                while (hi - lo > 1) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (keys[mid] < stop) lo = mid; else hi = mid;
                }
                ch.j0 = (uint32_t)j;
                ch.cnt = (uint32_t)(hi - j);
                ch.target = (int32_t)node;
                if (same_node_prev) {
                    ch.prev_ref = (uint32_t)(j - 1);
                    ch.g_first = 1.0f;                       // (a log row's decay comes with the run: s_dec)
                } else {
                    const NodeMeta m = meta[node];
                    const uint32_t c = m.ver & 1u;
                    ch.prev_ref = WREF_TABLE | c;
                    ch.g_first = decay_f32(lambda, Tb - m.tref[c]);
                }
            }
        }
        if (active) {
            // key of the second sort: chain heads by (window, decreasing length); everything else goes to the end
            const uint32_t c8 = ch.cnt > 255u ? 255u : ch.cnt;
            lk[j] = chain_head ? (((uint32_t)w << 8) | (255u - c8)) : lk_none;
            if (chain_head) p.chains_sparse[j] = ch;
        }
    }
}

// chain records in list order; the extra last block tables, per window, where its chains start in the (window, length)-sorted
// list, how many there are, how many are hubs
__global__ void k_gather_chains(WPlan p, const uint32_t* __restrict__ lk, const uint32_t* __restrict__ lv, int64_t nc,
                                int64_t nw, uint32_t heavy_threshold, uint32_t lk_none) {
    if (blockIdx.x == gridDim.x - 1) {
        auto lb = [&](uint32_t key) {
            int64_t lo = 0, hi = nc;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (lk[mid] < key) lo = mid + 1; else hi = mid;
            }
            return lo;
        };
        const uint32_t thr8 = heavy_threshold > 254u ? 254u : heavy_threshold;
        for (int64_t w = threadIdx.x; w < nw; w += blockDim.x) {
            const int64_t a = lb((uint32_t)w << 8);
            const int64_t h = lb(((uint32_t)w << 8) | (255u - thr8));     // lengths > thr8 sort before this key
            const int64_t z = lb((uint32_t)(w + 1) << 8);
            WinDesc D;
            D.start = (uint32_t)a;
            D.n_heavy = (uint32_t)(h - a);
            D.n_chains = (uint32_t)(z - a);
            D.n_ext = D.n_heavy;
            D.n_ext1 = D.n_heavy;
            {
                const int64_t m = lb(((uint32_t)w << 8) | (255u - (WIN_MED_MIN - 1u)));     // lengths >= WIN_MED_MIN sort before this key
                D.n_med = (uint32_t)((m > h ? m : h) - a);
            }
            D.pad1 = D.pad2 = 0;
            p.wdesc[w] = D;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += (int64_t)(gridDim.x - 1) * blockDim.x) {
        if (lk[i] >= lk_none) continue;
        p.chains[i] = p.chains_sparse[lv[i]];
    }
}

// readout references: the version of src / dst / neg of every edge before the edge's batch, decayed to the batch's
// readout clock (BatchDesc::now)
template <typename K>
__device__ __forceinline__ void edge_refs(uint32_t bx, uint32_t nbx, WPlan p, const K* __restrict__ keys, const int64_t* __restrict__ src,
                            const int64_t* __restrict__ dst, const int64_t* __restrict__ neg, int64_t Ec, int64_t B,
                            int64_t N, int batch_bits, double lambda, const NodeMeta* __restrict__ meta) {
    const int nwhich = neg ? 3 : 2;
    const int64_t tot = (int64_t)nwhich * Ec;
    const int64_t nc = 2 * Ec;
    for (int64_t x = (int64_t)bx * blockDim.x + threadIdx.x; x < tot; x += (int64_t)nbx * blockDim.x) {
        const int which = (int)(x / Ec);
        const int64_t e = x - (int64_t)which * Ec;
        int64_t node = which == 0 ? src[e] : (which == 1 ? dst[e] : neg[e]);
        if ((uint64_t)node >= (uint64_t)N) node = 0;               // the readout reports the bad id itself
        const int64_t b = e / B;
        uint32_t ref;
        float g;
        bool done = false;
        if (which < 2) {                                           // src / dst of the edge are targets in its batch
            const int64_t e0b = b * B;
            const int64_t neb = (Ec - e0b < B) ? (Ec - e0b) : B;
            const uint32_t orig = (uint32_t)(2 * e0b + (which ? neb : 0) + (e - e0b));
            done = version_before_own_run<K>(p, keys, batch_bits, node, b, p.inv[orig], p.base.desc, meta, lambda,
                                             p.base.desc[b].now, ref, g);
        }
        if (!done)
            resolve_version(keys, nc, batch_bits, node, b, p.base.desc, meta, lambda, p.base.desc[b].now, ref, g, p.node_lo, p.node_hi);
        p.e_ref[x] = ref;
        p.e_g[x] = g;
    }
}

// one launch: blocks [0, fgrid) resolve the sorted contributions (finish_w), the rest the readout references (edge_refs)
template <typename K>
__global__ void k_plan_w(WPlan p, const K* __restrict__ keys, const int64_t* __restrict__ src,
                         const int64_t* __restrict__ dst, const int64_t* __restrict__ neg, const double* __restrict__ t,
                         int64_t Ec, int64_t B, int64_t N, int batch_bits, double lambda, uint32_t* __restrict__ lk,
                         uint32_t* __restrict__ lv, uint32_t lk_none, const NodeMeta* __restrict__ meta, uint32_t* err,
                         uint32_t fgrid) {
    if (blockIdx.x < fgrid)
        finish_w<K>(blockIdx.x, fgrid, p, keys, src, dst, t, Ec, B, N, batch_bits, lambda, lk, lv, lk_none, meta, err);
    else
        edge_refs<K>(blockIdx.x - fgrid, gridDim.x - fgrid, p, keys, src, dst, neg, Ec, B, N, batch_bits, lambda, meta);
}

int wplan_build(const tpnet_state& st, const WPlan& p, const int64_t* src, const int64_t* dst, const int64_t* neg,
                const double* t, int64_t Ec, int64_t batch, double now_time, const double* t_prev_dev, double lambda,
                bool want_readout, hipStream_t s) {
    const int64_t nb = (Ec + batch - 1) / batch;
    const int64_t nc = 2 * Ec;
    const int64_t nw = (Ec + p.Ew - 1) / p.Ew;
    const int node_bits = ceil_log2_u64((uint64_t)st.N) < 1 ? 1 : ceil_log2_u64((uint64_t)st.N);
    // one spare value above the last batch: the chain-end bound (node, first batch of the next window) stays in the field
    const int batch_bits = ceil_log2_u64((uint64_t)nb + 1) < 1 ? 1 : ceil_log2_u64((uint64_t)nb + 1);
    const int key_bits = node_bits + batch_bits;
    if (key_bits > 63 || nb > 65535) return TPNET_ERR_BAD_ARG;
    int grid = (int)((nc + 255) / 256);
    if (grid > 4096) grid = 4096;
    const int dgrid = (int)((nb + 255) / 256) > 64 ? 64 : (int)((nb + 255) / 256);
    const DescArgs da{p.base.desc, t, nb, now_time, t_prev_dev, lambda, (int)st.L, (uint32_t)grid};
    const bool narrow = key_bits <= 31;       // (the chain-end bound (node + 1) << batch_bits must not wrap)
    uint32_t* k32_in = reinterpret_cast<uint32_t*>(p.base.keys_in);
    uint32_t* k32_out = reinterpret_cast<uint32_t*>(p.base.keys_out);
    size_t tmp = p.base.sort_tmp_bytes;
    if (narrow) {
        hipLaunchKernelGGL(k_make_keys_w<uint32_t>, dim3(grid + dgrid), dim3(256), 0, s, k32_in, p.base.vals_in, src, dst, Ec,
                           batch, st.N, batch_bits, da);
        TPNET_HIP_TRY(rocprim::radix_sort_pairs(p.base.sort_tmp, tmp, k32_in, k32_out, p.base.vals_in, p.base.vals_out,
                                                (size_t)nc, 0u, (unsigned)key_bits, s, false));
    } else {
        hipLaunchKernelGGL(k_make_keys_w<uint64_t>, dim3(grid + dgrid), dim3(256), 0, s, p.base.keys_in, p.base.vals_in, src,
                           dst, Ec, batch, st.N, batch_bits, da);
        TPNET_HIP_TRY(rocprim::radix_sort_pairs(p.base.sort_tmp, tmp, p.base.keys_in, p.base.keys_out, p.base.vals_in,
                                                p.base.vals_out, (size_t)nc, 0u, (unsigned)key_bits, s, false));
    }
    TPNET_HIP_TRY(hipGetLastError());
    const uint32_t thr = p.heavy_thr;   // contributions per (node, window) above which a workgroup per column part walks the chain
    const int egrid = (int)(((want_readout ? 3 : 0) * Ec + 255) / 256) > 4096 ? 4096 : (int)(((want_readout ? 3 : 0) * Ec + 255) / 256);
    const int win_bits = ceil_log2_u64((uint64_t)nw + 1) < 1 ? 1 : ceil_log2_u64((uint64_t)nw + 1);
    if (win_bits + 9 > 31) return TPNET_ERR_BAD_ARG;
    const uint32_t lk_none = 1u << (win_bits + 8);
    // (the second sort's payload array is the first sort's payload INPUT: dead since the first sort ran; its OUTPUT is read
    // by k_finish_w, which writes position j to lv_in[j] -- a different array)
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(st.meta);
    TPNET_HIP_TRY(hipMemsetAsync(p.node_lo, 0, (size_t)st.N * 4, s));
    TPNET_HIP_TRY(hipMemsetAsync(p.node_hi, 0, (size_t)st.N * 4, s));
    if (narrow)
        hipLaunchKernelGGL(k_node_ranges<uint32_t>, dim3(grid), dim3(256), 0, s, k32_out, p.base.vals_out, nc, batch_bits, p.node_lo,
                           p.node_hi, p.inv, p.rhead);
    else
        hipLaunchKernelGGL(k_node_ranges<uint64_t>, dim3(grid), dim3(256), 0, s, p.base.keys_out, p.base.vals_out, nc, batch_bits,
                           p.node_lo, p.node_hi, p.inv, p.rhead);
    if (narrow)
        hipLaunchKernelGGL(k_plan_w<uint32_t>, dim3(grid + (want_readout ? egrid : 0)), dim3(256), 0, s, p, k32_out, src, dst,
                           neg, t, Ec, batch, st.N, batch_bits, lambda, p.lk_in, p.lv_in, lk_none, meta, st.err, (uint32_t)grid);
    else
        hipLaunchKernelGGL(k_plan_w<uint64_t>, dim3(grid + (want_readout ? egrid : 0)), dim3(256), 0, s, p, p.base.keys_out,
                           src, dst, neg, t, Ec, batch, st.N, batch_bits, lambda, p.lk_in, p.lv_in, lk_none, meta, st.err,
                           (uint32_t)grid);
    TPNET_HIP_TRY(hipGetLastError());
    // second sort: chain heads by (window, decreasing length) -> every window's chain list, long chains first
    {
        size_t tmp2 = p.base.sort_tmp_bytes;
        TPNET_HIP_TRY(rocprim::radix_sort_pairs(p.base.sort_tmp, tmp2, p.lk_in, p.lk_out, p.lv_in, p.lv_out, (size_t)nc, 0u,
                                                (unsigned)(win_bits + 9), s, false));
        hipLaunchKernelGGL(k_gather_chains, dim3(grid + 1), dim3(256), 0, s, p, p.lk_out, p.lv_out, nc, nw, thr, lk_none);
    }
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

}  // namespace tpnet
