// Host-array entry points of the per-batch API (include/tpnet_hip.h: tpnet_stage_*, tpnet_host_pair_feature,
// tpnet_host_update).  The reference's callers hand HOST numpy arrays to every call (models/TPNet.py:74-77, 107, 117;
// train_link_prediction.py:359-373) and the stock path pays one pageable host->device copy per array.  Here the ids are
// checked and copied into a slot of a pinned, device-mapped ring on the host, and the kernels read the slot directly: the
// per-batch call is host-bound (a batch is microseconds of GPU work), so what counts is the number of runtime calls per
// API call -- one kernel launch for a readout, two launches for an update, and one event record per FILLED slot of the ring.
// (Measured and not kept: the update's plan kernel on a helper stream beside the batch's two readouts -- the GPU side would
// allow it, but two cross-stream event waits and two more event records cost the call 11 us of host time, and the per-batch
// loop is host-bound: C2 59-69 us per batch against 60.)
#include "tpnet_common.h"

#include <cstring>
#include <new>

struct tpnet_stage {
    char* base = nullptr;        // hipHostMalloc'ed (mapped, portable) -- or, device_memory: DEVICE memory the host writes through the BAR
    char* dev_base = nullptr;    // its device address
    bool device_memory = false;
    size_t slot_bytes = 0;
    int32_t n_slots = 0;
    int32_t pos = 0;             // the slot being filled
    size_t off = 0;              // bytes of it handed out
    bool open = false;           // launches that read the slot being filled have been enqueued (on `cur`) since it was entered
    hipStream_t cur = nullptr;
    hipEvent_t* ev = nullptr;    // recorded when a slot is LEFT: behind every launch that read it
    bool* used = nullptr;
};

namespace tpnet {

// A call's arrays are carved from the slot being filled; the slot is left -- ONE event record for all the launches that read
// it -- when the next request does not fit (or arrives on another stream), and a slot is entered again only after its event
// has completed.  A decoder-sized call (16-24 KB) therefore costs an event record once per ~10-16 calls instead of per call:
// an event record is a barrier packet in the queue, ~6 us of GPU timeline between two kernels (rocprofv3 kernel trace of the
// per-batch loop at C2: 66 us per batch with an event per call, 48 us of kernels).
static int stage_acquire(tpnet_stage* sg, size_t bytes, hipStream_t s, char** host, char** dev) {
    if (!sg || bytes > sg->slot_bytes) return TPNET_ERR_BAD_ARG;
    if (sg->off + bytes > sg->slot_bytes || (sg->open && s != sg->cur)) {
        if (sg->open) {
            TPNET_HIP_TRY(hipEventRecord(sg->ev[sg->pos], sg->cur));
            sg->used[sg->pos] = true;
            sg->open = false;
        }
        sg->pos = (sg->pos + 1) % sg->n_slots;
        sg->off = 0;
        if (sg->used[sg->pos]) {
            TPNET_HIP_TRY(hipEventSynchronize(sg->ev[sg->pos]));   // the launches that read this slot last are done
            sg->used[sg->pos] = false;
        }
    }
    *host = sg->base + (size_t)sg->pos * sg->slot_bytes + sg->off;
    *dev = sg->dev_base + (size_t)sg->pos * sg->slot_bytes + sg->off;
    return TPNET_OK;
}

// the host has written a call's arrays: a ring in device memory was written through write-combining stores, which must have left the
// core before the launch's doorbell is rung
static inline void stage_written(const tpnet_stage* sg) {
    if (sg->device_memory) __builtin_ia32_sfence();
}

// the launches that read `bytes` from the acquired address are enqueued on s
static int stage_release(tpnet_stage* sg, size_t bytes, hipStream_t s) {
    sg->off += (bytes + 255) / 256 * 256;
    sg->open = true;
    sg->cur = s;
    return TPNET_OK;
}

// ids -> the slot, range-checked; python-style negative ids wrap (ATen indexing semantics, as the reference's P[i][ids])
static bool copy_ids(int64_t* dst, const int64_t* src, int64_t n, int64_t N) {
    bool ok = true;
    for (int64_t i = 0; i < n; ++i) {
        int64_t x = src[i];
        if (x < 0) x += N;
        ok &= (uint64_t)x < (uint64_t)N;
        dst[i] = x;
    }
    return ok;
}

}  // namespace tpnet

using namespace tpnet;

extern "C" {

// mode: 0 = pinned, device-mapped host memory (the kernels read the ids over PCIe: one more ~2-us round trip at the head of every
// kernel's dependent chain); 1 = device memory that the host writes through the large BAR (fine-grained: the kernels read what the
// host wrote, not a stale cache line; +~1.6 us of host time per 16 KB call, -~1.5 us at the head of every kernel that reads a slot);
// -1 = 1 where the device has a large BAR, else 0
int tpnet_stage_create_ex(int32_t slots, size_t slot_bytes, int32_t mode, tpnet_stage** out) {
    if (!out || slots < 2 || slots > 1024 || slot_bytes < 64 || mode < -1 || mode > 1) return TPNET_ERR_BAD_ARG;
    *out = nullptr;
    tpnet_stage* sg = new (std::nothrow) tpnet_stage();
    if (!sg) return TPNET_ERR_BAD_ARG;
    sg->slot_bytes = (slot_bytes + 255) / 256 * 256;
    sg->n_slots = slots;
    void* mem = nullptr;
    bool bar = false;
    if (mode != 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.isLargeBar) bar = true;
        if (mode == 1 && !bar) { delete sg; return TPNET_ERR_NO_DEVICE; }
    }
    if (bar && hipExtMallocWithFlags(&mem, sg->slot_bytes * (size_t)slots, hipDeviceMallocFinegrained) == hipSuccess) {
        sg->device_memory = true;
        sg->base = reinterpret_cast<char*>(mem);
        sg->dev_base = sg->base;
    } else {
        if (mode == 1) { delete sg; return TPNET_ERR_HIP; }
        if (hipHostMalloc(&mem, sg->slot_bytes * (size_t)slots, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
            delete sg;
            return TPNET_ERR_HIP;
        }
        sg->base = reinterpret_cast<char*>(mem);
        void* dptr = nullptr;
        if (hipHostGetDevicePointer(&dptr, mem, 0) != hipSuccess) { (void)hipHostFree(mem); delete sg; return TPNET_ERR_HIP; }
        sg->dev_base = reinterpret_cast<char*>(dptr);
    }
    auto free_mem = [&]() { if (sg->device_memory) (void)hipFree(mem); else (void)hipHostFree(mem); };
    sg->ev = new (std::nothrow) hipEvent_t[slots];
    sg->used = new (std::nothrow) bool[slots];
    if (!sg->ev || !sg->used) { free_mem(); delete[] sg->ev; delete[] sg->used; delete sg; return TPNET_ERR_BAD_ARG; }
    for (int i = 0; i < slots; ++i) {
        sg->used[i] = false;
        if (hipEventCreateWithFlags(&sg->ev[i], hipEventDisableTiming) != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipEventDestroy(sg->ev[j]);
            free_mem();
            delete[] sg->ev; delete[] sg->used; delete sg;
            return TPNET_ERR_HIP;
        }
    }
    *out = sg;
    return TPNET_OK;
}

int tpnet_stage_create(int32_t slots, size_t slot_bytes, tpnet_stage** out) { return tpnet_stage_create_ex(slots, slot_bytes, 0, out); }

int tpnet_stage_in_device_memory(const tpnet_stage* sg) { return (sg && sg->device_memory) ? 1 : 0; }

int tpnet_stage_destroy(tpnet_stage* sg) {
    if (!sg) return TPNET_OK;
    if (sg->open) (void)hipDeviceSynchronize();          // launches of the slot being filled: no event covers them yet
    for (int i = 0; i < sg->n_slots; ++i) {
        if (sg->used[i]) (void)hipEventSynchronize(sg->ev[i]);
        (void)hipEventDestroy(sg->ev[i]);
    }
    if (sg->device_memory) (void)hipFree(sg->base);
    else (void)hipHostFree(sg->base);
    delete[] sg->ev;
    delete[] sg->used;
    delete sg;
    return TPNET_OK;
}

int64_t tpnet_stage_max_pairs(const tpnet_stage* sg) { return sg ? (int64_t)(sg->slot_bytes / 16) : 0; }

int64_t tpnet_stage_max_batch(const tpnet_stage* sg) {
    if (!sg) return 0;
    const int64_t fit = (int64_t)(sg->slot_bytes / 24), cap = plan_one_max_batch();
    return fit < cap ? fit : cap;
}

int tpnet_pair_feature(const tpnet_state* st, const int64_t* u, const int64_t* v, int64_t n, double now_time,
                       double lambda, uint32_t flags, const tpnet_mlp* mlp, float* out_gram, float* out, void* stream) {
    if (!st || !st->p0 || !st->q || !st->meta || !st->err || st->N < 1 || st->d < 1 || st->L < 1 || st->L > TPNET_MAX_LAYERS)
        return TPNET_ERR_BAD_ARG;
    if (n < 0 || !mlp || (n > 0 && (!u || !v || !out))) return TPNET_ERR_BAD_ARG;
    return launch_pair_feature(*st, u, v, n, now_time, lambda, flags, *mlp, out_gram, out, (hipStream_t)stream);
}

int tpnet_host_pair_feature(const tpnet_state* st, tpnet_stage* stage, const int64_t* h_u, const int64_t* h_v, int64_t n,
                            double now_time, double lambda, uint32_t flags, const tpnet_mlp* mlp, float* out_gram,
                            float* out, void* stream) {
    if (!st || !st->p0 || !st->q || !st->meta || !st->err || st->N < 1 || st->d < 1 || st->L < 1 || st->L > TPNET_MAX_LAYERS)
        return TPNET_ERR_BAD_ARG;
    if (n < 0 || (n > 0 && (!h_u || !h_v || !out)) || !stage) return TPNET_ERR_BAD_ARG;
    if (n == 0) return TPNET_OK;
    char *host = nullptr, *dev = nullptr;
    hipStream_t s = (hipStream_t)stream;
    int rc = stage_acquire(stage, (size_t)n * 16, s, &host, &dev);
    if (rc) return rc;
    int64_t* hu = reinterpret_cast<int64_t*>(host);
    if (!copy_ids(hu, h_u, n, st->N) || !copy_ids(hu + n, h_v, n, st->N)) return TPNET_ERR_INDEX;   // nothing consumed
    stage_written(stage);
    const int64_t* du = reinterpret_cast<const int64_t*>(dev);
    if (mlp)
        rc = launch_pair_feature(*st, du, du + n, n, now_time, lambda, flags, *mlp, out_gram, out, s);
    else
        rc = launch_pair_gram(*st, du, du + n, n, now_time, lambda, flags, out, s);
    if (rc) return rc;
    return stage_release(stage, (size_t)n * 16, s);
}

int tpnet_host_update(const tpnet_state* st, tpnet_stage* stage, const int64_t* h_src, const int64_t* h_dst,
                      const double* h_t, int64_t B, double now_time, double lambda, uint32_t launch_id, uint32_t flags,
                      void* workspace, size_t ws_bytes, void* stream) {
    if (!st || !st->p0 || !st->q || !st->meta || !st->err || st->N < 1 || st->d < 1 || st->L < 1 || st->L > TPNET_MAX_LAYERS)
        return TPNET_ERR_BAD_ARG;
    if (B < 1 || !h_src || !h_dst || !h_t || !stage) return TPNET_ERR_BAD_ARG;
    if (launch_id == 0 || launch_id >= 0x7FFFFFFFu) return TPNET_ERR_BAD_ARG;
    if (flags & TPNET_FLAG_EAGER_DECAY) return TPNET_ERR_BAD_ARG;     // (the exact mode's dense decay is tpnet_decay, before this call)
    if (B > tpnet_stage_max_batch(stage)) {
        // a batch too large for the single-workgroup plan: the staged arrays are copied to the tail of the workspace (the
        // chunk planner gathers from them at random: not something to do over PCIe) and the per-batch path plans + steps
        if ((size_t)B * 24 > stage->slot_bytes) return TPNET_ERR_BAD_ARG;
        const size_t pb = (plan_bytes(B, B) + 255) / 256 * 256;
        if (!workspace || pb + (size_t)B * 24 > ws_bytes) return TPNET_ERR_WORKSPACE;
        char *host = nullptr, *dev = nullptr;
        hipStream_t s = (hipStream_t)stream;
        int rc = stage_acquire(stage, (size_t)B * 24, s, &host, &dev);
        if (rc) return rc;
        int64_t* hs = reinterpret_cast<int64_t*>(host);
        if (!copy_ids(hs, h_src, B, st->N) || !copy_ids(hs + B, h_dst, B, st->N)) return TPNET_ERR_INDEX;
        memcpy(hs + 2 * B, h_t, (size_t)B * 8);
        stage_written(stage);
        int64_t* d = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(workspace) + pb);
        TPNET_HIP_TRY(hipMemcpyAsync(d, host, (size_t)B * 24, hipMemcpyHostToDevice, s));
        rc = stage_release(stage, (size_t)B * 24, s);
        if (rc) return rc;
        return tpnet_update(st, d, d + B, reinterpret_cast<const double*>(d + 2 * B), B, now_time, lambda, launch_id, flags,
                            workspace, pb, stream);
    }
    Plan p{};
    int rc = plan_carve(workspace, ws_bytes, B, B, &p);
    if (rc) return rc;
    char *host = nullptr, *dev = nullptr;
    hipStream_t s = (hipStream_t)stream;
    rc = stage_acquire(stage, (size_t)B * 24, s, &host, &dev);
    if (rc) return rc;
    int64_t* hs = reinterpret_cast<int64_t*>(host);
    if (!copy_ids(hs, h_src, B, st->N) || !copy_ids(hs + B, h_dst, B, st->N)) return TPNET_ERR_INDEX;
    memcpy(hs + 2 * B, h_t, (size_t)B * 8);
    stage_written(stage);
    const int64_t* ds = reinterpret_cast<const int64_t*>(dev);
    rc = plan_one(*st, p, ds, ds + B, reinterpret_cast<const double*>(ds + 2 * B), B, now_time, lambda, flags, s);
    if (rc) return rc;
    rc = stage_release(stage, (size_t)B * 24, s);                      // the plan kernel is the only reader of the slot
    if (rc) return rc;
    StreamArgs a{};
    a.own_mod = 1;
    return launch_step(*st, a, p, 0, B, (int32_t)B, lambda, launch_id, flags | ROLE_UPDATE, s);
}

// The encoder's whole readout for one (src, other) batch (tpnet_encoder_features) with the BATCH's arrays on the host -- what the
// reference's loop holds (train_link_prediction.py:325-340: numpy slices of the edge list): the three arrays go through the staging
// ring and the row set-up kernel reads them there, so the call enqueues no copy (a pinned copy + its cross-queue barrier is ~20 us
// of GPU timeline per batch in the rocprofv3 trace of bench.py's encoder-level loop).  Ids outside [0, N) give TPNET_ERR_INDEX.
int tpnet_host_encoder_features(const tpnet_state* st, tpnet_stage* stage, const void* sampler, int64_t E, int64_t num_nodes,
                                const int64_t* h_src, const int64_t* h_other, const double* h_t, int64_t B, int32_t K,
                                double now_time, double lambda, uint32_t flags, const tpnet_mlp* mlp, void* scratch,
                                size_t scratch_bytes, float* gram, float* out, void* stream) {
    if (!st || !stage || st->N < 1 || B < 0 || (B > 0 && (!h_src || !h_other || !h_t))) return TPNET_ERR_BAD_ARG;
    if (B == 0) return TPNET_OK;
    char *host = nullptr, *dev = nullptr;
    hipStream_t s = (hipStream_t)stream;
    int rc = stage_acquire(stage, (size_t)B * 24, s, &host, &dev);
    if (rc) return rc;
    int64_t* hs = reinterpret_cast<int64_t*>(host);
    for (int64_t i = 0; i < B; ++i) {
        if ((uint64_t)h_src[i] >= (uint64_t)st->N || (uint64_t)h_other[i] >= (uint64_t)st->N) return TPNET_ERR_INDEX;
        hs[i] = h_src[i];
        hs[B + i] = h_other[i];
    }
    memcpy(hs + 2 * B, h_t, (size_t)B * 8);
    stage_written(stage);
    const int64_t* ds = reinterpret_cast<const int64_t*>(dev);
    rc = tpnet_encoder_features(st, sampler, E, num_nodes, ds, ds + B, reinterpret_cast<const double*>(ds + 2 * B), B, K, now_time,
                                lambda, flags, mlp, scratch, scratch_bytes, gram, out, stream);
    if (rc) return rc;
    return stage_release(stage, (size_t)B * 24, s);                   // (the row set-up kernel is the only reader of the slot)
}

// Is (src, dst) of a get_pair_wise_feature call the ENCODER's pattern (models/TPNet.py:311-316): src = tile(neigh, 2), dst =
// concat(repeat(a1, K), repeat(a2, K)) with one K for both halves?  One pass over the two host arrays: the halves of src compared,
// K = gcd of the positions where dst changes inside a half (any K whose blocks are constant serves).  Returns K (>= 2), or 0 (not
// the pattern).  The ids' range is the caller's to check.
// The module then ships n/2 neighbour ids + 2 n/(2K) anchors instead of 2 n ids, and runs the anchored readout.
int64_t tpnet_host_encoder_pattern(const int64_t* src, const int64_t* dst, int64_t n, int64_t N) {
    if (!src || !dst || n < 4 || (n & 1) || N < 1) return 0;
    const int64_t h = n / 2;
    if (memcmp(src, src + h, (size_t)h * 8) != 0) return 0;
    (void)N;                                                             // (the caller checks the ids' range: numpy's reductions are SIMD)
    // the usual case first: K = the first run's length, every block of K constant in both halves (a branch-free pass again)
    int64_t k0 = 1;
    while (k0 < h && dst[k0] == dst[0]) ++k0;
    if (k0 >= 2 && h % k0 == 0) {
        uint64_t acc = 0;
        for (int64_t j = 0; j < n; j += k0) {
            const int64_t first = dst[j];
            for (int64_t k = 1; k < k0; ++k) acc |= (uint64_t)(dst[j + k] ^ first);
        }
        if (acc == 0) return k0;
    }
    // runs of several lengths (two rows with the same anchor make a run of 2 K): K = gcd of the positions where dst changes
    auto gcd = [](int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; };
    int64_t g = h;
    for (int half = 0; half < 2; ++half) {
        const int64_t* x = dst + half * h;
        for (int64_t i = 1; i < h; ++i)
            if (x[i] != x[i - 1]) { g = gcd(g, i); if (g < 2) return 0; }
    }
    return g;
}


// The encoder's call from the reference's OWN host index arrays (models/TPNet.py:311-316: src = tile(neigh, 2), dst =
// concat(repeat(a1, K), repeat(a2, K)), 2 x 640 KB at C2) in ONE crossing (round 4): the pattern check (the halves of src equal; dst
// made of constant blocks of K), the ids' range, the staging of the n / 2 neighbour ids + 2 n / (2 K) anchors into the pinned ring
// (the kernel reads them there: no copy is enqueued) and the launch of readout + self.mlp.  Until round 4 the module did the
// check in C (37 us), two numpy reductions for the range (10 us) and a pinned copy + an enqueued host-to-device copy from Python
// (30 us): 82 us of host time per call against 46 us of GPU time -- the loop was host-bound.  The loops below are written for
// the vectoriser (AVX2 where the CPU has it: the function is multi-versioned); what remains is reading the 1.28 MB once.
// *K_out = the pattern's K (>= 4) if the call was served; 0 = not this pattern / not served (the caller takes the general path,
// which also reports ids out of range).
#if defined(__HIP_DEVICE_COMPILE__)
#define TPNET_HOST_SIMD
#else
#define TPNET_HOST_SIMD __attribute__((target_clones("avx2", "default")))
#endif
TPNET_HOST_SIMD
static int64_t pattern_and_stage(const int64_t* src, const int64_t* dst, int64_t n, int64_t N, int64_t* out_neigh, int64_t k_given,
                                 int64_t* out_a1, int64_t* out_a2) {
    // second phase (k_given > 0): copy + range; first phase (k_given == 0): detection only.  Returns K, or 0.
    const int64_t h = n / 2;
    if (k_given == 0) {
        if (memcmp(src, src + h, (size_t)h * 8) != 0) return 0;
        int64_t k0 = 1;
        while (k0 < h && dst[k0] == dst[0]) ++k0;
        if (k0 < 2 || h % k0 != 0) return 0;
        uint64_t acc = 0;
        for (int64_t j = 0; j < n; j += k0) {
            const int64_t first = dst[j];
            uint64_t a = 0;
            for (int64_t k = 1; k < k0; ++k) a |= (uint64_t)(dst[j + k] ^ first);
            acc |= a;
        }
        return acc == 0 ? k0 : 0;
    }
    const int64_t K = k_given, m = h / K;
    int64_t mx = 0;
    uint64_t neg = 0;
    for (int64_t i = 0; i < h; ++i) {                       // neighbour ids: copy, max and sign in one pass
        const int64_t x = src[i];
        out_neigh[i] = x;
        neg |= (uint64_t)x;
        mx = x > mx ? x : mx;
    }
    for (int64_t j = 0; j < m; ++j) {
        const int64_t x = dst[j * K], y = dst[h + j * K];
        out_a1[j] = x;
        out_a2[j] = y;
        neg |= (uint64_t)x | (uint64_t)y;
        mx = x > mx ? x : mx;
        mx = y > mx ? y : mx;
    }
    return ((neg >> 63) == 0 && mx < N) ? K : 0;
}

int tpnet_host_anchored_features(const tpnet_state* st, tpnet_stage* stage, const int64_t* h_src, const int64_t* h_dst, int64_t n,
                                 double now_time, double lambda, uint32_t flags, const tpnet_mlp* mlp, float* gram, float* out,
                                 int32_t* K_out, void* stream) {
    if (!K_out) return TPNET_ERR_BAD_ARG;
    *K_out = 0;
    if (!st || !stage || !h_src || !h_dst || !mlp || !out || st->N < 1 || n < 8 || (n & 1)) return TPNET_OK;
    const int64_t h = n / 2;
    const int64_t K = pattern_and_stage(h_src, h_dst, n, st->N, nullptr, 0, nullptr, nullptr);
    if (K < 4 || K > (1 << 20)) return TPNET_OK;
    const int64_t m = h / K;
    // served where readout + dense layers are one launch, or the two-launch path has its feature buffer
    if (!(tpnet_encoder_fused_supported(st, m, (int32_t)K, mlp) || (gram && tpnet_pair_gram_anchored_supported(st)))) return TPNET_OK;
    const size_t bytes = (size_t)(h + 2 * m) * 8;
    if (bytes > stage->slot_bytes) return TPNET_OK;
    hipStream_t s = (hipStream_t)stream;
    char *host = nullptr, *dev = nullptr;
    int rc = stage_acquire(stage, bytes, s, &host, &dev);
    if (rc) return rc;
    int64_t* hn = reinterpret_cast<int64_t*>(host);
    if (pattern_and_stage(h_src, h_dst, n, st->N, hn, K, hn + h, hn + h + m) == 0) return TPNET_OK;   // an id out of range: general path
    stage_written(stage);
    const int64_t* dn = reinterpret_cast<const int64_t*>(dev);
    rc = tpnet_anchored_features(st, dn, dn + h, dn + h + m, m, (int32_t)K, now_time, lambda, flags, mlp, gram, out, stream);
    if (rc) return rc;
    rc = stage_release(stage, bytes, s);
    if (rc) return rc;
    *K_out = (int32_t)K;
    return TPNET_OK;
}

}  // extern "C"
