// C ABI (include/tpnet_hip.h): argument checks, workspace carving, launch sequencing.  No allocation, no implicit
// synchronisation; everything is enqueued on the caller's stream.
#include "tpnet_common.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace tpnet {
thread_local int g_last_hip_error = 0;

static int check_state(const tpnet_state* st) {
    if (!st || !st->p0 || !st->q || !st->meta || !st->err) return TPNET_ERR_BAD_ARG;
    if (st->N < 1 || st->d < 1 || st->L < 1 || st->L > TPNET_MAX_LAYERS) return TPNET_ERR_BAD_ARG;
    if (st->N >= (1ll << 31)) return TPNET_ERR_BAD_ARG;
    return TPNET_OK;
}

// largest chunk (multiple of `batch`, at most E) whose plan fits the workspace.  Contribution indices are 32-bit inside a
// chunk (Item::j0, the sort's payload), so a chunk never exceeds 2^30 edges however large the workspace is.
static int64_t max_chunk(size_t ws_bytes, int64_t E, int64_t batch) {
    const int64_t hard = ((int64_t)1 << 30) / batch * batch;
    const int64_t lim = (E < hard || hard < batch) ? E : hard;
    if (plan_bytes(lim, batch) <= ws_bytes) return lim;
    int64_t lo = 0, hi = (lim + batch - 1) / batch;  // in batches; lo fits (0), hi does not
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) / 2;
        if (plan_bytes(mid * batch, batch) <= ws_bytes) lo = mid; else hi = mid;
    }
    return lo * batch;
}

struct StepTimer {
    hipEvent_t* ev = nullptr;  // one pair per chunk, around its loop of step launches (plan kernels excluded)
    int64_t n = 0, cap = 0;    // pairs recorded / available
    int64_t launches = 0;      // step launches between the recorded pairs
    int64_t batches = 0;       // batches those launches covered
    int64_t edges = 0;         // edges those launches covered
    // windowed path: L+2 events per timed window (before the first update launch, after each launch)
    hipEvent_t* wev = nullptr;
    int64_t wn = 0, wcap = 0;  // windows recorded / available
    int64_t wedges = 0;        // edges of the recorded windows
};

// The plans of a multi-chunk stream side by side -- one region per chunk, each with every array of a plan but the version log --
// in front of ONE version log: what lets a stream of several chunks be replayed (run_stream_windowed).  At most 64 chunks.
constexpr int64_t ARENA_MAX_CHUNKS = 64;
static size_t arena_region_bytes(int64_t N, int d, int L, int64_t chunk, int64_t batch) {
    return (wplan_bytes(chunk, batch, N, d, L) - wplan_log_bytes(chunk, d, L) + 511) / 256 * 256;
}
static size_t arena_bytes(int64_t N, int d, int L, int64_t E, int64_t chunk, int64_t batch) {
    const int64_t n = (E + chunk - 1) / chunk;
    return (size_t)n * arena_region_bytes(N, d, L, chunk, batch) + wplan_log_bytes(chunk, d, L) + 512;
}

// edges per window (*Ew_out) and the most a chunk may cover whatever the workspace (whole windows, or all of E); 0: not windowed
static int64_t window_chunk_limit(int64_t N, int d, int L, int64_t E, int64_t batch, uint32_t flags, int* K_out, int64_t* Ew_out) {
    const int K = wplan_window_batches(batch, d, L);
    if (K == 0) return 0;
    *K_out = K;
    int64_t Ew = (int64_t)K * batch;
    // packed rows of (2L+2)(2L+3)/2 floats: every chunk's output must start on a 16-byte boundary (launch_wstep)
    const int NN = 2 * L + 2;
    if ((flags & TPNET_FLAG_PACKED) && ((NN * (NN + 1) / 2) % 4) != 0 && Ew % 4 != 0) Ew *= (Ew % 2 == 0) ? 2 : 4;
    *Ew_out = Ew;
    int64_t cap = wplan_max_chunk_edges(batch, d, L);                // the version log of a chunk is bounded
    // batches that fit one workgroup's LDS: chunks of at most 64 windows, which the hashed planner serves (wplan3.hip) -- ONE set
    // of plan kernels whatever the stream's length (a chunk that falls to the sorted planner meets rocPRIM's large-size sort
    // kernels for the first time in the middle of a long call: HIP resolves a kernel at its first launch, ~0.3 ms each)
    // (the dense planner, wplan_dense.hip: one launch, up to 256 windows)
    const bool dense = !(flags & (TPNET_FLAG_PLAN_SORTED | TPNET_FLAG_PLAN_HASHED)) && wplan_dense_eligible(N, d, L, batch);
    const int64_t maxw = dense ? 256 : WIN_MAX_WINDOWS;
    if (batch <= PLAN_ONE_MAX && !(flags & TPNET_FLAG_PLAN_SORTED) && cap > maxw * Ew) cap = maxw * Ew;
    const int64_t hard = cap / Ew * Ew;
    return (E <= cap) ? E : hard;
}

// The windowed path (wstep.hip) serves a stream when its arithmetic contract allows it (no eager decay / strictly
// sequential sums: those are the per-batch kernels' exact mode), the rows take 16-byte vectors, there are enough batches
// for a window to pay, and the caller's workspace holds the plan of at least one window.  Returns the chunk (edges per
// plan: whole windows of Kmax batches, or all of E) and Kmax; 0 = use the per-batch path.  `region_out` (may be null: the caller
// has no use for a replayable layout): set to the bytes per chunk region where the stream takes several chunks and the workspace
// holds them side by side (arena_bytes), else to 0.
static int64_t window_chunk(const tpnet_state& st, size_t ws_bytes, int64_t E, int64_t batch, uint32_t flags, int* K_out,
                            size_t* region_out = nullptr) {
    if (region_out) *region_out = 0;
    if (flags & (TPNET_FLAG_EAGER_DECAY | TPNET_FLAG_SEQUENTIAL | TPNET_FLAG_SCHED_BATCH)) return 0;
    const int64_t nb = (E + batch - 1) / batch;
    // short streams: the pipeline pays its plan (~31 us for 20 C2 batches with the dense planner, wplan_dense.hip) and L + 1 dependent
    // launches up front, the per-batch schedule ~20 us and ~6.6 us per batch (C2): the pipeline wins from 16 batches
    // (tools/short_sweep.py, round 4: 14 batches 138 against 134 us, 16: 135 against 145, 20: 143-158 against 173-179, 24: 157 against
    // 198; until round 4, with the hashed planner and lane-group walks of chains up to 52 contributions: from 24).  Larger batches keep
    // the chunk planner's crossover (two device-wide sorts).
    static const int min_nb3 = TPNET_DEV_INT(WIN_MIN_BATCHES, 16);
    const int min_nb = batch <= PLAN_ONE_MAX ? min_nb3 : 56;
    int64_t Ew = 0;
    const int64_t lim = window_chunk_limit(st.N, st.d, st.L, E, batch, flags, K_out, &Ew);
    if (lim == 0 || nb < ((flags & TPNET_FLAG_SCHED_WINDOWED) ? 4 : min_nb)) return 0;
    if ((lim + batch - 1) / batch < 4) return 0;
    const bool lim_fits = wplan_bytes(lim, batch, st.N, st.d, st.L) <= ws_bytes;
    if (lim_fits && lim >= E) return lim;
    if (region_out && batch <= PLAN_ONE_MAX && !(flags & TPNET_FLAG_PLAN_SORTED)) {      // (the sorted planner keeps nothing to replay)
        // several chunks: the largest chunk (whole windows, at least two chunks, at most ARENA_MAX_CHUNKS) whose arena fits, if any
        // (the log grows with the chunk, the per-node arrays of the regions with the number of chunks: not monotone, and at most
        // 256 candidates -- a chunk is at most 256 windows)
        int64_t hi = lim / Ew;                                          // (lim < E: whole windows; else lim = E did not fit)
        if (hi * Ew >= E) hi = (E - 1) / Ew;
        int64_t lo = (E + ARENA_MAX_CHUNKS * Ew - 1) / (ARENA_MAX_CHUNKS * Ew);
        if (lo < 1) lo = 1;
        if (lo * Ew < 4 * batch) lo = (4 * batch + Ew - 1) / Ew;
        for (int64_t w = hi; w >= lo; --w)
            if (arena_bytes(st.N, st.d, st.L, E, w * Ew, batch) <= ws_bytes) {
                *region_out = arena_region_bytes(st.N, st.d, st.L, w * Ew, batch);
                return w * Ew;
            }
    }
    if (lim_fits) return lim;
    int64_t lo = 0, hi = (lim + Ew - 1) / Ew;       // in windows; lo fits (0), hi does not
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) / 2;
        if (wplan_bytes(mid * Ew, batch, st.N, st.d, st.L) <= ws_bytes) lo = mid; else hi = mid;
    }
    if (lo > 0) return lo * Ew;
    // not even one window of Kmax batches (a caller who capped the version log below that): chunks of fewer batches -- the window
    // length is chosen per chunk (window_batches_for) -- down to what the pipeline still wins at; a replayable layout first
    const int K = *K_out;
    const int64_t u = Ew / K;                                              // one batch (packed rows: whole 16 bytes of output)
    const int64_t n_min = ((flags & TPNET_FLAG_SCHED_WINDOWED) ? 4 : min_nb) * batch;
    if (region_out && batch <= PLAN_ONE_MAX && !(flags & TPNET_FLAG_PLAN_SORTED))
        for (int64_t n = K - 1; n >= 1 && n * u >= n_min; --n) {
            const int64_t c = n * u;
            if (c < E && (E + c - 1) / c <= ARENA_MAX_CHUNKS && arena_bytes(st.N, st.d, st.L, E, c, batch) <= ws_bytes) {
                *region_out = arena_region_bytes(st.N, st.d, st.L, c, batch);
                return c;
            }
        }
    for (int64_t n = K - 1; n >= 1 && n * u >= n_min; --n)
        if (n * u < E && wplan_bytes(n * u, batch, st.N, st.d, st.L) <= ws_bytes) return n * u;
    return 0;
}

// Batches per window of a chunk of nb batches.  A pipeline of nw windows is nw + L dependent launches, each a fixed floor
// (~10 us: kernel boundary + the dependent loads of its longest unit) plus its share of the bandwidth work (~2.5 us per batch
// at C2): few long windows for short streams.  Measured (tools/short_trace.sh, sum of the pipeline launches): 20 batches as
// 1 / 2 / 4 windows 127 / 85 / 110 us; 158 batches at 12 / 16 / 24 per window 493 / 472 / 450 us; long streams: the cap of 24
// (tools/degree_sensitivity.py).
static int window_batches_for(int64_t nb, int Kmax) {
    static const int fixed = TPNET_DEV_INT(WINDOW_FIXED, 0);
    if (fixed > 0) return fixed < Kmax ? fixed : Kmax;
    static const int num = TPNET_DEV_INT(WINDOW_SQ, 5);
    int K = 2;
    while (K < Kmax && (int64_t)K * K < (int64_t)num * nb) ++K;         // K = ceil(sqrt(5 nb))
    const int64_t nw0 = (nb + K - 1) / K;                              // equal windows: ceil(nb / nw0) batches each
    return (int)((nb + nw0 - 1) / nw0);
}

// what a plan left in the workspace was built for (tpnet_plan_tag::built, opaque to the caller)
struct PlanBuilt {
    uint64_t valid;
    const void *src, *dst, *t, *ws;
    int64_t E, batch, N;
    uint64_t ws_bytes;
    double now_time, lambda;
    int32_t d, L, K;
    uint32_t flags;
    uint64_t table_sig, stream_sig;
    uint64_t have_readout;
    int64_t chunk;              // edges per chunk, and
    uint64_t region;            // bytes per chunk region where the stream is several chunks (0: one chunk)
};
static_assert(sizeof(PlanBuilt) <= sizeof(((tpnet_plan_tag*)nullptr)->built), "tpnet_plan_tag::built too small");

static int run_stream_windowed(const tpnet_state& st, const int64_t* src, const int64_t* dst, const int64_t* neg,
                               const double* t, int64_t E, int64_t batch, double now_time, double lambda,
                               uint32_t launch_id_base, uint32_t flags, float* out_pos, float* out_neg, void* ws,
                               size_t ws_bytes, int64_t chunk, int Kmax, hipStream_t s, StepTimer* timer,
                               tpnet_plan_tag* tag, size_t region) {
    const int NN = 2 * st.L + 2;
    const int NG = (flags & TPNET_FLAG_PACKED) ? NN * (NN + 1) / 2 : NN * NN;
    uint32_t lid = launch_id_base;
    const bool have_readout = out_pos || out_neg;
    // a plan may be replayed when the caller vouches (tag) that the stream arrays and the table's per-node state are what the
    // plan in this workspace was built for, and the workspace still holds the plan of EVERY chunk: the stream is one chunk, or
    // its chunks' plans lie side by side (`region` bytes each, window_chunk) in front of the one version log they share -- the
    // table state a later chunk's plan was built on follows from the first one's and the stream
    char* const arena = reinterpret_cast<char*>((reinterpret_cast<size_t>(ws) + 255) / 256 * 256);
    const int64_t n_chunks = (E + chunk - 1) / chunk;
    float* const shared_log = region ? reinterpret_cast<float*>(arena + (size_t)n_chunks * region) : nullptr;
    PlanBuilt now{};
    bool replay = false, all_planned_here = true;
    if (tag && (n_chunks == 1 || region) && tag->table_sig && tag->stream_sig) {
        now.valid = 1; now.src = src; now.dst = dst; now.t = t; now.ws = ws; now.E = E; now.batch = batch; now.N = st.N;
        now.ws_bytes = ws_bytes; now.now_time = now_time; now.lambda = lambda; now.d = st.d; now.L = st.L;
        now.K = window_batches_for(((E < chunk ? E : chunk) + batch - 1) / batch, Kmax);
        now.flags = flags & ~(uint32_t)TPNET_FLAG_SCHED_WINDOWED;
        now.table_sig = tag->table_sig; now.stream_sig = tag->stream_sig; now.have_readout = have_readout ? 1 : 0;
        now.chunk = chunk; now.region = region;
        replay = memcmp(&now, tag->built, sizeof(PlanBuilt)) == 0;
    }
    if (tag) {
        memset(tag->built, 0, sizeof(tag->built));                  // (invalid unless this call completes every chunk's plan)
        tag->replayed = replay ? 1 : 0;
    }
    for (int64_t c0 = 0, ci = 0; c0 < E; c0 += chunk, ++lid, ++ci) {
        const int64_t Ec = (E - c0 < chunk) ? (E - c0) : chunk;
        const int64_t nb = (Ec + batch - 1) / batch;
        const int K = window_batches_for(nb, Kmax);
        WPlan p{};
        int rc = region ? wplan_carve(arena + (size_t)ci * region, region, Ec, batch, st.N, st.d, st.L, K, &p, shared_log)
                        : wplan_carve(ws, ws_bytes, Ec, batch, st.N, st.d, st.L, K, &p);
        if (rc) return rc;
        static const int no3 = TPNET_DEV_INT(NO_PLAN3, 0);
        // the hashed planner costs ~45 us + 0.76 us per batch, the sorted one ~200 us + 0.63 us per batch (C2, profiles/r03_C2.md):
        // 4 % of a long stream's time, paid for having one set of plan kernels (see window_chunk)
        static const int max3 = TPNET_DEV_INT(PLAN3_MAX_BATCHES, 1 << 30);
        const bool dense = !(flags & (TPNET_FLAG_PLAN_SORTED | TPNET_FLAG_PLAN_HASHED)) && wplan_dense_applies(st, p, Ec, batch, K);
        const bool plan3 = !dense && !no3 && !(flags & TPNET_FLAG_PLAN_SORTED) && nb <= max3 && wplan3_applies(st, Ec, batch, K);
        // (the sorted planner keeps nothing to replay: a stream with such a chunk is planned again every time; a recorded plan had
        // none, and the same sizes choose the same planners)
        if (!(plan3 || dense)) all_planned_here = false;
        if (dense)
            rc = wplan_dense_build(st, p, src + c0, dst + c0, neg ? neg + c0 : nullptr, t + c0, Ec, batch, now_time,
                                   c0 > 0 ? t + c0 - 1 : nullptr, lambda, have_readout, replay, s);
        else if (plan3)
            rc = wplan3_build(st, p, src + c0, dst + c0, neg ? neg + c0 : nullptr, t + c0, Ec, batch, now_time,
                              c0 > 0 ? t + c0 - 1 : nullptr, lambda, have_readout, replay, s);
        else
            rc = wplan_build(st, p, src + c0, dst + c0, neg ? neg + c0 : nullptr, t + c0, Ec, batch, now_time,
                             c0 > 0 ? t + c0 - 1 : nullptr, lambda, have_readout, s);
        if (rc) return rc;
        StreamArgs a;
        a.src = src + c0;
        a.dst = dst + c0;
        a.neg = neg ? neg + c0 : nullptr;
        a.t = t + c0;
        a.out_pos = out_pos ? out_pos + c0 * NG : nullptr;
        a.out_neg = out_neg ? out_neg + c0 * NG : nullptr;
        a.own_mod = 1;
        a.own_rem = 0;
        const int64_t nw = (Ec + p.Ew - 1) / p.Ew;
        const int64_t nsteps = nw + (have_readout ? st.L : st.L - 1);
        const bool timed = timer && timer->n < timer->cap;
        if (timed) (void)hipEventRecord(timer->ev[2 * timer->n], s);
        for (int64_t j = 0; j < nsteps; ++j) {
            rc = launch_wstep(st, a, p, j, Ec, batch, lambda, flags, s);
            if (rc) return rc;
        }
        if (timed) {
            (void)hipEventRecord(timer->ev[2 * timer->n + 1], s);
            ++timer->n;
            timer->launches += nsteps;
            timer->batches += nb;
            timer->edges += Ec;
        }
        // the write-back: node by node where the hashed planner planned the chunk (wplan3.hip: 9 us for the 158-batch epoch), else
        // the scan of the sorted positions for the last-run flag (wstep.hip: 25 us).  (Measured and not kept, round 3: the
        // write-back as extra blocks of the chunk's last pipeline step -- readouts only, nothing it reads is written -- grew that
        // step by what the write-back takes alone, leading the grid or not, and every other step by ~2 us: 634 against 635 us for
        // the epoch; a write-back driven by chain records that carry (last chain of the node, table copy, last clock): 30 us.)
        if (!(dense ? wplan_dense_writeback(st, p, Ec, batch, lid, s) : (plan3 && wplan3_writeback(st, p, Ec, batch, lid, s)))) {
            rc = launch_wwriteback(st, p, Ec, lid, s);
            if (rc) return rc;
        }
    }
    if (now.valid && all_planned_here) memcpy(tag->built, &now, sizeof(PlanBuilt));
    return TPNET_OK;
}

static int run_stream_impl(const tpnet_state& st, const int64_t* src, const int64_t* dst, const int64_t* neg,
                           const double* t, int64_t E, int64_t batch, double now_time, double lambda,
                           uint32_t launch_id_base, uint32_t flags, float* out_pos, float* out_neg, void* ws,
                           size_t ws_bytes, hipStream_t s, StepTimer* timer, tpnet_plan_tag* tag = nullptr) {
    if (E == 0) return TPNET_OK;
    {
        int Kw = 0;
        size_t region = 0;
        const int64_t wchunk = window_chunk(st, ws_bytes, E, batch, flags, &Kw, tag ? &region : nullptr);
        if (wchunk > 0)
            return run_stream_windowed(st, src, dst, neg, t, E, batch, now_time, lambda, launch_id_base, flags, out_pos,
                                       out_neg, ws, ws_bytes, wchunk, Kw, s, timer, tag, region);
    }
    const int64_t chunk = max_chunk(ws_bytes, E, batch);
    // the per-batch plan of a stream (item lists, coefficients, batch descriptors: plan.hip) is a function of src / dst / t, the
    // clock at entry and the flags alone -- not of the table -- and the step kernels only read it: a stream that is ONE chunk (up
    // to ~2 M edges: every dataset of the reference at any batch size) replays it when the caller vouches for the arrays
    // (tag->stream_sig; table_sig is not looked at).  valid = 2 tells such a plan from a windowed one (valid = 1).
    PlanBuilt now{};
    bool replay = false;
    if (tag && chunk >= E && tag->stream_sig) {
        now.valid = 2; now.src = src; now.dst = dst; now.t = t; now.ws = ws; now.E = E; now.batch = batch; now.N = st.N;
        now.ws_bytes = ws_bytes; now.now_time = now_time; now.lambda = lambda; now.d = st.d; now.L = st.L; now.K = 0;
        now.flags = flags; now.table_sig = 0; now.stream_sig = tag->stream_sig;
        now.have_readout = out_pos ? 1 : 0;                     // (the edge-fused update's lists exist for a (src, dst) readout only)
        now.chunk = chunk; now.region = 0;
        replay = memcmp(&now, tag->built, sizeof(PlanBuilt)) == 0;
    }
    if (tag) { memset(tag->built, 0, sizeof(tag->built)); tag->replayed = replay ? 1 : 0; }   // (invalid unless the call completes)
    if (chunk < 1) return TPNET_ERR_WORKSPACE;
    const int NN = 2 * st.L + 2;
    const int NG = (flags & TPNET_FLAG_PACKED) ? NN * (NN + 1) / 2 : NN * NN;   // floats per feature row of a chunk's outputs
    uint32_t lid = launch_id_base;
    for (int64_t c0 = 0; c0 < E; c0 += chunk) {
        const int64_t Ec = (E - c0 < chunk) ? (E - c0) : chunk;
        Plan p{};
        int rc = plan_carve(ws, ws_bytes, Ec, batch, &p);
        if (rc) return rc;
        // edge-fused updates: every launch of this chunk must run BOTH roles with the (src,dst) readout on, and the sums
        // must not be order-sensitive by contract; worth it where the batch is bound by bytes, not by its longest chain
        static const int fuse_env = TPNET_DEV_INT(FUSE, -1);       // developer override: 0 / 1
        static const int role_mask0 = TPNET_DEV_INT(ROLE_MASK, 3);
        const bool fuse = out_pos && p.fuse_src && role_mask0 == 3 &&
                          !(flags & (TPNET_FLAG_EAGER_DECAY | TPNET_FLAG_SEQUENTIAL)) &&
                          (fuse_env >= 0 ? fuse_env == 1 : batch > 1024);
        // the clock before a later chunk is t[c0-1], read on device (no host copy of the timestamps is needed)
        if (!replay) {
            rc = plan_build(st, p, src + c0, dst + c0, t + c0, Ec, batch, now_time, c0 > 0 ? t + c0 - 1 : nullptr, lambda,
                            flags | (fuse ? PLAN_FUSE : 0u), s);
            if (rc) return rc;
        }
        StreamArgs a;
        a.src = src + c0;
        a.dst = dst + c0;
        a.neg = neg ? neg + c0 : nullptr;
        a.t = t + c0;
        a.out_pos = out_pos ? out_pos + c0 * NG : nullptr;
        a.out_neg = out_neg ? out_neg + c0 * NG : nullptr;
        a.own_mod = 1;
        a.own_rem = 0;
        const int64_t nb = (Ec + batch - 1) / batch;
        const bool have_readout = a.out_pos || a.out_neg;
        const bool timed = timer && timer->n < timer->cap && !(flags & TPNET_FLAG_EAGER_DECAY);
        if (timed) (void)hipEventRecord(timer->ev[2 * timer->n], s);
        for (int64_t b = 0; b < nb; ++b, ++lid) {
            const int32_t ne = (int32_t)((Ec - b * batch < batch) ? (Ec - b * batch) : batch);
            if (flags & TPNET_FLAG_EAGER_DECAY) {
                // reference order: readout on the pre-batch state, THEN decay + scatter-add (TPNet.py:83-96)
                if (have_readout) {
                    rc = launch_step(st, a, p, b, batch, ne, lambda, lid, flags | ROLE_READOUT, s);
                    if (rc) return rc;
                }
                rc = launch_decay_desc(st, p, b, s);
                if (rc) return rc;
                rc = launch_step(st, a, p, b, batch, ne, lambda, lid, flags | ROLE_UPDATE, s);
                if (rc) return rc;
            } else {
                static const int role_mask = TPNET_DEV_INT(ROLE_MASK, 3);
                const uint32_t roles = ((role_mask & 2) ? ROLE_UPDATE : 0u) |
                                       ((have_readout && (role_mask & 1)) ? ROLE_READOUT : 0u);
                rc = launch_step(st, a, p, b, batch, ne, lambda, lid, flags | roles | (fuse ? STEP_FUSE : 0u), s);
                if (rc) return rc;
            }
        }
        if (timed) {
            (void)hipEventRecord(timer->ev[2 * timer->n + 1], s);
            ++timer->n;
            timer->launches += nb;
            timer->batches += nb;
            timer->edges += Ec;
        }
    }
    if (now.valid) memcpy(tag->built, &now, sizeof(PlanBuilt));
    return TPNET_OK;
}

}  // namespace tpnet

using namespace tpnet;

extern "C" {

const char* tpnet_strerror(int status) {
    switch (status) {
        case TPNET_OK: return "ok";
        case TPNET_ERR_BAD_ARG: return "bad argument";
        case TPNET_ERR_WORKSPACE: return "workspace missing or too small";
        case TPNET_ERR_HIP: return "HIP runtime error";
        case TPNET_ERR_INDEX: return "node id out of range";
        case TPNET_ERR_NO_DEVICE: return "no HIP device";
        case TPNET_ERR_NEED_GRAM: return "the one-launch encoder kernel is not available: call again with a gram buffer";
        default: return "unknown status";
    }
}

int tpnet_abi_version(void) { return TPNET_ABI_VERSION; }
int tpnet_last_hip_error(void) { return g_last_hip_error; }

int tpnet_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

size_t tpnet_q_bytes(int64_t N, int32_t d, int32_t L) { return (size_t)2 * (size_t)N * (size_t)L * (size_t)d * sizeof(float); }
size_t tpnet_meta_bytes(int64_t N) { return (size_t)N * sizeof(tpnet_node_meta); }

int tpnet_state_init(const tpnet_state* st, double t0, void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    return launch_state_init(*st, t0, (hipStream_t)stream);
}

int tpnet_import_layers(const tpnet_state* st, const float* const* layers, double now_time, void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (!layers) return TPNET_ERR_BAD_ARG;
    for (int i = 0; i < st->L; ++i)
        if (!layers[i]) return TPNET_ERR_BAD_ARG;
    return launch_import(*st, layers, now_time, (hipStream_t)stream);
}

int tpnet_export_layers(const tpnet_state* st, float* const* layers, double now_time, double lambda, void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (!layers) return TPNET_ERR_BAD_ARG;
    for (int i = 0; i < st->L; ++i)
        if (!layers[i]) return TPNET_ERR_BAD_ARG;
    return launch_export(*st, layers, now_time, lambda, (hipStream_t)stream);
}

int tpnet_decay(const tpnet_state* st, const float* factors, double t_new, void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (!factors) return TPNET_ERR_BAD_ARG;
    return launch_decay(*st, factors, t_new, (hipStream_t)stream);
}

int tpnet_gather_rows(const tpnet_state* st, const int64_t* ids, int64_t n, double now_time, double lambda, float* out,
                      void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!ids || !out))) return TPNET_ERR_BAD_ARG;
    return launch_gather_rows(*st, ids, n, now_time, lambda, out, (hipStream_t)stream);
}

int tpnet_gather_elems(const tpnet_state* st, const int64_t* rows, const int64_t* cols, int64_t n, double now_time,
                       double lambda, float* out, void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!rows || !cols || !out))) return TPNET_ERR_BAD_ARG;
    return launch_gather_elems(*st, rows, cols, n, now_time, lambda, out, (hipStream_t)stream);
}

int tpnet_pair_gram(const tpnet_state* st, const int64_t* u, const int64_t* v, int64_t n, double now_time,
                    double lambda, uint32_t flags, float* out, void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!u || !v || !out))) return TPNET_ERR_BAD_ARG;
    return launch_pair_gram(*st, u, v, n, now_time, lambda, flags, out, (hipStream_t)stream);
}

int tpnet_pair_gram_shared(const tpnet_state* st, const int64_t* u, const int64_t* v1, const int64_t* v2, int64_t n,
                           double now_time, double lambda, uint32_t flags, float* out1, float* out2, void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!u || !v1 || !v2 || !out1 || !out2))) return TPNET_ERR_BAD_ARG;
    if (flags & TPNET_FLAG_PACKED) return TPNET_ERR_BAD_ARG;   // packed rows exist for the one-pair readout only
    return launch_pair_gram_shared(*st, u, v1, v2, n, now_time, lambda, flags, out1, out2, (hipStream_t)stream);
}

int tpnet_pair_gram_anchored(const tpnet_state* st, const int64_t* neigh, const int64_t* a1, const int64_t* a2,
                             int64_t n_rows, int32_t K, double now_time, double lambda, uint32_t flags, float* out1,
                             float* out2, void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (n_rows < 0 || K < 0 || (n_rows > 0 && K > 0 && (!neigh || !a1 || !a2 || !out1 || !out2))) return TPNET_ERR_BAD_ARG;
    if (flags & TPNET_FLAG_PACKED) return TPNET_ERR_BAD_ARG;
    if (n_rows > 0 && K > 0 && !pair_gram_anchored_supported(*st)) {
        // rows that are not one chunk of 16-byte vectors (d = 120, 140, use_matrix ...): the generic kernel, one launch per
        // anchor side, on index arrays the caller would otherwise build -- not available without them: report it
        return TPNET_ERR_BAD_ARG;
    }
    // rows of 64 / 128 floats: the same blocks on the matrix cores (encoder_mfma.hip, fp32 class)
    if ((flags & TPNET_FLAG_NO_MFMA_READOUT) == 0 && encoder_mfma_supported(*st, n_rows, K) && !((reinterpret_cast<uintptr_t>(out1) | reinterpret_cast<uintptr_t>(out2)) & 15))
        return launch_encoder_gram_mfma(*st, neigh, a1, a2, n_rows, K, now_time, lambda, flags, out1, out2, (hipStream_t)stream);
    return launch_pair_gram_anchored(*st, neigh, a1, a2, n_rows, K, now_time, lambda, flags, out1, out2, (hipStream_t)stream);
}

int tpnet_pair_gram_anchored_supported(const tpnet_state* st) {
    if (check_state(st)) return 0;
    return pair_gram_anchored_supported(*st) ? 1 : 0;
}

size_t tpnet_workspace_bytes(int64_t max_edges, int64_t batch) { return plan_bytes(max_edges, batch); }

// bytes for a stream of max_edges on the windowed schedule in chunks of at most chunk_cap edges: one chunk's plan, or -- several
// chunks -- their plans side by side in front of one version log (the layout a replay needs, window_chunk), up to ARENA_MAX_CHUNKS
static size_t windowed_workspace_bytes(int64_t N, int d, int L, int64_t max_edges, int64_t batch, int64_t chunk_cap) {
    int K = 0;
    int64_t Ew = 0;
    int64_t lim = window_chunk_limit(N, d, L, max_edges, batch, 0u, &K, &Ew);
    if (lim == 0) return 0;
    if (chunk_cap > 0 && lim > chunk_cap) lim = chunk_cap;
    const size_t one = wplan_bytes(lim, batch, N, d, L);
    int64_t c = lim / Ew * Ew;                                          // chunks of a stream that takes several: whole windows,
    if (c == 0) c = lim / (Ew / K) * (Ew / K);                          // or, under a cap below one window of K batches, whole batches
    if (lim >= max_edges || c < 4 * batch || (max_edges + c - 1) / c > ARENA_MAX_CHUNKS) return one;
    const size_t all = arena_bytes(N, d, L, max_edges, c, batch);
    // bounded (include/tpnet_hip.h: TPNET_ARENA_MAX_RATIO): the chunks' plans side by side are ~0.9 KB per edge of the WHOLE stream --
    // without a bound a 100 M-edge C2 stream asked for 108 GB, a 64-chunk one for more than the GPU has
    if (all > (size_t)TPNET_ARENA_MAX_RATIO * one) return one;
    return all > one ? all : one;
}

size_t tpnet_stream_workspace_bytes(int64_t N, int32_t d, int32_t L, int64_t max_edges, int64_t batch) {
    if (max_edges < 1) max_edges = 1;
    if (batch < 1) batch = 1;
    // a plan never covers more than a chunk: ~2 M edges on the per-batch schedule, what the version log allows on the
    // windowed one; longer streams are walked chunk by chunk
    const int64_t cap_a = (2000000 / batch > 0 ? 2000000 / batch : 1) * batch;
    const size_t a = plan_bytes(max_edges < cap_a ? max_edges : cap_a, batch);
    const size_t b = windowed_workspace_bytes(N, d, L, max_edges, batch, 0);
    return a > b ? a : b;
}

size_t tpnet_stream_workspace_bytes_capped(int64_t N, int32_t d, int32_t L, int64_t max_edges, int64_t batch,
                                           size_t log_cap_bytes) {
    if (log_cap_bytes == 0 || L < 1 || d < 1) return tpnet_stream_workspace_bytes(N, d, L, max_edges, batch);
    if (max_edges < 1) max_edges = 1;
    if (batch < 1) batch = 1;
    int64_t e = (int64_t)(log_cap_bytes / (2 * (size_t)L * (size_t)d * 4));
    e = e / batch * batch;
    if (e < 4 * batch) e = 4 * batch;                      // (the windowed schedule needs at least four batches per chunk)
    const int64_t cap_a = (2000000 / batch > 0 ? 2000000 / batch : 1) * batch;
    const int64_t ea = max_edges < e ? max_edges : e;
    const size_t a = plan_bytes(ea < cap_a ? ea : cap_a, batch);
    const size_t b = windowed_workspace_bytes(N, d, L, max_edges, batch, e);
    return a > b ? a : b;
}

int tpnet_stream_schedule(int64_t N, int32_t d, int32_t L, int64_t E, int64_t batch, uint32_t flags, size_t ws_bytes) {
    if (N < 1 || d < 1 || L < 1 || L > TPNET_MAX_LAYERS || E < 0 || batch < 1) return TPNET_ERR_BAD_ARG;
    if (E == 0) return 0;
    tpnet_state st{};
    st.N = N; st.d = d; st.L = L;
    int K = 0;
    return window_chunk(st, ws_bytes, E, batch, flags, &K) > 0 ? 1 : 0;
}

int tpnet_update(const tpnet_state* st, const int64_t* src, const int64_t* dst, const double* t, int64_t B,
                 double now_time, double lambda, uint32_t launch_id, uint32_t flags, void* workspace, size_t ws_bytes,
                 void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (B < 1 || !src || !dst || !t) return TPNET_ERR_BAD_ARG;  // the reference raises on an empty batch (t[-1])
    if (launch_id == 0 || launch_id >= 0x7FFFFFFFu) return TPNET_ERR_BAD_ARG;
    if (plan_bytes(B, B) > ws_bytes) return TPNET_ERR_WORKSPACE;
    if (B <= plan_one_max_batch()) {
        // one batch: the plan is ONE single-workgroup kernel (plan.hip, k_plan_one) instead of keys + device sort + finish
        Plan p{};
        rc = plan_carve(workspace, ws_bytes, B, B, &p);
        if (rc) return rc;
        hipStream_t s = (hipStream_t)stream;
        rc = plan_one(*st, p, src, dst, t, B, now_time, lambda, flags, s);
        if (rc) return rc;
        if (flags & TPNET_FLAG_EAGER_DECAY) {
            rc = launch_decay_desc(*st, p, 0, s);
            if (rc) return rc;
        }
        StreamArgs a{};
        a.own_mod = 1;
        return launch_step(*st, a, p, 0, B, (int32_t)B, lambda, launch_id, flags | ROLE_UPDATE, s);
    }
    return run_stream_impl(*st, src, dst, nullptr, t, B, B, now_time, lambda, launch_id, flags, nullptr, nullptr,
                           workspace, ws_bytes, (hipStream_t)stream, nullptr);
}

int tpnet_run_stream(const tpnet_state* st, const int64_t* src, const int64_t* dst, const int64_t* neg,
                     const double* t, int64_t E, int64_t batch, double now_time, double lambda,
                     uint32_t launch_id_base, uint32_t flags, float* out_pos, float* out_neg, void* workspace,
                     size_t ws_bytes, double* t_end_out, void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (E < 0 || batch < 1 || (E > 0 && (!src || !dst || !t))) return TPNET_ERR_BAD_ARG;
    if (out_neg && !neg) return TPNET_ERR_BAD_ARG;
    const int64_t nb = (E + batch - 1) / batch;
    if (launch_id_base == 0 || (uint64_t)launch_id_base + (uint64_t)nb >= 0x7FFFFFFFull) return TPNET_ERR_BAD_ARG;
    rc = run_stream_impl(*st, src, dst, neg, t, E, batch, now_time, lambda, launch_id_base, flags, out_pos, out_neg,
                         workspace, ws_bytes, (hipStream_t)stream, nullptr);
    if (rc) return rc;
    if (t_end_out && E > 0) {
        TPNET_HIP_TRY(hipMemcpyAsync(t_end_out, t + E - 1, sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream));
        TPNET_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    }
    return TPNET_OK;
}

int tpnet_run_stream_tagged(const tpnet_state* st, const int64_t* src, const int64_t* dst, const int64_t* neg,
                            const double* t, int64_t E, int64_t batch, double now_time, double lambda,
                            uint32_t launch_id_base, uint32_t flags, float* out_pos, float* out_neg, void* workspace,
                            size_t ws_bytes, double* t_end_out, void* stream, tpnet_plan_tag* tag) {
    int rc = check_state(st);
    if (rc) return rc;
    if (E < 0 || batch < 1 || (E > 0 && (!src || !dst || !t))) return TPNET_ERR_BAD_ARG;
    if (out_neg && !neg) return TPNET_ERR_BAD_ARG;
    const int64_t nb = (E + batch - 1) / batch;
    if (launch_id_base == 0 || (uint64_t)launch_id_base + (uint64_t)nb >= 0x7FFFFFFFull) return TPNET_ERR_BAD_ARG;
    rc = run_stream_impl(*st, src, dst, neg, t, E, batch, now_time, lambda, launch_id_base, flags, out_pos, out_neg,
                         workspace, ws_bytes, (hipStream_t)stream, nullptr, tag);
    if (rc) return rc;
    if (t_end_out && E > 0) {
        TPNET_HIP_TRY(hipMemcpyAsync(t_end_out, t + E - 1, sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream));
        TPNET_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    }
    return TPNET_OK;
}

// a kernel that keeps its queue busy for `ticks` of the 100 MHz wall clock, and one that does nothing
__global__ void k_warm_spin(uint64_t ticks) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
struct WarmArgs { uint64_t w[64]; };                        // 512 bytes of kernel arguments, like the step kernels'
__global__ void k_warm_noop(WarmArgs a, uint64_t* sink) { if (sink && a.w[0] == 0x1234567ull) *sink = a.w[1]; }

int tpnet_runtime_warmup(int32_t launches, int32_t spin_us, void* stream) {
    if (launches < 0 || launches > 4096 || spin_us < 0 || spin_us > 5000) return TPNET_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (spin_us > 0) hipLaunchKernelGGL(k_warm_spin, dim3(1), dim3(64), 0, s, (uint64_t)spin_us * 100ull);
    WarmArgs wa{};
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_warm_noop, dim3(1), dim3(64), 0, s, wa, (uint64_t*)nullptr);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int tpnet_plan_stream(const tpnet_state* st, const int64_t* src, const int64_t* dst, const double* t, int64_t E,
                      int64_t batch, double now_time, double lambda, uint32_t flags, void* workspace, size_t ws_bytes,
                      void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (E < 1 || batch < 1 || !src || !dst || !t) return TPNET_ERR_BAD_ARG;
    if (E >= ((int64_t)1 << 30)) return TPNET_ERR_BAD_ARG;            // 32-bit contribution indices inside one plan
    if (plan_bytes(E, batch) > ws_bytes) return TPNET_ERR_WORKSPACE;
    Plan p{};
    rc = plan_carve(workspace, ws_bytes, E, batch, &p);
    if (rc) return rc;
    return plan_build(*st, p, src, dst, t, E, batch, now_time, nullptr, lambda, flags, (hipStream_t)stream);
}

int tpnet_step_batch(const tpnet_state* st, const int64_t* src, const int64_t* dst, const int64_t* neg,
                     const double* t, int64_t E, int64_t batch, int64_t b, double lambda, uint32_t launch_id,
                     uint32_t flags, int32_t own_mod, int32_t own_rem, float* out_pos, float* out_neg,
                     void* workspace, size_t ws_bytes, void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    const int64_t nb = (E + batch - 1) / batch;
    if (E < 1 || batch < 1 || b < 0 || b >= nb || !src || !dst || !t) return TPNET_ERR_BAD_ARG;
    if (own_mod < 0 || own_rem < 0 || (own_mod > 0 && own_rem >= own_mod)) return TPNET_ERR_BAD_ARG;
    if (out_neg && !neg) return TPNET_ERR_BAD_ARG;
    if (launch_id == 0 || launch_id >= 0x7FFFFFFFu) return TPNET_ERR_BAD_ARG;
    if (flags & TPNET_FLAG_EAGER_DECAY) return TPNET_ERR_BAD_ARG;   // the sharded path carries the decay lazily
    Plan p{};
    rc = plan_carve(workspace, ws_bytes, E, batch, &p);
    if (rc) return rc;
    StreamArgs a;
    a.src = src; a.dst = dst; a.neg = neg; a.t = t;
    a.out_pos = out_pos; a.out_neg = out_neg;
    a.own_mod = own_mod; a.own_rem = own_rem;
    const int32_t ne = (int32_t)((E - b * batch < batch) ? (E - b * batch) : batch);
    const bool have_readout = out_pos || out_neg;
    return launch_step(*st, a, p, b, batch, ne, lambda, launch_id,
                       flags | ROLE_UPDATE | (have_readout ? ROLE_READOUT : 0u), (hipStream_t)stream);
}

int tpnet_pack_rows(const tpnet_state* st, const int64_t* ids, int64_t n, double now_time, double lambda, float* out,
                    void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!ids || !out))) return TPNET_ERR_BAD_ARG;
    return launch_pack_rows(*st, ids, n, now_time, lambda, out, (hipStream_t)stream);
}

int tpnet_unpack_rows(const tpnet_state* st, const int64_t* ids, int64_t n, double now_time, const float* in,
                      void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!ids || !in))) return TPNET_ERR_BAD_ARG;
    return launch_unpack_rows(*st, ids, n, now_time, in, (hipStream_t)stream);
}

int tpnet_pack_bundles(const tpnet_state* st, const int64_t* ids, int64_t n, double now_time, double lambda, float* out,
                       void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!ids || !out))) return TPNET_ERR_BAD_ARG;
    return launch_pack_bundles(*st, ids, n, now_time, lambda, out, (hipStream_t)stream);
}

int tpnet_unpack_bundles(const tpnet_state* st, const int64_t* local_ids, int64_t n, double now_time, const float* recv,
                         int64_t maxc, const int64_t* offs, int32_t G, void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (n < 0 || G < 1 || maxc < 0 || (n > 0 && (!local_ids || !recv || !offs))) return TPNET_ERR_BAD_ARG;
    return launch_unpack_bundles(*st, local_ids, n, now_time, recv, maxc, offs, G, (hipStream_t)stream);
}

int tpnet_unpack_gathered(const tpnet_state* st, const int64_t* ids, int64_t n, double now_time, const float* recv,
                          int64_t maxc, const int64_t* offs, int32_t G, int32_t me, void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (n < 0 || G < 1 || me < 0 || me >= G || maxc < 0 || (n > 0 && (!ids || !recv || !offs))) return TPNET_ERR_BAD_ARG;
    return launch_unpack_gathered(*st, ids, n, now_time, recv, maxc, offs, G, me, (hipStream_t)stream);
}

int tpnet_gram_finish(float* x, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && !x)) return TPNET_ERR_BAD_ARG;
    return tpnet::launch_gram_finish(x, n, (hipStream_t)stream);
}

int tpnet_gram_unpack(const float* packed, int64_t n, int32_t L, uint32_t flags, float* out, void* stream) {
    if (n < 0 || L < 1 || L > TPNET_MAX_LAYERS || (n > 0 && (!packed || !out))) return TPNET_ERR_BAD_ARG;
    return tpnet::launch_gram_unpack(packed, n, L, flags, out, (hipStream_t)stream);
}

int tpnet_check_errors(const tpnet_state* st, void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    uint32_t h[4] = {0, 0, 0, 0};
    TPNET_HIP_TRY(hipMemcpyAsync(h, st->err, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    TPNET_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (h[0] != 0) {
        TPNET_HIP_TRY(hipMemsetAsync(st->err, 0, sizeof(h), (hipStream_t)stream));
        return TPNET_ERR_INDEX;
    }
    return TPNET_OK;
}

int tpnet_time_stream(const tpnet_state* st, const int64_t* src, const int64_t* dst, const int64_t* neg,
                      const double* t, int64_t E, int64_t batch, double now_time, double lambda,
                      uint32_t launch_id_base, uint32_t flags, float* out_pos, float* out_neg, void* workspace,
                      size_t ws_bytes, int reps, float* total_ms_out, float* kernel_ms_out, int64_t* launches_out,
                      int64_t* edges_out, void* stream) {
    int rc = check_state(st);
    if (rc) return rc;
    if (reps < 1 || E < 1 || batch < 1) return TPNET_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nb = (E + batch - 1) / batch;
    if ((uint64_t)launch_id_base + (uint64_t)nb * (uint64_t)(reps + 1) >= 0x7FFFFFFFull) return TPNET_ERR_BAD_ARG;
    hipEvent_t e0, e1;
    TPNET_HIP_TRY(hipEventCreate(&e0));
    TPNET_HIP_TRY(hipEventCreate(&e1));
    // event pairs around the step-launch loop of every chunk of the LAST rep (at most 64 chunks)
    StepTimer tm;
    tm.cap = 256;
    std::vector<hipEvent_t> evs((size_t)(2 * tm.cap));
    for (auto& e : evs) TPNET_HIP_TRY(hipEventCreate(&e));
    tm.ev = evs.data();
    uint32_t lid = launch_id_base;
    TPNET_HIP_TRY(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) {
        rc = run_stream_impl(*st, src, dst, neg, t, E, batch, now_time, lambda, lid, flags, out_pos, out_neg, workspace,
                             ws_bytes, s, (kernel_ms_out && r == reps - 1) ? &tm : nullptr);
        if (rc) return rc;
        lid += (uint32_t)nb;
    }
    TPNET_HIP_TRY(hipEventRecord(e1, s));
    TPNET_HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    TPNET_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (total_ms_out) *total_ms_out = ms;
    if (kernel_ms_out) {
        double sum = 0.0;
        for (int64_t i = 0; i < tm.n; ++i) {
            float k = 0.f;
            TPNET_HIP_TRY(hipEventElapsedTime(&k, evs[2 * i], evs[2 * i + 1]));
            sum += k;
        }
        *kernel_ms_out = tm.launches ? (float)(sum / (double)tm.launches) : 0.f;
    }
    if (launches_out) *launches_out = tm.launches;
    if (edges_out) *edges_out = tm.edges;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    for (auto& e : evs) (void)hipEventDestroy(e);
    return TPNET_OK;
}

}  // extern "C"
