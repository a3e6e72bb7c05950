/*
 * tpnet_hip.h -- C ABI of the MI355X (gfx950) implementation of TPNet's temporal-walk-matrix hot path.
 *
 * Drop-in boundary: the functions below are what a binding for the reference's
 * `RandomProjectionModule` (reference: models/TPNet.py:9-157) calls instead of the stock ATen ops.
 * Plain C: device pointers, sizes, a hipStream_t passed as void*; no torch types.  All device buffers
 * are owned by the caller (e.g. the PyTorch caching allocator); nothing is allocated or freed inside;
 * kernels are enqueued on the given stream and the functions do not synchronise unless stated.
 *
 * Every function returns 0 on success or a negative tpnet_status code; tpnet_strerror() names it.
 *
 * ---------------------------------------------------------------------------------------------------
 * State layout in HBM (one instance per RandomProjectionModule):
 *   p0   float [N][d]            layer 0 (the random projection matrix P[0]; never written by update)
 *   q    float [2][N][L][d]      layers 1..L as per-node "bundles" (L*d contiguous floats), two copies
 *                                (ping-pong): a batch reads the pre-batch copy and writes the other one,
 *                                which removes the read/write hazard between layers inside ONE launch
 *   meta tpnet_node_meta [N]     per node: which copy is current + the time its bundle was last decayed to
 * Mathematical state (what the reference keeps eagerly, models/TPNet.py:83-85):
 *   P[i][n] = q[c][n][i-1] * exp(-lambda * i * (now_time - meta[n].tref[c])),  c = cur(n), i = 1..L
 * i.e. the dense per-batch decay of the reference is carried lazily per row and applied on read.
 * ---------------------------------------------------------------------------------------------------
 */
#ifndef TPNET_HIP_H
#define TPNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TPNET_ABI_VERSION 7 /* 2: + tpnet_gather_elems, tpnet_gram_finish, tpnet_gram_unpack, tpnet_decoder_bf16, TPNET_FLAG_PACKED;
                               3: + tpnet_stream_workspace_bytes (windowed schedule of tpnet_run_stream);
                               4: + tpnet_pair_feature (readout + self.mlp in one launch), host-array entry points
                                    (tpnet_stage_*, tpnet_host_pair_feature, tpnet_host_update), tpnet_pair_gram_anchored;
                               5: + tpnet_run_stream_tagged / tpnet_plan_tag (a stream's plan replayed across epochs), the encoder's call as one
                                    crossing (tpnet_anchored_features, tpnet_encoder_features, tpnet_host_encoder_features), the targeted
                                    exchange's plan on the device (tpnet_xplan_targeted);
                               6: + tpnet_mlp::wimg / tpnet_mlp_prepare_image (the encoder's readout and self.mlp in ONE launch on the matrix
                                    cores), TPNET_FLAG_NO_MFMA_READOUT, tpnet_rows_stream_targeted, tpnet_mlp64_bwd_f32, tpnet_host_anchored_features;
                                    tpnet_run_stream_tagged replays streams of up to 64 chunks;
                               7: + tpnet_stream_schedule (which schedule tpnet_run_stream would take); tpnet_xplan_targeted serves G = 1;
                                    tpnet_xplan_targeted_large (batches beyond one workgroup's lists); tpnet_wshard_* (a row shard on the windowed pipeline);
                                    the side-by-side plans of a multi-chunk stream are bounded (TPNET_ARENA_MAX_RATIO) */
#define TPNET_MAX_LAYERS 4 /* num_layer L in 1..4 (reference default 3, utils/load_configs.py:70) */

typedef enum tpnet_status {
    TPNET_OK = 0,
    TPNET_ERR_BAD_ARG = -1,     /* null pointer, negative size, L out of range, d < 1 */
    TPNET_ERR_WORKSPACE = -2,   /* workspace too small (see tpnet_workspace_bytes) */
    TPNET_ERR_HIP = -3,         /* a HIP runtime call failed; tpnet_last_hip_error() has the hipError_t */
    TPNET_ERR_INDEX = -4,       /* a node id outside [0, N) was found (checked on device; such ids are skipped,
                                   never dereferenced) -- reported by tpnet_check_errors() */
    TPNET_ERR_NO_DEVICE = -5,
    TPNET_ERR_NEED_GRAM = -6    /* tpnet_anchored_features / tpnet_encoder_features / tpnet_host_*: called with gram == NULL (allowed
                                   where tpnet_encoder_fused_supported), but the one-launch kernel could not be launched on this
                                   runtime: call again with a gram buffer (readout and dense layers then run as two launches) */
} tpnet_status;

/* 32-byte per-node record.  tref is kept per copy so that a launch rewriting a node never disturbs what a
 * concurrent reader of the same launch needs (the pre-batch copy and its reference time). */
typedef struct tpnet_node_meta {
    uint32_t ver;   /* (launch_id << 1) | current_copy */
    uint32_t pad0;
    double tref[2]; /* time each copy's bundle is expressed at (f64, like now_time: models/TPNet.py:36-37,99) */
    uint64_t pad1;
} tpnet_node_meta;

typedef struct tpnet_state {
    float* p0;             /* [N][d] */
    float* q;              /* [2][N][L][d] */
    tpnet_node_meta* meta; /* [N] */
    int64_t N;             /* rows, including padding row 0 (reference node_num) */
    int32_t d;             /* projection dimension (any d >= 1; d % 4 == 0 takes the vector path) */
    int32_t L;             /* num_layer */
    uint32_t* err;         /* [4] device error words, zeroed by tpnet_state_init; err[0] = bad-id count */
} tpnet_state;

/* flags */
#define TPNET_FLAG_NOT_SCALE 1u     /* readout: return the raw Gram (reference not_scale=True, TPNet.py:124-125) */
#define TPNET_FLAG_EAGER_DECAY 2u   /* update: dense decay of every row first, exactly models/TPNet.py:83-85 */
#define TPNET_FLAG_SEQUENTIAL 4u    /* update: never split a target's contributions over several wave groups:
                                       the sum is then accumulated in the reference's index order (src side,
                                       then dst side, models/TPNet.py:93-96) */
#define TPNET_FLAG_PACKED 8u        /* readout (tpnet_pair_gram, tpnet_run_stream, tpnet_step_batch): write only the
                                       (2L+2)(2L+3)/2 distinct entries a <= b of the symmetric Gram, raw (no x<0 -> 0 /
                                       log), row-major upper triangle, row stride (2L+2)(2L+3)/2 floats -- the wire
                                       format of the column-sharded table, whose partial inner products are summed
                                       across GPUs before tpnet_gram_unpack finishes them */

#define TPNET_FLAG_SCHED_WINDOWED 16u /* tpnet_run_stream: take the windowed schedule whenever it applies (>= 4 batches), not
                                       only for streams long enough for it to pay (>= 28 batches of <= 2048 edges, >= 56 larger
                                       ones) */
#define TPNET_FLAG_SCHED_BATCH 32u    /* tpnet_run_stream: one launch per batch, always */
#define TPNET_FLAG_PLAN_SORTED 64u    /* tpnet_run_stream, windowed schedule: plan every chunk with the chunk planner (two device-wide
                                       radix sorts) even where the three-launch planner applies (batches of <= 2048 edges, <= 64
                                       windows per chunk).  Both give the same bits; this one is the slower start-up */
#define TPNET_FLAG_PLAN_HASHED 128u   /* tpnet_run_stream, windowed schedule: plan with the hashed planner (a fill + four kernels, <= 64
                                       windows per chunk) even where the one-launch dense planner applies (table small against the
                                       stream: N * 12 bytes <= batch * L * d * 4).  Same bits again */

#define TPNET_FLAG_NO_MFMA_READOUT 256u /* tpnet_pair_gram_anchored and the encoder calls built on it: keep the vector-ALU readout where
                                       the matrix-core one applies (rows of 64 / 128 floats, L = 3, K >= 4: split-bf16 operands,
                                       three pieces per value, fp32 accumulation -- fp32 class, other summation order) */

const char* tpnet_strerror(int status);
int tpnet_abi_version(void);
int tpnet_last_hip_error(void);
/* number of HIP devices visible (0 without a GPU; never initialises a context) */
int tpnet_device_count(void);

/* First-use costs of the HIP runtime, paid when the caller chooses (e.g. once per process and device, when the module's engine
 * is created) instead of inside the first long call: measured, the first 20-launch call of a fresh process takes 60-70 us
 * longer than every later one unless a burst of launches FOLLOWED BY A SYNCHRONISE came before it.  Enqueues one kernel that
 * keeps the stream busy for spin_us microseconds and `launches` kernels with 512 bytes of arguments behind it; does not
 * synchronise itself (the caller does, once).  Touches no caller memory. */
int tpnet_runtime_warmup(int32_t launches, int32_t spin_us, void* stream);

/* Bytes the caller must allocate for q and meta. */
size_t tpnet_q_bytes(int64_t N, int32_t d, int32_t L);
size_t tpnet_meta_bytes(int64_t N);

/* Zero q, set meta[n] = {ver 0, tref = t0}, zero err.  (reset_random_projections, models/TPNet.py:131-139,
 * minus the P[0] redraw which stays with the caller's RNG.) */
int tpnet_state_init(const tpnet_state* st, double t0, void* stream);

/* layers[i-1] (device, row-major [N][d], i = 1..L)  ->  q (copy 0), meta = {0, now_time}.
 * (reload_random_projections / load_state_dict / .to(): models/TPNet.py:149-157) */
int tpnet_import_layers(const tpnet_state* st, const float* const* layers, double now_time, void* stream);

/* q -> layers[i-1][N][d] with the pending decay applied: the eager matrices the reference would hold at
 * now_time.  (backup_random_projections / state_dict: models/TPNet.py:141-147) */
int tpnet_export_layers(const tpnet_state* st, float* const* layers, double now_time, double lambda, void* stream);

/* Eager dense decay, P[i] <- P[i] * factors[i-1] (host array, L floats), tref <- t_new for all rows.
 * (models/TPNet.py:83-85; factors = f32(exp(-lambda*(t_new-now))^i) computed by the caller in f64) */
int tpnet_decay(const tpnet_state* st, const float* factors, double t_new, void* stream);

/* get_random_projections (models/TPNet.py:101-110): out[(i*n + k)*d + :] = P[i][ids[k]] at now_time, i = 0..L. */
int tpnet_gather_rows(const tpnet_state* st, const int64_t* ids, int64_t n, double now_time, double lambda,
                      float* out, void* stream);

/* Single elements of every layer: out[k][i] = P[i][rows[k]][cols[k]] at now_time, i = 0..L (out: device float[n][L+1]).
 * With a square table (d = N) this is the readout of the reference's sibling walk-matrix state, PINT's
 * MatrixMemory.get_memory before its normalisation (models/MemoryModel.py:396-405: `matrix[src, dst]`). */
int tpnet_gather_elems(const tpnet_state* st, const int64_t* rows, const int64_t* cols, int64_t n, double now_time,
                       double lambda, float* out, void* stream);

/* get_pair_wise_feature up to (not including) self.mlp (models/TPNet.py:112-128):
 * out[p][(2L+2)*a + b] = <R_a, R_b>, R = [P0[u_p]..PL[u_p], P0[v_p]..PL[v_p]] at now_time; then, unless
 * TPNET_FLAG_NOT_SCALE, x<0 -> 0 and log(x+1).  u, v: device int64[n]; out: device float[n][(2L+2)^2]. */
int tpnet_pair_gram(const tpnet_state* st, const int64_t* u, const int64_t* v, int64_t n, double now_time,
                    double lambda, uint32_t flags, float* out, void* stream);

/* Two pairwise readouts that share their first node, u's rows loaded once: out1[p] = G(u_p, v1_p), out2[p] =
 * G(u_p, v2_p) (same layout and scaling as tpnet_pair_gram).  This is the shape of the reference's callers: the
 * decoder's (src,dst)/(src,neg) pairs (models/modules.py:112) and the encoder's relative encodings, where every
 * neighbour is paired with the edge's src AND dst (models/TPNet.py:311-316). */
int tpnet_pair_gram_shared(const tpnet_state* st, const int64_t* u, const int64_t* v1, const int64_t* v2, int64_t n,
                           double now_time, double lambda, uint32_t flags, float* out1, float* out2, void* stream);

/* self.mlp = Linear(F, 4F) -> ReLU -> Linear(4F, F), F = (2L+2)^2 (models/TPNet.py:63-65), as device arrays in the layout
 * the fused kernel reads coalesced: w1t[k][j] = mlp[0].weight[j][k] ([F][H]), w2t[k][o] = mlp[2].weight[o][k] ([H][F]). */
typedef struct tpnet_mlp {
    const float* w1t; /* [F][H] */
    const float* b1;  /* [H] */
    const float* w2t; /* [H][F] */
    const float* b2;  /* [F] */
    int32_t F;        /* (2L+2)^2 */
    int32_t H;        /* 4 F */
    /* optional (NULL: never the matrix-core kernel), L = 3 only: the layouts the fp32 matrix-core kernel reads --
     * w1 = mlp[0].weight as is ([256][64]); w2f[((w * 2 + t) * 64 + lane) * 16 + s] = mlp[2].weight[32 t + (lane & 31)]
     * [32 w + (s & 3) + 8 (s >> 2) + 4 (lane >> 5)]  (w = 0..7 hidden tile, t = 0..1 output tile, s = 0..15 k-step) */
    const float* w1;
    const float* w2f;
    /* optional (NULL: the encoder's calls run readout and dense layers as two launches), F = 64 / H = 256 only: the image
     * tpnet_mlp_prepare_image writes -- both weight matrices split into two bf16 pieces per value, every 16-byte element the
     * operand of one lane of one v_mfma_f32_16x16x32_bf16, then b1 and b2 (tpnet_mlp_image_bytes() bytes, 16-byte aligned) */
    const void* wimg;
} tpnet_mlp;

/* get_pair_wise_feature INCLUDING self.mlp (models/TPNet.py:112-129) in one launch: the features of tpnet_pair_gram stay
 * in LDS and go through both dense layers in fp32 (differs from the torch layers in summation order only): on the fp32
 * matrix cores (v_mfma_f32_32x32x2_f32; L = 3, rows of >= 36 floats in 16-byte vectors, mlp->w1 / w2f given) for lists of any
 * length, else on the vector ALUs (any L; meant for short lists).
 * out: device float[n][F]; out_gram (optional, may be NULL): the pre-mlp features [n][F], what a backward pass needs.
 * Meant for the decoder's short pair lists (models/modules.py:112: n = batch size); any L in 1..4. */
int tpnet_pair_feature(const tpnet_state* st, const int64_t* u, const int64_t* v, int64_t n, double now_time,
                       double lambda, uint32_t flags, const tpnet_mlp* mlp, float* out_gram, float* out, void* stream);

/* self.mlp alone on the fp32 matrix cores (L = 3: 64 -> 256 -> 64; mlp->w1 / w2f required): y[n][64] = mlp(x[n][64]), for
 * features that a separate readout produced (the anchored / shared-first-node kernels of the encoder's call). */
int tpnet_mlp64_f32(const float* x, int64_t n, const tpnet_mlp* mlp, float* y, void* stream);

/* The derived layouts of a tpnet_mlp from the Parameters of self.mlp = Linear(F, H) -> ReLU -> Linear(H, F) (models/TPNet.py:63-65)
 * in ONE launch: w1 = mlp[0].weight [H][F], w2 = mlp[2].weight [F][H] (device, contiguous) -> w1t [F][H], w2t [H][F] and, for
 * F = 64 / H = 256 (else NULL), w2f in the gathered order tpnet_mlp documents.  b1, b2 and tpnet_mlp::w1 are the Parameters'
 * own storage.  A training loop calls this after every optimizer step (train_link_prediction.py:384-386). */
int tpnet_mlp_prepare(const float* w1, const float* w2, int32_t F, int32_t H, float* w1t, float* w2t, float* w2f, void* stream);
/* tpnet_mlp::wimg from the Parameters of self.mlp = Linear(64, 256) -> ReLU -> Linear(256, 64): w1 [256][64], b1 [256], w2 [64][256],
 * b2 [64] (device, contiguous) -> img (device, tpnet_mlp_image_bytes() bytes), one launch; to be called again whenever a
 * Parameter changed (the biases are copied into the image too). */
size_t tpnet_mlp_image_bytes(void);
int tpnet_mlp_prepare_image(const float* w1, const float* b1, const float* w2, const float* b2, void* img, void* stream);

/* ---- host-array entry points: what the reference's per-batch calls hand over are HOST numpy arrays (models/TPNet.py:
 * 74-77, 107, 117; train_link_prediction.py:359-373).  These variants take host pointers, check the ids on the host
 * (TPNET_ERR_INDEX for an id outside [-N, N); negative ids wrap like ATen indexing), copy them into a slot of a pinned,
 * device-mapped staging ring and launch kernels that read the slot directly -- no separate host->device copy is enqueued,
 * and a call costs one FFI crossing.  A tpnet_stage is the ONLY object this library allocates (pinned host memory + one
 * event per slot); a slot is reused only after the launch that read it has finished (the call waits if the ring wrapped).
 * A stage serves ONE calling thread and one device (the reference's loop is single-threaded under the GIL). */
typedef struct tpnet_stage tpnet_stage;
int tpnet_stage_create(int32_t slots, size_t slot_bytes, tpnet_stage** out);
/* (ABI 7) where the ring lives: mode 0 = pinned, device-mapped HOST memory (tpnet_stage_create: the kernels read the ids over PCIe,
 * one more ~2-us round trip at the head of every kernel's chain of dependent loads); 1 = DEVICE memory that the host writes through
 * the large BAR (fine-grained; +~1.6 us of host stores per 16 KB call, a local read at the head of the kernels; TPNET_ERR_NO_DEVICE
 * without a large BAR); -1 = 1 where the device has a large BAR, else 0.  Meant for the small per-batch ring: host stores through
 * the BAR run at ~7 GB/s. */
int tpnet_stage_create_ex(int32_t slots, size_t slot_bytes, int32_t mode, tpnet_stage** out);
int tpnet_stage_in_device_memory(const tpnet_stage* stage);
int tpnet_stage_destroy(tpnet_stage* stage);
/* largest n / B the host entry points accept for a stage (slot_bytes / 16, slot_bytes / 24 capped at 2048) */
int64_t tpnet_stage_max_pairs(const tpnet_stage* stage);
int64_t tpnet_stage_max_batch(const tpnet_stage* stage);

/* get_pair_wise_feature from host ids: mlp == NULL -> out = the pre-mlp features (tpnet_pair_gram); else out = mlp(features)
 * and out_gram (optional) the pre-mlp features (tpnet_pair_feature). */
int tpnet_host_pair_feature(const tpnet_state* st, tpnet_stage* stage, const int64_t* h_u, const int64_t* h_v, int64_t n,
                            double now_time, double lambda, uint32_t flags, const tpnet_mlp* mlp, float* out_gram,
                            float* out, void* stream);

/* update (models/TPNet.py:67-99) from host arrays, B <= tpnet_stage_max_batch: ONE plan kernel (a single workgroup sorts
 * the batch's 2B contributions in LDS and writes the item lists) + the step kernel.  workspace: tpnet_workspace_bytes(B, B).
 * Larger batches (up to slot_bytes / 24 edges): the staged arrays are copied to the tail of the workspace (which must then hold
 * tpnet_workspace_bytes(B, B) rounded up to 256 + 24 B bytes) and tpnet_update plans and steps from there. */
int tpnet_host_update(const tpnet_state* st, tpnet_stage* stage, const int64_t* h_src, const int64_t* h_dst,
                      const double* h_t, int64_t B, double now_time, double lambda, uint32_t launch_id, uint32_t flags,
                      void* workspace, size_t ws_bytes, void* stream);

/* The encoder's readout (models/TPNet.py:311-324): row i has two anchors a1[i], a2[i] (the edge's src and dst) and K sampled
 * neighbours neigh[i*K .. i*K+K); out1[(i*K + k)] = G(neigh[i*K+k], a1[i]), out2[(i*K + k)] = G(neigh[i*K+k], a2[i]), rows of
 * (2L+2)^2 floats laid out and scaled like tpnet_pair_gram's -- with out2 = out1 + n_rows*K*(2L+2)^2 this IS the reference's
 * get_pair_wise_feature(tile(neigh, 2), concat(repeat(a1, K), repeat(a2, K))) before self.mlp.  One lane group walks a row:
 * the anchors' rows are fetched once per row (not once per pair) and stay in registers, their own Gram blocks are reduced
 * once per row.  Needs rows of exactly one chunk of 16-byte vectors (d = 64, 128, 256, 512: tpnet_pair_gram_anchored_supported
 * returns 1); other shapes take tpnet_pair_gram_shared / tpnet_pair_gram.  Rows of 64 / 128 floats with L = 3 and K >= 4 are
 * served by the matrix cores (csrc/encoder_mfma.hip: 16 x 16 x 32 bf16 products on operands split into three bf16 pieces, fp32
 * accumulation over d inside the pipe -- no cross-lane reduction; TPNET_FLAG_NO_MFMA_READOUT keeps the vector-ALU walk). */
int tpnet_pair_gram_anchored(const tpnet_state* st, const int64_t* neigh, const int64_t* a1, const int64_t* a2,
                             int64_t n_rows, int32_t K, double now_time, double lambda, uint32_t flags, float* out1,
                             float* out2, void* stream);
int tpnet_pair_gram_anchored_supported(const tpnet_state* st);

/* Workspace for tpnet_update / tpnet_run_stream with at most max_edges edges per call. */
size_t tpnet_workspace_bytes(int64_t max_edges, int64_t batch);

/* Workspace for tpnet_run_stream on a table of N rows x d columns x L layers and a stream of up to max_edges edges: the
 * larger of the per-batch plan (capped at a chunk of ~2 M edges; longer streams are walked chunk by chunk) and what the
 * WINDOWED schedule needs: its plan + a version log of 2*L*d*4 bytes per edge of a chunk (at most 16 GiB).  With at
 * least this much, tpnet_run_stream runs a stream of small batches (<= 4096 edges, d % 4 == 0) as a software pipeline
 * over windows of batches -- one launch per window carrying one layer of the update for each of L consecutive windows
 * plus the readouts of the window behind them -- instead of one launch per batch; with less it falls back to per-batch
 * launches.  The two schedules differ in f32 summation order only (both within 1e-4 of the reference). */
size_t tpnet_stream_workspace_bytes(int64_t N, int32_t d, int32_t L, int64_t max_edges, int64_t batch);
/* The same with the version log of a chunk capped at log_cap_bytes (0 = the library's 16 GiB): the windowed schedule costs
 * 2*L*d*4 bytes of log per edge of a chunk (C2: 3 KB per edge, 484 MB for one Wikipedia epoch) plus ~0.5 KB per edge of plan; a
 * caller short of memory trades chunk length (more pipeline fills and drains) for workspace.  tpnet_run_stream takes whatever
 * chunk the workspace it is given holds.  Both functions add, for a stream of 2 .. 64 chunks, the room that keeps every chunk's
 * plan (everything but the log) resident: what tpnet_run_stream_tagged needs to replay such a stream. */
size_t tpnet_stream_workspace_bytes_capped(int64_t N, int32_t d, int32_t L, int64_t max_edges, int64_t batch,
                                           size_t log_cap_bytes);

/* The room tpnet_stream_workspace_bytes(_capped) adds so that a stream of several chunks can be replayed (every chunk's plan kept
 * side by side in front of one version log) is bounded: it is granted only while the whole workspace stays within
 * TPNET_ARENA_MAX_RATIO times the workspace of ONE chunk; beyond that the functions return the one-chunk size (the stream then
 * runs chunk by chunk and is planned again every time).  A 100 M-edge C2 stream asks for 21 GB, not 108. */
#define TPNET_ARENA_MAX_RATIO 3

/* Which schedule tpnet_run_stream / tpnet_run_stream_tagged would take for a stream of E edges in batches of `batch` on a table of
 * N x d x L with these flags and a workspace of ws_bytes: 1 = the windowed pipeline (k_wpipe: one launch per window of batches),
 * 0 = one launch per batch (k_step).  Host-side arithmetic only; lets a caller warm up the kernels the real call will run. */
int tpnet_stream_schedule(int64_t N, int32_t d, int32_t L, int64_t E, int64_t batch, uint32_t flags, size_t ws_bytes);

/* update (models/TPNet.py:67-99) for one batch: src, dst device int64[B], t device double[B] (absolute times,
 * chronological; t[B-1] is the new now_time).  now_time = the module's clock before the call; launch_id = a
 * caller-kept counter, strictly increasing over calls that write the state (start at 1).  The new clock t[B-1] is
 * read on the device; the caller keeps its own host copy of it. */
int tpnet_update(const tpnet_state* st, const int64_t* src, const int64_t* dst, const double* t, int64_t B,
                 double now_time, double lambda, uint32_t launch_id, uint32_t flags, void* workspace,
                 size_t ws_bytes, void* stream);

/* The caller loop of train_link_prediction.py:253-373 / evaluate_models_utils.py:56-184 for a device-resident
 * edge stream: for each chronological batch of `batch` edges (last one partial): readout (src,dst) and
 * (src,neg) on the pre-batch state, then update.  out_pos/out_neg: device float[E][(2L+2)^2] (pre-mlp
 * features); either may be NULL to skip that readout.  neg may be NULL (then out_neg must be NULL).
 * now_time: in = clock before the stream; the new clock is t[E-1] (returned through *t_end_out if non-NULL,
 * which costs one 8-byte device->host copy + stream sync at the end; pass NULL to stay asynchronous).
 * launch_id_base: first launch id; the call uses ids launch_id_base .. launch_id_base + ceil(E/batch) - 1. */
int tpnet_run_stream(const tpnet_state* st, const int64_t* src, const int64_t* dst, const int64_t* neg,
                     const double* t, int64_t E, int64_t batch, double now_time, double lambda,
                     uint32_t launch_id_base, uint32_t flags, float* out_pos, float* out_neg,
                     void* workspace, size_t ws_bytes, double* t_end_out, void* stream);

/* The same loop for a stream that is run AGAIN: train_link_prediction.py:234-253 replays the same chronological stream every
 * epoch (reset_random_projections at the epoch's start, then the same batches), so the plan of the update -- sorted
 * contributions, chains, version references of src / dst -- is the same every epoch; only the negatives are drawn anew.  A call
 * that runs the windowed schedule on the whole stream leaves its plan in the workspace and describes it in tag->built -- the stream
 * is ONE chunk, or (since ABI 6) up to 64 chunks whose plans find room side by side in front of the one version log they share
 * (tpnet_stream_workspace_bytes sizes the workspace for that) --; a later call with equal arguments, the same workspace and the same two signatures skips the planning and only
 * resolves the negatives' readout references again (tag->replayed = 1).  The CALLER vouches with the signatures:
 *   stream_sig: identifies the CONTENTS of src / dst / t (equal value = unchanged arrays; 0 = never replay);
 *   table_sig:  identifies the table's per-node (current copy, reference time) state before the call -- e.g. one constant
 *               for "just after tpnet_state_init(t0)" per t0, a fresh value after anything else wrote the state (0 = never).
 * A stream on the per-batch schedule that is ONE chunk (up to ~2 M edges) replays its plan too: that plan is a function of src / dst / t,
 * now_time and the flags alone, so stream_sig decides and table_sig is not looked at.
 * A caller that lets anything else use the workspace in between clears the tag (memset 0).  tag == NULL: tpnet_run_stream. */
typedef struct tpnet_plan_tag {
    uint64_t table_sig;   /* in */
    uint64_t stream_sig;  /* in */
    uint64_t replayed;    /* out: 1 = this call reused the plan in the workspace */
    uint64_t built[20];   /* library-owned description of the plan the workspace holds (zero = none) */
} tpnet_plan_tag;
int tpnet_run_stream_tagged(const tpnet_state* st, const int64_t* src, const int64_t* dst, const int64_t* neg,
                            const double* t, int64_t E, int64_t batch, double now_time, double lambda,
                            uint32_t launch_id_base, uint32_t flags, float* out_pos, float* out_neg,
                            void* workspace, size_t ws_bytes, double* t_end_out, void* stream, tpnet_plan_tag* tag);

/* ---- row-sharded multi-GPU building blocks (one process per GPU; the exchange itself is the caller's RCCL call) ----
 * Rows are owned cyclically: owner(n) = n % G.  Every rank holds the full stream; per batch it (1) packs the
 * touched rows it owns, (2) all-gathers them over RCCL, (3) unpacks the other ranks' rows into its local table,
 * (4) runs the fused step restricted to the targets / pairs it owns.  See tpnet_amd/sharded.py. */

/* Plan the update of a whole device-resident stream once (the per-batch item lists tpnet_step_batch consumes).
 * The workspace must hold the whole stream: tpnet_workspace_bytes(E, batch). */
int tpnet_plan_stream(const tpnet_state* st, const int64_t* src, const int64_t* dst, const double* t, int64_t E,
                      int64_t batch, double now_time, double lambda, uint32_t flags, void* workspace, size_t ws_bytes,
                      void* stream);

/* One fused launch for batch b of a planned stream: readout of the pairs whose src node this rank owns (other
 * output rows are left untouched) + update of the targets it owns (id % own_mod == own_rem; own_mod = 1: all;
 * own_mod = 0: ids < own_rem, the owned rows of a compact local table). */
int tpnet_step_batch(const tpnet_state* st, const int64_t* src, const int64_t* dst, const int64_t* neg,
                     const double* t, int64_t E, int64_t batch, int64_t b, double lambda, uint32_t launch_id,
                     uint32_t flags, int32_t own_mod, int32_t own_rem, float* out_pos, float* out_neg,
                     void* workspace, size_t ws_bytes, void* stream);

/* out[k][i][:] = P[i+1][ids[k]] at now_time (i = 0..L-1; layer 0 is static and replicated, never exchanged). */
int tpnet_pack_rows(const tpnet_state* st, const int64_t* ids, int64_t n, double now_time, double lambda, float* out,
                    void* stream);
/* Local copy of row ids[k] <- in[k] (expressed at now_time).  Only for rows owned by another rank. */
int tpnet_unpack_rows(const tpnet_state* st, const int64_t* ids, int64_t n, double now_time, const float* in,
                      void* stream);

/* All peers' rows in one launch after the all-gather: ids = the batch's touched nodes ordered by (owner, node)
 * (device int64[n]), recv = device float [G][maxc][L*d] as gathered, offs = device int64[G] start of each owner's
 * run inside ids.  Rows owned by `me` are skipped. */
int tpnet_unpack_gathered(const tpnet_state* st, const int64_t* ids, int64_t n, double now_time, const float* recv,
                          int64_t maxc, const int64_t* offs, int32_t G, int32_t me, void* stream);

/* Compact row shards: a rank's table holds ONLY the rows it owns (local row = global id / G) followed by halo rows, which
 * per batch receive copies of the other ranks' rows that the batch reads.  Layer 0 is sharded like the others, so whole
 * bundles travel:
 *   tpnet_pack_bundles:   out[k][i][:] = P[i][ids[k]] at now_time, i = 0..L          (ids: LOCAL rows of this rank's table)
 *   tpnet_unpack_bundles: list entry k of the batch's touched nodes (ordered by (owner, node); offs[r] = start of owner r's run)
 *                         sits at recv[r][k - offs[r]] ([G][maxc][L+1][d] as all-gathered) and is written to local row
 *                         local_ids[k] (< 0: skip): layer 0 into p0, layers 1..L into the row's current copy, as of now_time.
 * tpnet_step_batch with own_mod = 0 then restricts the step to the targets / pair sources with local id < own_rem. */
int tpnet_pack_bundles(const tpnet_state* st, const int64_t* ids, int64_t n, double now_time, double lambda, float* out,
                       void* stream);
int tpnet_unpack_bundles(const tpnet_state* st, const int64_t* local_ids, int64_t n, double now_time, const float* recv,
                         int64_t maxc, const int64_t* offs, int32_t G, void* stream);

/* One batch of the compact row-sharded stream in ONE call, the exchange included: tpnet_pack_bundles(pack_ids, n_pack) into
 * `send` ([maxc][(L+1)*d], padded), ONE RCCL all-gather into `recv` ([G][maxc][(L+1)*d]) on `stream`, tpnet_unpack_bundles
 * (unpack_ids, n_unpack, offs), then tpnet_step_batch(b) with own_mod = 0, own_rem = n_owned.  comm: a communicator from
 * tpnet_rccl_comm_create (NULL, or maxc = 0: no exchange, just the step).  RCCL is resolved at run time (dlopen of lib_path,
 * else "librccl.so" -- the library PyTorch already loaded): tpnet_rccl_unique_id on one rank, the 128 bytes broadcast by the
 * caller, tpnet_rccl_comm_create on every rank (collective: ncclCommInitRank). */
int tpnet_rccl_unique_id(const char* lib_path, void* id128);
int tpnet_rccl_comm_create(const char* lib_path, const void* id128, int32_t nranks, int32_t rank, void** comm);
int tpnet_rccl_comm_destroy(void* comm);
int tpnet_rows_step(const tpnet_state* st, void* comm, const int64_t* pack_ids, int64_t n_pack, float* send, float* recv,
                    int64_t maxc, const int64_t* unpack_ids, int64_t n_unpack, const int64_t* offs, int32_t G,
                    double now_time, const int64_t* src, const int64_t* dst, const int64_t* neg, const double* t, int64_t E,
                    int64_t batch, int64_t b, double lambda, uint32_t launch_id, uint32_t flags, int32_t n_owned,
                    float* out_pos, float* out_neg, void* workspace, size_t ws_bytes, void* stream);

/* The same batch with the TARGETED exchange: every owned row travels only to the ranks that read it, as two
 * messages (layer 0; layers 1..L decayed to now_time) that the reader receives STRAIGHT INTO the halo rows of its p0 and q
 * arrays -- so a sharded batch is two launches (pack, step) around one grouped ncclSend / ncclRecv, with no unpack.
 *   pack_ids:  device int64, the LOCAL rows to send, ordered by (reader rank, node): send_cnt[r] of them for rank r;
 *   send_cnt / recv_cnt: HOST int64[G] (entry `me` = 0); the rows from rank r land in halo rows
 *              [n_owned + sum(recv_cnt[:r]), ...) in the sender's order -- the local ids the caller relabelled the batch with;
 *   send_p0 [sum(send_cnt)][d], send_q [sum(send_cnt)][L*d]: device scratch.
 * tpnet_pack_split is the pack launch alone (a caller that moves the rows with another transport, e.g. the gloo tests): it
 * also stamps halo rows [halo0, halo0 + n_halo) as "as of now_time". */
int tpnet_pack_split(const tpnet_state* st, const int64_t* ids, int64_t n, double now_time, double lambda, float* out_p0,
                     float* out_q, int64_t halo0, int64_t n_halo, void* stream);
int tpnet_rows_step_targeted(const tpnet_state* st, void* comm, const int64_t* pack_ids, float* send_p0, float* send_q,
                             const int64_t* send_cnt, const int64_t* recv_cnt, int32_t G, int32_t me, double now_time,
                             const int64_t* src, const int64_t* dst, const int64_t* neg, const double* t, int64_t E,
                             int64_t batch, int64_t b, double lambda, uint32_t launch_id, uint32_t flags, int32_t n_owned,
                             float* out_pos, float* out_neg, void* workspace, size_t ws_bytes, void* stream);
/* Batches [b0, b1) of a stream through tpnet_rows_step_targeted in ONE call (the per-batch loop of tpnet_amd/sharded.py was one
 * FFI crossing, ~10 us of host time, per batch): pack_start[b] = offset of batch b's rows in pack_ids, send_cnt / recv_cnt
 * HOST int64[nb][G], t_last HOST double[nb] (the clock batch b leaves: its rows of batch b + 1 are packed at it), launch ids
 * launch_id_base + b. */
int tpnet_rows_stream_targeted(const tpnet_state* st, void* comm, const int64_t* pack_ids, const int64_t* pack_start,
                               float* send_p0, float* send_q, const int64_t* send_cnt, const int64_t* recv_cnt, int32_t G,
                               int32_t me, double now_time, const double* t_last, const int64_t* src, const int64_t* dst,
                               const int64_t* neg, const double* t, int64_t E, int64_t batch, int64_t b0, int64_t b1,
                               double lambda, uint32_t launch_id_base, uint32_t flags, int32_t n_owned, float* out_pos,
                               float* out_neg, void* workspace, size_t ws_bytes, void* stream);

/* The targeted exchange's plan on the device (what tpnet_rows_step_targeted's callers need per batch): for rank `me` of G (owner(n)
 * = n % G), from the stream's device arrays, two launches and no synchronisation:
 *   recv_keys [nb][cap] uint32 (cap = tpnet_xplan_capacity()): the distinct remote nodes the rank reads in batch b, as (owner <<
 *       bits(N) | node), ascending = the order the peers' messages land in the halo rows;
 *   pack_ids  [nb][cap] int64: the local rows (node / G) the rank sends in batch b, ordered (reader, node), a row once per reader;
 *   cnt       [nb][2][G] int64: rows received from each owner / rows sent to each reader (the host reads these back);
 *   status    [2] uint32: ids outside [0, N) seen; batches whose lists exceeded cap (then the caller plans another way);
 *   local_src / local_dst / local_neg [E] int64: every endpoint relabelled (owned: n / G; received in its batch: n_owned + its
 *       place in the batch's receive list; else n_owned).
 * TPNET_ERR_BAD_ARG when bits(N) + bits(G) > 31 or G > 64 (the caller plans another way). */
int64_t tpnet_xplan_capacity(void);
int tpnet_xplan_targeted(const int64_t* src, const int64_t* dst, const int64_t* neg, int64_t E, int64_t batch, int64_t N, int32_t G,
                         int32_t me, int32_t n_owned, uint32_t* recv_keys, int64_t* pack_ids, int64_t* cnt, uint32_t* status,
                         int64_t* local_src, int64_t* local_dst, int64_t* local_neg, void* stream);

/* The same plan for batches whose lists exceed tpnet_xplan_capacity() (C4's law over 8 ranks: ~25 000 rows received per batch of
 * 80 000 edges): one device-wide radix sort of the call's (list, batch, peer, node) keys instead of a workgroup per batch.  The
 * lists are COMPACT: pack_ids holds the local rows to send, all batches back to back in (batch, reader, node) order (at most 3 E;
 * batch b starts at the sum of the earlier batches' send counts), and a remote node's halo row is n_owned + its place in its
 * batch's receive list ((owner, node) order) -- both as tpnet_rows_stream_targeted consumes them.  cnt / status / local_* as
 * above.  scratch: tpnet_xplan_large_bytes(E, batch, G) device bytes.  Synchronises `stream` ONCE in the middle (the number of
 * keys sizes the sort); the caller reads cnt / status back afterwards as for tpnet_xplan_targeted. */
size_t tpnet_xplan_large_bytes(int64_t E, int64_t batch, int32_t G);
int tpnet_xplan_targeted_large(const int64_t* src, const int64_t* dst, const int64_t* neg, int64_t E, int64_t batch, int64_t N,
                               int32_t G, int32_t me, int32_t n_owned, void* scratch, size_t scratch_bytes, int64_t* pack_ids,
                               int64_t* cnt, uint32_t* status, int64_t* local_src, int64_t* local_dst, int64_t* local_neg,
                               void* stream);

/* ---- a row shard on the WINDOWED pipeline (ABI 7; csrc/wshard.hip) -------------------------------------------
 * The per-batch shard above exchanges rows and launches a kernel per batch.  Here one chunk of the stream (the whole call) runs as
 * the software pipeline of tpnet_run_stream's windowed schedule -- one launch per window of batches -- and the ranks meet once per
 * launch: every node the chunk touches gets ONE halo row for the whole chunk (filled at its start with the owner's pre-chunk row),
 * every rank plans from the whole stream and derives, without a request round, which of its runs' results another rank reads; after
 * each launch those rows are packed, exchanged with ONE grouped ncclSend / ncclRecv, and unpacked into the version log.  A log slot
 * travels bit for bit: G shards compute what one GPU computes on the windowed schedule.
 *   st: the rank's LOCAL table (n_owned rows + halo rows: tpnet_state::N = n_owned + halo capacity); src / dst / neg: GLOBAL ids.
 *   tpnet_wshard_plan returns TPNET_OK and *out, or 1 = this call is not served (batches beyond 8 192 edges, fewer than 4 batches,
 *   more remote nodes in the chunk than halo rows, a table too large for the dense planner, exact / packed modes ...): the caller
 *   takes tpnet_rows_stream_targeted.  It synchronises `stream` twice (halo counts, message counts).  A tpnet_wshard is a HOST object
 *   (counts and views into the caller's workspace; the second thing besides tpnet_stage this library allocates); device memory is
 *   the caller's: workspace (tpnet_wshard_workspace_bytes) and the exchange buffers (tpnet_wshard_info -> tpnet_wshard_set_buffers).
 *   phases of begin / step (bit mask): 1 = launch the pipeline step, 2 = pack, 4 = exchange over `comm` (tpnet_rccl_comm_create),
 *   8 = unpack; a caller with another transport runs 2, moves the rows itself (sendbuf rows [sum of send_cnt[j][:r], +send_cnt[j][r])
 *   go to peer r's recvbuf rows [sum of ITS recv_cnt[j][:me], ...); the chunk's halo rows: send_p0 / send_q of owner o to halo rows
 *   [n_owned + hstart(o), ...) of p0 and of copy 0 of q, hstart(o) = sum of chunk_cnt[o'] over o' < o, o' != me), then 8. */
typedef struct tpnet_wshard tpnet_wshard;
size_t tpnet_wshard_workspace_bytes(int64_t n_local, int32_t d, int32_t L, int64_t E, int64_t batch, int32_t G, int32_t n_owned);
int tpnet_wshard_plan(const tpnet_state* st, const int64_t* src, const int64_t* dst, const int64_t* neg, const double* t, int64_t E,
                      int64_t batch, int64_t N_global, int32_t G, int32_t me, int32_t n_owned, double now_time, double lambda,
                      uint32_t flags, int32_t want_pos, int32_t want_neg, void* workspace, size_t ws_bytes, void* stream,
                      tpnet_wshard** out);
int tpnet_wshard_info(const tpnet_wshard* w, int64_t* n_steps, int64_t* halo_rows, int64_t* max_send_rows, int64_t* max_recv_rows,
                      const int64_t** chunk_cnt, const int64_t** send_cnt, const int64_t** recv_cnt);
int tpnet_wshard_set_buffers(tpnet_wshard* w, float* send_p0, float* send_q, float* sendbuf, float* recvbuf);
int tpnet_wshard_begin(tpnet_wshard* w, void* comm, uint32_t phases, void* stream);
int tpnet_wshard_step(tpnet_wshard* w, void* comm, int64_t j, uint32_t phases, float* out_pos, float* out_neg, void* stream);
int tpnet_wshard_finish(tpnet_wshard* w, uint32_t launch_id, void* stream);
/* begin + every step + finish in one call (comm may be NULL with one rank) */
int tpnet_wshard_run(tpnet_wshard* w, void* comm, float* out_pos, float* out_neg, uint32_t launch_id, void* stream);
void tpnet_wshard_destroy(tpnet_wshard* w);

/* ---- the step in front of the path: 'recent' historical-neighbour sampling on the device ----------
 * Replaces NeighborSampler('recent') + get_neighbor_sampler (utils/utils.py:82-224, 293-312): undirected adjacency,
 * per node sorted by time (stable: ties keep the reference's append order), neighbours strictly BEFORE the query
 * time, the K most recent at the back of the row, zeros in front. */
size_t tpnet_sampler_bytes(int64_t E, int64_t num_nodes);
/* sampler: caller-owned device buffer of tpnet_sampler_bytes(E, num_nodes); src/dst/t (and optional edge_ids, NULL =
 * 1..E) are device arrays of E interactions; num_nodes = largest node id + 1. */
int tpnet_sampler_build(void* sampler, size_t sampler_bytes, const int64_t* src, const int64_t* dst, const double* t,
                        const int64_t* edge_ids, int64_t E, int64_t num_nodes, void* stream);
/* out_ids [n][K] int64 (required), out_eids [n][K] int64 and out_times [n][K] double (optional, may be NULL). */
int tpnet_sample_recent(const void* sampler, int64_t E, int64_t num_nodes, const int64_t* node_ids, const double* times,
                        int64_t n, int32_t K, int64_t* out_ids, int64_t* out_eids, double* out_times, void* stream);

/* The encoder's readout with the ids resident on the device end to end (models/TPNet.py:280-324, SURVEY §8 f-3 -> f-2): for one
 * (src, other) batch of B edges -- other = dst or the negatives -- the 2B nodes [src; other] at times tile(t, 2) draw their K most
 * recent neighbours from the device sampler, and every neighbour is paired with the edge's two endpoints: out[0][(i*K + k)] =
 * G(neigh[i][k], src[i % B]), out[1][...] = G(neigh[i][k], other[i % B]), i in [0, 2B) -- the reference's
 * get_pair_wise_feature(tile(neigh, 2), concat(repeat(tile(src,2),K), repeat(tile(other,2),K))) before self.mlp, without any index
 * array and without the neighbour ids visiting the host.  One call, two launches (rows + sampler, readout).  scratch: tpnet_encoder_scratch_bytes(B, K)
 * device bytes; the sampled neighbour ids [2B][K] (int64) are left at (scratch rounded up to 256) + 64 B bytes for the caller's
 * other neighbour features.  Needs tpnet_pair_gram_anchored_supported(st).  src / other / t: device arrays of B. */
size_t tpnet_encoder_scratch_bytes(int64_t B, int32_t K);
int tpnet_encoder_gram(const tpnet_state* st, const void* sampler, int64_t E, int64_t num_nodes, const int64_t* src,
                       const int64_t* other, const double* t, int64_t B, int32_t K, double now_time, double lambda,
                       uint32_t flags, void* scratch, size_t scratch_bytes, float* out, void* stream);

/* The first half of tpnet_encoder_gram alone: the rows [src; other] at tile(t, 2), their anchors and their K sampled neighbours,
 * laid out in `scratch` (rounded up to 256): nodes int64[2B] | times double[2B] | a1 int64[2B] | a2 int64[2B] | neigh int64[2B][K]
 * | (rows of <= 128 floats: the call's pairs spelled out for the generic readout) u int64[4BK] | v int64[4BK]. */
int tpnet_encoder_rows(const tpnet_state* st, const void* sampler, int64_t E, int64_t num_nodes, const int64_t* src,
                       const int64_t* other, const double* t, int64_t B, int32_t K, void* scratch, size_t scratch_bytes,
                       void* stream);

/* The encoder's call INCLUDING self.mlp (models/TPNet.py:311-324, 129; L = 3, rows of one chunk of 16-byte vectors): gram =
 * tpnet_pair_gram_anchored's output [2][n_rows*K][64] (kept: what a backward pass needs), out = mlp(gram) in the fp32 class
 * (tpnet_mlp64_f32's kernel), both on `stream`: the encoder's call as ONE crossing.  Where tpnet_encoder_fused_supported
 * returns 1 (rows of 64 / 128 floats, K >= 4, mlp->wimg given) readout and dense layers are ONE launch on the matrix cores
 * (csrc/encoder_mfma.hip) and gram may be NULL: the pre-mlp features are then never written. */
int tpnet_encoder_fused_supported(const tpnet_state* st, int64_t n_rows, int32_t K, const tpnet_mlp* mlp);
int tpnet_anchored_features(const tpnet_state* st, const int64_t* neigh, const int64_t* a1, const int64_t* a2, int64_t n_rows,
                            int32_t K, double now_time, double lambda, uint32_t flags, const tpnet_mlp* mlp, float* gram,
                            float* out, void* stream);
/* tpnet_encoder_rows + tpnet_anchored_features: the encoder's whole readout for one (src, other) batch, ids resident on the
 * device, self.mlp included, ONE call. */
int tpnet_encoder_features(const tpnet_state* st, const void* sampler, int64_t E, int64_t num_nodes, const int64_t* src,
                           const int64_t* other, const double* t, int64_t B, int32_t K, double now_time, double lambda,
                           uint32_t flags, const tpnet_mlp* mlp, void* scratch, size_t scratch_bytes, float* gram, float* out,
                           void* stream);

/* tpnet_encoder_features with the batch's src / other / times as HOST arrays (the reference's loop slices them from the edge
 * list on the host, train_link_prediction.py:325-340): staged through `stage` (B <= slot_bytes / 24), read in place by the row
 * set-up kernel; no copy is enqueued.  TPNET_ERR_INDEX for an id outside [0, N). */
int tpnet_host_encoder_features(const tpnet_state* st, tpnet_stage* stage, const void* sampler, int64_t E, int64_t num_nodes,
                                const int64_t* h_src, const int64_t* h_other, const double* h_t, int64_t B, int32_t K,
                                double now_time, double lambda, uint32_t flags, const tpnet_mlp* mlp, void* scratch,
                                size_t scratch_bytes, float* gram, float* out, void* stream);

/* Host-side check (no GPU work): is (src, dst) of a get_pair_wise_feature call of n pairs the encoder's pattern (models/TPNet.py:
 * 311-316: src = tile(neigh, 2), dst = concat(repeat(a1, K), repeat(a2, K)))?  Returns K >= 2, else 0 (N is unused: the ids' range
 * is the caller's to check). */
int64_t tpnet_host_encoder_pattern(const int64_t* src, const int64_t* dst, int64_t n, int64_t N);
/* The encoder's call from the reference's own HOST index arrays in one crossing (ABI 6): pattern check, range check, the n / 2
 * neighbour ids + the anchors staged through `stage` (slot_bytes >= (n / 2 + n / K) * 8; the kernel reads them there) and
 * tpnet_anchored_features launched on `stream`.  *K_out = the pattern's K (>= 4) if the call was served, 0 if not (not the
 * pattern, K < 4, an id outside [0, N), a shape the kernels do not serve): the caller then takes the general path.  gram: as for
 * tpnet_anchored_features (may be NULL where tpnet_encoder_fused_supported). */
int tpnet_host_anchored_features(const tpnet_state* st, tpnet_stage* stage, const int64_t* h_src, const int64_t* h_dst, int64_t n,
                                 double now_time, double lambda, uint32_t flags, const tpnet_mlp* mlp, float* gram, float* out,
                                 int32_t* K_out, void* stream);

/* ---- the step behind the path (BASELINE config 5): self.mlp = Linear(64,256)->ReLU->Linear(256,64)
 * (models/TPNet.py:64-65,129) fused in one kernel on the bf16 matrix cores, fp32 accumulate, L = 3 only.
 * x [n][64] f32 (the readout's features), y [n][64] f32.  w1_bf16: [256][64] bf16 = mlp[0].weight; w2p_bf16: [64][256]
 * bf16 = mlp[2].weight with its hidden axis permuted per 32-tile to the accumulator order
 * k' = 16 s + 8 h + j  <-  k = 16 s + 8 (j>>2) + 4 h + (j&3)   (s in 0..1, h in 0..1, j in 0..7). */
int tpnet_mlp64_bf16(const float* x, int64_t n, const void* w1_bf16, const float* b1, const void* w2p_bf16,
                     const float* b2, float* y, void* stream);

/* Backward of self.mlp with respect to its weights on the bf16 matrix cores (training; the projections carry no gradient,
 * models/TPNet.py:49-62).  x: the pre-mlp features [n][64] f32, gy: the gradient of the output [n][64] f32; w1_bf16 [256][64]
 * = mlp[0].weight, w2t_bf16 [256][64] = mlp[2].weight transposed.  Every workgroup writes ONE partial result of
 * tpnet_mlp64_bwd_partial_floats() floats, laid out gW1 [256][64] | gW2 [64][256] | gb1 [256] (gb2 = the column sums of gy is
 * the caller's); the call returns the
 * number of partials written (<= n_partial; negative = error) and the caller sums them (fixed order: deterministic). */
int64_t tpnet_mlp64_bwd_partial_floats(void);
int tpnet_mlp64_bwd_bf16(const float* x, const float* gy, int64_t n, const void* w1_bf16, const float* b1,
                         const void* w2t_bf16, float* partial, int32_t n_partial, void* stream);
/* The same backward in the fp32 class of the default forward paths (round 4): every operand in two bf16 pieces, every product as
 * lo*hi + hi*lo + hi*hi with fp32 accumulation; weights from the f32 layouts of `mlp` (mlp->w1 = mlp[0].weight [256][64], mlp->w2t =
 * mlp[2].weight transposed [256][64], mlp->b1).  Same partial layout and return value as tpnet_mlp64_bwd_bf16. */
int tpnet_mlp64_bwd_f32(const float* x, const float* gy, int64_t n, const tpnet_mlp* mlp, float* partial, int32_t n_partial,
                        void* stream);

/* get_pair_wise_feature with self.mlp on the bf16 matrix cores INSIDE the readout kernel (L = 3, d % 4 == 0 and d >= 64):
 * 8 waves form the features of 32 pairs into an LDS tile, which is the B operand of layer 1; wave w owns hidden units
 * [32w, 32w+32) of both layers, the 8 partial outputs are added in a fixed order.  The features never touch HBM (out_gram,
 * optional: a copy for a backward pass).  Weights as for tpnet_mlp64_bf16 (w1_bf16 [256][64], w2p_bf16 [64][256] permuted).
 * bf16 operands, fp32 accumulation: ~1e-2 relative, opt-in (RandomProjectionModule.fused_mlp). */
int tpnet_pair_feature_bf16(const tpnet_state* st, const int64_t* u, const int64_t* v, int64_t n, double now_time,
                            double lambda, uint32_t flags, const void* w1_bf16, const float* b1, const void* w2p_bf16,
                            const float* b2, float* out_gram, float* out, void* stream);

/* LinkPredictor_v1 (models/modules.py:73-117): out[p] = fc2(relu(fc1(concat[src_emb[p], dst_emb[p], feat[p]]))) with ONE
 * output unit, both layers in one bf16 matrix-core kernel (fp32 accumulate); neither the concatenation nor the hidden
 * layer touches memory.  src_emb, dst_emb: device f32 [n][D] (D % 4 == 0; both NULL = the reference's not_encode mode,
 * zeros); feat: device f32 [n][F] (F % 16 == 0; NULL if the decoder has no pairwise feature).  w1p: bf16
 * [32*hidden_tiles][2*DP + F], DP = 16*ceil(D/16): fc1.weight with its input axis laid out [src | pad | dst | pad | feat]
 * and zero rows beyond the hidden width; b1p, w2p: f32 [32*hidden_tiles] (fc1.bias, fc2.weight[0], zero-padded); b2 =
 * fc2.bias[0].  hidden_tiles in 1..8 (hidden width <= 256).  out: device f32 [n]. */
int tpnet_decoder_bf16(const float* src_emb, const float* dst_emb, int32_t D, const float* feat, int32_t F, int64_t n,
                       const void* w1p_bf16, const float* b1p, const float* w2p, float b2, int32_t hidden_tiles,
                       float* out, void* stream);

/* The readout's element-wise tail, in place on x[n]: x<0 -> 0, then log(x + 1) (models/TPNet.py:126-128).  For callers
 * that ran the readout with TPNET_FLAG_NOT_SCALE because the raw Gram entries still had to be summed across GPUs
 * (column-sharded table: every rank holds d/G columns of every row, the entries are partial inner products). */
int tpnet_gram_finish(float* x, int64_t n, void* stream);

/* packed[n][(2L+2)(2L+3)/2] (TPNET_FLAG_PACKED rows, summed over the ranks) -> out[n][(2L+2)^2]: mirrors the upper
 * triangle and, unless TPNET_FLAG_NOT_SCALE, applies x<0 -> 0, log(x + 1). */
int tpnet_gram_unpack(const float* packed, int64_t n, int32_t L, uint32_t flags, float* out, void* stream);

/* Copies st->err to the host (synchronises the stream): returns TPNET_ERR_INDEX if any bad id was seen since
 * the last call (and clears the words), TPNET_OK otherwise. */
int tpnet_check_errors(const tpnet_state* st, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TPNET_HIP_H */
