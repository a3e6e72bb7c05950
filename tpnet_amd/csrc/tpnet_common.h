// Shared declarations of the gfx950 implementation (internal; the public ABI is include/tpnet_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/tpnet_hip.h"
#include "tpnet_dev.h"

// Developer knobs (window length, thresholds, roles switched off for timing experiments ...) exist only in builds made with
// -DTPNET_DEV (make VARIANT=dev VARFLAGS=-DTPNET_DEV -> libtpnet_hip_dev.so, loaded through TPNET_DEV_LIB by tools/): the product
// library never reads the environment, and none of the knob names is compiled into it (tests/test_abi.py checks).
#ifdef TPNET_DEV
static inline int tpnet_dev_env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#define TPNET_DEV_INT(NAME, DFLT) tpnet_dev_env_int("TPNET_DEV_" #NAME, (DFLT))
#define TPNET_DEV_STR(NAME) getenv("TPNET_DEV_" #NAME)
#else
#define TPNET_DEV_INT(NAME, DFLT) (DFLT)
#define TPNET_DEV_STR(NAME) ((const char*)nullptr)
#endif

namespace tpnet {

// ---- per-node record, 32 bytes: {ver, pad, tref[2], pad} ---------------------------------------------------
// tref is kept PER COPY: a launch that rewrites node n writes tref[new copy] and ver only, so a concurrent reader
// of the same launch (which must see the pre-batch row) finds tref[old copy] untouched whatever ver it observes.
struct NodeMeta {
    uint32_t ver;   // (launch_id << 1) | current copy
    uint32_t pad0;
    double tref[2];
    uint64_t pad1;
};
static_assert(sizeof(NodeMeta) == 32, "NodeMeta must be 32 bytes");

// Per-batch descriptor written by the plan kernels, read by the step kernel.
struct BatchDesc {
    int64_t e0;        // first edge of the batch
    int32_t ne;        // edges in the batch
    int32_t pad;
    double now;        // module clock before this batch's update (readout time)
    double t_last;     // t[e0+ne-1]: clock after the update (models/TPNet.py:76,99)
    uint32_t n_light;  // number of light items (targets whose contributions one wave group sums)
    uint32_t n_heavy;  // number of heavy items (one workgroup each)
    float decay[TPNET_MAX_LAYERS];  // eager mode: f32(exp(-lambda*(t_last-now))^i), i = 1..L
};

// One update item: a distinct target node of the batch and its run of contributions in the sorted arrays.  The first
// two contributions ride in the record itself, so that a typical target (1-2 contributions) needs no further
// dependent load before its rows can be fetched.
struct Item {
    uint32_t j0;     // first contribution (index into the sorted arrays)
    uint32_t cnt;    // number of contributions (0 = empty slot)
    int32_t target;  // the node
    int32_t p0;      // partner / weight of contribution j0
    float w0;
    int32_t p1;      // partner / weight of contribution j0+1 (if cnt >= 2)
    float w1;
    uint32_t pad;
};
static_assert(sizeof(Item) == 32, "Item must be 32 bytes");

// Views into the caller's workspace for one chunk of the stream.
struct Plan {
    uint64_t* keys_in;
    uint64_t* keys_out;
    uint32_t* vals_in;
    uint32_t* vals_out;
    int32_t* s_partner;   // [2*Ec] partner node of each contribution, sorted by (batch, target, side, edge)
    float* s_coef;        // [2*Ec] time weight w_e = exp(-lambda (t_last - t_e))  (models/TPNet.py:78)
    int32_t* s_target;    // [2*Ec]
    Item* light;          // [2*Ec] batch b owns [2*e0(b), 2*e0(b)+n_light)
    Item* heavy;          // [2*Ec]
    BatchDesc* desc;      // [nb]
    // edge-fused updates (plan_build with PLAN_FUSE): a target with exactly ONE contribution in its batch is not put on the
    // light list; the readout group of that edge's (src,dst) pair, which has both bundles in registers anyway, writes the
    // target's new bundle.  Per edge of the chunk: does the src / dst endpoint take its update this way, and the edge's
    // time weight.  The arrays live in keys_in, which is dead once the sort has run.
    uint8_t* fuse_src;    // [Ec]
    uint8_t* fuse_dst;    // [Ec]
    float* fuse_w;        // [Ec]
    void* sort_tmp;
    size_t sort_tmp_bytes;
    unsigned long long* dbg;  // first TPNET_DBG_BYTES of the workspace: in-kernel stamps of diagnostic builds (-DTPNET_STAMPS)
};
static constexpr size_t TPNET_DBG_BYTES = 1u << 20;   // (diagnostic builds: 512 KB of per-wave stamps + 512 KB of per-block stamps)

struct StreamArgs {
    const int64_t* src;
    const int64_t* dst;
    const int64_t* neg;
    const double* t;
    float* out_pos;
    float* out_neg;
    int32_t own_mod;   // row sharding: this rank computes targets / pairs (by their src node) with id % own_mod == own_rem
    int32_t own_rem;   // (own_mod = 1: everything; own_mod = 0: compact local table, the rank owns ids < own_rem)
    // the windowed pipeline of a row shard (wshard.hip): the edges whose src node this rank owns, window by window (ascending): the
    // readout role of window w walks own_list[own_start[w] .. own_start[w + 1]) instead of every edge of the window
    const uint32_t* own_list = nullptr;
    const uint32_t* own_start = nullptr;
};

// Launch geometry of the fast paths: LPP lanes cooperate on one row (one pair / one target), each lane owning VPL
// vectors of W floats per column chunk.
struct Geom {
    int lpp, vpl, w;
};
// `many` = a throughput-bound launch (a long list of independent pairs): rows of 17..32 vectors then take 16 lanes x 2
// vectors -- four units per wave, a reduction over 16 lanes: the standalone readout of 80 000 pairs at d=128 runs
// 1.36x faster that way, while the latency-bound per-batch step is 11 % slower with it (both measured).
inline Geom pick_geom(int d, bool many = false) {
    if (d % 4 != 0) return {64, 1, 1};
    const int nvec = d / 4;
    static const char* g0 = TPNET_DEV_STR(GEOM);
    if (g0 && g0[0] == '1' && g0[3] == '2' && nvec <= 32) return {16, 2, 4};   // developer override "16x2"
    if (many && nvec > 16 && nvec <= 32) return {16, 2, 4};
    // narrow rows (the column slices of a dim-sharded table, tpnet_amd/sharded.py): 4 / 8 lanes per row, so that a wave
    // carries 16 / 8 pairs instead of 4 with three quarters of its lanes idle
    if (nvec <= 4) return {4, 1, 4};
    if (nvec <= 8) return {8, 1, 4};
    if (nvec <= 16) return {16, 1, 4};
    if (nvec <= 32) return {32, 1, 4};
    static const char* g = TPNET_DEV_STR(GEOM);          // developer override: "64x1", "64x2", "32x2"
    if (g && g[0] == '6' && g[3] == '1' && nvec <= 64) return {64, 1, 4};
    if (g && g[0] == '6') return {64, nvec <= 64 ? 1 : 2, 4};
    if (g && g[0] == '3') return {32, 2, 4};
    // d <= 256: 32 lanes x 2 vectors, not 64 x 1 -- the 64-value Gram reduction over 64 lanes needs v_permlane32_swap
    // and costs 1 281 cycles against 545 over 32 lanes (tools/probes/reduce_probe.hip), and a wave then carries two
    // pairs / targets instead of one (measured: C4 +19 %, encoder readout at d=256 +18 %, C3 unchanged).
    if (nvec <= 64) return {32, 2, 4};
    return {64, 2, 4};                                        // d = 512 (and chunks of 512 beyond): C5 is 1.6x faster so
}

// host-side launchers implemented in step.hip / readout.hip / tables.hip / plan.hip
int launch_state_init(const tpnet_state& st, double t0, hipStream_t s);
int launch_import(const tpnet_state& st, const float* const* layers_dev, double now, hipStream_t s);
int launch_export(const tpnet_state& st, float* const* layers_dev, double now, double lambda, hipStream_t s);
int launch_decay(const tpnet_state& st, const float* factors_host, double t_new, hipStream_t s);
int launch_gather_elems(const tpnet_state& st, const int64_t* rows, const int64_t* cols, int64_t n, double now,
                        double lambda, float* out, hipStream_t s);
int launch_gram_finish(float* x, int64_t n, hipStream_t s);
int launch_gram_unpack(const float* packed, int64_t n, int L, uint32_t flags, float* out, hipStream_t s);
int launch_gather_rows(const tpnet_state& st, const int64_t* ids, int64_t n, double now, double lambda, float* out,
                       hipStream_t s);
int launch_pair_gram(const tpnet_state& st, const int64_t* u, const int64_t* v, int64_t n, double now, double lambda,
                     uint32_t flags, float* out, hipStream_t s);
int launch_pair_gram_shared(const tpnet_state& st, const int64_t* u, const int64_t* v1, const int64_t* v2, int64_t n,
                            double now, double lambda, uint32_t flags, float* out1, float* out2, hipStream_t s);
bool pair_gram_anchored_supported(const tpnet_state& st);
int launch_pair_gram_anchored(const tpnet_state& st, const int64_t* neigh, const int64_t* a1, const int64_t* a2,
                              int64_t n_rows, int K, double now, double lambda, uint32_t flags, float* out1, float* out2,
                              hipStream_t s);
// encoder_mfma.hip: the same readout on the matrix cores (split-bf16 operands, fp32 class) for rows of 64 / 128 floats, L = 3, K >= 4
bool encoder_mfma_supported(const tpnet_state& st, int64_t n_rows, int K);
int launch_encoder_gram_mfma(const tpnet_state& st, const int64_t* neigh, const int64_t* a1, const int64_t* a2, int64_t n_rows,
                             int K, double now, double lambda, uint32_t flags, float* out1, float* out2, hipStream_t s);
// the same + self.mlp in ONE launch (gram: optional copy of the pre-mlp features [2][n_rows * K][64]; out [2 * n_rows * K][64])
bool encoder_fused_supported(const tpnet_state& st, int64_t n_rows, int K, const tpnet_mlp* mlp);
int launch_encoder_fused(const tpnet_state& st, const int64_t* neigh, const int64_t* a1, const int64_t* a2, int64_t n_rows, int K,
                         double now, double lambda, uint32_t flags, const tpnet_mlp* mlp, float* gram, float* out, hipStream_t s);
// One launch: readout of batch b (if out_pos/out_neg) on the pre-batch state + update of batch b.
int launch_step(const tpnet_state& st, const StreamArgs& a, const Plan& p, int64_t b, int64_t batch, int32_t ne,
                double lambda, uint32_t launch_id, uint32_t flags, hipStream_t s);
int launch_decay_desc(const tpnet_state& st, const Plan& p, int64_t b, hipStream_t s);
int launch_pack_rows(const tpnet_state& st, const int64_t* ids, int64_t n, double now, double lambda, float* out,
                     hipStream_t s);
int launch_unpack_gathered(const tpnet_state& st, const int64_t* ids, int64_t n, double now, const float* recv,
                           int64_t maxc, const int64_t* offs, int G, int me, hipStream_t s);
int launch_unpack_rows(const tpnet_state& st, const int64_t* ids, int64_t n, double now, const float* in, hipStream_t s);
int launch_pack_bundles(const tpnet_state& st, const int64_t* ids, int64_t n, double now, double lambda, float* out,
                        hipStream_t s);
int launch_pack_split(const tpnet_state& st, const int64_t* ids, int64_t n, double now, double lambda, float* out_p0,
                      float* out_q, int64_t halo0, int64_t n_halo, hipStream_t s);
int launch_unpack_bundles(const tpnet_state& st, const int64_t* local_ids, int64_t n, double now, const float* recv,
                          int64_t maxc, const int64_t* offs, int G, hipStream_t s);

int launch_pair_feature(const tpnet_state& st, const int64_t* u, const int64_t* v, int64_t n, double now, double lambda,
                        uint32_t flags, const tpnet_mlp& m, float* out_gram, float* out, hipStream_t s);

// readout + mlp on the matrix cores (feature_mfma.hip): bf16 (opt-in class) or fp32 (f32 = true: w1 = f32 [256][64], w2 = the
// gathered f32 layout of tpnet_mlp::w2f)
bool pair_feature_mfma_supported(const tpnet_state& st);
// mlp_x3.hip: the fp32-class dense layers on existing feature rows, every wave its own 32-row tiles, split weights in LDS
bool mlp_x3_available();
int64_t mlp_x3_from();
int launch_mlp_rows_x3(const float* x, int64_t n, const float* w1f, const float* b1, const float* w2f, const float* b2, float* y,
                       hipStream_t s);
int launch_pair_feature_bf16(const tpnet_state& st, const int64_t* u, const int64_t* v, int64_t n, double now, double lambda,
                             uint32_t flags, const void* w1, const float* b1, const void* w2p, const float* b2,
                             float* out_gram, float* out, hipStream_t s, int mode, const float* feat_in = nullptr);
int mlp_f32_mode();

// the plan of ONE batch by one workgroup (plan.hip, k_plan_one): same Plan contents as plan_build for batch 0 of a chunk
static constexpr int64_t PLAN_ONE_MAX = 2048;
bool encoder_generic_readout(const tpnet_state& st);   // sampler.hip: the encoder's calls on rows of <= 128 floats
int64_t plan_one_max_batch();      // PLAN_ONE_MAX (0 with the developer override TPNET_DEV_NO_PLAN_ONE)
int plan_one(const tpnet_state& st, const Plan& p, const int64_t* src, const int64_t* dst, const double* t, int64_t B,
             double now_time, double lambda, uint32_t flags, hipStream_t s);

size_t plan_bytes(int64_t max_edges, int64_t batch);
int plan_carve(void* ws, size_t ws_bytes, int64_t Ec, int64_t batch, Plan* out);
int plan_build(const tpnet_state& st, const Plan& p, const int64_t* src, const int64_t* dst, const double* t,
               int64_t Ec, int64_t batch, double now_time, const double* t_prev_dev, double lambda, uint32_t flags,
               hipStream_t s);

// internal flag bits of launch_step (above the public TPNET_FLAG_* bits)
static constexpr uint32_t ROLE_READOUT = 1u << 16;
static constexpr uint32_t ROLE_UPDATE = 1u << 17;
static constexpr uint32_t STEP_FUSE = 1u << 18;     // launch_step: the plan was built with PLAN_FUSE and this launch has both roles
static constexpr uint32_t STEP_ITEMS_FIRST = 1u << 20;   // launch_step -> k_step: light items lead the work index space
static constexpr uint32_t STEP_HAS_POS = 1u << 21;   // launch_step -> k_step: out_pos / out_neg exist (decides the roles from
static constexpr uint32_t STEP_HAS_NEG = 1u << 22;   // preloaded kernel arguments alone)
static constexpr uint32_t PLAN_FUSE = 1u << 19;     // plan_build: divert single-contribution targets to the edge-fused path

// ---------------------------------------------------------------------------------------------------------------
// Windowed stream path (plan.hip: wplan_*, wstep.hip; DESIGN.md "windows").  A chunk of the stream is cut into windows
// of K batches, and the work of a window into L+1 ROLES: update of layer 1, ..., update of layer L, readout.  What
// makes a role a dependency-free launch: layer i of a node after batch b depends only on layer i-1 of its partners
// BEFORE batch b and on its own layer i after its previous batch, and layer 0 never changes.  Every (node, batch) run
// writes its result to its own slot of a version LOG that lives as long as the chunk, the table stays frozen at its
// pre-chunk state until a write-back pass at the end of the chunk, and the plan resolves every read "row of node n
// before batch b" to a log slot or to the frozen table.  So role i of window w only needs role i-1 of windows <= w and
// role i of windows < w -- and ONE launch carries {layer 1 of window j, layer 2 of window j-1, ..., readout of window
// j-L}: a software pipeline over the windows in which the long dependent chain of a hub node hides behind the
// bandwidth-bound readout of an older window.  The arithmetic per run is the reference's (models/TPNet.py:83-96) in a
// fixed association that does not depend on how the stream is cut into windows or chunks.
// ---------------------------------------------------------------------------------------------------------------
static constexpr int WIN_MAX_BATCHES = 64;          // batches per window (the workgroup-walked chains table their runs by batch)
static constexpr uint32_t WREF_TABLE = 1u << 31;    // the version to read is the table's (frozen) pre-chunk row; bit 0 = which copy
static constexpr uint32_t WREF_RUN_HEAD = 1u << 30; // first / last contribution of a (node, batch) run
static constexpr uint32_t WREF_RUN_TAIL = 1u << 29;
static constexpr uint32_t WREF_BLK_HEAD = 1u << 28; // first / last contribution of a block of WIN_BLOCK inside a run
static constexpr uint32_t WREF_BLK_TAIL = 1u << 27;
static constexpr uint32_t WREF_LAST_RUN = 1u << 26; // the run is its node's last one in the chunk: the write-back copies it to the table
static constexpr uint32_t WREF_SLOT_MASK = (1u << 26) - 1;   // bits 25..0: log slot (= chunk-relative sorted position of a run's tail)
static constexpr int WIN_MAX_WINDOWS = 64;          // windows per chunk of the three-launch planner (wplan3.hip: a 64-bit mask per node)
static constexpr int WIN_BLOCK = 8;                 // contributions summed on their own before they join the row (fixed
                                                    // association: the result of a run does not depend on who sums it)

static constexpr uint32_t WIN_MED_MIN = 16;         // contributions from which a chain gets a workgroup of its own (two rounds of
                                                    // WIN_BLOCK rows are what one lane group walks without a tail)
// length classes 0..wchain_med_classes(thr) hold exactly the chains of >= WIN_MED_MIN contributions (class k >= 1 of wchain_class
// holds the lengths of bit length bitlen(thr) - k + 1)
__host__ __device__ inline int wchain_med_classes(uint32_t thr) {
    const int c = (32 - __builtin_clz(thr | 1u)) - 4;
    return c < 0 ? 0 : (c > 6 ? 6 : c);       // (class 7 also holds everything shorter)
}
// length class of a chain inside its window's list (wplan3.hip): 0 = walked by a workgroup per column part (more than
// `thr` contributions), then classes of halving length -- a block's chains are alike, the long ones lead
__host__ __device__ inline int wchain_class(uint32_t cnt, uint32_t thr) {
    if (cnt > thr) return 0;
    const int dl = (32 - __builtin_clz(thr | 1u)) - (32 - __builtin_clz(cnt | 1u));
    return 1 + (dl > 6 ? 6 : (dl < 0 ? 0 : dl));
}

struct WinDesc {          // per window: its slice of the chain list (sorted by decreasing length: the plan's second sort)
    uint32_t start;       // first chain of the window in WPlan::chains
    uint32_t n_heavy;     // the first n_heavy are walked by a workgroup per column part ...
    uint32_t n_chains;    // ... the rest by one lane group each
    uint32_t n_ext;       // chains of length classes 0..2 (more than ~thr / 4 contributions): what a step with spare workgroups
                          // walks by workgroups (>= n_heavy; the sorted planner: = n_heavy)
    uint32_t n_ext1;      // chains of length classes 0..1 (more than ~thr / 2)
    uint32_t n_med;       // chains of at least WIN_MED_MIN contributions (>= n_heavy): beyond the heavy ones, each is walked by ONE
                          // workgroup, its blocks of WIN_BLOCK dealt to the lane groups (wstep.hip: chain_medium)
    uint32_t pad1, pad2;
};

struct Chain {            // all contributions of ONE node inside ONE window: sorted positions [j0, j0 + cnt)
    uint32_t j0;          // chunk-relative
    uint32_t cnt;
    int32_t target;
    uint32_t prev_ref;    // the node's row before the window: log slot of its previous run in the chunk, or WREF_TABLE | copy
    float g_first;        // prev_ref a table row: its pending decay to the first run's clock (layer 1; layer i: ^i)
    uint32_t pad0, pad1, pad2;
};
static_assert(sizeof(Chain) == 32, "Chain must be 32 bytes");

struct WPlan {
    Plan base;            // desc, sorted keys / payload, s_partner, s_coef, sort temp; light / heavy hold the Chain lists
    uint32_t* s_ref;      // [2*Ec] per sorted contribution: flags | log slot of the partner's version
    float* s_g;           // [2*Ec] decay of the partner's (log) version to the run's clock: exp(-lambda (T_b - T_version))
    float* s_dec;         // [2*Ec] at run heads: decay of the node's own previous run (in the chunk) to this run's clock
    uint16_t* s_bc;       // [2*Ec] batch (in the chunk) of the contribution
    WinDesc* wdesc;       // [nw]
    Chain* chains;        // [<= 2*Ec] every window's chains, window by window, longest first
    Chain* chains_sparse; // [2*Ec] plan scratch: the chain record at its head's sorted position
    uint32_t* lk_in;      // [2*Ec] x 4: keys / payload of the plan's second sort (chain heads by (window, length))
    uint32_t* lk_out;
    uint32_t* lv_in;
    uint32_t* lv_out;
    uint32_t* e_ref;      // [3][Ec] readout: version of src / dst / neg of every edge before its batch
    float* e_g;           // [3][Ec]
    float* log;           // [2*Ec][L][d] version log of the chunk (only the slots of run tails are ever touched)
    uint32_t* node_lo;    // [N] first / one-past-last sorted position of every node's contributions in the chunk (0, 0: none)
    uint32_t* node_hi;
    uint32_t* inv;        // [2*Ec] pre-sort index of a contribution -> its sorted position
    uint32_t* rhead;      // [2*Ec] sorted position -> first position of its (node, batch) run
    unsigned long long* wmask;   // [N] hashed planner (wplan3.hip): bit w = the node is a target in window w of the chunk
    uint32_t* wcls;       // [2][WIN_MAX_WINDOWS][8] chains per (window, length class); placement cursors; then [WIN_MAX_WINDOWS] position cursors
    void* wtab;           // the chain table, then the runs' per-batch segments: wplan3_table_bytes(Ec, batch) bytes in all
    size_t wzero_bytes;   // wmask, wcls and wtab are contiguous: what one fill has to zero before a plan
    uint32_t* wblk;       // per-workgroup counts / bases of the hashed planner (wplan3_blk_bytes)
    void* dense;          // the dense planner's matrices (wplan_dense.hip: wplan_dense_bytes; nullptr: not eligible)
    uint32_t heavy_thr;   // contributions per (node, window) above which a workgroup per column part walks the chain
    int32_t K;            // batches per window
    int64_t Ew;           // edges per full window = K * batch
};

size_t wplan_bytes(int64_t max_edges, int64_t batch, int64_t N, int d, int L);
int wplan_window_batches(int64_t batch, int d, int L);                 // 0 = the windowed path does not apply
int64_t wplan_max_chunk_edges(int64_t batch, int d, int L);            // edges one plan (and its log) may cover
int wplan_carve(void* ws, size_t ws_bytes, int64_t Ec, int64_t batch, int64_t N, int d, int L, int K, WPlan* out,
                float* shared_log = nullptr, bool shard = false);   // K <= wplan_window_batches (shard: chosen by the caller); shared_log: the version log lives outside the region
size_t wplan_bytes_shard(int64_t max_edges, int64_t batch, int64_t N, int d, int L);   // workspace of a row shard's chunk (wshard.hip)
size_t wplan_log_bytes(int64_t max_edges, int d, int L);       // the version log's share of wplan_bytes
int wplan_build(const tpnet_state& st, const WPlan& p, const int64_t* src, const int64_t* dst, const int64_t* neg,
                const double* t, int64_t Ec, int64_t batch, double now_time, const double* t_prev_dev, double lambda,
                bool want_readout, hipStream_t s);
// the same plan without a device-wide sort (wplan3.hip: one fill + three kernels) for batches that fit one workgroup's LDS and chunks of <= 64 windows;
// replay = the workspace still holds this plan of the SAME stream on the SAME table state: only the negatives' readout
// references are formed again
bool wplan3_applies(const tpnet_state& st, int64_t Ec, int64_t batch, int K);
size_t wplan3_table_bytes(int64_t Ec, int64_t batch);
size_t wplan3_blk_bytes(int64_t Ec, int64_t batch);
int wplan3_build(const tpnet_state& st, const WPlan& p, const int64_t* src, const int64_t* dst, const int64_t* neg,
                 const double* t, int64_t Ec, int64_t batch, double now_time, const double* t_prev_dev, double lambda,
                 bool want_readout, bool replay, hipStream_t s);
// the same plan in ONE launch through a dense (batch x node) matrix of run lengths (wplan_dense.hip), where the table is small
// against the stream (wplan_dense_eligible: every dataset of the reference); any number of windows up to 256
bool wplan_dense_eligible(int64_t N, int d, int L, int64_t batch);
size_t wplan_dense_bytes(int64_t Ec, int64_t batch, int64_t N, int d, int L);          // 0: not eligible
bool wplan_dense_applies(const tpnet_state& st, const WPlan& p, int64_t Ec, int64_t batch, int K);
// own (row shard, wshard.hip; -1: everything): the stream holds LOCAL ids, rows < own are this rank's, the rows behind them halo
// rows of other ranks' nodes -- only contributions to owned targets are planned (a halo node's run is ONE log slot, filled by the
// exchange), only owned nodes are written back; status (shard): [0] += batches whose owned contributions exceeded the sort
int wplan_dense_build(const tpnet_state& st, const WPlan& p, const int64_t* src, const int64_t* dst, const int64_t* neg,
                      const double* t, int64_t Ec, int64_t batch, double now_time, const double* t_prev_dev, double lambda,
                      bool want_readout, bool replay, hipStream_t s, int64_t own = -1, uint32_t* status = nullptr);
bool wplan_dense_writeback(const tpnet_state& st, const WPlan& p, int64_t Ec, int64_t batch, uint32_t launch_id, hipStream_t s,
                           int64_t own = -1);
bool wplan_dense_applies_shard(const tpnet_state& st, const WPlan& p, int64_t Ec, int64_t batch, int K);

uint32_t wplan_heavy_threshold(int K, int64_t batch, int d);
bool wplan_medium_chains(int d);          // rows that are exactly one chunk of their geometry: chain_medium serves (wstep.hip)
// pipeline step j of a chunk of nw windows: layer i of window j-i+1 (i = 1..L) and the readout of window j-L, whichever
// exist, in ONE launch; j = 0 .. nw+L-1.
int launch_wstep(const tpnet_state& st, const StreamArgs& a, const WPlan& p, int64_t j, int64_t Ec, int64_t batch,
                 double lambda, uint32_t flags, hipStream_t s);
// end of the chunk: every touched node's last version -> the table's other copy, meta published under launch_id
int launch_wwriteback(const tpnet_state& st, const WPlan& p, int64_t Ec, uint32_t launch_id, hipStream_t s);
// the same for a chunk the hashed planner planned, node by node (false: not served, take launch_wwriteback)
bool wplan3_writeback(const tpnet_state& st, const WPlan& p, int64_t Ec, int64_t batch, uint32_t launch_id, hipStream_t s);

extern thread_local int g_last_hip_error;
#define TPNET_HIP_TRY(expr)                                   \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) {                               \
            tpnet::g_last_hip_error = (int)_e;                \
            return TPNET_ERR_HIP;                             \
        }                                                     \
    } while (0)

}  // namespace tpnet
