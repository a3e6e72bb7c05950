// A ROW SHARD on the windowed pipeline (include/tpnet_hip.h: tpnet_wshard_*; new capability -- the reference is
// single-device, utils/load_configs.py:88).  Rank `me` of G owns the rows n with n % G == me (local row n / G of a compact table of
// n_owned rows + halo rows).  The per-batch shard (rows_rccl.hip) exchanges rows and launches a kernel per BATCH; here a chunk of the
// stream runs as the software pipeline of wstep.hip, one launch per WINDOW of batches, and the ranks meet once per launch:
//
//   relabel   every node the chunk touches gets ONE local id for the whole chunk: owned -> n / G, another rank's -> a halo row
//             (halo rows ordered by (owner, node): every owner's nodes are one contiguous range, in the order the owner lists them);
//   begin     the chunk's halo rows receive the owners' pre-chunk rows (all layers, as tpnet_pack_split packs them), in place;
//   plan      the dense planner (wplan_dense.hip) on the local ids, restricted to the contributions to OWNED targets; a halo node's
//             run in a batch is ONE slot of the version log.  Every rank plans from the whole stream, so each derives, without a
//             request round, which of its runs' results another rank reads (k_ws_needs: a partner row of one of the reader's targets
//             -> layers 1..L-1; a row of one of its readouts -> layers 1..L) and which remote runs it reads itself -- as two sorted
//             lists per (window, peer) that sender and receiver enumerate in the same order (node, then batch);
//   step j    k_wpipe (layer i of window j-i+1, readouts of window j-L; chains of owned nodes, pairs whose src is owned), then the
//             results of this launch that other ranks read: pack -> ONE grouped ncclSend / ncclRecv -> unpack into the log slots of
//             the halo nodes' runs.  A launch reads only what earlier launches (and their exchanges) wrote: the pipeline's invariant;
//   finish    the write-back of the owned nodes.
// The rows travel as they are (a log slot is copied bit for bit), so G shards compute exactly what one GPU computes on the windowed
// schedule -- bit for bit where the chunk's halo rows need no decay (a table just reset or imported), else up to the one extra f32
// rounding of tpnet_pack_split's decay, as on the per-batch shard.
#include "wplan_dense.hpp"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <dlfcn.h>
#include <cstring>
#include <memory>
#include <vector>

namespace tpnet {

// rows_rccl.hip's RCCL entry points (resolved there; this file only calls through them)
struct WsRccl {
    int (*send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
};
bool rows_rccl_api(WsRccl* out);          // rows_rccl.hip: false until a communicator has been created

static constexpr int WS_MAXG = 64;
static constexpr int kWsFloat32 = 7;      // rccl.h: ncclFloat32

struct WsOwners {                         // by value in kernel arguments: who owns a LOCAL id
    int32_t G, me, n_owned;
    int32_t hstart[WS_MAXG + 1];          // halo rows [n_owned + hstart[o], n_owned + hstart[o + 1]) hold owner o's nodes (o == me: empty)
};
__device__ __forceinline__ int ws_owner(const WsOwners& w, int64_t lid) {
    if (lid < w.n_owned) return w.me;
    const int32_t h = (int32_t)(lid - w.n_owned);
    int o = 0;
#pragma unroll 1
    for (int k = 1; k < w.G; ++k)
        if (h >= w.hstart[k]) o = k;
    return o;
}

// ---- relabel -------------------------------------------------------------------------------------------------------------
// index space q = owner * n_cap + n / G  (n % G == owner): every owner's nodes contiguous, ascending
__global__ __launch_bounds__(256) void k_ws_mark(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                                 const int64_t* __restrict__ neg, int64_t E, int64_t N, int32_t G, int64_t n_cap,
                                                 uint8_t* __restrict__ mark, uint32_t* __restrict__ status) {
    const int nk = neg ? 3 : 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nk * E; i += (int64_t)gridDim.x * blockDim.x) {
        const int kind = (int)(i / E);
        const int64_t e = i - (int64_t)kind * E;
        const int64_t n = kind == 0 ? src[e] : (kind == 1 ? dst[e] : neg[e]);
        if ((uint64_t)n >= (uint64_t)N) { atomicAdd(status, 1u); continue; }
        mark[(n % G) * n_cap + n / G] = 1;
    }
}

struct WsFlag {
    __device__ uint32_t operator()(uint8_t x) const { return x ? 1u : 0u; }
};

// cnt[o] = touched nodes of owner o; hstart; the owned touched rows (pack list, ascending)
__global__ __launch_bounds__(256) void k_ws_counts(const uint32_t* __restrict__ rank, const uint8_t* __restrict__ mark, int32_t G,
                                                   int32_t me, int64_t n_cap, int64_t* __restrict__ cnt_out /* [G + 1]: counts, then total */,
                                                   int64_t* __restrict__ pack_ids) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    if (gid <= G) {
        if (gid < G) cnt_out[gid] = (int64_t)(rank[(gid + 1) * n_cap] - rank[gid * n_cap]);
        else cnt_out[G] = (int64_t)rank[(int64_t)G * n_cap];
    }
    const uint32_t r0 = rank[(int64_t)me * n_cap];
    for (int64_t k = gid; k < n_cap; k += gsz)
        if (mark[(int64_t)me * n_cap + k]) pack_ids[rank[(int64_t)me * n_cap + k] - r0] = k;
}

__global__ __launch_bounds__(256) void k_ws_relabel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                                    const int64_t* __restrict__ neg, int64_t E, int64_t N, int32_t G, int32_t me,
                                                    int64_t n_cap, const uint32_t* __restrict__ rank, int64_t* __restrict__ lsrc,
                                                    int64_t* __restrict__ ldst, int64_t* __restrict__ lneg) {
    const int nk = neg ? 3 : 2;
    const uint32_t mine = rank[(int64_t)(me + 1) * n_cap] - rank[(int64_t)me * n_cap];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nk * E; i += (int64_t)gridDim.x * blockDim.x) {
        const int kind = (int)(i / E);
        const int64_t e = i - (int64_t)kind * E;
        const int64_t n = kind == 0 ? src[e] : (kind == 1 ? dst[e] : neg[e]);
        int64_t loc = n_cap;                                         // (an id out of range: the first halo row -- the plan call fails anyway)
        if ((uint64_t)n < (uint64_t)N) {
            const int32_t o = (int32_t)(n % G);
            if (o == me) loc = n / G;
            else loc = n_cap + (int64_t)(rank[(int64_t)o * n_cap + n / G] - (o > me ? mine : 0u));
        }
        (kind == 0 ? lsrc : (kind == 1 ? ldst : lneg))[e] = loc;
    }
}

// ---- the edges whose src node this rank owns, in order (the readout role of a window walks its slice) ----------------------------
struct WsOwnFlag {
    const int64_t* lsrc;
    int64_t n_owned;
    __device__ uint32_t operator()(int64_t e) const { return lsrc[e] < n_owned ? 1u : 0u; }
};
__global__ __launch_bounds__(256) void k_ws_ownlist(const int64_t* __restrict__ lsrc, int64_t n_owned, int64_t E, int64_t Ew, int64_t nw,
                                                    const uint32_t* __restrict__ rank /* [E + 1] exclusive scan of the flags */,
                                                    uint32_t* __restrict__ own_list, uint32_t* __restrict__ own_start) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = gid; e < E; e += gsz)
        if (lsrc[e] < n_owned) own_list[rank[e]] = (uint32_t)e;
    for (int64_t w = gid; w <= nw; w += gsz) own_start[w] = rank[(w * Ew < E) ? w * Ew : E];
}

// ---- what travels ----------------------------------------------------------------------------------------------------------
// item key: list (0 = this rank receives, 1 = it sends) | kind (0 = A: read as a partner row or by a readout, layers 1..L-1 travel;
// 1 = B: read by a readout, layer L travels too) | window | peer | log slot
static constexpr int WS_SLOT_BITS = 26, WS_PEER_BITS = 6, WS_WIN_BITS = 10;
__device__ __host__ __forceinline__ unsigned long long ws_key(int list, int kind, uint32_t w, uint32_t peer, uint32_t slot) {
    return ((unsigned long long)list << (1 + WS_WIN_BITS + WS_PEER_BITS + WS_SLOT_BITS)) |
           ((unsigned long long)kind << (WS_WIN_BITS + WS_PEER_BITS + WS_SLOT_BITS)) |
           ((unsigned long long)w << (WS_PEER_BITS + WS_SLOT_BITS)) | ((unsigned long long)peer << WS_SLOT_BITS) | slot;
}

struct WsNeed {
    const int64_t* gsrc;                  // the call's GLOBAL ids: owner(n) = n % G (a lookup of the owner of a halo row through the
    const int64_t* gdst;                  // by-value table of halo ranges was an indexed load from a stack copy per probe)
    const int64_t* gneg;
    const int64_t* lsrc;
    const int64_t* ldst;
    const int64_t* lneg;
    int64_t E, B;
    int32_t K, have_pos, have_neg, G;
    // marks, plain idempotent stores (no atomics: a hub's version is read by thousands of edges of a batch -- per-slot atomic masks
    // were ~48 000 same-address returning atomics per plan on the hub's owner at G = 8, 145 us of the plan):
    uint8_t* fa;                          // [2 E][G] slot x reader: another rank reads this rank's run (kind A)
    uint8_t* fb;                          // [2 E][G] ... by one of its readouts (kind B)
    uint8_t* wa;                          // [2 E] slot: this rank reads another rank's run: its owner + 1 (kind A)
    uint8_t* wb;                          // [2 E] ... by one of its readouts (kind B)
    uint16_t* ww;                         // [2 E] slot: the window of the run
    unsigned long long* keys;             // [6 E] the items (k_ws_items)
    uint32_t* counter;
};

// reader `r` reads node x (owner o) in its version before batch b (level 1: as a partner row of one of its targets; level 2: by one
// of its readouts)
__device__ __forceinline__ void ws_need(const WsNeed& a, const WsOwners& ow, const DView& D, bool live, int r, int o, int64_t x,
                                        int64_t b, int level) {
    if (!live || o == r || !(o == ow.me || r == ow.me)) return;
    const uint2 m = D.m[b * D.Ns + x];
    if (m.x == 0u) return;                                   // (the table's pre-chunk row: it came with the chunk's halo rows)
    const uint32_t slot = D.basef[x] + m.x - 1u;
    a.ww[slot] = (uint16_t)((m.y >> 16) / (uint32_t)a.K);
    if (r == ow.me) {                                        // this rank reads another rank's run
        a.wa[slot] = (uint8_t)(o + 1);
        if (level == 2) a.wb[slot] = (uint8_t)(o + 1);
    } else {                                                 // another rank reads this rank's run
        a.fa[(size_t)slot * a.G + r] = 1;
        if (level == 2) a.fb[(size_t)slot * a.G + r] = 1;
    }
}

__global__ __launch_bounds__(256) void k_ws_needs(WsNeed a, WsOwners ow, DView D) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < a.E; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = a.lsrc[e], d = a.ldst[e];
        const int64_t g = a.lneg ? a.lneg[e] : 0;
        const int64_t b = e / a.B;
        const int os = (int)((uint64_t)a.gsrc[e] % (uint32_t)ow.G), od = (int)((uint64_t)a.gdst[e] % (uint32_t)ow.G);
        const int og = a.gneg ? (int)((uint64_t)a.gneg[e] % (uint32_t)ow.G) : 0;
        ws_need(a, ow, D, true, os, od, d, b, 1);                          // target s, partner d   (models/TPNet.py:90-93)
        ws_need(a, ow, D, true, od, os, s, b, 1);                          // target d, partner s   (models/TPNet.py:94-96)
        ws_need(a, ow, D, a.have_pos != 0, os, od, d, b, 2);               // readout (s, d): the owner of s computes it
        ws_need(a, ow, D, a.have_neg && a.lneg, os, og, g, b, 2);          // readout (s, neg)
    }
}

// the marks -> items.  A thread takes 16 consecutive marks of both kinds (two 16-byte loads), a wave appends its items with ONE atomic
// on the counter (a thread per mark was one atomic per 64 marks: ~45 000 same-address atomics per plan at G = 8, 66 us)
__device__ __forceinline__ int ws_nonzero_bytes(uint4 v) {
    int n = 0;
    const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t x = q[k];
        x |= x >> 4; x |= x >> 2; x |= x >> 1;               // bit 0 of every byte = the byte is non-zero
        n += __popc(x & 0x01010101u);
    }
    return n;
}
__global__ __launch_bounds__(256) void k_ws_items(WsNeed a) {
    const int64_t nslot = 2 * a.E;
    const int64_t n_send = nslot * a.G;                      // marks of the send lists; behind them the nslot marks of the receive lists
    const int64_t chunks_s = (n_send + 15) / 16, chunks_r = (nslot + 15) / 16;
    const int64_t nchunk = chunks_s + chunks_r;
    const int64_t per = (int64_t)gridDim.x * blockDim.x;
    const int64_t rounds = (nchunk + per - 1) / per;
    const int lane = threadIdx.x & 63;
    for (int64_t it = 0; it < rounds; ++it) {                // (every lane of a wave runs every round: the wave scans together)
        const int64_t c = it * per + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        uint4 va = make_uint4(0u, 0u, 0u, 0u), vb = va;
        const bool send = c < chunks_s;
        const int64_t i0 = send ? c * 16 : (c - chunks_s) * 16;
        if (c < nchunk) {                                     // (the arrays are 256-byte aligned and padded: whole 16-byte loads)
            va = *reinterpret_cast<const uint4*>((send ? a.fa : a.wa) + i0);
            vb = *reinterpret_cast<const uint4*>((send ? a.fb : a.wb) + i0);
        }
        const int lim = (int)((send ? n_send : nslot) - i0 < 16 ? (send ? n_send : nslot) - i0 : 16);
        const uint8_t* ba = reinterpret_cast<const uint8_t*>(&va);
        const uint8_t* bb = reinterpret_cast<const uint8_t*>(&vb);
        int cnt = 0;
        if (c < nchunk) {
            if (lim == 16) cnt = ws_nonzero_bytes(va) + ws_nonzero_bytes(vb);
            else
                for (int k = 0; k < lim; ++k) cnt += (ba[k] ? 1 : 0) + (bb[k] ? 1 : 0);
        }
        int inc = cnt;
#pragma unroll
        for (int o2 = 1; o2 < 64; o2 <<= 1) {
            const int v = __shfl_up(inc, o2, 64);
            if (lane >= o2) inc += v;
        }
        const int total = __shfl(inc, 63, 64);
        if (total == 0) continue;
        uint32_t base = 0;
        if (lane == 63) base = atomicAdd(a.counter, (uint32_t)total);
        uint32_t at = (uint32_t)__shfl((int)base, 63, 64) + (uint32_t)(inc - cnt);
        if (cnt > 0) {
            for (int k = 0; k < lim; ++k) {
                const uint32_t fa = ba[k], fb = bb[k];
                if (!(fa | fb)) continue;
                const int64_t i = i0 + k;
                const uint32_t slot = send ? (uint32_t)(i / a.G) : (uint32_t)i;
                const uint32_t w = a.ww[slot];
                const int list = send ? 1 : 0;
                if (fa) a.keys[at++] = ws_key(list, 0, w, send ? (uint32_t)(i % a.G) : fa - 1u, slot);
                if (fb) a.keys[at++] = ws_key(list, 1, w, send ? (uint32_t)(i % a.G) : fb - 1u, slot);
            }
        }
    }
}

__device__ __forceinline__ uint32_t ws_lower(const unsigned long long* __restrict__ u, uint32_t lo, uint32_t hi, unsigned long long key) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (u[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// off[((list * 2 + kind) * nw + w) * G + peer] = first sorted item of that list; off[last] = the number of items
__global__ __launch_bounds__(256) void k_ws_offsets(const unsigned long long* __restrict__ sorted, uint32_t cap, int64_t nw, int32_t G,
                                                    uint32_t* __restrict__ off) {
    const int64_t nq = 4 * nw * G;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q <= nq; q += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long key;
        if (q == nq) key = ws_key(1, 1, (1u << WS_WIN_BITS) - 1u, (1u << WS_PEER_BITS) - 1u, (1u << WS_SLOT_BITS) - 1u) + 1ull;
        else {
            const int lk = (int)(q / (nw * G));
            const int64_t r = q - (int64_t)lk * nw * G;
            key = ws_key(lk >> 1, lk & 1, (uint32_t)(r / G), (uint32_t)(r % G), 0u);
        }
        off[q] = ws_lower(sorted, 0u, cap, key);
    }
}

// the entries of every step: item k of list (list, kind, w, peer) travels at step j = w + i - 1 for its layers i (kind A: 1..L-1,
// kind B: L), in the order (peer, layer, rank in its list); ent[list][stepbase[list][j] + ...] = slot | (layer - 1) << 26
struct WsEnt {
    const unsigned long long* sorted;
    const uint32_t* off;
    const uint32_t* pbase;                // [2][nsteps][G]: first entry of (list, step, peer) in the list's entry array
    uint32_t* ent[2];
    int64_t nw;
    int32_t G, L, nsteps;
};
__device__ __forceinline__ uint32_t ws_cnt(const WsEnt& a, int list, int kind, int64_t w, int peer) {
    if (w < 0 || w >= a.nw) return 0u;
    const int64_t q = ((int64_t)(list * 2 + kind) * a.nw + w) * a.G + peer;
    return a.off[q + 1] - a.off[q];
}
__global__ __launch_bounds__(256) void k_ws_entries(WsEnt a) {
    const uint32_t n_items = a.off[4 * a.nw * a.G];
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_items; k += gridDim.x * blockDim.x) {
        const unsigned long long key = a.sorted[k];
        const uint32_t slot = (uint32_t)(key & ((1ull << WS_SLOT_BITS) - 1ull));
        const int peer = (int)((key >> WS_SLOT_BITS) & ((1u << WS_PEER_BITS) - 1u));
        const int64_t w = (int64_t)((key >> (WS_SLOT_BITS + WS_PEER_BITS)) & ((1u << WS_WIN_BITS) - 1u));
        const int kind = (int)((key >> (WS_SLOT_BITS + WS_PEER_BITS + WS_WIN_BITS)) & 1u);
        const int list = (int)((key >> (WS_SLOT_BITS + WS_PEER_BITS + WS_WIN_BITS + 1)) & 1u);
        const int64_t q = ((int64_t)(list * 2 + kind) * a.nw + w) * a.G + peer;
        const uint32_t rank = k - a.off[q];
        const int i0 = kind ? a.L : 1, i1 = kind ? a.L : a.L - 1;
        for (int i = i0; i <= i1; ++i) {
            const int64_t j = w + i - 1;
            uint32_t at = a.pbase[((int64_t)list * a.nsteps + j) * a.G + peer];
            for (int ip = 1; ip < i; ++ip) at += ws_cnt(a, list, ip == a.L ? 1 : 0, j - ip + 1, peer);
            a.ent[list][at + rank] = slot | ((uint32_t)(i - 1) << WS_SLOT_BITS);
        }
    }
}

// rows of the log <-> a contiguous buffer, in entry order (16 lanes x 16 bytes per row pass)
template <bool PACK>
__global__ __launch_bounds__(256) void k_ws_copy(const uint32_t* __restrict__ ent, uint32_t n, float* __restrict__ log, float* __restrict__ buf,
                                                 int L, int d) {
    const int dv = d / 4;
    const int lanes = 16;
    const int g = threadIdx.x / lanes, gl = threadIdx.x % lanes;
    for (uint32_t k = blockIdx.x * (256 / lanes) + g; k < n; k += gridDim.x * (256 / lanes)) {
        const uint32_t en = ent[k];
        const uint32_t slot = en & ((1u << WS_SLOT_BITS) - 1u), layer = en >> WS_SLOT_BITS;
        float4* lrow = reinterpret_cast<float4*>(log + ((int64_t)slot * L + layer) * (int64_t)d);
        float4* brow = reinterpret_cast<float4*>(buf + (int64_t)k * d);
        for (int x = gl; x < dv; x += lanes) {
            if (PACK) brow[x] = lrow[x];
            else lrow[x] = brow[x];
        }
    }
}

}  // namespace tpnet

using namespace tpnet;

// the plan of one chunk on one rank (host object: sizes, counts and views into the CALLER's workspace; no device memory of its own)
struct tpnet_wshard {
    tpnet_state st;                       // the local table
    WPlan p;
    StreamArgs a;                         // local ids, ownership
    int64_t E, batch, N_global;
    int32_t G, me, n_owned, K;
    int64_t nw, nsteps;
    double lambda;
    uint32_t flags;
    bool have_readout;
    std::vector<int64_t> chunk_cnt;       // [G] touched nodes per owner (chunk_cnt[me]: the rows this rank packs for everybody)
    std::vector<int32_t> hstart;          // [G + 1]
    std::vector<int64_t> send_cnt, recv_cnt;       // [nsteps][G] rows per step and peer
    std::vector<int64_t> send_base, recv_base;     // [nsteps] first entry of the step
    int64_t max_send = 0, max_recv = 0;   // rows of the largest step (what the exchange buffers must hold)
    int64_t* pack_ids = nullptr;          // device: the owned touched rows
    uint32_t* ent[2] = {nullptr, nullptr};         // device: [0] receive entries, [1] send entries
    float *send_p0 = nullptr, *send_q = nullptr, *sendbuf = nullptr, *recvbuf = nullptr;   // the caller's exchange buffers
    double now_time;
};

static size_t ws_al(size_t x) { return (x + 255) / 256 * 256; }
static size_t ws_sort_tmp(size_t cap, size_t nq) {
    size_t b1 = 0, b2 = 0;
    unsigned long long* k = nullptr;
    (void)rocprim::radix_sort_keys(nullptr, b1, k, k, cap, 0u, 64u, (hipStream_t)0, false);
    uint8_t* m = nullptr;
    uint32_t* r = nullptr;
    (void)rocprim::exclusive_scan(nullptr, b2, rocprim::make_transform_iterator(m, WsFlag()), r, 0u, nq, rocprim::plus<uint32_t>(),
                                  (hipStream_t)0, false);
    return ws_al(b1 > b2 ? b1 : b2);
}
static size_t ws_scratch_bytes(int64_t n_cap, int32_t G, int64_t E, int64_t nb, int L) {
    const size_t nq = (size_t)G * (size_t)n_cap + 1;
    const size_t cap = 6 * (size_t)E;
    const size_t nsteps = (size_t)nb + (size_t)L + 1;
    return ws_al(nq) + ws_al(nq * 4) + 3 * ws_al((size_t)E * 8) + ws_al((size_t)n_cap * 8) + 2 * ws_al(2 * (size_t)E * G) +
           2 * ws_al(2 * (size_t)E) + ws_al(4 * (size_t)E) + 2 * ws_al(cap * 8) + ws_al((4 * (size_t)nb * G + 1) * 4) + ws_al(2 * nsteps * G * 4) +
           2 * ws_al(cap * 3 * 4) + ws_al((size_t)(G + 2) * 8) + ws_al(16) + ws_sort_tmp(cap, nq > (size_t)E + 1 ? nq : (size_t)E + 1) +
           2 * ws_al(((size_t)E + 1) * 4) + ws_al(((size_t)nb + 2) * 4) + 1024;
}

extern "C" {

size_t tpnet_wshard_workspace_bytes(int64_t n_local, int32_t d, int32_t L, int64_t E, int64_t batch, int32_t G, int32_t n_owned) {
    if (n_local < 1 || d < 1 || L < 1 || L > TPNET_MAX_LAYERS || E < 1 || batch < 1 || G < 1 || G > WS_MAXG || n_owned < 1) return 0;
    const int64_t nb = (E + batch - 1) / batch;
    return ws_al(wplan_bytes_shard(E, batch, n_local, d, L)) + ws_scratch_bytes(n_owned, G, E, nb, L) + 512;
}

void tpnet_wshard_destroy(tpnet_wshard* w) { delete w; }

// returns TPNET_OK and *out = the plan; or 1 = the windowed shard does not serve this call (the caller takes the per-batch shard);
// or a negative tpnet_status.  Synchronises `stream` three times (the chunk's halo counts; the number of items; the exchange lists' counts).
int tpnet_wshard_plan(const tpnet_state* st, const int64_t* src, const int64_t* dst, const int64_t* neg, const double* t, int64_t E,
                      int64_t batch, int64_t N_global, int32_t G, int32_t me, int32_t n_owned, double now_time, double lambda,
                      uint32_t flags, int32_t want_pos, int32_t want_neg, void* workspace, size_t ws_bytes, void* stream,
                      tpnet_wshard** out) {
    if (!st || !st->p0 || !st->q || !st->meta || !st->err || !src || !dst || !t || !out || !workspace) return TPNET_ERR_BAD_ARG;
    if (E < 1 || batch < 1 || G < 1 || G > WS_MAXG || me < 0 || me >= G || n_owned < 1 || n_owned >= st->N) return TPNET_ERR_BAD_ARG;
    if (N_global < 1 || (N_global + G - 1) / G != n_owned) return TPNET_ERR_BAD_ARG;
    *out = nullptr;
    const int d = st->d, L = st->L;
    if (flags & (TPNET_FLAG_EAGER_DECAY | TPNET_FLAG_SEQUENTIAL | TPNET_FLAG_PACKED)) return 1;
    if (d % 4 != 0 || 2 * batch > 16 * 1024) return 1;
    const int64_t nb = (E + batch - 1) / batch;
    if (nb < 4 || E >= ((int64_t)1 << 25)) return 1;                      // (slots are 26-bit positions < 2 E)
    if (ws_bytes < tpnet_wshard_workspace_bytes(st->N, d, L, E, batch, G, n_owned)) return TPNET_ERR_WORKSPACE;
    // batches per window: ~24 K edges of THIS rank's share per launch (wplan_window_batches: of a GPU's own batches), equal windows
    int64_t Kmax = 24576 * (int64_t)G / batch;
    if (Kmax > WIN_MAX_BATCHES) Kmax = WIN_MAX_BATCHES;
    if (Kmax < 2) return 1;
    int K = 2;
    while (K < Kmax && (int64_t)K * K < 5 * nb) ++K;                      // ceil(sqrt(5 nb)), as api.hip's window_batches_for
    const int64_t nw0 = (nb + K - 1) / K;
    K = (int)((nb + nw0 - 1) / nw0);
    const int64_t nw = (nb + K - 1) / K;
    if (nw >= (1 << WS_WIN_BITS) || nw > 256) return 1;
    hipStream_t s = (hipStream_t)stream;
    std::unique_ptr<tpnet_wshard> holder(new tpnet_wshard());       // (every early return below frees it)
    tpnet_wshard* w = holder.get();
    w->st = *st; w->E = E; w->batch = batch; w->N_global = N_global; w->G = G; w->me = me; w->n_owned = n_owned; w->K = K;
    w->nw = nw; w->lambda = lambda; w->flags = flags; w->now_time = now_time;
    w->have_readout = want_pos || want_neg;
    w->nsteps = nw + (w->have_readout ? L : L - 1);
    auto fail = [&](int rc) { return rc; };
    const size_t plan_b = ws_al(wplan_bytes_shard(E, batch, st->N, d, L));
    char* base = reinterpret_cast<char*>((reinterpret_cast<size_t>(workspace) + 255) / 256 * 256);
    int rc = wplan_carve(base, plan_b, E, batch, st->N, d, L, K, &w->p, nullptr, true);
    if (rc) return fail(rc == TPNET_ERR_BAD_ARG ? 1 : rc);
    if (!wplan_dense_applies_shard(*st, w->p, E, batch, K)) return fail(1);
    char* c = base + plan_b;
    auto take = [&](size_t bytes) { void* r = c; c += ws_al(bytes); return r; };
    const int64_t n_cap = n_owned;
    const size_t nq = (size_t)G * (size_t)n_cap + 1;
    const size_t cap = 6 * (size_t)E;
    uint8_t* mark = (uint8_t*)take(nq);
    uint32_t* rank = (uint32_t*)take(nq * 4);
    int64_t* lsrc = (int64_t*)take((size_t)E * 8);
    int64_t* ldst = (int64_t*)take((size_t)E * 8);
    int64_t* lneg = (int64_t*)take((size_t)E * 8);
    w->pack_ids = (int64_t*)take((size_t)n_cap * 8);
    uint8_t* fa = (uint8_t*)take(2 * (size_t)E * G);
    uint8_t* fb = (uint8_t*)take(2 * (size_t)E * G);
    uint8_t* wa = (uint8_t*)take(2 * (size_t)E);
    uint8_t* wb = (uint8_t*)take(2 * (size_t)E);
    uint16_t* ww = (uint16_t*)take(4 * (size_t)E);
    unsigned long long* keys = (unsigned long long*)take(cap * 8);
    unsigned long long* sorted = (unsigned long long*)take(cap * 8);
    uint32_t* off = (uint32_t*)take((4 * (size_t)nw * G + 1) * 4);
    uint32_t* pbase = (uint32_t*)take(2 * (size_t)w->nsteps * G * 4);
    w->ent[0] = (uint32_t*)take(cap * 3 * 4);
    w->ent[1] = (uint32_t*)take(cap * 3 * 4);
    int64_t* cnt_dev = (int64_t*)take((size_t)(G + 2) * 8);
    uint32_t* status = (uint32_t*)take(16);
    uint32_t* own_list = (uint32_t*)take(((size_t)E + 1) * 4);
    uint32_t* erank = (uint32_t*)take(((size_t)E + 1) * 4);
    uint32_t* own_start = (uint32_t*)take(((size_t)nw + 2) * 4);
    void* tmp = c;
    size_t tmp_bytes = ws_sort_tmp(cap, nq > (size_t)E + 1 ? nq : (size_t)E + 1);
    if ((size_t)(c + tmp_bytes - reinterpret_cast<char*>(workspace)) > ws_bytes) return fail(TPNET_ERR_WORKSPACE);

    // ---- relabel: the chunk's touched nodes, one local id each
    TPNET_HIP_TRY(hipMemsetAsync(mark, 0, nq, s));
    TPNET_HIP_TRY(hipMemsetAsync(status, 0, 16, s));
    const int64_t items = (neg ? 3 : 2) * E;
    int grid = (int)((items + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_ws_mark, dim3(grid), dim3(256), 0, s, src, dst, neg, E, N_global, G, n_cap, mark, status);
    {
        size_t tb = tmp_bytes;
        if (rocprim::exclusive_scan(tmp, tb, rocprim::make_transform_iterator(mark, WsFlag()), rank, 0u, nq, rocprim::plus<uint32_t>(), s,
                                    false) != hipSuccess)
            return fail(TPNET_ERR_HIP);
    }
    hipLaunchKernelGGL(k_ws_counts, dim3((unsigned)((n_cap + 255) / 256)), dim3(256), 0, s, rank, mark, G, me, n_cap, cnt_dev, w->pack_ids);
    hipLaunchKernelGGL(k_ws_relabel, dim3(grid), dim3(256), 0, s, src, dst, neg, E, N_global, G, me, n_cap, rank, lsrc, ldst,
                       neg ? lneg : nullptr);
    std::vector<int64_t> hc((size_t)G + 1);
    uint32_t hstat[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(hc.data(), cnt_dev, (size_t)(G + 1) * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(hstat, status, 16, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return fail(TPNET_ERR_HIP);
    if (hstat[0]) return fail(TPNET_ERR_INDEX);
    w->chunk_cnt.assign(hc.begin(), hc.begin() + G);
    w->hstart.assign((size_t)G + 1, 0);
    for (int o = 0; o < G; ++o) w->hstart[o + 1] = w->hstart[o] + (o == me ? 0 : (int32_t)hc[o]);
    if ((int64_t)n_owned + w->hstart[G] > st->N) return fail(1);              // more remote nodes in the chunk than halo rows: per batch

    // the halo rows in use are "as of now" from here on (the planner reads their records; begin() fills them with the owners' rows
    // decayed to this clock)
    rc = launch_pack_split(*st, nullptr, 0, now_time, lambda, nullptr, nullptr, n_owned, w->hstart[G], s);
    if (rc) return fail(rc);
    // ---- the plan of the pipeline on the local ids, owned targets only
    rc = wplan_dense_build(*st, w->p, lsrc, ldst, neg ? lneg : nullptr, t, E, batch, now_time, nullptr, lambda, w->have_readout, false, s,
                           n_owned, status + 1);
    if (rc) return fail(rc);
    StreamArgs& a = w->a;
    a.src = lsrc; a.dst = ldst; a.neg = neg ? lneg : nullptr; a.t = t;
    a.out_pos = nullptr; a.out_neg = nullptr;
    a.own_mod = 0; a.own_rem = n_owned;
    {
        // the rank's own edges (src node owned), in order, and every window's slice of them
        size_t tb = tmp_bytes;
        auto flags_it = rocprim::make_transform_iterator(rocprim::counting_iterator<int64_t>(0), WsOwnFlag{lsrc, (int64_t)n_owned});
        if (rocprim::exclusive_scan(tmp, tb, flags_it, erank, 0u, (size_t)E + 1, rocprim::plus<uint32_t>(), s, false) != hipSuccess)
            return fail(TPNET_ERR_HIP);
        int g1 = (int)((E + 255) / 256);
        if (g1 > 4096) g1 = 4096;
        hipLaunchKernelGGL(k_ws_ownlist, dim3(g1), dim3(256), 0, s, lsrc, (int64_t)n_owned, E, w->p.Ew, nw, erank, own_list, own_start);
        a.own_list = own_list;
        a.own_start = own_start;
    }

    // ---- what travels after every step: items, sorted; counts per (list, kind, window, peer)
    // (fa .. wb are contiguous in the workspace: one fill)
    TPNET_HIP_TRY(hipMemsetAsync(fa, 0, (size_t)(reinterpret_cast<char*>(ww) - reinterpret_cast<char*>(fa)), s));
    TPNET_HIP_TRY(hipMemsetAsync(status + 2, 0, 4, s));
    WsOwners ow{};
    ow.G = G; ow.me = me; ow.n_owned = n_owned;
    for (int o = 0; o <= G; ++o) ow.hstart[o] = w->hstart[o];
    const DView D = dview_of(w->p, E, batch, st->N);
    if (G > 1) {
        WsNeed nd{};
        nd.gsrc = src; nd.gdst = dst; nd.gneg = neg;
        nd.lsrc = lsrc; nd.ldst = ldst; nd.lneg = neg ? lneg : nullptr;
            nd.E = E; nd.B = batch; nd.K = K; nd.have_pos = want_pos ? 1 : 0; nd.have_neg = want_neg ? 1 : 0; nd.G = G;
        nd.fa = fa; nd.fb = fb; nd.wa = wa; nd.wb = wb; nd.ww = ww; nd.keys = keys; nd.counter = status + 2;
        int g2 = (int)((E + 255) / 256);
        if (g2 > 4096) g2 = 4096;
        hipLaunchKernelGGL(k_ws_needs, dim3(g2), dim3(256), 0, s, nd, ow, D);
        int64_t gi = ((2 * E * (G + 1) + 31) / 16 + 255) / 256;
        if (gi > 8192) gi = 8192;
        hipLaunchKernelGGL(k_ws_items, dim3((unsigned)gi), dim3(256), 0, s, nd);
        // how many items there are sizes the sort (a sort of the 6 E slots the array has room for was 8 x the work at G = 8)
        if (hipMemcpyAsync(hstat, status, 16, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
            return fail(TPNET_ERR_HIP);
        if (hstat[1]) return fail(1);                                // a batch's owned contributions exceeded the sort: per batch
        const uint32_t n_items = hstat[2];
        if (n_items > cap) return fail(TPNET_ERR_WORKSPACE);
        if (n_items > 0) {
            size_t tb = tmp_bytes;
            if (rocprim::radix_sort_keys(tmp, tb, keys, sorted, (size_t)n_items, 0u, (unsigned)(2 + WS_WIN_BITS + WS_PEER_BITS + WS_SLOT_BITS), s,
                                         false) != hipSuccess)
                return fail(TPNET_ERR_HIP);
        }
        hipLaunchKernelGGL(k_ws_offsets, dim3((unsigned)((4 * nw * G + 256) / 256)), dim3(256), 0, s, sorted, n_items, nw, G, off);
    } else {
        TPNET_HIP_TRY(hipMemsetAsync(off, 0, (4 * (size_t)nw * G + 1) * 4, s));
    }
    std::vector<uint32_t> hoff(4 * (size_t)nw * G + 1);
    if (hipMemcpyAsync(hoff.data(), off, hoff.size() * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(hstat, status, 16, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return fail(TPNET_ERR_HIP);
    if (hstat[1]) return fail(1);                                    // a batch's owned contributions exceeded the sort: per batch
    auto cnt_of = [&](int list, int kind, int64_t ww, int peer) -> int64_t {
        if (ww < 0 || ww >= nw) return 0;
        const size_t q = ((size_t)(list * 2 + kind) * (size_t)nw + (size_t)ww) * (size_t)G + (size_t)peer;
        return (int64_t)hoff[q + 1] - (int64_t)hoff[q];
    };
    const int64_t ns = w->nsteps;
    w->send_cnt.assign((size_t)ns * G, 0);
    w->recv_cnt.assign((size_t)ns * G, 0);
    w->send_base.assign((size_t)ns + 1, 0);
    w->recv_base.assign((size_t)ns + 1, 0);
    std::vector<uint32_t> hpb(2 * (size_t)ns * G, 0u);
    for (int list = 0; list < 2; ++list) {
        int64_t run = 0;
        for (int64_t j = 0; j < ns; ++j) {
            (list ? w->send_base : w->recv_base)[j] = run;
            int64_t step_rows = 0;
            for (int r = 0; r < G; ++r) {
                int64_t n = 0;
                for (int i = 1; i <= L; ++i) n += cnt_of(list, i == L ? 1 : 0, j - i + 1, r);
                (list ? w->send_cnt : w->recv_cnt)[(size_t)j * G + r] = n;
                hpb[((size_t)list * ns + j) * G + r] = (uint32_t)run;
                run += n;
                step_rows += n;
            }
            if (list) w->max_send = step_rows > w->max_send ? step_rows : w->max_send;
            else w->max_recv = step_rows > w->max_recv ? step_rows : w->max_recv;
        }
        (list ? w->send_base : w->recv_base)[ns] = run;
        if ((size_t)run > cap * 3) return fail(TPNET_ERR_WORKSPACE);
    }
    if (G > 1 && hoff.back() > 0) {
        TPNET_HIP_TRY(hipMemcpyAsync(pbase, hpb.data(), hpb.size() * 4, hipMemcpyHostToDevice, s));
        WsEnt ea{};
        ea.sorted = sorted; ea.off = off; ea.pbase = pbase; ea.ent[0] = w->ent[0]; ea.ent[1] = w->ent[1];
        ea.nw = nw; ea.G = G; ea.L = L; ea.nsteps = (int32_t)ns;
        int g3 = (int)((hoff.back() + 255) / 256);
        if (g3 > 4096) g3 = 4096;
        hipLaunchKernelGGL(k_ws_entries, dim3(g3), dim3(256), 0, s, ea);
        TPNET_HIP_TRY(hipStreamSynchronize(s));                      // (hpb leaves scope)
    }
    TPNET_HIP_TRY(hipGetLastError());
    *out = holder.release();
    return TPNET_OK;
}

/* sizes the caller needs: steps of the pipeline, halo rows in use, rows of the largest step's messages, the chunk's row counts per
 * owner [G], rows per (step, peer) [nsteps][G] */
int tpnet_wshard_info(const tpnet_wshard* w, int64_t* n_steps, int64_t* halo_rows, int64_t* max_send_rows, int64_t* max_recv_rows,
                      const int64_t** chunk_cnt, const int64_t** send_cnt, const int64_t** recv_cnt) {
    if (!w) return TPNET_ERR_BAD_ARG;
    if (n_steps) *n_steps = w->nsteps;
    if (halo_rows) *halo_rows = w->hstart[w->G];
    if (max_send_rows) *max_send_rows = w->max_send;
    if (max_recv_rows) *max_recv_rows = w->max_recv;
    if (chunk_cnt) *chunk_cnt = w->chunk_cnt.data();
    if (send_cnt) *send_cnt = w->send_cnt.data();
    if (recv_cnt) *recv_cnt = w->recv_cnt.data();
    return TPNET_OK;
}

/* the caller's exchange buffers (device): send_p0 [chunk_cnt[me]][d], send_q [chunk_cnt[me]][L d] for the chunk's halo rows;
 * sendbuf [max_send_rows][d], recvbuf [max_recv_rows][d] for the steps */
int tpnet_wshard_set_buffers(tpnet_wshard* w, float* send_p0, float* send_q, float* sendbuf, float* recvbuf) {
    if (!w) return TPNET_ERR_BAD_ARG;
    w->send_p0 = send_p0; w->send_q = send_q; w->sendbuf = sendbuf; w->recvbuf = recvbuf;
    return TPNET_OK;
}

/* phases (bit mask) of tpnet_wshard_begin / tpnet_wshard_step: a caller with its own transport runs pack, moves the rows, runs unpack */
#define WS_PH_LAUNCH 1
#define WS_PH_PACK 2
#define WS_PH_EXCHANGE 4
#define WS_PH_UNPACK 8

/* the chunk's halo rows: PACK = this rank's touched rows into send_p0 / send_q (decayed to the chunk's clock; the halo rows in use
 * stamped "as of now"), EXCHANGE = every peer receives them straight into its halo rows of p0 and of copy 0 of q (comm) */
int tpnet_wshard_begin(tpnet_wshard* w, void* comm, uint32_t phases, void* stream) {
    if (!w) return TPNET_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int G = w->G, me = w->me;
    const int64_t mine = w->chunk_cnt[me], halo = w->hstart[G];
    if (G == 1) return TPNET_OK;
    if (phases & WS_PH_PACK) {
        if (mine > 0 && (!w->send_p0 || !w->send_q)) return TPNET_ERR_BAD_ARG;
        int rc = launch_pack_split(w->st, w->pack_ids, mine, w->now_time, w->lambda, w->send_p0, w->send_q, w->n_owned, halo, s);
        if (rc) return rc;
    }
    if (phases & WS_PH_EXCHANGE) {
        WsRccl api;
        if (!comm || !rows_rccl_api(&api)) return TPNET_ERR_BAD_ARG;
        const size_t d = (size_t)w->st.d, Ld = (size_t)w->st.L * d;
        float* halo_p0 = w->st.p0 + (size_t)w->n_owned * d;
        float* halo_q = w->st.q + (size_t)w->n_owned * Ld;
        int bad = api.group_start();
        for (int r = 0; r < G; ++r) {
            if (r == me) continue;
            if (mine) {
                bad |= api.send(w->send_p0, (size_t)mine * d, kWsFloat32, r, comm, s);
                bad |= api.send(w->send_q, (size_t)mine * Ld, kWsFloat32, r, comm, s);
            }
            if (w->chunk_cnt[r]) {
                bad |= api.recv(halo_p0 + (size_t)w->hstart[r] * d, (size_t)w->chunk_cnt[r] * d, kWsFloat32, r, comm, s);
                bad |= api.recv(halo_q + (size_t)w->hstart[r] * Ld, (size_t)w->chunk_cnt[r] * Ld, kWsFloat32, r, comm, s);
            }
        }
        bad |= api.group_end();
        if (bad) return TPNET_ERR_HIP;
    }
    return TPNET_OK;
}

/* pipeline step j: LAUNCH = k_wpipe; PACK = this launch's results that other ranks read -> sendbuf (peer by peer); EXCHANGE = ONE
 * grouped ncclSend / ncclRecv (comm); UNPACK = recvbuf -> the log slots of the halo nodes' runs */
int tpnet_wshard_step(tpnet_wshard* w, void* comm, int64_t j, uint32_t phases, float* out_pos, float* out_neg, void* stream) {
    if (!w || j < 0 || j >= w->nsteps) return TPNET_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int G = w->G, me = w->me, d = w->st.d, L = w->st.L;
    if (phases & WS_PH_LAUNCH) {
        StreamArgs a = w->a;
        a.out_pos = out_pos;
        a.out_neg = out_neg;
        if ((out_pos || out_neg) != w->have_readout) return TPNET_ERR_BAD_ARG;
        int rc = launch_wstep(w->st, a, w->p, j, w->E, w->batch, w->lambda, w->flags, s);
        if (rc) return rc;
    }
    if (G == 1) return TPNET_OK;
    const int64_t ns = w->send_base[j + 1] - w->send_base[j], nr = w->recv_base[j + 1] - w->recv_base[j];
    if ((phases & WS_PH_PACK) && ns > 0) {
        if (!w->sendbuf) return TPNET_ERR_BAD_ARG;
        int g = (int)((ns + 15) / 16);
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL((k_ws_copy<true>), dim3(g), dim3(256), 0, s, w->ent[1] + w->send_base[j], (uint32_t)ns, w->p.log, w->sendbuf, L, d);
    }
    if ((phases & WS_PH_EXCHANGE) && (ns > 0 || nr > 0)) {
        WsRccl api;
        if (!comm || !rows_rccl_api(&api)) return TPNET_ERR_BAD_ARG;
        int bad = api.group_start();
        size_t so = 0, ro = 0;
        for (int r = 0; r < G; ++r) {
            const size_t cs = (size_t)w->send_cnt[(size_t)j * G + r], cr = (size_t)w->recv_cnt[(size_t)j * G + r];
            if (r != me && cs) bad |= api.send(w->sendbuf + so * d, cs * d, kWsFloat32, r, comm, s);
            if (r != me && cr) bad |= api.recv(w->recvbuf + ro * d, cr * d, kWsFloat32, r, comm, s);
            so += cs;
            ro += cr;
        }
        bad |= api.group_end();
        if (bad) return TPNET_ERR_HIP;
    }
    if ((phases & WS_PH_UNPACK) && nr > 0) {
        if (!w->recvbuf) return TPNET_ERR_BAD_ARG;
        int g = (int)((nr + 15) / 16);
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL((k_ws_copy<false>), dim3(g), dim3(256), 0, s, w->ent[0] + w->recv_base[j], (uint32_t)nr, w->p.log, w->recvbuf, L, d);
    }
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

/* the write-back of the owned nodes (the halo rows are scratch of the chunk) */
int tpnet_wshard_finish(tpnet_wshard* w, uint32_t launch_id, void* stream) {
    if (!w || launch_id == 0 || launch_id >= 0x7FFFFFFFu) return TPNET_ERR_BAD_ARG;
    if (!wplan_dense_writeback(w->st, w->p, w->E, w->batch, launch_id, (hipStream_t)stream, w->n_owned)) return TPNET_ERR_HIP;
    return TPNET_OK;
}

/* begin + every step + finish in ONE call (comm: an RCCL communicator of tpnet_rccl_comm_create; NULL with one rank) */
int tpnet_wshard_run(tpnet_wshard* w, void* comm, float* out_pos, float* out_neg, uint32_t launch_id, void* stream) {
    if (!w) return TPNET_ERR_BAD_ARG;
    if (w->G > 1 && !comm) return TPNET_ERR_BAD_ARG;
    int rc = tpnet_wshard_begin(w, comm, WS_PH_PACK | WS_PH_EXCHANGE, stream);
    if (rc) return rc;
    for (int64_t j = 0; j < w->nsteps; ++j) {
        rc = tpnet_wshard_step(w, comm, j, WS_PH_LAUNCH | WS_PH_PACK | WS_PH_EXCHANGE | WS_PH_UNPACK, out_pos, out_neg, stream);
        if (rc) return rc;
    }
    return tpnet_wshard_finish(w, launch_id, stream);
}

/* measurement aid (tpnet_dev.h): tpnet_wshard_run with HIP events on `stream` around every step's launch and around its pack +
 * exchange + unpack; one synchronise at the end.  launch_ms_out / exchange_ms_out: averages per step. */
int tpnet_time_wshard_run(tpnet_wshard* w, void* comm, float* out_pos, float* out_neg, uint32_t launch_id, void* stream,
                          float* total_ms_out, float* launch_ms_out, float* exchange_ms_out) {
    if (!w) return TPNET_ERR_BAD_ARG;
    if (w->G > 1 && !comm) return TPNET_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)w->nsteps;
    std::vector<hipEvent_t> ev(3 * n + 1);
    for (auto& e : ev) TPNET_HIP_TRY(hipEventCreate(&e));
    (void)hipEventRecord(ev[3 * n], s);
    int rc = tpnet_wshard_begin(w, comm, WS_PH_PACK | WS_PH_EXCHANGE, stream);
    for (size_t j = 0; j < n && rc == TPNET_OK; ++j) {
        (void)hipEventRecord(ev[3 * j], s);
        rc = tpnet_wshard_step(w, comm, (int64_t)j, WS_PH_LAUNCH, out_pos, out_neg, stream);
        (void)hipEventRecord(ev[3 * j + 1], s);
        if (rc == TPNET_OK) rc = tpnet_wshard_step(w, comm, (int64_t)j, WS_PH_PACK | WS_PH_EXCHANGE | WS_PH_UNPACK, out_pos, out_neg, stream);
        (void)hipEventRecord(ev[3 * j + 2], s);
    }
    if (rc == TPNET_OK) rc = tpnet_wshard_finish(w, launch_id, stream);
    if (rc == TPNET_OK) {
        TPNET_HIP_TRY(hipStreamSynchronize(s));
        double la = 0.0, xc = 0.0;
        float ms = 0.f;
        for (size_t j = 0; j < n; ++j) {
            TPNET_HIP_TRY(hipEventElapsedTime(&ms, ev[3 * j], ev[3 * j + 1]));
            la += ms;
            TPNET_HIP_TRY(hipEventElapsedTime(&ms, ev[3 * j + 1], ev[3 * j + 2]));
            xc += ms;
        }
        TPNET_HIP_TRY(hipEventElapsedTime(&ms, ev[3 * n], ev[3 * n - 1]));
        if (total_ms_out) *total_ms_out = ms;
        if (launch_ms_out) *launch_ms_out = (float)(la / (double)n);
        if (exchange_ms_out) *exchange_ms_out = (float)(xc / (double)n);
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

}  // extern "C"
