// Shared by the dense planner (wplan_dense.hip) and the row shard's exchange plan (wshard.hip): the dense planner's arrays and
// the lookup "version of node v before batch b".
#pragma once
#include "wplan_common.hpp"

namespace tpnet {

static constexpr int DCH = 256;                     // nodes per chunk = threads per workgroup of every kernel but the sort
static constexpr int DENSE_MAX_WINDOWS = 256;       // (window, class) counters live in LDS: 8 KB
static constexpr int DENSE_TL = 512;
static constexpr int DENSE_MAX_CHUNKS = 2048;       // chunks of 256 nodes (a thread per node in phase B): N <= 524 288

struct DView {                 // the dense planner's arrays (WPlan::dense, carved by wplan_carve)
    uint16_t* len;             // [nb][Ns]
    uint2* m;                  // [nb][Ns] {pre, len | prevb << 16}; prevb = 0xFFFF: no run before batch b
    uint32_t* crank;           // [nw][Ns] B1: rank of the chain (n, w) among the chains of its (chunk of nodes, window, class)
    uint2* wsl;                // [nw][Ns] B1a: {contributions of n in window w, batch of its last run there (0xFFFF: none)}
    uint32_t* tot;             // [Ns] contributions of node n in the chunk
    uint32_t* lastb;           // [Ns] batch of its last run (0xFFFF: none)
    uint32_t* base;            // [Ns] B1: exclusive prefix of tot inside the node's chunk of BS nodes
    uint32_t* basef;           // [Ns] C: first position of node n (base + the chunk's base): what the write-back and a replay read
    uint32_t* ctot;            // [nchunks] B1: contributions per chunk of nodes
    uint32_t* ccnt;            // [nchunks][nw * 8] B1: chains per (chunk, window, class)
    uint32_t* cpre;            // [nchunks][nw * 8] B2: the same, summed over the earlier chunks
    uint32_t* ptot;            // [nw * 8] B2: chains per (window, class)
    int64_t Ns;                // row stride (N rounded up to 64)
};

static inline int64_t dense_ns(int64_t N) { return (N + 63) / 64 * 64; }

static inline DView dview_of(const WPlan& p, int64_t Ec, int64_t batch, int64_t N) {
    DView v;
    const size_t nb = (size_t)((Ec + batch - 1) / batch);
    const size_t Ns = (size_t)dense_ns(N);
    size_t nw = (nb + 1) / 2 + 1;
    if (nw > (size_t)DENSE_MAX_WINDOWS) nw = DENSE_MAX_WINDOWS;
    const size_t nchunks = (Ns + 255) / 256;
    char* c = reinterpret_cast<char*>(p.dense);
    auto take = [&](size_t bytes) { void* r = c; c += (bytes + 255) / 256 * 256; return r; };
    v.len = (uint16_t*)take(nb * Ns * 2);
    v.m = (uint2*)take(nb * Ns * 8);
    v.crank = (uint32_t*)take(nw * Ns * 4);
    v.wsl = (uint2*)take(nw * Ns * 8);
    v.tot = (uint32_t*)take(Ns * 4);
    v.lastb = (uint32_t*)take(Ns * 4);
    v.base = (uint32_t*)take(Ns * 4);
    v.basef = (uint32_t*)take(Ns * 4);
    v.ctot = (uint32_t*)take(nchunks * 4);
    v.ccnt = (uint32_t*)take(nchunks * nw * 8 * 4);
    v.cpre = (uint32_t*)take(nchunks * nw * 8 * 4);
    v.ptot = (uint32_t*)take(nw * 8 * 4);
    v.Ns = (int64_t)Ns;
    return v;
}

// exclusive scan of one value per thread over the workgroup (wave shuffles + one LDS word per wave); total = the sum
template <int BS>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wsum /* [BS / 64] */, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)inc, o, 64);
        if (lane >= o) inc += u;
    }
    __syncthreads();                                        // (wsum may still be read from an earlier scan)
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < BS / 64; ++i) {
        const uint32_t s = wsum[i];
        if (i < wave) base += s;
        tot += s;
    }
    total = tot;
    return base + inc - v;
}

struct DVRef { uint32_t ref; double t_src; };

// the version of node v before batch b, given m[b][v] (and, speculatively loaded, base[v] and the node's meta record)
// (tl: the batches' closing clocks staged in LDS by the caller's workgroup, or nullptr -> read from the descriptors)
__device__ __forceinline__ DVRef dense_version(const BatchDesc* __restrict__ desc, const double* tl, uint2 m, uint32_t basev,
                                               uint4 m0, uint4 m1) {
    DVRef r;
    if (m.x > 0u) {
        r.ref = basev + m.x - 1u;                           // the tail of its previous run
        r.t_src = tl ? tl[m.y >> 16] : desc[m.y >> 16].t_last;
    } else {
        const uint32_t c = m0.x & 1u;
        r.ref = WREF_TABLE | c;
        r.t_src = c ? __hiloint2double((int)m1.y, (int)m1.x) : __hiloint2double((int)m0.w, (int)m0.z);
    }
    return r;
}


}  // namespace tpnet
