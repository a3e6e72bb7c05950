// Micro-benchmark behind DESIGN.md's C4 argument: a kernel with k_step's ACCESS MIX on a table of C4's size (10 M nodes, d = 256,
// L = 3: layer 0 = 10 GB of 1 KB rows, layers 1..3 = two copies of 10 M bundles of 3 KB = 61 GB) and no arithmetic beyond the adds
// that keep the loads alive.  One unit (a group of 32 lanes x 2 float4 per KB, k_step's geometry at d = 256) stands for one edge:
//   reads  3 nodes x (its 1 KB layer-0 row + its 3 KB bundle of the current copy)          = 12 KB   (src, dst, neg)
//   writes 2 nodes x (3 KB bundle of the other copy) + 2 x 256 B of features                =  6.5 KB (both endpoints updated)
// -- the 2 : 1 read : write mix of profiles/r03_C4.md (131 MB read + 66 MB written per 10 000-edge launch).  Node ids are hashed
// (uniform), every row access is an HBM miss as at C4.  What it answers: the rate the memory system gives THIS mix at this
// launch size, to hold k_step's 4.3 TB/s against (the guide's 5.5-5.8 TB/s is for pure reads).
//   usage: c4_mix_probe [units=10000] [reps=40] [block=256] [inflight=12|6|4] [nt_store=0|1] [nt_load=0|1] [grid_cap=0] [N=10000000] [meta=0|1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

template <bool NTL>
__device__ __forceinline__ v4f ld(const v4f* p) {
    if constexpr (NTL) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NTS>
__device__ __forceinline__ void st(v4f* p, v4f v) {
    if constexpr (NTS) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// INFL = 1 KB rows in flight per group before they are summed: 12 (a whole unit), 6 or 4
// META: every node's current copy comes from a 32-byte per-node record (a dependent random read in front of the rows, as in k_step)
template <int INFL, bool NTS, bool NTL, bool META = false>
__global__ void k_mix(const float* __restrict__ p0, float* __restrict__ q, float* __restrict__ feat, int64_t N, int64_t units,
                      uint64_t seed, const uint4* __restrict__ meta = nullptr) {
    const int gl = threadIdx.x & 31;
    const int64_t gpb = blockDim.x / 32;
    for (int64_t u = (int64_t)blockIdx.x * gpb + threadIdx.x / 32; u < units; u += (int64_t)gridDim.x * gpb) {
        int64_t node[3];
        int cur[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint64_t h = mix64(seed + (uint64_t)u * 3 + k);
            node[k] = (int64_t)(h % (uint64_t)N);
            cur[k] = (int)((h >> 40) & 1);
        }
        if constexpr (META) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint4 m = meta[node[k] * 2];
                cur[k] = (int)((m.x + (uint32_t)cur[k]) & 1u);
            }
        }
        // the 12 rows of the unit: row r = (node r / 4, layer r % 4); layer 0 in p0, layers 1..3 in the node's bundle
        const v4f* rp[12];
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            const int k = r / 4, layer = r % 4;
            const float* base = layer == 0 ? p0 + node[k] * 256 : q + (((int64_t)cur[k] * N + node[k]) * 3 + (layer - 1)) * 256;
            rp[r] = reinterpret_cast<const v4f*>(base) + gl;
        }
        v4f acc[2] = {v4f{0, 0, 0, 0}, v4f{0, 0, 0, 0}};
        v4f keep[6][2];                                  // what the two written bundles are made of (rows of nodes 0 and 1)
#pragma unroll
        for (int r0 = 0; r0 < 12; r0 += INFL) {
            v4f x[INFL][2];
#pragma unroll
            for (int r = 0; r < INFL; ++r) {
                x[r][0] = ld<NTL>(rp[r0 + r]);
                x[r][1] = ld<NTL>(rp[r0 + r] + 32);
            }
#pragma unroll
            for (int r = 0; r < INFL; ++r) {
                acc[0] += x[r][0];
                acc[1] += x[r][1];
                const int rr = r0 + r, k = rr / 4, layer = rr % 4;
                if (k < 2 && layer > 0) { keep[k * 3 + layer - 1][0] = x[r][0]; keep[k * 3 + layer - 1][1] = x[r][1]; }
            }
        }
        // new bundles of nodes 0 and 1 into the OTHER copy, features streamed out
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float* qb = q + (((int64_t)(cur[k] ^ 1) * N + node[k]) * 3) * 256;
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                st<NTS>(reinterpret_cast<v4f*>(qb + l * 256) + gl, keep[k * 3 + l][0] + acc[0]);
                st<NTS>(reinterpret_cast<v4f*>(qb + l * 256) + gl + 32, keep[k * 3 + l][1] + acc[1]);
            }
        }
        if (gl < 16) {
            __builtin_nontemporal_store(acc[0], reinterpret_cast<v4f*>(feat + u * 128) + gl);
            __builtin_nontemporal_store(acc[1], reinterpret_cast<v4f*>(feat + u * 128 + 64) + gl);
        }
    }
}

template <int INFL>
static void launch(bool nts, bool ntl, int grid, int block, hipStream_t s, const float* p0, float* q, float* feat, int64_t N, int64_t units,
                   uint64_t seed, const uint4* meta) {
    if (meta) { hipLaunchKernelGGL((k_mix<INFL, false, false, true>), dim3(grid), dim3(block), 0, s, p0, q, feat, N, units, seed, meta); return; }
    if (nts && ntl) hipLaunchKernelGGL((k_mix<INFL, true, true>), dim3(grid), dim3(block), 0, s, p0, q, feat, N, units, seed);
    else if (nts) hipLaunchKernelGGL((k_mix<INFL, true, false>), dim3(grid), dim3(block), 0, s, p0, q, feat, N, units, seed);
    else if (ntl) hipLaunchKernelGGL((k_mix<INFL, false, true>), dim3(grid), dim3(block), 0, s, p0, q, feat, N, units, seed);
    else hipLaunchKernelGGL((k_mix<INFL, false, false>), dim3(grid), dim3(block), 0, s, p0, q, feat, N, units, seed);
}

int main(int argc, char** argv) {
    const int64_t units = argc > 1 ? atoll(argv[1]) : 10000;
    const int reps = argc > 2 ? atoi(argv[2]) : 40;
    const int block = argc > 3 ? atoi(argv[3]) : 256;
    const int infl = argc > 4 ? atoi(argv[4]) : 12;
    const bool nts = argc > 5 ? atoi(argv[5]) != 0 : false;
    const bool ntl = argc > 6 ? atoi(argv[6]) != 0 : false;
    const int grid_cap = argc > 7 ? atoi(argv[7]) : 0;
    const int64_t N = argc > 8 ? atoll(argv[8]) : 10000000;
    const bool use_meta = argc > 9 ? atoi(argv[9]) != 0 : false;
    float *p0, *q, *feat;
    CK(hipMalloc(&p0, (size_t)N * 1024));
    CK(hipMalloc(&q, (size_t)N * 3072 * 2));
    CK(hipMalloc(&feat, (size_t)units * 512));
    CK(hipMemset(p0, 0, (size_t)N * 1024));
    CK(hipMemset(q, 0, (size_t)N * 3072 * 2));
    uint4* meta = nullptr;
    if (use_meta) { CK(hipMalloc(&meta, (size_t)N * 32)); CK(hipMemset(meta, 0, (size_t)N * 32)); }
    CK(hipDeviceSynchronize());
    const int gpb = block / 32;
    int grid = (int)((units + gpb - 1) / gpb);
    if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](uint64_t seed) {
        if (infl == 12) launch<12>(nts, ntl, grid, block, s, p0, q, feat, N, units, seed, meta);
        else if (infl == 6) launch<6>(nts, ntl, grid, block, s, p0, q, feat, N, units, seed, meta);
        else launch<4>(nts, ntl, grid, block, s, p0, q, feat, N, units, seed, meta);
    };
    for (int i = 0; i < 5; ++i) run(1000 + i);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) run(77 * (uint64_t)i + 5);      // a new set of rows per launch: every access an HBM miss
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    const double rd = (double)units * 12288.0, wr = (double)units * (6144.0 + 512.0);
    if (use_meta) printf("[dependent 32-byte meta read per node] ");
    printf("units %lld block %d grid %d inflight %d nt_store %d nt_load %d: %.2f us per launch, read %.1f MB + written %.1f MB -> %.2f TB/s "
           "(read side %.2f, write side %.2f)\n", (long long)units, block, grid, infl, (int)nts, (int)ntl, us, rd / 1e6, wr / 1e6,
           (rd + wr) / us / 1e6, rd / us / 1e6, wr / us / 1e6);
    return 0;
}
