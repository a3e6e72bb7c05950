// Probe for the NEXT design step: cost and safety of handing a row from one wave to waves on other CUs
// INSIDE a launch, with write-through (sc1) stores, a version word and sc1 loads -- the primitive a persistent,
// dependency-driven stream kernel would use instead of one launch per batch.
//   * a hot row (512 B, like a d=128 layer row) lives in a ring of 3 copies; version v is in copy v % 3, every float = v
//   * writer of step s (a different wave -- and CU/XCD -- each step) polls ver == s, reads copy s%3, writes copy
//     (s+1)%3 with sc1 stores, drains (s_waitcnt vmcnt(0)), publishes ver = s+1 (agent-scope atomic store)
//   * every other wave is a reader: polls ver, reads the copy with sc1 loads, checks all 128 floats == ver (unless the
//     version moved on by >= 2 meanwhile), optionally under background streaming load
// build: hipcc -O3 --offload-arch=gfx950 handoff_probe.hip -o handoff_probe ; run: ./handoff_probe [steps] [bg]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f4;

__device__ __forceinline__ f4 ld_sc1(const f4* p) {
    f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_sc1(f4* p, f4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned ld_ver(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(512) void probe(f4* ring /*[3][32]*/, unsigned* ver, unsigned steps, unsigned long long* stats,
                                             const f4* bg, size_t bg_n, int bg_on, unsigned long long* tstamp, int pollers) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const unsigned nwaves = gridDim.x * (blockDim.x / 64);
    unsigned long long bad = 0, reads = 0, spins = 0;
    f4 sink = {0, 0, 0, 0};
    size_t bgi = (size_t)wave * 64 + lane;
    // writer schedule: step s is written by wave (s * 37) % nwaves  (hops CUs and XCDs)
    unsigned my_next = 0xFFFFFFFFu;
    for (unsigned s = 0; s < steps; ++s)
        if ((s * 37u) % nwaves == wave) { my_next = s; break; }
    for (unsigned guard = 0; guard < 40000000u; ++guard) {
        const unsigned v = ld_ver(ver);
        if (v >= steps) break;
        if (v == my_next) {                                   // my turn to write version v+1
            if (v == 0 && lane == 0) tstamp[0] = __builtin_amdgcn_s_memrealtime();
            f4 x = {0, 0, 0, 0};
            if (lane < 32) x = ld_sc1(ring + (v % 3) * 32 + lane);
            bool ok = lane >= 32 || (x.x == (float)v && x.y == (float)v && x.z == (float)v && x.w == (float)v);
            if (!__all(ok)) bad += 1000000;                   // a writer must never see stale data
            x.x += 1.0f; x.y += 1.0f; x.z += 1.0f; x.w += 1.0f;
            if (lane < 32) st_sc1(ring + ((v + 1) % 3) * 32 + lane, x);
            drain();
            if (lane == 0) __hip_atomic_store(ver, v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v + 1 == steps && lane == 0) tstamp[1] = __builtin_amdgcn_s_memrealtime();
            my_next = 0xFFFFFFFFu;
            for (unsigned s = v + 1; s < steps; ++s)
                if ((s * 37u) % nwaves == wave) { my_next = s; break; }
        } else if ((int)(threadIdx.x >> 6) >= pollers && my_next - v > 64) {   // not a poller now: stream / sleep, look rarely
            if (bg_on) {
                for (int k = 0; k < 64; ++k) { f4 t = bg[bgi % bg_n]; sink += t; bgi += (size_t)nwaves * 64; }
            } else {
                __builtin_amdgcn_s_sleep(64);
            }
        } else {                                              // reader
            f4 x = {0, 0, 0, 0};
            if (lane < 32) x = ld_sc1(ring + (v % 3) * 32 + lane);
            const unsigned v2 = ld_ver(ver);
            bool ok = lane >= 32 || (x.x == (float)v && x.y == (float)v && x.z == (float)v && x.w == (float)v);
            if (v2 - v < 2) { ++reads; if (!__all(ok)) ++bad; }
            if (bg_on) {                                      // background streaming load on the same CU
                for (int k = 0; k < 8; ++k) { f4 t = bg[bgi % bg_n]; sink += t; bgi += (size_t)nwaves * 64; }
            } else {
                __builtin_amdgcn_s_sleep(2);
            }
            ++spins;
        }
    }
    if (lane == 0) {
        atomicAdd(&stats[0], bad);
        atomicAdd(&stats[1], reads);
        atomicAdd(&stats[2], spins);
        if (sink.x == 123.456f) stats[3] = 1;
    }
}

int main(int argc, char** argv) {
    const unsigned steps = argc > 1 ? (unsigned)atoi(argv[1]) : 2000;
    const int bg_on = argc > 2 ? atoi(argv[2]) : 0;
    const int pollers = argc > 3 ? atoi(argv[3]) : 8;
    f4* ring; unsigned* ver; unsigned long long *stats, *ts; f4* bg;
    const size_t bg_n = (size_t)64 << 20;                      // 1 GiB of float4
    hipMalloc(&ring, 3 * 32 * sizeof(f4)); hipMalloc(&ver, 256); hipMalloc(&stats, 64); hipMalloc(&ts, 64);
    hipMalloc(&bg, bg_n * sizeof(f4));
    hipMemset(ring, 0, 3 * 32 * sizeof(f4)); hipMemset(ver, 0, 256); hipMemset(stats, 0, 64); hipMemset(ts, 0, 64);
    hipMemset(bg, 0, bg_n * sizeof(f4));
    hipDeviceSynchronize();
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, ring, ver, steps, stats, bg, bg_n, bg_on, ts, pollers);
    hipError_t e = hipDeviceSynchronize();
    unsigned long long h[4], t[2]; unsigned v;
    hipMemcpy(h, stats, 32, hipMemcpyDeviceToHost); hipMemcpy(t, ts, 16, hipMemcpyDeviceToHost);
    hipMemcpy(&v, ver, 4, hipMemcpyDeviceToHost);
    printf("handoff probe: pollers/block=%d steps=%u bg=%d status=%s final ver=%u bad=%llu checked reads=%llu spins=%llu; chain %.3f us per hand-off\n",
           pollers, steps, bg_on, hipGetErrorString(e), v, h[0], h[1], h[2], (double)(t[1] - t[0]) * 0.01 / steps);
    return (v == steps && h[0] == 0) ? 0 : 1;
}
