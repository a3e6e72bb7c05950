// Probe for the NEXT design step: can consecutive per-batch launches OVERLAP their fixed costs?
// Today batch b+1's kernel starts only after batch b's kernel has ended (one in-order stream): launch processing, wave
// ramp, the first round trip (ids / item records, which do not depend on the state) and the end-of-kernel flush are all
// serial.  Here the launches alternate between TWO streams and the dependency is carried by a flag in memory:
//   kernel(b): preamble (one independent global load) -> spin until flag[b-1] == 1 (BOUNDED: a wave that waits longer
//   than max_spin polls gives up and records it, so the probe cannot hang the GPU) -> `work_trips` dependent loads
//   (stand-in for meta -> rows -> ...) -> write-through store, drain -> done[b]++; the last workgroup sets flag[b].
// Footprint as k_step at C2: 256 workgroups of 512 threads, ONE workgroup per CU (96 KB of LDS each), so a workgroup of
// kernel b+1 becomes resident only where one of kernel b has retired.
// build: hipcc -O3 --offload-arch=gfx950 overlap_probe.hip -o overlap_probe
// run:   ./overlap_probe [batches] [work_trips] [mode]   mode 0 = one stream, no flags; 1 = two streams + flags;
//                                                        2 = one stream + flags (cost of the flag protocol alone)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512) void batch_kernel(const unsigned* __restrict__ chain, unsigned n_chain, int work_trips,
                                                    unsigned* flags, unsigned* done, int b, int use_flags,
                                                    unsigned max_spin, unsigned* gave_up, float* sink) {
    extern __shared__ float lds[];
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    // preamble: independent of the previous batch
    unsigned idx = chain[(tid * 2654435761u + (unsigned)b * 97u) % n_chain];
    if (use_flags && b > 0) {
        // ONE lane per workgroup polls (relaxed, agent scope: served by memory, not by this XCD's L2); the others wait at
        // the barrier.  No acquire fence: the real kernel would read what the previous batch wrote with sc1 loads.
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(&flags[b - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > max_spin) { atomicAdd(gave_up, 1u); break; }
            }
        }
        __syncthreads();
    }
    // dependent round trips
    for (int t = 0; t < work_trips; ++t) idx = chain[idx % n_chain];
    lds[threadIdx.x] = (float)idx;
    if (idx == 0xFFFFFFFFu) sink[tid] = lds[(threadIdx.x + 1) % blockDim.x];
    if (use_flags) {
        __syncthreads();
        if (threadIdx.x == 0) {                     // (the real kernel's stores would be write-through and drained here)
            const unsigned prev = __hip_atomic_fetch_add(&done[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == gridDim.x - 1) __hip_atomic_store(&flags[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 2000;
    const int trips = argc > 2 ? atoi(argv[2]) : 5;
    const int mode = argc > 3 ? atoi(argv[3]) : 1;
    const unsigned n_chain = 1u << 22;                       // 16 MB of indices: misses L2 like the state at C2
    std::vector<unsigned> h(n_chain);
    unsigned x = 12345u;
    for (unsigned i = 0; i < n_chain; ++i) { x = x * 1664525u + 1013904223u; h[i] = x % n_chain; }
    unsigned *chain, *flags, *done, *gave_up;
    float* sink;
    CHECK(hipMalloc(&chain, n_chain * 4));
    CHECK(hipMemcpy(chain, h.data(), n_chain * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&flags, nb * 4)); CHECK(hipMalloc(&done, nb * 4)); CHECK(hipMalloc(&gave_up, 4));
    CHECK(hipMalloc(&sink, 256 * 512 * 4));
    hipStream_t s[2];
    CHECK(hipStreamCreate(&s[0])); CHECK(hipStreamCreate(&s[1]));
    const size_t lds = 96 * 1024;
    CHECK(hipFuncSetAttribute((const void*)batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(flags, 0, nb * 4)); CHECK(hipMemset(done, 0, nb * 4)); CHECK(hipMemset(gave_up, 0, 4));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, s[0]));
        for (int b = 0; b < nb; ++b) {
            hipStream_t st = (mode == 1) ? s[b & 1] : s[0];
            hipLaunchKernelGGL(batch_kernel, dim3(256), dim3(512), lds, st, chain, n_chain, trips, flags, done, b,
                               mode != 0, 20000u, gave_up, sink);
        }
        if (mode == 1) {                                      // join stream 1 into stream 0 before the stop event
            hipEvent_t j; CHECK(hipEventCreate(&j)); CHECK(hipEventRecord(j, s[1])); CHECK(hipStreamWaitEvent(s[0], j, 0));
        }
        CHECK(hipEventRecord(e1, s[0]));
        CHECK(hipDeviceSynchronize());
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned g = 0; CHECK(hipMemcpy(&g, gave_up, 4, hipMemcpyDeviceToHost));
        unsigned last = 0; CHECK(hipMemcpy(&last, flags + nb - 1, 4, hipMemcpyDeviceToHost));
        printf("mode %d trips %d rep %d: %.2f us per batch  (waves that gave up waiting: %u, last flag %u)\n", mode, trips,
               rep, ms * 1000.0f / nb, g, last);
    }
    return 0;
}
