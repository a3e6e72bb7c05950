// Developer probe: does a hipGraph shorten a chain of DEPENDENT small kernels on this GPU (the per-batch schedule's 21 launches)?
// 20 dependent kernels of 256 workgroups x 512 threads, each three dependent loads deep (ids -> meta -> row) + a store:
// (a) 20 launches on one stream, (b) the same 20 launches captured once and replayed as a graph.  Wall clock and HIP events.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/graph_chain.hip -o gpurun_out/graph_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(512) void k_link(const int* __restrict__ ids, const int* __restrict__ meta, const float4* __restrict__ rows,
                                              float4* __restrict__ out, int n, int step) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = i >> 5, l = i & 31;
    if (g >= n) return;
    const int id = ids[(g + step * 977) % n];
    const int c = meta[id] & 1;
    float4 acc = rows[((size_t)id * 2 + c) * 64 + l];
    const float4 b = rows[((size_t)id * 2 + c) * 64 + 32 + l];
    acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    out[((size_t)id * 2 + (c ^ 1)) * 64 + l] = acc;
}

int main() {
    const int n = 4096, N = 9228, K = 20;
    int *ids, *meta; float4 *rows;
    CK(hipMalloc(&ids, n * 4)); CK(hipMalloc(&meta, N * 4)); CK(hipMalloc(&rows, (size_t)N * 2 * 64 * 16));
    std::vector<int> h(n); for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 2654435761ll) % N);
    CK(hipMemcpy(ids, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(meta, 0, N * 4)); CK(hipMemset(rows, 0, (size_t)N * 2 * 64 * 16));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto chain = [&]() { for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_link, dim3(n * 32 / 512), dim3(512), 0, s, ids, meta, rows, rows, n, k); };
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); chain(); CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<double> wall, dev;
        for (int r = 0; r < 60; ++r) {
            CK(hipStreamSynchronize(s));
            const auto t0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(e0, s));
            if (mode == 0) chain(); else CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            const auto t1 = std::chrono::steady_clock::now();
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 10) { wall.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count()); dev.push_back(ms * 1e3); }
        }
        std::sort(wall.begin(), wall.end()); std::sort(dev.begin(), dev.end());
        printf("%s: %d dependent kernels, wall %.1f us, events %.1f us (%.2f us per kernel)\n", mode ? "graph " : "stream", K,
               wall[wall.size() / 2], dev[dev.size() / 2], dev[dev.size() / 2] / K);
    }
    return 0;
}
