// Probe: do matrix instructions of one wave and vector instructions of ANOTHER wave of the same SIMD overlap on gfx950?
// A workgroup of 8 waves per CU (wave w and w + 4 share a SIMD): waves 0..3 run chains of v_mfma_f32_16x16x32_bf16 (NCH independent
// accumulators), waves 4..7 run chains of v_fma_f32 (8 independent), each for a fixed instruction count.  Times: matrix waves alone,
// vector waves alone, both.  If the two overlap, "both" ~ max(alone); if the issue port serialises them, "both" ~ sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
#ifdef BIG
#define ACC_T f32x16
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define PIPE 32
#else
#define ACC_T f32x4
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define PIPE 16
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NCH>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    extern __shared__ char smem[];
    const int wave = threadIdx.x >> 6;
    const bool mat = wave < 4;
    if ((mode == 1 && !mat) || (mode == 2 && mat)) return;
    if (mode == 4) {
        // ONE instruction stream with both: every matrix instruction followed by three independent vector instructions
        if (!mat) return;
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x - j)); }
        ACC_T acc[NCH];
        for (int c = 0; c < NCH; ++c) for (int q = 0; q < (int)(sizeof(ACC_T) / 4); ++q) acc[c][q] = 0;
        float x[6];
        for (int j = 0; j < 6; ++j) x[j] = 0.001f * (threadIdx.x + j);
        const float m = 1.0000001f, ad = 1e-9f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    acc[c] = MFMA(a, b, acc[c]);
                    x[(3 * c) % 6] = __builtin_fmaf(x[(3 * c) % 6], m, ad);
                    x[(3 * c + 1) % 6] = __builtin_fmaf(x[(3 * c + 1) % 6], m, ad);
                    x[(3 * c + 2) % 6] = __builtin_fmaf(x[(3 * c + 2) % 6], m, ad);
                }
        }
        float s = 0;
        for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][3];
        for (int j = 0; j < 6; ++j) s += x[j];
        if (s == 12345.678f) out[threadIdx.x] = s;
        return;
    }
    if (mat) {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x - j)); }
        ACC_T acc[NCH];
        for (int c = 0; c < NCH; ++c) for (int q = 0; q < (int)(sizeof(ACC_T) / 4); ++q) acc[c][q] = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < NCH; ++c) acc[c] = MFMA(a, b, acc[c]);
        }
        float s = 0;
        for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][3];
        if (s == 12345.678f) out[threadIdx.x] = s;
    } else {
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = 0.001f * (threadIdx.x + j);
        const float m = 1.0000001f, a = 1e-9f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8 * NCH / 2; ++r) {          // as many vector instructions x 4 cycles as the matrix waves' pipe cycles / 2
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], m, a);
            }
        }
        float s = 0;
        for (int j = 0; j < 8; ++j) s += x[j];
        if (s == 12345.678f) out[threadIdx.x] = s;
    }
}

int main() {
    float* out;
    CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    auto run = [&](auto kern, int mode, const char* name, int nch) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 100 * 1024, 0, out, 10, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 100 * 1024, 0, out, iters, mode);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (double)iters * 8 * nch, valu = (double)iters * 8 * nch / 2 * 8;
        printf("chains %d, %-22s %8.1f us   (per matrix wave %.0f mfma = %.0f pipe cycles; per vector wave %.0f fma = %.0f issue cycles)\n", nch, name,
               ms * 1e3, mfma, mfma * PIPE, valu, valu * 4);
        return 0;
    };
    for (int mode = 1; mode <= 3; ++mode) {
        const char* nm = mode == 1 ? "matrix waves alone" : mode == 2 ? "vector waves alone" : "both";
        run(k<1>, mode, nm, 1);
    }
    for (int mode = 1; mode <= 3; ++mode) {
        const char* nm = mode == 1 ? "matrix waves alone" : mode == 2 ? "vector waves alone" : "both";
        run(k<2>, mode, nm, 2);
    }
    for (int mode = 1; mode <= 4; ++mode) {
        const char* nm = mode == 1 ? "matrix waves alone" : mode == 2 ? "vector waves alone" : mode == 3 ? "both" : "one stream: mfma + 3 fma";
        run(k<4>, mode, nm, 4);
    }
    return 0;
}
