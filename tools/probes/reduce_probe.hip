// Micro-benchmark of the 64-value recursive-halving reduction over 32- and 64-lane groups: cycles per reduction for
// (a) the product form (v_permlane32/16_swap + DPP), (b) ds_bpermute shuffles (__shfl_xor), (c) ds_swizzle for M=16.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int M>
__device__ __forceinline__ float pair_sum(float a) {
    constexpr int ctrl = (M == 8) ? 0x128 : (M == 4) ? 0x141 : (M == 2) ? 0x4E : 0xB1;
    const float p = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), ctrl, 0xF, 0xF, false));
    return a + p;
}
template <int C, int M, int MODE>
struct Halve {
    static __device__ __forceinline__ void run(float* v, int gl) {
        if constexpr (MODE == 0 && M == 32) {
#pragma unroll
            for (int i = 0; i < C / 2; ++i) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + C / 2]), false, false);
                v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
        } else if constexpr (MODE == 0 && M == 16) {
#pragma unroll
            for (int i = 0; i < C / 2; ++i) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + C / 2]), false, false);
                v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
        } else if constexpr (MODE == 2 && M == 16) {
            const bool upper = (gl & M) != 0;
#pragma unroll
            for (int i = 0; i < C / 2; ++i) {
                const float keep = upper ? v[i + C / 2] : v[i];
                const float send = upper ? v[i] : v[i + C / 2];
                v[i] = keep + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(send), 0x401F));
            }
        } else if constexpr (MODE == 1 || M >= 16) {
            const bool upper = (gl & M) != 0;
#pragma unroll
            for (int i = 0; i < C / 2; ++i) {
                const float keep = upper ? v[i + C / 2] : v[i];
                const float send = upper ? v[i] : v[i + C / 2];
                v[i] = keep + __shfl_xor(send, M, 64);
            }
        } else {
            const bool upper = (gl & M) != 0;
#pragma unroll
            for (int i = 0; i < C / 2; ++i) {
                const float x = pair_sum<M>(v[i]);
                const float y = pair_sum<M>(v[i + C / 2]);
                v[i] = upper ? y : x;
            }
        }
        if constexpr (M > 1) Halve<C / 2, M / 2, MODE>::run(v, gl);
    }
};

template <int LPP, int MODE>
__global__ void probe(float* out, unsigned long long* cyc, int reps) {
    const int gl = threadIdx.x % LPP;
    float v[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = (float)(threadIdx.x * 64 + i) * 1e-3f;
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        float w[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) w[i] = v[i] + acc;
        Halve<64, LPP / 2, MODE>::run(w, gl);
        acc += w[0] * 1e-9f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int LPP, int MODE>
void run(const char* name, int waves_per_simd) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 64);
    const int reps = 2000;
    hipLaunchKernelGGL((probe<LPP, MODE>), dim3(256), dim3(256 * waves_per_simd), 0, 0, out, cyc, reps);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("LPP=%d %-28s waves/SIMD=%d: %.0f cycles per 64-value reduction (incl. 64 adds of setup)\n", LPP, name, waves_per_simd,
           (double)h / reps);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<32, 0>("permlane16_swap + DPP", w);
        run<32, 1>("__shfl_xor (ds_bpermute)", w);
        run<32, 2>("ds_swizzle(M=16) + DPP", w);
        run<64, 0>("permlane32/16_swap + DPP", w);
        run<64, 1>("__shfl_xor (ds_bpermute)", w);
    }
    return 0;
}
