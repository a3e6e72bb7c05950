// self.mlp = Linear(64,256) -> ReLU -> Linear(256,64) (models/TPNet.py:63-65,129) on rows of features that already exist ([n][64]
// f32: the encoder's call, models/TPNet.py:311-324 -- 4*B*K rows per call), fp32 class, for LONG lists.
// Same arithmetic as MODE 2 of feature_mfma.hip (every f32 operand x = hi + lo, hi = bf16(x), lo = bf16(x - hi); every product as
// lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulators), other mapping: there a 32-row tile is spread over the 8
// waves of a workgroup (one hidden slice each, weights in registers) and costs three workgroup barriers and an 8-way partial-sum
// exchange through LDS per tile -- ~3 us per tile of which 0.3 us is matrix work.  Here every WAVE owns whole 32-row tiles: it
// walks the 8 hidden slices itself, layer 2's accumulators never leave its registers, and the split weights (128 KB: W1 and W2,
// hi and lo, in exactly the per-lane operand order) live in the CU's LDS (160 KB on gfx950), staged once per workgroup.  No
// barrier after the staging; 192 matrix instructions per tile and wave against 128 ds_read_b128 (lanes consecutive: conflict-free).
// The hidden slices are summed inside the accumulator (slice 0..7 in order) instead of the other kernel's fixed tree: the two
// kernels differ in the last bits, both within the fp32-class bound (<= 2e-5 of the output scale against the torch layers).
// Measured: 80 000 rows 32.5 -> 23 us, 800 000 rows 224 -> 150 us (load + split + store alone: 12 / 97 us; matrix pipe busy 4.8 M x
// 32 cycles per launch = 42 % of a 2.4 GHz clock).  Not kept: twelve waves per workgroup (157 us); slice w + 1's layer-1 products
// issued ahead of slice w's vector work, with and without sched_group_barrier interleaving (162 us: more registers, no overlap won).
// Nor: the weights of a step read from LDS one step ahead of their products behind scheduling barriers (164 us); one LDS block per
// slice (every read base + 16-bit offset), the accumulators started from the bias and a one-instruction ReLU -- 130 vector instructions
// per two slices instead of 225 -- at 153 us: neither the LDS round trips nor the vector work is what the launch waits for.
#include "tpnet_common.h"

namespace tpnet {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

#ifndef TPNET_X3_THREADS
#define TPNET_X3_THREADS 512
#endif
static constexpr int XB = TPNET_X3_THREADS;           // threads per workgroup: 8 waves, two per SIMD
static constexpr int XW = XB / 64;
static constexpr int XF = 64, XH = 256;
static constexpr int X_W1H = 0, X_W1L = 32768, X_W2H = 65536, X_W2L = 98304, X_B1 = 131072, X_B2 = X_B1 + 1024;
static constexpr int X_LDS = X_B2 + 256;             // 132 352 bytes

__device__ __forceinline__ void split8v(const float4 a, const float4 b, bf16x8& hi, bf16x8& lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 t = (__bf16)v[j];
        hi[j] = t;
        lo[j] = (__bf16)(v[j] - (float)t);
    }
}

// w1f = f32 [256][64] (mlp[0].weight as is); w2f = f32 [8 slices][2 output tiles][64 lanes][16] (tpnet_mlp::w2f, include/tpnet_hip.h)
__global__ __launch_bounds__(XB) void k_mlp64_x3(const float* __restrict__ X, int64_t n, const float* __restrict__ w1f,
                                                 const float* __restrict__ b1, const float* __restrict__ w2f,
                                                 const float* __restrict__ b2, float* __restrict__ Y, int nact_arg) {
#ifdef TPNET_DEV
    const int nact = nact_arg & 0xFF, dbg_mode = nact_arg >> 8;     // diagnostic builds: 1 = no matrix work, 2 = no stores
#else
    const int nact = nact_arg;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t ntiles = (n + 31) / 32;
    // tiles of the active waves: wave v of workgroup b is wave v * grid + b of the launch (a list's last, partial round of tiles
    // then lands on different CUs, not on the eight waves of a few)
    const int64_t aw = (int64_t)gridDim.x * nact;
    int64_t tile = (int64_t)wave * gridDim.x + blockIdx.x;
    const bool active = wave < nact;
    // ---- this wave's first tile ahead of the staging: lane (r, h) holds X[row r][16 s + 8 h + j], the B operand of k-step s
    float4 xa[4], xb[4];
    auto load_tile = [&](int64_t t) {
        const int64_t row = t * 32 + r;
        const bool ok = active && t < ntiles && row < n;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float* p = X + row * XF + 16 * s + 8 * h;
            xa[s] = ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
            xb[s] = ok ? *reinterpret_cast<const float4*>(p + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_tile(tile);
    // ---- the split weights -> LDS, every thread 4 + 4 pieces of 8 floats
    // (walked in the order of the LDS image: consecutive lanes write consecutive 16-byte slots -- in the sources' own order eight
    // consecutive lanes hit one bank, 4 096 conflict cycles per workgroup -- and read 32-byte pieces 256 / 64 bytes apart)
    for (int i = tid; i < 2048; i += XB) {                                // W1: slot ((w * 4 + s) * 64 + lane (h, r))
        const int w = i >> 8, sx = (i >> 6) & 3, hh = (i >> 5) & 1, rr = i & 31;
        const float* p = w1f + ((w * 32 + rr) * XF + 16 * sx + 8 * hh);
        bf16x8 hi, lo;
        split8v(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4), hi, lo);
        *reinterpret_cast<bf16x8*>(smem + X_W1H + i * 16) = hi;
        *reinterpret_cast<bf16x8*>(smem + X_W1L + i * 16) = lo;
    }
    for (int i = tid; i < 2048; i += XB) {                                // W2: slot (((w * 2 + s2) * 2 + t2) * 64 + lane)
        const int w = i >> 8, s2 = (i >> 7) & 1, t2 = (i >> 6) & 1, ln = i & 63;
        const float* p = w2f + ((((w * 2 + t2) * 64 + ln) * 16) + 8 * s2);
        bf16x8 hi, lo;
        split8v(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4), hi, lo);
        *reinterpret_cast<bf16x8*>(smem + X_W2H + i * 16) = hi;
        *reinterpret_cast<bf16x8*>(smem + X_W2L + i * 16) = lo;
    }
    if (tid < XH) {
        // bias of layer 1 in accumulator order: register q of lane half hh of slice w = hidden unit 32 w + (q&3) + 8 (q>>2) + 4 hh
        const int w = tid >> 5, hh = (tid >> 4) & 1, q = tid & 15;
        reinterpret_cast<float*>(smem + X_B1)[tid] = b1[w * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh];
    } else if (tid < XH + XF) {
        reinterpret_cast<float*>(smem + X_B2)[tid - XH] = b2[tid - XH];
    }
    __syncthreads();
    if (!active) return;
    const bf16x8* W1H = reinterpret_cast<const bf16x8*>(smem + X_W1H) + lane;
    const bf16x8* W1L = reinterpret_cast<const bf16x8*>(smem + X_W1L) + lane;
    const bf16x8* W2H = reinterpret_cast<const bf16x8*>(smem + X_W2H) + lane;
    const bf16x8* W2L = reinterpret_cast<const bf16x8*>(smem + X_W2L) + lane;
    const float4* B1 = reinterpret_cast<const float4*>(smem + X_B1) + h * 4;
    const float4* B2 = reinterpret_cast<const float4*>(smem + X_B2);
    for (; tile < ntiles; tile += aw) {
        bf16x8 bxh[4], bxl[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) split8v(xa[s], xb[s], bxh[s], bxl[s]);
        load_tile(tile + aw);                                             // the next tile rides under this one's matrix work
        f32x16 y0, y1;
#pragma unroll
        for (int q = 0; q < 16; ++q) { y0[q] = 0.0f; y1[q] = 0.0f; }
        // layer 1 of hidden slice w: H^T = W1[32 w .., :] . X^T
        auto layer1 = [&](int w) {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 ah = W1H[(w * 4 + s) * 64], al = W1L[(w * 4 + s) * 64];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bxh[s], acc, 0, 0, 0);      // the small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bxl[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bxh[s], acc, 0, 0, 0);
            }
            return acc;
        };
        // bias, ReLU, split of slice w's accumulators (register q = hidden unit 32 w + (q&3) + 8 (q>>2) + 4 h of row r) -> the B
        // operand of layer 2; then this slice's share of both output tiles
        auto layer2 = [&](int w, const f32x16& a) {
            bf16x8 bhh[2], bhl[2];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 bb = B1[w * 8 + q4];
                const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = 4 * q4 + j;
                    float x = a[q] + bv[j];
                    x = x > 0.0f ? x : 0.0f;
                    const __bf16 hi = (__bf16)x;
                    bhh[q >> 3][q & 7] = hi;
                    bhl[q >> 3][q & 7] = (__bf16)(x - (float)hi);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 a0h = W2H[((w * 2 + s2) * 2 + 0) * 64], a0l = W2L[((w * 2 + s2) * 2 + 0) * 64];
                const bf16x8 a1h = W2H[((w * 2 + s2) * 2 + 1) * 64], a1l = W2L[((w * 2 + s2) * 2 + 1) * 64];
                y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, bhh[s2], y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, bhh[s2], y1, 0, 0, 0);
                y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, bhl[s2], y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, bhl[s2], y1, 0, 0, 0);
                y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, bhh[s2], y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, bhh[s2], y1, 0, 0, 0);
            }
        };
#ifdef TPNET_DEV
        if (dbg_mode == 1) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { y0[q] = (float)bxh[q & 3][q >> 2] + (float)bxl[q & 3][q >> 2]; y1[q] = (float)bxh[q & 3][4 + (q >> 2)]; }
        } else
#endif
#pragma unroll 2
        for (int w = 0; w < 8; ++w) {
            const f32x16 acc = layer1(w);
            layer2(w, acc);
        }
        // y0[4 i .. 4 i + 3] = outputs 8 i + 4 h + (0..3) of row r, y1: + 32
        const int64_t row = tile * 32 + r;
#ifdef TPNET_DEV
        if (dbg_mode == 2 && y0[0] != 12345.678f) continue;
#endif
        if (row < n) {
            float* yo = Y + row * XF;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = 8 * i + 4 * h;
                const float4 c0 = B2[o >> 2], c1 = B2[(32 + o) >> 2];
                *reinterpret_cast<float4*>(yo + o) =
                    make_float4(y0[4 * i] + c0.x, y0[4 * i + 1] + c0.y, y0[4 * i + 2] + c0.z, y0[4 * i + 3] + c0.w);
                *reinterpret_cast<float4*>(yo + 32 + o) =
                    make_float4(y1[4 * i] + c1.x, y1[4 * i + 1] + c1.y, y1[4 * i + 2] + c1.z, y1[4 * i + 3] + c1.w);
            }
        }
    }
}

// 0: not decided; 1: available; -1: this device / runtime does not give a workgroup 132 KB of LDS
static int mlp_x3_state = 0;

bool mlp_x3_available() {
    if (mlp_x3_state == 0) {
        static const int off = TPNET_DEV_INT(NO_MLP_X3, 0);
        const bool ok = !off && hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp64_x3),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS) == hipSuccess;
        (void)hipGetLastError();
        mlp_x3_state = ok ? 1 : -1;
    }
    return mlp_x3_state == 1;
}

// rows from which the per-wave-tile kernel takes a list (below: feature_mfma.hip's kernel spreads the few tiles over more CUs)
int64_t mlp_x3_from() {
    static const int64_t from = (int64_t)TPNET_DEV_INT(MLP_X3_FROM, 8192);
    return from;
}

int launch_mlp_rows_x3(const float* x, int64_t n, const float* w1f, const float* b1, const float* w2f, const float* b2, float* y,
                       hipStream_t s) {
    if (n == 0) return TPNET_OK;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w1f) |
         reinterpret_cast<uintptr_t>(w2f)) & 15)
        return TPNET_ERR_BAD_ARG;
    if (!mlp_x3_available()) return TPNET_ERR_BAD_ARG;
    const int64_t tiles = (n + 31) / 32;
    int64_t nact = (tiles + 255) / 256;                // waves per workgroup that take tiles: a CU's share of one round
    nact = nact > XW ? XW : nact;
    static const int nact_dev = TPNET_DEV_INT(X3_WAVES, 0);
    static const int mode_dev = TPNET_DEV_INT(X3_MODE, 0);
    if (nact_dev > 0 && nact_dev < nact) nact = nact_dev;
    int64_t grid = (tiles + nact - 1) / nact;
    grid = grid > 256 ? 256 : grid;
    hipLaunchKernelGGL(k_mlp64_x3, dim3((unsigned)grid), dim3(XB), X_LDS, s, x, n, w1f, b1, w2f, b2, y, (int)nact | (mode_dev << 8));
    if (hipGetLastError() != hipSuccess) {             // (a runtime that refuses the launch: the callers fall back for good)
        mlp_x3_state = -1;
        return TPNET_ERR_BAD_ARG;
    }
    return TPNET_OK;
}

}  // namespace tpnet
