// gfx950 kernels that walk or gather the table itself: init / import / export / dense decay (models/TPNet.py:83-85,
// :131-157), row and element gathers (:101-110; models/MemoryModel.py:396-405), the row exchange of the row-sharded
// layout, and the readout's element-wise tail on summed partial Gram entries (:126-128).
#include "device_common.hpp"

namespace tpnet {

// ---------------------------------------------------------------------------------------------------------------
// dense passes
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_state_init(tpnet_state S, double t0) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t nq = 2 * S.N * (int64_t)S.L * S.d;
    for (int64_t i = tid; i < nq; i += stride) S.q[i] = 0.0f;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    for (int64_t n = tid; n < S.N; n += stride) {
        NodeMeta m;
        m.ver = 0; m.pad0 = 0; m.tref[0] = t0; m.tref[1] = t0; m.pad1 = 0;
        meta[n] = m;
    }
    if (tid < 4) S.err[tid] = 0;
}

struct LayerPtrs {
    float* p[TPNET_MAX_LAYERS];
};

// layers (row-major [N][d] each) -> copy 0 bundles; meta = {0, now}
__global__ void k_import(tpnet_state S, LayerPtrs lp, double now) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t d = S.d, L = S.L;
    const int64_t tot = S.N * L * d;
    for (int64_t x = tid; x < tot; x += stride) {
        const int64_t n = x / (L * d);
        const int64_t r = x - n * (L * d);
        const int64_t i = r / d, k = r - i * d;
        S.q[x] = lp.p[i][n * d + k];
    }
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    for (int64_t n = tid; n < S.N; n += stride) {
        NodeMeta m;
        m.ver = 0; m.pad0 = 0; m.tref[0] = now; m.tref[1] = now; m.pad1 = 0;
        meta[n] = m;
    }
}

// current bundles with the pending decay applied -> layers
__global__ void k_export(tpnet_state S, LayerPtrs lp, double now, double lambda) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t d = S.d, L = S.L;
    const int64_t tot = S.N * L * d;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(S.meta);
    for (int64_t x = tid; x < tot; x += stride) {
        const int64_t n = x / (L * d);
        const int64_t r = x - n * (L * d);
        const int64_t i = r / d, k = r - i * d;
        const MetaView m = read_meta(meta, n, READER_BID, now, lambda);
        float g = m.g;
        for (int64_t z = 0; z < i; ++z) g *= m.g;
        lp.p[i][n * d + k] = S.q[((int64_t)m.copy * S.N) * (L * d) + x] * g;
    }
}

struct DecayFactors {
    float f[TPNET_MAX_LAYERS];
};

// eager dense decay (models/TPNet.py:83-85): current copy of every row *= f[i]; tref = t_new.
// One block-stride loop over nodes; LPP-agnostic (one thread per element of the bundle).
__global__ void k_decay(tpnet_state S, DecayFactors df, double t_new) {
    const int64_t d = S.d, L = S.L;
    const int64_t per = L * d;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    for (int64_t n = blockIdx.x; n < S.N; n += gridDim.x) {
        const uint32_t ver = meta[n].ver;
        const int c = ver & 1;
        float* qb = S.q + ((int64_t)c * S.N + n) * per;
        for (int64_t r = threadIdx.x; r < per; r += blockDim.x) {
            const int64_t i = r / d;
            qb[r] = qb[r] * df.f[i];
        }
        __syncthreads();
        if (threadIdx.x == 0) meta[n].tref[c] = t_new;
    }
}

__global__ void k_decay_desc(tpnet_state S, const BatchDesc* __restrict__ desc, int64_t b) {
    const int64_t d = S.d, L = S.L;
    const int64_t per = L * d;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    const float* __restrict__ decay = desc[b].decay;  // indexed from memory (a private copy would be demoted to LDS)
    const double t_last = desc[b].t_last;
    for (int64_t n = blockIdx.x; n < S.N; n += gridDim.x) {
        const uint32_t ver = meta[n].ver;
        const int c = ver & 1;
        float* qb = S.q + ((int64_t)c * S.N + n) * per;
        for (int64_t r = threadIdx.x; r < per; r += blockDim.x) {
            const int64_t i = r / d;
            qb[r] = qb[r] * decay[i];
        }
        __syncthreads();
        if (threadIdx.x == 0) meta[n].tref[c] = t_last;
    }
}

// get_random_projections (models/TPNet.py:101-110): out[(i*n + k)*d + :] = P[i][ids[k]]
__global__ void k_gather_rows(tpnet_state S, const int64_t* __restrict__ ids, int64_t n, double now, double lambda,
                              float* __restrict__ out) {
    const int64_t d = S.d, L = S.L;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(S.meta);
    for (int64_t k = blockIdx.x; k < n; k += gridDim.x) {
        int64_t id = ids[k];
        const bool ok = (uint64_t)id < (uint64_t)S.N;
        if (!ok) {
            if (threadIdx.x == 0) atomicAdd(S.err, 1u);
            id = 0;
        }
        const MetaView m = read_meta(meta, id, READER_BID, now, lambda);
        const float* qb = S.q + ((int64_t)m.copy * S.N + id) * (L * d);
        for (int64_t r = threadIdx.x; r < (L + 1) * d; r += blockDim.x) {
            const int64_t i = r / d, c = r - i * d;
            float x;
            if (i == 0) {
                x = S.p0[id * d + c];
            } else {
                float g = m.g;
                for (int64_t z = 1; z < i; ++z) g *= m.g;
                x = qb[(i - 1) * d + c] * g;
            }
            out[(i * n + k) * d + c] = ok ? x : __builtin_nanf("");
        }
    }
}

// single elements of the layers: out[k][i] = P[i][rows[k]][cols[k]] at `now`, i = 0..L (the walk-matrix readout of
// models/MemoryModel.py:396-405, `matrix[src, dst]`, when the table is square: tpnet_amd/matrix_memory.py)
__global__ void k_gather_elems(tpnet_state S, const int64_t* __restrict__ rows, const int64_t* __restrict__ cols,
                               int64_t n, double now, double lambda, float* __restrict__ out) {
    const int64_t d = S.d, L = S.L;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(S.meta);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        int64_t r = rows[k], c = cols[k];
        const bool ok = (uint64_t)r < (uint64_t)S.N && (uint64_t)c < (uint64_t)d;
        if (!ok) {
            atomicAdd(S.err, 1u);
            r = 0; c = 0;
        }
        const MetaView m = read_meta(meta, r, READER_BID, now, lambda);
        const float* qb = S.q + ((int64_t)m.copy * S.N + r) * (L * d);
        float g = 1.0f;
        out[k * (L + 1)] = ok ? S.p0[r * d + c] : __builtin_nanf("");
        for (int64_t i = 1; i <= L; ++i) {
            g *= m.g;
            out[k * (L + 1) + i] = ok ? qb[(i - 1) * d + c] * g : __builtin_nanf("");
        }
    }
}

int launch_gather_elems(const tpnet_state& st, const int64_t* rows, const int64_t* cols, int64_t n, double now,
                        double lambda, float* out, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    hipLaunchKernelGGL(k_gather_elems, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, st, rows, cols, n, now, lambda, out);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

// row exchange of the sharded state: out[k][i][:] = P[i+1][ids[k]] at `now` (decay applied), i = 0..L-1
__global__ void k_pack_rows(tpnet_state S, const int64_t* __restrict__ ids, int64_t n, double now, double lambda,
                            float* __restrict__ out) {
    const int64_t d = S.d, L = S.L;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(S.meta);
    for (int64_t k = blockIdx.x; k < n; k += gridDim.x) {
        int64_t id = ids[k];
        if ((uint64_t)id >= (uint64_t)S.N) {
            if (threadIdx.x == 0) atomicAdd(S.err, 1u);
            id = 0;
        }
        const MetaView m = read_meta(meta, id, READER_BID, now, lambda);
        const float* qb = S.q + ((int64_t)m.copy * S.N + id) * (L * d);
        for (int64_t r = threadIdx.x; r < L * d; r += blockDim.x) {
            const int64_t i = r / d;
            float g = m.g;
            for (int64_t z = 0; z < i; ++z) g *= m.g;
            out[k * L * d + r] = qb[r] * g;
        }
    }
}

// the inverse: the current copy of row ids[k] <- in[k], expressed at `now` (rows owned by another rank: never the
// target of a local update, so no launch of this rank rewrites them concurrently)
__global__ void k_unpack_rows(tpnet_state S, const int64_t* __restrict__ ids, int64_t n, double now,
                              const float* __restrict__ in) {
    const int64_t d = S.d, L = S.L;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    for (int64_t k = blockIdx.x; k < n; k += gridDim.x) {
        const int64_t id = ids[k];
        if ((uint64_t)id >= (uint64_t)S.N) {
            if (threadIdx.x == 0) atomicAdd(S.err, 1u);
            continue;
        }
        const int c = (int)(meta[id].ver & 1u);
        float* qb = S.q + ((int64_t)c * S.N + id) * (L * d);
        for (int64_t r = threadIdx.x; r < L * d; r += blockDim.x) qb[r] = in[k * L * d + r];
        if (threadIdx.x == 0) meta[id].tref[c] = now;
    }
}

// all peers in one launch: ids = the batch's touched nodes ordered by (owner, node); recv = [G][maxc][L*d] as the
// all-gather delivered it; offs[r] = start of owner r's run inside ids.  Rows owned by `me` are skipped.
__global__ void k_unpack_gathered(tpnet_state S, const int64_t* __restrict__ ids, int64_t n, double now,
                                  const float* __restrict__ recv, int64_t maxc, const int64_t* __restrict__ offs, int G,
                                  int me) {
    const int64_t d = S.d, L = S.L;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    for (int64_t k = blockIdx.x; k < n; k += gridDim.x) {
        const int64_t id = ids[k];
        if ((uint64_t)id >= (uint64_t)S.N) {
            if (threadIdx.x == 0) atomicAdd(S.err, 1u);
            continue;
        }
        const int r = (int)(id % G);
        if (r == me) continue;
        const int64_t slot = k - offs[r];
        const float* in = recv + ((int64_t)r * maxc + slot) * (L * d);
        const int c = (int)(meta[id].ver & 1u);
        float* qb = S.q + ((int64_t)c * S.N + id) * (L * d);
        for (int64_t x = threadIdx.x; x < L * d; x += blockDim.x) qb[x] = in[x];
        if (threadIdx.x == 0) meta[id].tref[c] = now;
    }
}

// compact row shards (tpnet_amd/sharded.py): whole bundles, layer 0 included -- a rank holds ONLY its own rows plus a halo of
// the rows the current batch reads from other ranks, so the static layer travels with the others.
//   pack:   out[k][i][:] = P[i][ids[k]] at `now` (decay applied), i = 0..L
__global__ void k_pack_bundles(tpnet_state S, const int64_t* __restrict__ ids, int64_t n, double now, double lambda,
                               float* __restrict__ out) {
    const int64_t d = S.d, L = S.L;
    const NodeMeta* meta = reinterpret_cast<const NodeMeta*>(S.meta);
    for (int64_t k = blockIdx.x; k < n; k += gridDim.x) {
        int64_t id = ids[k];
        if ((uint64_t)id >= (uint64_t)S.N) {
            if (threadIdx.x == 0) atomicAdd(S.err, 1u);
            id = 0;
        }
        const MetaView m = read_meta(meta, id, READER_BID, now, lambda);
        const float* qb = S.q + ((int64_t)m.copy * S.N + id) * (L * d);
        float* o = out + k * (L + 1) * d;
        for (int64_t r = threadIdx.x; r < (L + 1) * d; r += blockDim.x) {
            const int64_t i = r / d;
            if (i == 0) {
                o[r] = S.p0[id * d + r];
            } else {
                float g = m.g;
                for (int64_t z = 1; z < i; ++z) g *= m.g;
                o[r] = qb[r - d] * g;
            }
        }
    }
}

//   unpack: the gathered bundles of one batch -> the halo rows of the local table.  List entry k (the batch's touched nodes
//   ordered by (owner, node)) of owner r sits at recv[r][k - offs[r]]; it goes to local row local_ids[k] (< 0: skip --
//   the rank's own nodes).  Layer 0 into p0, layers 1..L into the row's current copy, expressed at `now`.
__global__ void k_unpack_bundles(tpnet_state S, const int64_t* __restrict__ local_ids, int64_t n, double now,
                                 const float* __restrict__ recv, int64_t maxc, const int64_t* __restrict__ offs, int G) {
    const int64_t d = S.d, L = S.L;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    for (int64_t k = blockIdx.x; k < n; k += gridDim.x) {
        const int64_t id = local_ids[k];
        if (id < 0) continue;
        if (id >= S.N) {
            if (threadIdx.x == 0) atomicAdd(S.err, 1u);
            continue;
        }
        int r = 0;
        for (int z = 1; z < G; ++z) r = (offs[z] <= k) ? z : r;
        const float* in = recv + ((int64_t)r * maxc + (k - offs[r])) * ((L + 1) * d);
        const int c = (int)(meta[id].ver & 1u);
        float* qb = S.q + ((int64_t)c * S.N + id) * (L * d);
        for (int64_t x = threadIdx.x; x < (L + 1) * d; x += blockDim.x) {
            if (x < d) S.p0[id * d + x] = in[x]; else qb[x - d] = in[x];
        }
        if (threadIdx.x == 0) meta[id].tref[c] = now;
    }
}

// pack for the TARGETED exchange: layer 0 and layers 1..L of row ids[k] go to two buffers (what the peers receive straight
// into the halo rows of their p0 and q arrays: no unpack launch on the other side); the blocks behind the n rows stamp the
// n_halo halo rows this rank is about to receive as "as of `now`" (the received layers are decayed to the batch's clock)
__global__ void k_pack_split(tpnet_state S, const int64_t* __restrict__ ids, int64_t n, double now, double lambda,
                             float* __restrict__ out_p0, float* __restrict__ out_q, int64_t halo0, int64_t n_halo) {
    const int64_t d = S.d, L = S.L;
    NodeMeta* meta = reinterpret_cast<NodeMeta*>(S.meta);
    if ((int64_t)blockIdx.x >= n) {
        const int64_t j = ((int64_t)blockIdx.x - n) * blockDim.x + threadIdx.x;
        if (j < n_halo && halo0 + j < S.N) {
            NodeMeta* m = meta + halo0 + j;
            m->tref[m->ver & 1u] = now;
        }
        return;
    }
    const int64_t k = blockIdx.x;
    int64_t id = ids[k];
    if ((uint64_t)id >= (uint64_t)S.N) {
        if (threadIdx.x == 0) atomicAdd(S.err, 1u);
        id = 0;
    }
    const MetaView m = read_meta(meta, id, READER_BID, now, lambda);
    const float* qb = S.q + ((int64_t)m.copy * S.N + id) * (L * d);
    for (int64_t r = threadIdx.x; r < d; r += blockDim.x) out_p0[k * d + r] = S.p0[id * d + r];
    for (int64_t r = threadIdx.x; r < L * d; r += blockDim.x) {
        const int64_t i = r / d;
        float g = m.g;
        for (int64_t z = 0; z < i; ++z) g *= m.g;
        out_q[k * (L * d) + r] = qb[r] * g;
    }
}

int launch_pack_split(const tpnet_state& st, const int64_t* ids, int64_t n, double now, double lambda, float* out_p0,
                      float* out_q, int64_t halo0, int64_t n_halo, hipStream_t s) {
    const int64_t hb = (n_halo + 255) / 256;
    if (n + hb == 0) return TPNET_OK;
    if (n + hb > 0x7FFFFFFF) return TPNET_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_pack_split, dim3((unsigned)(n + hb)), dim3(256), 0, s, st, ids, n, now, lambda, out_p0, out_q, halo0,
                       n_halo);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_pack_bundles(const tpnet_state& st, const int64_t* ids, int64_t n, double now, double lambda, float* out,
                        hipStream_t s) {
    if (n == 0) return TPNET_OK;
    hipLaunchKernelGGL(k_pack_bundles, dim3(grid_for(n, 1, 8192)), dim3(256), 0, s, st, ids, n, now, lambda, out);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_unpack_bundles(const tpnet_state& st, const int64_t* local_ids, int64_t n, double now, const float* recv,
                          int64_t maxc, const int64_t* offs, int G, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    hipLaunchKernelGGL(k_unpack_bundles, dim3(grid_for(n, 1, 8192)), dim3(256), 0, s, st, local_ids, n, now, recv, maxc,
                       offs, G);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------------------
int launch_unpack_gathered(const tpnet_state& st, const int64_t* ids, int64_t n, double now, const float* recv,
                           int64_t maxc, const int64_t* offs, int G, int me, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    hipLaunchKernelGGL(k_unpack_gathered, dim3(grid_for(n, 1, 8192)), dim3(256), 0, s, st, ids, n, now, recv, maxc, offs,
                       G, me);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_pack_rows(const tpnet_state& st, const int64_t* ids, int64_t n, double now, double lambda, float* out,
                     hipStream_t s) {
    if (n == 0) return TPNET_OK;
    hipLaunchKernelGGL(k_pack_rows, dim3(grid_for(n, 1, 8192)), dim3(256), 0, s, st, ids, n, now, lambda, out);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_unpack_rows(const tpnet_state& st, const int64_t* ids, int64_t n, double now, const float* in, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    hipLaunchKernelGGL(k_unpack_rows, dim3(grid_for(n, 1, 8192)), dim3(256), 0, s, st, ids, n, now, in);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_state_init(const tpnet_state& st, double t0, hipStream_t s) {
    const int64_t nq = 2 * st.N * (int64_t)st.L * st.d;
    hipLaunchKernelGGL(k_state_init, dim3(grid_for(nq, 256 * 4, 4096)), dim3(256), 0, s, st, t0);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_import(const tpnet_state& st, const float* const* layers, double now, hipStream_t s) {
    LayerPtrs lp{};
    for (int i = 0; i < st.L; ++i) lp.p[i] = const_cast<float*>(layers[i]);
    hipLaunchKernelGGL(k_import, dim3(grid_for(st.N * (int64_t)st.L * st.d, 256 * 4, 4096)), dim3(256), 0, s, st, lp,
                       now);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_export(const tpnet_state& st, float* const* layers, double now, double lambda, hipStream_t s) {
    LayerPtrs lp{};
    for (int i = 0; i < st.L; ++i) lp.p[i] = layers[i];
    hipLaunchKernelGGL(k_export, dim3(grid_for(st.N * (int64_t)st.L * st.d, 256 * 4, 4096)), dim3(256), 0, s, st, lp,
                       now, lambda);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_decay(const tpnet_state& st, const float* factors_host, double t_new, hipStream_t s) {
    DecayFactors df{};
    for (int i = 0; i < st.L; ++i) df.f[i] = factors_host[i];
    hipLaunchKernelGGL(k_decay, dim3(grid_for(st.N, 1, 8192)), dim3(256), 0, s, st, df, t_new);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_decay_desc(const tpnet_state& st, const Plan& p, int64_t b, hipStream_t s) {
    hipLaunchKernelGGL(k_decay_desc, dim3(grid_for(st.N, 1, 8192)), dim3(256), 0, s, st, p.desc, b);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_gather_rows(const tpnet_state& st, const int64_t* ids, int64_t n, double now, double lambda, float* out,
                       hipStream_t s) {
    if (n == 0) return TPNET_OK;
    hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(n, 1, 8192)), dim3(256), 0, s, st, ids, n, now, lambda, out);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

// element-wise tail of the readout on a buffer of raw Gram entries (same two operations, in the same order, as the
// fused store of gram_pair)
__global__ void k_gram_finish(float* __restrict__ x, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = x[i];
        v = (v < 0.0f) ? 0.0f : v;
        x[i] = logf(v + 1.0f);
    }
}

// packed rows (TPNET_FLAG_PACKED: NN(NN+1)/2 raw entries a <= b) -> full [NN*NN] feature rows, with the element-wise tail
__global__ void k_gram_unpack(const float* __restrict__ packed, int64_t n, int NN, int do_scale, float* __restrict__ out) {
    const int NG = NN * NN, NT = NN * (NN + 1) / 2;
    const int64_t total = n * NG;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t p = i / NG;
        const int idx = (int)(i - p * NG);
        int a = idx / NN, b = idx - a * NN;
        if (a > b) { const int z = a; a = b; b = z; }
        float v = packed[p * NT + a * NN - (a * (a - 1)) / 2 + (b - a)];
        if (do_scale) {
            v = (v < 0.0f) ? 0.0f : v;
            v = logf(v + 1.0f);
        }
        out[i] = v;
    }
}

int launch_gram_unpack(const float* packed, int64_t n, int L, uint32_t flags, float* out, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    const int NN = 2 * L + 2;
    hipLaunchKernelGGL(k_gram_unpack, dim3(grid_for(n * NN * NN, 256, 256 * 16)), dim3(256), 0, s, packed, n, NN,
                       (flags & TPNET_FLAG_NOT_SCALE) ? 0 : 1, out);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}

int launch_gram_finish(float* x, int64_t n, hipStream_t s) {
    if (n == 0) return TPNET_OK;
    hipLaunchKernelGGL(k_gram_finish, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, x, n);
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}


}  // namespace tpnet
