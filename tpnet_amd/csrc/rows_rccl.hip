// Row-sharded stream, one batch per FFI call (include/tpnet_hip.h: tpnet_rccl_*, tpnet_rows_step): pack the owned rows the
// batch touches -> ONE RCCL all-gather over xGMI -> unpack the other ranks' rows into the halo -> the fused step on the local
// table, all enqueued from C on the caller's stream.  The Python runner (tpnet_amd/sharded.py) issued these as three ctypes
// calls plus a torch.distributed collective per batch; the collective's own bookkeeping was most of the host time.
// RCCL is resolved at run time from the library the process already has (PyTorch's librccl.so): no link-time dependency, and
// a build without RCCL still loads.  The communicator is created here from a 128-byte unique id that the caller broadcasts
// with whatever channel it has (torch.distributed in tpnet_amd/sharded.py).
#include "tpnet_common.h"

#include <dlfcn.h>
#include <cstring>
#include <vector>

namespace tpnet {

struct RcclApi {
    void* handle = nullptr;
    int (*get_unique_id)(void*) = nullptr;                                         // ncclGetUniqueId(ncclUniqueId*)
    int (*comm_init_rank)(void**, int, void*, int) = nullptr;                      // resolved through a by-value shim below
    int (*comm_destroy)(void*) = nullptr;
    int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;      // ncclSend(buf, count, type, peer, comm, stream)
    int (*recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;            // ncclRecv
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
    void* init_sym = nullptr;
};

struct UniqueId { char internal[128]; };      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed BY VALUE to ncclCommInitRank
static constexpr int kNcclFloat32 = 7;        // rccl.h: ncclFloat32 = 7

static RcclApi g_rccl;

static int rccl_load(const char* lib_path) {
    if (g_rccl.handle) return TPNET_OK;
    void* h = nullptr;
    if (lib_path && lib_path[0]) h = dlopen(lib_path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return TPNET_ERR_NO_DEVICE;
    g_rccl.get_unique_id = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.init_sym = dlsym(h, "ncclCommInitRank");
    g_rccl.comm_destroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.all_gather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(h, "ncclAllGather"));
    g_rccl.send = reinterpret_cast<int (*)(const void*, size_t, int, int, void*, hipStream_t)>(dlsym(h, "ncclSend"));
    g_rccl.recv = reinterpret_cast<int (*)(void*, size_t, int, int, void*, hipStream_t)>(dlsym(h, "ncclRecv"));
    g_rccl.group_start = reinterpret_cast<int (*)()>(dlsym(h, "ncclGroupStart"));
    g_rccl.group_end = reinterpret_cast<int (*)()>(dlsym(h, "ncclGroupEnd"));
    if (!g_rccl.get_unique_id || !g_rccl.init_sym || !g_rccl.comm_destroy || !g_rccl.all_gather || !g_rccl.send || !g_rccl.recv ||
        !g_rccl.group_start || !g_rccl.group_end) {
        dlclose(h);
        return TPNET_ERR_NO_DEVICE;
    }
    g_rccl.handle = h;
    return TPNET_OK;
}

// wshard.hip calls the exchange through the same entry points (false until a communicator has been created here)
struct WsRccl {
    int (*send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
};
bool rows_rccl_api(WsRccl* out) {
    if (!g_rccl.handle || !out) return false;
    out->send = g_rccl.send;
    out->recv = g_rccl.recv;
    out->group_start = g_rccl.group_start;
    out->group_end = g_rccl.group_end;
    return true;
}

}  // namespace tpnet

using namespace tpnet;

extern "C" {

int tpnet_rccl_unique_id(const char* lib_path, void* id128) {
    if (!id128) return TPNET_ERR_BAD_ARG;
    int rc = rccl_load(lib_path);
    if (rc) return rc;
    UniqueId id;
    memset(&id, 0, sizeof(id));
    if (g_rccl.get_unique_id(&id) != 0) return TPNET_ERR_HIP;
    memcpy(id128, &id, sizeof(id));
    return TPNET_OK;
}

int tpnet_rccl_comm_create(const char* lib_path, const void* id128, int32_t nranks, int32_t rank, void** comm) {
    if (!id128 || !comm || nranks < 1 || rank < 0 || rank >= nranks) return TPNET_ERR_BAD_ARG;
    int rc = rccl_load(lib_path);
    if (rc) return rc;
    UniqueId id;
    memcpy(&id, id128, sizeof(id));
    auto init = reinterpret_cast<int (*)(void**, int, UniqueId, int)>(g_rccl.init_sym);
    void* c = nullptr;
    if (init(&c, nranks, id, rank) != 0 || !c) return TPNET_ERR_HIP;
    *comm = c;
    return TPNET_OK;
}

int tpnet_rccl_comm_destroy(void* comm) {
    if (!comm) return TPNET_OK;
    if (!g_rccl.handle) return TPNET_ERR_BAD_ARG;
    return g_rccl.comm_destroy(comm) == 0 ? TPNET_OK : TPNET_ERR_HIP;
}

int tpnet_rows_step(const tpnet_state* st, void* comm, const int64_t* pack_ids, int64_t n_pack, float* send, float* recv,
                    int64_t maxc, const int64_t* unpack_ids, int64_t n_unpack, const int64_t* offs, int32_t G,
                    double now_time, const int64_t* src, const int64_t* dst, const int64_t* neg, const double* t, int64_t E,
                    int64_t batch, int64_t b, double lambda, uint32_t launch_id, uint32_t flags, int32_t n_owned,
                    float* out_pos, float* out_neg, void* workspace, size_t ws_bytes, void* stream) {
    if (!st || !st->p0 || !st->q || !st->meta || !st->err || st->N < 1 || st->d < 1 || st->L < 1 || st->L > TPNET_MAX_LAYERS)
        return TPNET_ERR_BAD_ARG;
    if (G < 1 || maxc < 0 || n_pack < 0 || n_pack > maxc || n_unpack < 0) return TPNET_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (comm && maxc > 0) {
        if (!g_rccl.handle || !pack_ids || !send || !recv || !unpack_ids || !offs) return TPNET_ERR_BAD_ARG;
        const size_t bundle = (size_t)(st->L + 1) * (size_t)st->d;
        rc = launch_pack_bundles(*st, pack_ids, n_pack, now_time, lambda, send, s);
        if (rc) return rc;
        // recv = [G][maxc][bundle]: every rank's padded block, in rank order (same layout as all_gather_into_tensor)
        if (g_rccl.all_gather(send, recv, (size_t)maxc * bundle, kNcclFloat32, comm, s) != 0) return TPNET_ERR_HIP;
        rc = launch_unpack_bundles(*st, unpack_ids, n_unpack, now_time, recv, maxc, offs, G, s);
        if (rc) return rc;
    }
    return tpnet_step_batch(st, src, dst, neg, t, E, batch, b, lambda, launch_id, flags, 0, n_owned, out_pos, out_neg,
                            workspace, ws_bytes, stream);
}

int tpnet_pack_split(const tpnet_state* st, const int64_t* ids, int64_t n, double now_time, double lambda, float* out_p0,
                     float* out_q, int64_t halo0, int64_t n_halo, void* stream) {
    if (!st || !st->p0 || !st->q || !st->meta || !st->err || st->N < 1 || st->d < 1 || st->L < 1 || st->L > TPNET_MAX_LAYERS)
        return TPNET_ERR_BAD_ARG;
    if (n < 0 || n_halo < 0 || halo0 < 0 || halo0 + n_halo > st->N || (n > 0 && (!ids || !out_p0 || !out_q))) return TPNET_ERR_BAD_ARG;
    return launch_pack_split(*st, ids, n, now_time, lambda, out_p0, out_q, halo0, n_halo, (hipStream_t)stream);
}

// the exchange of one batch (pack -> grouped ncclSend / ncclRecv into the halo rows); nothing to do without peers
static int rows_exchange_targeted(const tpnet_state* st, void* comm, const int64_t* pack_ids, float* send_p0, float* send_q,
                                  const int64_t* send_cnt, const int64_t* recv_cnt, int32_t G, int32_t me, double now_time,
                                  double lambda, int32_t n_owned, hipStream_t s) {
    if (!(comm && G > 1)) return TPNET_OK;
    if (!g_rccl.handle || !send_cnt || !recv_cnt) return TPNET_ERR_BAD_ARG;
    int64_t ns = 0, nr = 0;
    for (int r = 0; r < G; ++r) {
        if (send_cnt[r] < 0 || recv_cnt[r] < 0 || (r == me && (send_cnt[r] || recv_cnt[r]))) return TPNET_ERR_BAD_ARG;
        ns += send_cnt[r];
        nr += recv_cnt[r];
    }
    if (n_owned + nr > st->N) return TPNET_ERR_WORKSPACE;            // more rows to receive than the shard has halo rows
    if (ns > 0 && (!pack_ids || !send_p0 || !send_q)) return TPNET_ERR_BAD_ARG;
    int rc = launch_pack_split(*st, pack_ids, ns, now_time, lambda, send_p0, send_q, n_owned, nr, s);
    if (rc) return rc;
    // every row travels only to the ranks that read it, and lands where the step kernel reads it: the halo rows of p0 and
    // of copy 0 of q (halo rows are never targets, so their current copy stays 0)
    const size_t d = (size_t)st->d, Ld = (size_t)st->L * d;
    float* halo_p0 = st->p0 + (size_t)n_owned * d;
    float* halo_q = st->q + (size_t)n_owned * Ld;
    int bad = 0;
    bad |= g_rccl.group_start();
    size_t so = 0, ro = 0;
    for (int r = 0; r < G; ++r) {
        if (send_cnt[r]) {
            bad |= g_rccl.send(send_p0 + so * d, (size_t)send_cnt[r] * d, kNcclFloat32, r, comm, s);
            bad |= g_rccl.send(send_q + so * Ld, (size_t)send_cnt[r] * Ld, kNcclFloat32, r, comm, s);
            so += (size_t)send_cnt[r];
        }
        if (recv_cnt[r]) {
            bad |= g_rccl.recv(halo_p0 + ro * d, (size_t)recv_cnt[r] * d, kNcclFloat32, r, comm, s);
            bad |= g_rccl.recv(halo_q + ro * Ld, (size_t)recv_cnt[r] * Ld, kNcclFloat32, r, comm, s);
            ro += (size_t)recv_cnt[r];
        }
    }
    bad |= g_rccl.group_end();
    return bad ? TPNET_ERR_HIP : TPNET_OK;
}

int tpnet_rows_step_targeted(const tpnet_state* st, void* comm, const int64_t* pack_ids, float* send_p0, float* send_q,
                             const int64_t* send_cnt, const int64_t* recv_cnt, int32_t G, int32_t me, double now_time,
                             const int64_t* src, const int64_t* dst, const int64_t* neg, const double* t, int64_t E,
                             int64_t batch, int64_t b, double lambda, uint32_t launch_id, uint32_t flags, int32_t n_owned,
                             float* out_pos, float* out_neg, void* workspace, size_t ws_bytes, void* stream) {
    if (!st || !st->p0 || !st->q || !st->meta || !st->err || st->N < 1 || st->d < 1 || st->L < 1 || st->L > TPNET_MAX_LAYERS)
        return TPNET_ERR_BAD_ARG;
    if (G < 1 || me < 0 || me >= G || n_owned < 0 || n_owned > st->N) return TPNET_ERR_BAD_ARG;
    const int rc = rows_exchange_targeted(st, comm, pack_ids, send_p0, send_q, send_cnt, recv_cnt, G, me, now_time, lambda, n_owned,
                                          (hipStream_t)stream);
    if (rc) return rc;
    return tpnet_step_batch(st, src, dst, neg, t, E, batch, b, lambda, launch_id, flags, 0, n_owned, out_pos, out_neg,
                            workspace, ws_bytes, stream);
}

// batches [b0, b1) of a prepared stream; ev (optional): 3 events per batch -- before the exchange, before the step, behind it
static int rows_stream_loop(const tpnet_state* st, void* comm, const int64_t* pack_ids, const int64_t* pack_start, float* send_p0,
                            float* send_q, const int64_t* send_cnt, const int64_t* recv_cnt, int32_t G, int32_t me, double now_time,
                            const double* t_last, const int64_t* src, const int64_t* dst, const int64_t* neg, const double* t,
                            int64_t E, int64_t batch, int64_t b0, int64_t b1, double lambda, uint32_t launch_id_base, uint32_t flags,
                            int32_t n_owned, float* out_pos, float* out_neg, void* workspace, size_t ws_bytes, void* stream,
                            hipEvent_t* ev) {
    const int64_t nb = batch > 0 ? (E + batch - 1) / batch : 0;
    if (batch < 1 || b0 < 0 || b1 > nb || b0 > b1 || !t_last || !send_cnt || !recv_cnt || !pack_start || G < 1) return TPNET_ERR_BAD_ARG;
    if (launch_id_base == 0 || (uint64_t)launch_id_base + (uint64_t)nb >= 0x7FFFFFFFull) return TPNET_ERR_BAD_ARG;
    if (!st || !st->p0 || !st->q || !st->meta || !st->err || st->N < 1 || st->d < 1 || st->L < 1 || st->L > TPNET_MAX_LAYERS)
        return TPNET_ERR_BAD_ARG;
    if (me < 0 || me >= G || n_owned < 0 || n_owned > st->N) return TPNET_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    for (int64_t b = b0; b < b1; ++b) {
        // the clock a batch's rows are packed at is the one the previous batch left (models/TPNet.py:99)
        const double now = b == 0 ? now_time : t_last[b - 1];
        if (ev) (void)hipEventRecord(ev[3 * (b - b0)], s);
        int rc = rows_exchange_targeted(st, comm, pack_ids ? pack_ids + pack_start[b] : nullptr, send_p0, send_q,
                                        send_cnt + (size_t)G * (size_t)b, recv_cnt + (size_t)G * (size_t)b, G, me, now, lambda,
                                        n_owned, s);
        if (rc) return rc;
        if (ev) (void)hipEventRecord(ev[3 * (b - b0) + 1], s);
        rc = tpnet_step_batch(st, src, dst, neg, t, E, batch, b, lambda, launch_id_base + (uint32_t)b, flags, 0, n_owned, out_pos,
                              out_neg, workspace, ws_bytes, stream);
        if (rc) return rc;
        if (ev) (void)hipEventRecord(ev[3 * (b - b0) + 2], s);
    }
    return TPNET_OK;
}

int tpnet_rows_stream_targeted(const tpnet_state* st, void* comm, const int64_t* pack_ids, const int64_t* pack_start,
                               float* send_p0, float* send_q, const int64_t* send_cnt, const int64_t* recv_cnt, int32_t G,
                               int32_t me, double now_time, const double* t_last, const int64_t* src, const int64_t* dst,
                               const int64_t* neg, const double* t, int64_t E, int64_t batch, int64_t b0, int64_t b1,
                               double lambda, uint32_t launch_id_base, uint32_t flags, int32_t n_owned, float* out_pos,
                               float* out_neg, void* workspace, size_t ws_bytes, void* stream) {
    return rows_stream_loop(st, comm, pack_ids, pack_start, send_p0, send_q, send_cnt, recv_cnt, G, me, now_time, t_last, src, dst, neg,
                            t, E, batch, b0, b1, lambda, launch_id_base, flags, n_owned, out_pos, out_neg, workspace, ws_bytes, stream,
                            nullptr);
}

// measurement aid (tpnet_dev.h): the same loop with HIP events on `stream` around every batch's exchange and step
int tpnet_time_rows_stream_targeted(const tpnet_state* st, void* comm, const int64_t* pack_ids, const int64_t* pack_start,
                                    float* send_p0, float* send_q, const int64_t* send_cnt, const int64_t* recv_cnt, int32_t G,
                                    int32_t me, double now_time, const double* t_last, const int64_t* src, const int64_t* dst,
                                    const int64_t* neg, const double* t, int64_t E, int64_t batch, int64_t b0, int64_t b1,
                                    double lambda, uint32_t launch_id_base, uint32_t flags, int32_t n_owned, float* out_pos,
                                    float* out_neg, void* workspace, size_t ws_bytes, void* stream, float* total_ms_out,
                                    float* step_ms_out, float* exchange_ms_out) {
    if (b1 <= b0 || b1 - b0 > 4096) return TPNET_ERR_BAD_ARG;
    const size_t n = (size_t)(b1 - b0);
    std::vector<hipEvent_t> ev(3 * n);
    for (auto& e : ev) TPNET_HIP_TRY(hipEventCreate(&e));
    int rc = rows_stream_loop(st, comm, pack_ids, pack_start, send_p0, send_q, send_cnt, recv_cnt, G, me, now_time, t_last, src, dst,
                              neg, t, E, batch, b0, b1, lambda, launch_id_base, flags, n_owned, out_pos, out_neg, workspace, ws_bytes,
                              stream, ev.data());
    if (rc == TPNET_OK) {
        TPNET_HIP_TRY(hipEventSynchronize(ev[3 * n - 1]));
        double step = 0.0, xch = 0.0;
        float ms = 0.f;
        for (size_t i = 0; i < n; ++i) {
            TPNET_HIP_TRY(hipEventElapsedTime(&ms, ev[3 * i], ev[3 * i + 1]));
            xch += ms;
            TPNET_HIP_TRY(hipEventElapsedTime(&ms, ev[3 * i + 1], ev[3 * i + 2]));
            step += ms;
        }
        TPNET_HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[3 * n - 1]));
        if (total_ms_out) *total_ms_out = ms;
        if (step_ms_out) *step_ms_out = (float)(step / (double)n);
        if (exchange_ms_out) *exchange_ms_out = (float)(xch / (double)n);
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

}  // extern "C"
