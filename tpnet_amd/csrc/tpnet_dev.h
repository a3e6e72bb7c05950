/* tpnet_dev.h -- measurement aids exported by libtpnet_hip.so that are NOT part of the drop-in boundary
 * (include/tpnet_hip.h): bench.py and the tools under tools/ bind them; a binding of the reference never needs them. */
#ifndef TPNET_DEV_H
#define TPNET_DEV_H
#include "../../include/tpnet_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Timing aid for bench.py (not part of the drop-in surface): elapsed milliseconds of `reps` back-to-back tpnet_run_stream
 * calls measured with hipEvents recorded on `stream` (the stream the kernels run on).  The state is advanced `reps` times;
 * the caller resets it.  For the LAST rep, hipEvent pairs around each chunk's loop of launches of the dominant kernel (planning
 * kernels and the write-back excluded) give: kernel_ms_out = their summed time / the number of launches, i.e. the average
 * launch PERIOD (kernel duration + inter-kernel boundary) of k_step (per-batch schedule: one launch per batch) or k_wpipe
 * (windowed schedule: one launch per pipeline step); launches_out = those launches; edges_out = the edges they covered
 * (up to 256 chunks).  Any of the out pointers may be NULL. */
int tpnet_time_stream(const tpnet_state* st, const int64_t* src, const int64_t* dst, const int64_t* neg,
                      const double* t, int64_t E, int64_t batch, double now_time, double lambda,
                      uint32_t launch_id_base, uint32_t flags, float* out_pos, float* out_neg,
                      void* workspace, size_t ws_bytes, int reps, float* total_ms_out, float* kernel_ms_out,
                      int64_t* launches_out, int64_t* edges_out, void* stream);

/* tpnet_rows_stream_targeted (same arguments) with three HIP events per batch on `stream` -- before the batch's exchange (pack +
 * grouped send / recv), before its step kernel, behind it -- and one synchronise at the end: total_ms_out = first event to last,
 * step_ms_out / exchange_ms_out = the average per batch of the step launch / of pack + exchange.  bench.py's `roofline` at N > 1
 * (every rank calls it: the exchange is collective).  At most 4096 batches. */
int tpnet_time_rows_stream_targeted(const tpnet_state* st, void* comm, const int64_t* pack_ids, const int64_t* pack_start,
                                    float* send_p0, float* send_q, const int64_t* send_cnt, const int64_t* recv_cnt, int32_t G,
                                    int32_t me, double now_time, const double* t_last, const int64_t* src, const int64_t* dst,
                                    const int64_t* neg, const double* t, int64_t E, int64_t batch, int64_t b0, int64_t b1,
                                    double lambda, uint32_t launch_id_base, uint32_t flags, int32_t n_owned, float* out_pos,
                                    float* out_neg, void* workspace, size_t ws_bytes, void* stream, float* total_ms_out,
                                    float* step_ms_out, float* exchange_ms_out);

/* tpnet_wshard_run with HIP events on `stream` around every pipeline step's launch and around its pack + exchange + unpack; one
 * synchronise at the end.  total_ms_out = the whole call (the chunk's halo exchange included); launch_ms_out / exchange_ms_out =
 * averages per step. */
int tpnet_time_wshard_run(tpnet_wshard* w, void* comm, float* out_pos, float* out_neg, uint32_t launch_id, void* stream,
                          float* total_ms_out, float* launch_ms_out, float* exchange_ms_out);

#ifdef __cplusplus
}
#endif
#endif /* TPNET_DEV_H */
