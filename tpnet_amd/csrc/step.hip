// gfx950 fused per-batch step (tpnet_update / tpnet_run_stream / tpnet_step_batch): the 512-thread instantiation of
// step_impl.hpp and the choice of the kernel variant per batch.
#include "step_impl.hpp"

namespace tpnet {

#define TPNET_STEP_ARGS const tpnet_state&, const StreamArgs&, const Plan&, int64_t, int64_t, int32_t, double, uint32_t, \
                        uint32_t, hipStream_t
template int launch_step_bs<BLOCK, false>(TPNET_STEP_ARGS);
extern template int launch_step_bs<BLOCK_SMALL, false>(TPNET_STEP_ARGS);
extern template int launch_step_bs<BLOCK_SMALL, true>(TPNET_STEP_ARGS);

int launch_step(const tpnet_state& st, const StreamArgs& a, const Plan& p, int64_t b, int64_t batch, int32_t ne,
                double lambda, uint32_t launch_id, uint32_t flags, hipStream_t s) {
    // a launch of a fused plan (STEP_FUSE: api.hip turns it on for batches > 1024 edges) takes the variant that carries
    // the edge-fused update path
    if (flags & STEP_FUSE) return launch_step_bs<BLOCK_SMALL, true>(st, a, p, b, batch, ne, lambda, launch_id, flags, s);
    // 512-thread workgroups for a batch that is ONE pass of them -- 256 resident workgroups hold the pairs, the items and
    // 40 hub-layer units of up to ~1000 edges (see BLOCK_SMALL in device_common.hpp for the measurements; just above
    // that the 512-thread kernel falls off a cliff: d=128, B=1100 takes 9.3 us against 7.5 us) -- and 256-thread ones
    // for everything larger and for tiny batches
    static const char* env = TPNET_DEV_STR(BLOCK);                  // developer override: "256" / "512"
    const bool small_wg = env ? env[0] == '2' : (ne > 1024 || ne < 400);
    return small_wg ? launch_step_bs<BLOCK_SMALL, false>(st, a, p, b, batch, ne, lambda, launch_id, flags, s)
                    : launch_step_bs<BLOCK, false>(st, a, p, b, batch, ne, lambda, launch_id, flags, s);
}

}  // namespace tpnet
