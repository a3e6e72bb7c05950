// gfx950 fused per-batch step: the 256-thread instantiation of step_impl.hpp WITH the edge-fused update path (plans of
// batches > 1024 edges: a target's lone contribution is applied by its edge's readout group).
#include "step_impl.hpp"

namespace tpnet {

template int launch_step_bs<BLOCK_SMALL, true>(const tpnet_state&, const StreamArgs&, const Plan&, int64_t, int64_t,
                                               int32_t, double, uint32_t, uint32_t, hipStream_t);

}  // namespace tpnet
