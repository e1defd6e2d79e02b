// Instantiations of sse::sweep_fast_kernel (sse_fast.hip.h): the diagonal-pass launch of the headline geometry.
#include "sse_device.hip.h"
namespace sse {
template <int K, int PHASE, bool LABEL, bool COMPACT>
static hipError_t launch_fast_one(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_fast_kernel<K, PHASE, LABEL, COMPACT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sweep_fast_kernel<K, PHASE, LABEL, COMPACT>), dim3(B.R), dim3(256), c.lds_bytes, c.stream, B, A);
    return hipGetLastError();
}
hipError_t launch_sweep_fast(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A) {
    if (c.W != 4 || c.mode != SSE_MODE_LDS_EDGES) return hipErrorInvalidValue;
    const bool label = (A.domask & SSE_DO_LABEL) != 0u; // segment labelling for the cluster update of the same timestep
    if (label && !B.lite) return hipErrorInvalidValue;
    const bool compact = (A.domask & SSE_DO_COMPACT) != 0u; // dense op list for the cluster update of the same timestep
    if (compact && (!B.cops || label)) return hipErrorInvalidValue;
    if (c.K == 4) {
        if (label) return c.phase ? launch_fast_one<4, 1, true, false>(c, B, A) : launch_fast_one<4, 0, true, false>(c, B, A);
        if (compact) return c.phase ? launch_fast_one<4, 1, false, true>(c, B, A) : launch_fast_one<4, 0, false, true>(c, B, A);
        return c.phase ? launch_fast_one<4, 1, false, false>(c, B, A) : launch_fast_one<4, 0, false, false>(c, B, A);
    }
    if (c.K == 2) {
        if (label) return launch_fast_one<2, 0, true, false>(c, B, A);
        return compact ? launch_fast_one<2, 0, false, true>(c, B, A) : launch_fast_one<2, 0, false, false>(c, B, A);
    }
    return hipErrorInvalidValue;
}
} // namespace sse
