// Instantiations of sse::sweep_fast_kernel (sse_fast.hip.h): the diagonal-pass launch of the headline geometry.
#include "sse_device.hip.h"
namespace sse {
template <int K, int PHASE, bool LABEL>
static hipError_t launch_fast_one(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_fast_kernel<K, PHASE, LABEL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sweep_fast_kernel<K, PHASE, LABEL>), dim3(B.R), dim3(256), c.lds_bytes, c.stream, B, A);
    return hipGetLastError();
}
hipError_t launch_sweep_fast(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A) {
    if (c.W != 4 || c.mode != SSE_MODE_LDS_EDGES) return hipErrorInvalidValue;
    const bool label = (A.domask & SSE_DO_LABEL) != 0u; // segment labelling for the cluster update of the same timestep
    if (label && !B.lite) return hipErrorInvalidValue;
    if (c.K == 4) {
        if (label) return c.phase ? launch_fast_one<4, 1, true>(c, B, A) : launch_fast_one<4, 0, true>(c, B, A);
        return c.phase ? launch_fast_one<4, 1, false>(c, B, A) : launch_fast_one<4, 0, false>(c, B, A);
    }
    if (c.K == 2) return label ? launch_fast_one<2, 0, true>(c, B, A) : launch_fast_one<2, 0, false>(c, B, A);
    return hipErrorInvalidValue;
}
} // namespace sse
