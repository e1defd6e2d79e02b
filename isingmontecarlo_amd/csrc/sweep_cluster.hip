// Instantiations of sse::cluster_kernel (sse_cluster.hip.h): the cluster-update launch of the headline geometry.
#include "sse_device.hip.h"
#include "sse_cluster.hip.h"
namespace sse {
template <int K, bool HL, int PHASE>
static hipError_t launch_cluster_one(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&cluster_kernel<K, HL, PHASE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((cluster_kernel<K, HL, PHASE>), dim3(B.R), dim3(SSE_CLW * 64), c.lds_bytes, c.stream, B, A);
    return hipGetLastError();
}
size_t cluster_fixed_words(uint32_t N, uint32_t nwords, uint32_t Nb) { return cl_fixed_words(N, nwords, Nb); }
hipError_t launch_cluster(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A) {
    if (c.mode != SSE_MODE_LDS_EDGES || B.N > SSE_CL_MAX_VARS || !(A.domask & SSE_DO_CLUSTER)) return hipErrorInvalidValue;
    const bool hl = B.has_long != 0u;
    if (c.K == 4) {
        if (hl) return c.phase ? launch_cluster_one<4, true, 1>(c, B, A) : launch_cluster_one<4, true, 0>(c, B, A);
        return c.phase ? launch_cluster_one<4, false, 1>(c, B, A) : launch_cluster_one<4, false, 0>(c, B, A);
    }
    if (c.K == 2) return hl ? launch_cluster_one<2, true, 0>(c, B, A) : launch_cluster_one<2, false, 0>(c, B, A);
    return hipErrorInvalidValue;
}
} // namespace sse
