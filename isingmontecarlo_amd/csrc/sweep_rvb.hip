// Instantiations of sse::rvb_grow_kernel / sse::rvb_main_kernel (sse_rvb_split.hip.h): the RVB sweep as two launches.
#include "sse_device.hip.h"
#include "sse_rvb_split.hip.h"
namespace sse {
template <bool CL>
static hipError_t launch_grow_one(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rvb_grow_kernel<CL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((rvb_grow_kernel<CL>), dim3(B.R), dim3(1024), c.lds_bytes, c.stream, B, A);
    return hipGetLastError();
}
template <int W, bool CL>
static hipError_t launch_main_one(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rvb_main_kernel<W, CL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((rvb_main_kernel<W, CL>), dim3(B.R), dim3(W * 64), c.lds_bytes, c.stream, B, A);
    return hipGetLastError();
}
size_t rvb_split_grow_fixed_words(uint32_t N, uint32_t nwords, uint32_t ledges) {
    return (size_t)2 * nwords + (N + 3) / 4 + 4 * 16 + 16 + 2 * SSE_MAX_CHUNKS + ledges + rvb_grow_fixed_words(N); // Lds<16>::carve up to o_cur, then rvb_carve_grow
}
size_t rvb_split_main_words(uint32_t W, uint32_t N, uint32_t nwords, uint32_t ledges, uint32_t E, uint32_t Nb) { return rvb_main_words(W, N, nwords, ledges, E, Nb); }
size_t rvb_split_prod_stride(uint32_t Nb) { return rvb_bm_words(Nb) <= SSE_RVB_BM_MAX ? SSE_RVB_PROD_STRIDE + rvb_bm_words(Nb) : 0u; }
hipError_t launch_rvb_grow(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A) {
    if (!B.rvb_prod) return hipErrorInvalidValue;
    if (c.mode == SSE_MODE_LDS_EDGES) return launch_grow_one<true>(c, B, A);
    if (c.mode == SSE_MODE_GENERAL) return launch_grow_one<false>(c, B, A);
    return hipErrorInvalidValue;
}
hipError_t launch_rvb_main(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A) {
    if (!B.rvb_prod) return hipErrorInvalidValue;
    const bool cl = c.mode == SSE_MODE_LDS_EDGES;
    if (!cl && c.mode != SSE_MODE_GENERAL) return hipErrorInvalidValue;
    switch (c.W) {
    case 4: return cl ? launch_main_one<4, true>(c, B, A) : launch_main_one<4, false>(c, B, A);
    case 8: return cl ? launch_main_one<8, true>(c, B, A) : launch_main_one<8, false>(c, B, A);
    case 16: return cl ? launch_main_one<16, true>(c, B, A) : launch_main_one<16, false>(c, B, A);
    default: return hipErrorInvalidValue;
    }
}
} // namespace sse
