// Instantiations of sse::sweep_kernel for W = 6 wave64s per replica (one translation unit per W so that
// the variants compile in parallel).
#include "sse_device.hip.h"
namespace sse {
hipError_t launch_sweep_w6(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A) { return launch_w<6>(c, B, A); }
} // namespace sse
