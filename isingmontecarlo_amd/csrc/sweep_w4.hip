// Instantiations of sse::sweep_kernel for W = 4 wave64s per replica (one translation unit per W so that
// the variants compile in parallel).
#include "sse_device.hip.h"
namespace sse {
hipError_t launch_sweep_w4(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A) { return launch_w<4>(c, B, A); }
} // namespace sse
