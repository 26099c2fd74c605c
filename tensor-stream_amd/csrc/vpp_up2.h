// vpp_up2.h -- launcher of the streaming 1 : 2 up-scale kernel (vpp_bilinear_up2.hip), shared with the launch selection (vpp_select.hip) only.
#pragma once
#include "vpp_kernels.h"

namespace tsvpp {

// d.r32 == 10: BILINEAR at exactly 1 : 2 on both axes (launch_fused)
hipError_t launch_bilinear_up2(OutKind out, const LaunchDesc &d, const FrameTable &t, hipStream_t stream, LaunchInfo *info);

} // namespace tsvpp
