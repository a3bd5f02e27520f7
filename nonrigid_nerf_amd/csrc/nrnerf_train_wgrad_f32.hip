// nrnerf_train_wgrad_f32.hip -- the fp32 mode's weight-gradient kernel (trunk_wgrad_f32, nrnerf_train.h) for both trunk widths.
#include "nrnerf_train.h"

namespace nrn {
hipError_t launch_trunk_wgrad_f32(const WgradArgs& a, hipStream_t stream) { return launch_trunk_wgrad_f32<ArchDefault>(a, stream); }
hipError_t launch_trunk_wgrad_f32_a5(const WgradArgs& a, hipStream_t stream) { return launch_trunk_wgrad_f32<ArchNarrow>(a, stream); }
hipError_t launch_trunk_wgrad_f32_views(const WgradArgs& a, hipStream_t stream) { return launch_trunk_wgrad_f32<ArchDefault, true>(a, stream); }
}  // namespace nrn
