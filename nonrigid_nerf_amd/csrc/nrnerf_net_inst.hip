// nrnerf_net_inst.hip -- one instantiation of the network kernel per translation unit, so the variants
// (precision x bender x view-dependent head) compile in parallel.  Build with
//   -DNRN_POL=PolBF16 -DNRN_BEND=1 -DNRN_VIEWS=0 -DNRN_WAVES=8 -DNRN_ARCH=0 -DNRN_NAME=launch_net_a0_bf16_bend
#include "nrnerf_net_impl.h"

namespace nrn {
hipError_t NRN_NAME(const NetArgs& a, int num_cus, hipStream_t stream) {
    return launch_one<NRN_POL, ArchById<NRN_ARCH>::type, (NRN_BEND != 0), (NRN_VIEWS != 0), NRN_WAVES>(a, num_cus, stream);
}
}  // namespace nrn
