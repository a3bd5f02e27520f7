// nrnerf_net_x16.hip -- instantiations of the 16x16x32 trunk-only kernel (nrnerf_net_x16.h): compiled architecture 0's trunk, bf16 and f16.
#include "nrnerf_net_x16.h"

namespace nrn {
hipError_t launch_net_x16(int precision, const NetArgs& a, int num_cus, hipStream_t stream) {
    if (precision == PREC_BF16) return launch_net_x16_t<PolBF16, ArchDefault>(a, num_cus, stream);
    if (precision == PREC_F16) return launch_net_x16_t<PolF16, ArchDefault>(a, num_cus, stream);
    return hipErrorInvalidValue;
}
}  // namespace nrn
