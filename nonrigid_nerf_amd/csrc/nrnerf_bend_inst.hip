// nrnerf_bend_inst.hip -- one instantiation of the stand-alone bender kernel (nrnerf_bend.h) per translation unit.  Build with
//   -DNRN_POL=PolBF16 -DNRN_ARCH=0 -DNRN_NAME=launch_bend_a0_bf16
#include "nrnerf_bend.h"

namespace nrn {
hipError_t NRN_NAME(const BendArgs& a, int num_cus, hipStream_t stream) {
    return launch_bend_one<NRN_POL, ArchById<NRN_ARCH>::type, (NRN_POL::KH == 1) ? 4 : 8>(a, num_cus, stream);
}
}  // namespace nrn
