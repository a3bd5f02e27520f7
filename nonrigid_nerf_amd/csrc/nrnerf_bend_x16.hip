// nrnerf_bend_x16.hip -- instantiations of the 16x16x32 stand-alone bender (nrnerf_bend_x16.h): the reference's 5 x 64 bender (arch 0)
// and the 7 x 64 one of BASELINE config 4 (arch 1); f16 operands ("bf16" mode's single-product bender).
#include "nrnerf_bend_x16.h"
#include "nrnerf_x16_api.h"

namespace nrn {
hipError_t launch_bend_x16(int arch, const BendArgs& a, int num_cus, hipStream_t stream) {
    if (arch == 0) return launch_bend_x16_t<ArchDefault>(a, num_cus, stream);
    if (arch == 1) return launch_bend_x16_t<ArchDeepBend>(a, num_cus, stream);
    return hipErrorInvalidValue;
}
}  // namespace nrn
