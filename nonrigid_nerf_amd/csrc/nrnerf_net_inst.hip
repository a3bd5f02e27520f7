// nrnerf_net_inst.hip -- one instantiation of the network kernel per translation unit, so the variants
// (precision x bender x view-dependent head) compile in parallel.  Build with
//   -DNRN_POL=PolBF16 -DNRN_BEND=1 -DNRN_VIEWS=0 -DNRN_WAVES=8 -DNRN_MB=1 -DNRN_NAME=launch_net_bf16_bend
// NRN_MB > 1 selects the multi-block kernel (nrnerf_net_mb_impl.h: NRN_MB blocks of 32 samples per wave).
#ifndef NRN_MB
#define NRN_MB 1
#endif
#if NRN_MB > 1
#include "nrnerf_net_mb_impl.h"
#else
#include "nrnerf_net_impl.h"
#endif

namespace nrn {
hipError_t NRN_NAME(const NetArgs& a, int num_cus, hipStream_t stream) {
#if NRN_MB > 1
    static_assert(NRN_VIEWS == 0, "the multi-block kernel has no view-dependent head");
    return launch_one_mb<NRN_POL, ArchDefault, (NRN_BEND != 0), NRN_WAVES, NRN_MB>(a, num_cus, stream);
#else
    return launch_one<NRN_POL, ArchDefault, (NRN_BEND != 0), (NRN_VIEWS != 0), NRN_WAVES>(a, num_cus, stream);
#endif
}
}  // namespace nrn
