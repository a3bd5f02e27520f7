// nrnerf_train_inst.hip -- the two training kernels of nrnerf_train.h for one precision and one trunk (architecture 0: width
// 256, shared by the bender variants; architecture 5: width 128).
// Build with -DNRN_POL=PolBF16 -DNRN_TAG=bf16 [-DNRN_ARCH=5 -DNRN_TAG=bf16_a5] [-DNRN_VIEWS=1 -DNRN_TAG=bf16_views: with the
// view-dependent head behind the trunk]
#ifndef NRN_ARCH
#define NRN_ARCH 0
#endif
#ifndef NRN_VIEWS
#define NRN_VIEWS 0
#endif
#include "nrnerf_train.h"

#define NRN_CAT2(a, b) a##b
#define NRN_CAT(a, b) NRN_CAT2(a, b)
namespace nrn {
hipError_t NRN_CAT(launch_trunk_fwd_train_, NRN_TAG)(const TrunkArgs& a, int num_cus, hipStream_t stream) {
    return launch_trunk_train<NRN_POL, ArchById<NRN_ARCH>::type, (NRN_POL::KH == 1) ? 4 : 8, false, NRN_VIEWS != 0>(a, num_cus, stream);
}
hipError_t NRN_CAT(launch_trunk_bwd_, NRN_TAG)(const TrunkArgs& a, int num_cus, hipStream_t stream) {
    return launch_trunk_train<NRN_POL, ArchById<NRN_ARCH>::type, (NRN_POL::KH == 1) ? 4 : 8, true, NRN_VIEWS != 0>(a, num_cus, stream);
}
#ifdef NRN_WGRAD      // bf16 units only: the weight-gradient kernel of that trunk width
hipError_t NRN_CAT(launch_trunk_wgrad_, NRN_TAG)(const WgradArgs& a, hipStream_t stream) {
    return launch_trunk_wgrad<ArchById<NRN_ARCH>::type, NRN_VIEWS != 0>(a, stream);
}
#endif
}  // namespace nrn
