// nrnerf_train_inst.hip -- the two training kernels of nrnerf_train.h for one precision (the default architecture's trunk).
// Build with -DNRN_POL=PolBF16 -DNRN_TAG=bf16
#include "nrnerf_train.h"

#define NRN_CAT2(a, b) a##b
#define NRN_CAT(a, b) NRN_CAT2(a, b)
namespace nrn {
hipError_t NRN_CAT(launch_trunk_fwd_train_, NRN_TAG)(const TrunkArgs& a, int num_cus, hipStream_t stream) {
    return launch_trunk_train<NRN_POL, ArchDefault, (NRN_POL::KH == 1) ? 4 : 8, false>(a, num_cus, stream);
}
hipError_t NRN_CAT(launch_trunk_bwd_, NRN_TAG)(const TrunkArgs& a, int num_cus, hipStream_t stream) {
    return launch_trunk_train<NRN_POL, ArchDefault, (NRN_POL::KH == 1) ? 4 : 8, true>(a, num_cus, stream);
}
}  // namespace nrn
