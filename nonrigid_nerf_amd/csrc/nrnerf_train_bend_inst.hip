// nrnerf_train_bend_inst.hip -- the training kernels of the ray bender (forward, backward, divergence regulariser forward / backward) (nrnerf_train_bend.h) for one compiled bender
// architecture.  Build with -DNRN_ARCH=0 (reference default) or 1 (deeper offset MLP).
#include "nrnerf_train_bend.h"

#define NRN_CAT2(a, b) a##b
#define NRN_CAT(a, b) NRN_CAT2(a, b)
namespace nrn {
using BendArch = ArchById<NRN_ARCH>::type;
hipError_t NRN_CAT(launch_bend_fwd_train_a, NRN_ARCH)(const BendTrainArgs& a, int num_cus, hipStream_t stream, bool bf16_arrays) {
    return bf16_arrays ? launch_bend_train<BendArch, false, PolBF16>(a, num_cus, stream) : launch_bend_train<BendArch, false, PolF32>(a, num_cus, stream);
}
hipError_t NRN_CAT(launch_bend_bwd_a, NRN_ARCH)(const BendTrainArgs& a, int num_cus, hipStream_t stream, bool bf16_arrays) {
    return bf16_arrays ? launch_bend_train<BendArch, true, PolBF16>(a, num_cus, stream) : launch_bend_train<BendArch, true, PolF32>(a, num_cus, stream);
}
hipError_t NRN_CAT(launch_bend_div_fwd_a, NRN_ARCH)(const BendDivArgs& a, int num_cus, hipStream_t stream, bool bf16_arrays) {
    return bf16_arrays ? launch_bend_div<BendArch, false, PolBF16>(a, num_cus, stream) : launch_bend_div<BendArch, false, PolF32>(a, num_cus, stream);
}
hipError_t NRN_CAT(launch_bend_div_bwd_a, NRN_ARCH)(const BendDivArgs& a, int num_cus, hipStream_t stream, bool bf16_arrays) {
    return bf16_arrays ? launch_bend_div<BendArch, true, PolBF16>(a, num_cus, stream) : launch_bend_div<BendArch, true, PolF32>(a, num_cus, stream);
}
#if NRN_ARCH == 0     // the weight-gradient kernel does not depend on the bender's depth: one copy
hipError_t launch_bend_wgrad(const BendWgradArgs& a, hipStream_t stream, bool bf16_operands) {
    if (a.njobs <= 0 || a.nparts < 4 || a.nparts % 4 != 0 || a.m <= 0) return hipErrorInvalidValue;
    if (bf16_operands) {        // bend_wgrad16 addresses every array by 32-bit byte offsets off its base (raw buffer loads)
        const unsigned long long lim = 0xffffff00ull;
        for (int j = 0; j < a.njobs; ++j) {
            const BendWgradJob& jb = a.job[j];
            if ((unsigned long long)a.m * (unsigned)jb.ldz * (jb.dz16 ? 2 : 4) >= lim || (unsigned long long)a.m * (unsigned)jb.ldx * (jb.x16 ? 2 : 4) >= lim) return hipErrorInvalidValue;
        }
        if ((unsigned long long)a.m * 4 >= lim || (unsigned long long)(a.m / (a.S > 0 ? a.S : 1) + 1) * (unsigned)(a.ray_stride > a.lat_stride ? a.ray_stride : a.lat_stride) * 4 >= lim) return hipErrorInvalidValue;
    }
    if (bf16_operands) hipLaunchKernelGGL(bend_wgrad16<0>, dim3(a.nparts / 4, a.njobs), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(bend_wgrad<8>, dim3(a.nparts / 4, a.njobs), dim3(256), 0, stream, a);
    return hipGetLastError();
}
#endif
}  // namespace nrn
