// nrnerf_generic.hip -- instantiations of the run-time-parameterised network kernel (nrnerf_generic.h): one per precision.
#include "nrnerf_generic.h"

namespace nrn {
// precision ids as nrnerf_precision; the ray bender (mode 0) always takes the fp32 instantiation
hipError_t launch_generic(int precision, const GenArgs& a, int num_cus, hipStream_t stream) {
    if (a.mode == 0 || precision == PREC_F32) return launch_gen<PolF32, 1, false>(a, num_cus, stream);
    // 16-bit: 64 samples per workgroup, two workgroups per CU up to width 256.  (128 samples per workgroup -- each weight fragment
    // feeding four MFMAs, half the L2 -> CU weight traffic per sample, one workgroup per CU -- was measured and is slower: 115.9 vs
    // 83.7 ms per 512x384 frame at 64 + 128 samples, W 256, bf16: the kernel is bound by latency per layer, not by L2 bandwidth.)
    if (precision == PREC_BF16) return launch_gen<PolBF16, 2, false>(a, num_cus, stream);
    if (precision == PREC_F16) return launch_gen<PolF16, 2, false>(a, num_cus, stream);
    return hipErrorInvalidValue;
}
// the training entry points (nrnerf_generic_trunk_forward / _backward; declared in nrnerf_api.cpp): fp32 and bf16 handles
hipError_t launch_generic_train(int precision, const GenArgs& a, int num_cus, hipStream_t stream) {
    if (a.mode != 1 && a.mode != 2) return hipErrorInvalidValue;
    if (precision == PREC_F32) return launch_gen<PolF32, 1, true>(a, num_cus, stream);
    if (precision == PREC_BF16) return launch_gen<PolBF16, 2, true>(a, num_cus, stream);
    return hipErrorInvalidValue;
}
}  // namespace nrn
