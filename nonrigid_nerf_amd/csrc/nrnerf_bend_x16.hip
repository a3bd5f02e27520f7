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

#ifdef NRN_TIMING
// reads and clears the per-phase cycle counters of bend_kernel_x16 (tools/timing_probe_bender.py): out[8 waves][8 slots]
extern "C" int nrnerf_debug_timing_bend_x16(unsigned long long* out) {
    static const unsigned long long zero[64] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nrn::g_nrn_timing), 64 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(nrn::g_nrn_timing), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif
