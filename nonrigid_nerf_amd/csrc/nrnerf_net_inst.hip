// nrnerf_net_inst.hip -- one instantiation of the network kernel per translation unit, so the variants
// (precision x bender x view-dependent head) compile in parallel.  Build with
//   -DNRN_POL=PolBF16 -DNRN_BEND=1 -DNRN_VIEWS=0 -DNRN_WAVES=8 -DNRN_ARCH=0 -DNRN_NAME=launch_net_a0_bf16_bend
#ifndef NRN_MB
#define NRN_MB 1
#endif
#if NRN_MB > 1
#include "nrnerf_net_mb.h"
#else
#include "nrnerf_net_impl.h"
#endif
#ifndef NRN_EXACT
#define NRN_EXACT 0
#endif

namespace nrn {
hipError_t NRN_NAME(const NetArgs& a, int num_cus, hipStream_t stream) {
#if NRN_MB > 1
    static_assert(NRN_EXACT == 0, "two blocks per wave: exact view directions stay on the one-block kernel");
    return launch_one_mb<NRN_POL, ArchById<NRN_ARCH>::type, (NRN_BEND != 0), (NRN_VIEWS != 0), NRN_WAVES, NRN_MB>(a, num_cus, stream);
#else
    return launch_one<NRN_POL, ArchById<NRN_ARCH>::type, (NRN_BEND != 0), (NRN_VIEWS != 0), NRN_WAVES, (NRN_EXACT != 0)>(a, num_cus, stream);
#endif
}
}  // namespace nrn

#ifdef NRN_TIMING
#define NRN_CAT2(a, b) a##b
#define NRN_CAT(a, b) NRN_CAT2(a, b)
// reads and clears the per-phase cycle counters of this variant: out[8 waves][8 slots]
extern "C" int NRN_CAT(nrnerf_debug_timing_, NRN_NAME)(unsigned long long* out) {
    static const unsigned long long zero[64] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nrn::g_nrn_timing), 64 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(nrn::g_nrn_timing), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif
