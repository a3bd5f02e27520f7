// nrnerf_gx16.hip -- one width class of the width-class trunk kernel (nrnerf_gx16.h) per object: -DNRN_GX_WC=64 ... 512, bf16 and f16.
#include "nrnerf_gx16.h"
#include "nrnerf_gx16_bwd.h"

#ifndef NRN_GX_WC
#error "compile with -DNRN_GX_WC=<width class> (Makefile)"
#endif
#define NRN_CAT2(a, b) a##b
#define NRN_CAT(a, b) NRN_CAT2(a, b)

namespace nrn {
hipError_t NRN_CAT(launch_gx16_w, NRN_GX_WC)(int precision, bool views, const GxArgs& a, int num_cus, hipStream_t stream) {
    if (views) {
        if (precision == PREC_BF16) return launch_gx16_t<PolBF16, NRN_GX_WC, true>(a, num_cus, stream);
        if (precision == PREC_F16) return launch_gx16_t<PolF16, NRN_GX_WC, true>(a, num_cus, stream);
        return hipErrorInvalidValue;
    }
    if (precision == PREC_BF16) return launch_gx16_t<PolBF16, NRN_GX_WC, false>(a, num_cus, stream);
    if (precision == PREC_F16) return launch_gx16_t<PolF16, NRN_GX_WC, false>(a, num_cus, stream);
    return hipErrorInvalidValue;
}
hipError_t NRN_CAT(launch_gx16_bwd_w, NRN_GX_WC)(const GxBwdArgs& a, int num_cus, hipStream_t stream) {
    return launch_gx16_bwd_t<PolBF16, NRN_GX_WC>(a, num_cus, stream);
}
}  // namespace nrn
