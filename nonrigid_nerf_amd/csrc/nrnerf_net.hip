// nrnerf_net.hip -- host-side dispatch over the compiled network-kernel variants.
#include "nrnerf_kernels.h"

namespace nrn {
#define NRN_DECL(n) hipError_t launch_net_##n(const NetArgs&, int, hipStream_t);
NRN_DECL(f32_bend) NRN_DECL(f32_nobend) NRN_DECL(bf16_bend) NRN_DECL(bf16_nobend) NRN_DECL(f16_bend) NRN_DECL(f16_nobend)
NRN_DECL(f32_bend_views) NRN_DECL(f32_nobend_views) NRN_DECL(bf16_bend_views) NRN_DECL(bf16_nobend_views)
NRN_DECL(f16_bend_views) NRN_DECL(f16_nobend_views)
#undef NRN_DECL

hipError_t launch_net(int precision, bool has_bend, bool views, int arch_id, const NetArgs& a, int num_cus, hipStream_t stream) {
    if (arch_id != 0) return hipErrorInvalidValue;
#define NRN_PICK(p) (views ? (has_bend ? launch_net_##p##_bend_views(a, num_cus, stream) : launch_net_##p##_nobend_views(a, num_cus, stream)) \
                           : (has_bend ? launch_net_##p##_bend(a, num_cus, stream) : launch_net_##p##_nobend(a, num_cus, stream)))
    switch (precision) {
        case PREC_F32:  return NRN_PICK(f32);
        case PREC_BF16: return NRN_PICK(bf16);
        case PREC_F16:  return NRN_PICK(f16);
    }
#undef NRN_PICK
    return hipErrorInvalidValue;
}
}  // namespace nrn
