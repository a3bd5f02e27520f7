// nrnerf_net.hip -- host-side dispatch over the compiled network-kernel variants.
#include "nrnerf_kernels.h"

namespace nrn {
hipError_t launch_net_f32_bend(const NetArgs&, int, hipStream_t);
hipError_t launch_net_f32_nobend(const NetArgs&, int, hipStream_t);
hipError_t launch_net_bf16_bend(const NetArgs&, int, hipStream_t);
hipError_t launch_net_bf16_nobend(const NetArgs&, int, hipStream_t);
hipError_t launch_net_f16_bend(const NetArgs&, int, hipStream_t);
hipError_t launch_net_f16_nobend(const NetArgs&, int, hipStream_t);

hipError_t launch_net(int precision, bool has_bend, int arch_id, const NetArgs& a, int num_cus, hipStream_t stream) {
    if (arch_id != 0) return hipErrorInvalidValue;
    switch (precision) {
        case PREC_F32:  return has_bend ? launch_net_f32_bend(a, num_cus, stream) : launch_net_f32_nobend(a, num_cus, stream);
        case PREC_BF16: return has_bend ? launch_net_bf16_bend(a, num_cus, stream) : launch_net_bf16_nobend(a, num_cus, stream);
        case PREC_F16:  return has_bend ? launch_net_f16_bend(a, num_cus, stream) : launch_net_f16_nobend(a, num_cus, stream);
    }
    return hipErrorInvalidValue;
}
}  // namespace nrn
