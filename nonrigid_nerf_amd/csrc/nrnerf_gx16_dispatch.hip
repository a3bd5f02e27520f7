// nrnerf_gx16_dispatch.hip -- width class -> the object that holds its kernels (nrnerf_gx16.hip, one per class)
#include "nrnerf_gx16.h"
#include "nrnerf_gx16_bwd.h"
#include "nrnerf_x16_api.h"

namespace nrn {
#define NRN_GX_DECL(WC) hipError_t launch_gx16_w##WC(int, bool, const GxArgs&, int, hipStream_t);
NRN_GX_DECL(64) NRN_GX_DECL(128) NRN_GX_DECL(192) NRN_GX_DECL(256) NRN_GX_DECL(320) NRN_GX_DECL(384) NRN_GX_DECL(448) NRN_GX_DECL(512)
hipError_t launch_gx16(int precision, int wc, bool views, const GxArgs& a, int num_cus, hipStream_t stream) {
    switch (wc) {
        case 64: return launch_gx16_w64(precision, views, a, num_cus, stream);
        case 128: return launch_gx16_w128(precision, views, a, num_cus, stream);
        case 192: return launch_gx16_w192(precision, views, a, num_cus, stream);
        case 256: return launch_gx16_w256(precision, views, a, num_cus, stream);
        case 320: return launch_gx16_w320(precision, views, a, num_cus, stream);
        case 384: return launch_gx16_w384(precision, views, a, num_cus, stream);
        case 448: return launch_gx16_w448(precision, views, a, num_cus, stream);
        case 512: return launch_gx16_w512(precision, views, a, num_cus, stream);
        default: return hipErrorInvalidValue;
    }
}
#define NRN_GXB_DECL(WC) hipError_t launch_gx16_bwd_w##WC(const GxBwdArgs&, int, hipStream_t);
NRN_GXB_DECL(64) NRN_GXB_DECL(128) NRN_GXB_DECL(192) NRN_GXB_DECL(256) NRN_GXB_DECL(320) NRN_GXB_DECL(384) NRN_GXB_DECL(448) NRN_GXB_DECL(512)
hipError_t launch_gx16_bwd(int wc, const GxBwdArgs& a, int num_cus, hipStream_t stream) {
    switch (wc) {
        case 64: return launch_gx16_bwd_w64(a, num_cus, stream);
        case 128: return launch_gx16_bwd_w128(a, num_cus, stream);
        case 192: return launch_gx16_bwd_w192(a, num_cus, stream);
        case 256: return launch_gx16_bwd_w256(a, num_cus, stream);
        case 320: return launch_gx16_bwd_w320(a, num_cus, stream);
        case 384: return launch_gx16_bwd_w384(a, num_cus, stream);
        case 448: return launch_gx16_bwd_w448(a, num_cus, stream);
        case 512: return launch_gx16_bwd_w512(a, num_cus, stream);
        default: return hipErrorInvalidValue;
    }
}
long long gx16_rays_per_group(int wc, int S) { return gx16_rays_per_group_of(wc, S); }
}  // namespace nrn
