// nrnerf_net.hip -- host-side dispatch over the compiled network-kernel variants
// (architecture x precision x bender x view-dependent head; see the VARIANTS list in the Makefile).
#include "nrnerf_kernels.h"

namespace nrn {
typedef hipError_t (*launch_fn)(const NetArgs&, int, hipStream_t);
#define NRN_DECL(n) hipError_t launch_net_##n(const NetArgs&, int, hipStream_t);
#define NRN_ARCH0(p) NRN_DECL(a0_##p##_bend) NRN_DECL(a0_##p##_nobend) NRN_DECL(a0_##p##_bend_views) NRN_DECL(a0_##p##_nobend_views)
#define NRN_ARCH1(p) NRN_DECL(a1_##p##_bend) NRN_DECL(a1_##p##_bend_views)
#define NRN_ARCH2(p) NRN_DECL(a2_##p##_nobend) NRN_DECL(a2_##p##_nobend_views)
#define NRN_ARCH3(p) NRN_DECL(a3_##p##_bend_views) NRN_DECL(a4_##p##_bend_views)
NRN_ARCH3(f32) NRN_ARCH3(bf16) NRN_ARCH3(f16)
#define NRN_ARCH5(p) NRN_DECL(a5_##p##_bend) NRN_DECL(a5_##p##_nobend)
NRN_ARCH5(f32) NRN_ARCH5(bf16) NRN_ARCH5(f16)
NRN_ARCH0(f32) NRN_ARCH0(bf16) NRN_ARCH0(f16) NRN_ARCH1(f32) NRN_ARCH1(bf16) NRN_ARCH1(f16) NRN_ARCH2(f32) NRN_ARCH2(bf16) NRN_ARCH2(f16)
#undef NRN_DECL

// [arch][precision][has_bend][views]; nullptr = not compiled (arch 1 is a bender variant, arch 2 excludes a bender)
#define NRN_ROW0(p) {{launch_net_a0_##p##_nobend, launch_net_a0_##p##_nobend_views}, {launch_net_a0_##p##_bend, launch_net_a0_##p##_bend_views}}
#define NRN_ROW1(p) {{nullptr, nullptr}, {launch_net_a1_##p##_bend, launch_net_a1_##p##_bend_views}}
#define NRN_ROW2(p) {{launch_net_a2_##p##_nobend, launch_net_a2_##p##_nobend_views}, {nullptr, nullptr}}
// rows 3, 4: architectures 0, 1 with exact (Jacobian) view directions -- same packed weights as rows 0, 1
#define NRN_ROW3(p) {{nullptr, nullptr}, {nullptr, launch_net_a3_##p##_bend_views}}
#define NRN_ROW4(p) {{nullptr, nullptr}, {nullptr, launch_net_a4_##p##_bend_views}}
// row 5: architecture 5 (trunk width 128), no view-dependent head
#define NRN_ROW5(p) {{launch_net_a5_##p##_nobend, nullptr}, {launch_net_a5_##p##_bend, nullptr}}
static const launch_fn TABLE[6][3][2][2] = {
    {NRN_ROW0(f32), NRN_ROW0(bf16), NRN_ROW0(f16)},
    {NRN_ROW1(f32), NRN_ROW1(bf16), NRN_ROW1(f16)},
    {NRN_ROW2(f32), NRN_ROW2(bf16), NRN_ROW2(f16)},
    {NRN_ROW3(f32), NRN_ROW3(bf16), NRN_ROW3(f16)},
    {NRN_ROW4(f32), NRN_ROW4(bf16), NRN_ROW4(f16)},
    {NRN_ROW5(f32), NRN_ROW5(bf16), NRN_ROW5(f16)},
};

// stand-alone bender kernels: [arch 0 = 5-layer, 1 = 7-layer offset MLP][precision]
typedef hipError_t (*bend_fn)(const BendArgs&, int, hipStream_t);
#define NRN_BDECL(n) hipError_t launch_bend_##n(const BendArgs&, int, hipStream_t);
NRN_BDECL(a0_f32) NRN_BDECL(a0_bf16) NRN_BDECL(a0_f16) NRN_BDECL(a1_f32) NRN_BDECL(a1_bf16) NRN_BDECL(a1_f16)
#undef NRN_BDECL
static const bend_fn BEND_TABLE[2][3] = {
    {launch_bend_a0_f32, launch_bend_a0_bf16, launch_bend_a0_f16},
    {launch_bend_a1_f32, launch_bend_a1_bf16, launch_bend_a1_f16},
};
hipError_t launch_bend(int precision, int arch_id, const BendArgs& a, int num_cus, hipStream_t stream) {
    if (arch_id < 0 || arch_id > 1 || precision < 0 || precision > 2) return hipErrorInvalidValue;
    return BEND_TABLE[arch_id][precision](a, num_cus, stream);
}

hipError_t launch_net(int precision, bool has_bend, bool views, int arch_id, const NetArgs& a, int num_cus, hipStream_t stream) {
    if (arch_id < 0 || arch_id > 5 || precision < 0 || precision > 2) return hipErrorInvalidValue;
    const launch_fn f = TABLE[arch_id][precision][has_bend ? 1 : 0][views ? 1 : 0];
    return f ? f(a, num_cus, stream) : hipErrorInvalidValue;
}
}  // namespace nrn
