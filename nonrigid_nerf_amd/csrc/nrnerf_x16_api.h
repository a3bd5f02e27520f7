// nrnerf_x16_api.h -- host-side entry points of the 16x16x32 trunk-only kernel (nrnerf_net_x16.h / .hip), seen by the API layer only
// (kept out of nrnerf_kernels.h: every translation unit depends on that one).
#pragma once
#include "nrnerf_kernels.h"

namespace nrn {
// arch: 0 = the default trunk (8 x 256), 5 = --netwidth 128 (the ids of nrnerf_net.hip's dispatch table); views: the view-dependent head
// (width 256 only; directions = finite differences of the points of NetArgs::pts4 along the ray, rnh:339-351)
hipError_t launch_net_x16(int precision, int arch, bool views, const NetArgs& a, int num_cus, hipStream_t stream);
// rays of one fused-compositing group of that kernel: its waves per workgroup x the fewest rays whose 16-sample blocks fill whole
// iterations -- the API layer's "enough rays to fuse" threshold asks here instead of restating the kernel's mapping
long long x16_rays_per_group(int arch, int S);
// the largest coarse pass (samples per ray) that has a kernel with the coarse epilogue (NetArgs::fuse with n_importance > 0: compositing +
// sample_pdf + merge behind the trunk)
int x16_coarse_epilogue_max_samples();
// the stand-alone ray bender on 16x16x32 MFMAs (nrnerf_bend_x16.h): arch 0 = the 5 x 64 bender, 1 = 7 x 64; BendArgs::wstream / bias =
// the image of pack_pass_x16_bend (f16 fragments of PlanX16Bend)
hipError_t launch_bend_x16(int arch, const BendArgs& a, int num_cus, hipStream_t stream);
// the width-class trunk kernel for architectures outside the compiled set (nrnerf_gx16.h): wc = gx_width_class(trunk width), 64 .. 512
struct GxArgs {
    const float* pts4;          // [N, S, 4] points of the pass
    float* raw4;                // [N, S, 4] rgb, sigma (workspace, for the composite kernel)
    float* raw_out;             // [N, S, raw_ch] or null (retraw)
    int raw_ch;
    int n_rays, S;
    const void* wstream;        // the layers' fragment blocks back to back (+ a copy of the first two units behind the last)
    const float* bias;          // [total tiles][16]
    int depth, skip, L;         // pts_linears count; index after which [input, h] is concatenated (-1: never); encoding frequencies
    int n_bias_tiles;
    int LV;                     // (view-dependent head) direction-encoding frequencies
    int fuse_on;                // the pass' compositing as the kernel's epilogue (a FINAL pass of <= 256 samples; raw4 may then be null)
    CompositeArgs fuse;         // its arguments (raw4 unused, n_importance 0)
    void* save;                 // (training forward, bf16, plain head) [depth][save_stride] rows of save_w 16-bit values: every hidden activation; or null
    long long save_stride;      // elements between two layers' arrays (>= n_rays * S * save_w)
    int save_w;                 // the trunk's width (% 4 == 0)
    void* relu_bits;            // (with save; optional) [depth][n_blocks of 16 samples][64 lanes][4 * ceil(wc / 128)] bytes: which values passed the relu,
                                // one byte per lane and tile pair -- what the backward-data kernel (nrnerf_gx16_bwd.h) masks with
};
// (the backward-data kernel of such a trunk: nrnerf_gx16_bwd.h / nrnerf_gx16_bwd_api.h; bf16 only)
constexpr int gx16_bits_bytes_per_lane(int wc) { return 4 * ((wc / 32 + 3) / 4); }
hipError_t launch_gx16(int precision, int wc, bool views, const GxArgs& a, int num_cus, hipStream_t stream);
long long gx16_rays_per_group(int wc, int S);          // rays of one fused-compositing group of that kernel
}  // namespace nrn
