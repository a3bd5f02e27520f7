// nrnerf_raygen.hip -- camera rays of one frame, generated on the device (reference get_rays,
// run_nerf_helpers.py:588-605, plus the packing render() does, train.py:380-399).
//
// Pixel (row j, column i) -> direction in camera frame ((i-cx)/fx, -(j-cy)/fy, -1), rotated by c2w[:3,:3];
// origin = c2w[:3,3].  Output row (j*W + i) = [o3, d3, near, far (, d/|d|)]: exactly the `rays` tensor
// batchify_rays receives, so a frame costs 12 floats of input instead of 32-44 B/ray of HBM reads by a torch
// meshgrid pipeline.
#include <hip/hip_runtime.h>

#include "nrnerf_kernels.h"

namespace nrn {

__global__ void __launch_bounds__(256) raygen_kernel(const RayGenArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)a.H * a.W;
    if (idx >= n) return;
    const int j = (int)(idx / a.W), i = (int)(idx % a.W);
    const float d0 = __fdiv_rn(__fsub_rn((float)i, a.cx), a.fx);
    const float d1 = -__fdiv_rn(__fsub_rn((float)j, a.cy), a.fy);
    const float d2 = -1.0f;
    float d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)     // torch.sum(dirs[..., None, :] * c2w[:3,:3], -1)
        d[r] = __fadd_rn(__fadd_rn(__fmul_rn(d0, a.c2w[r * 4 + 0]), __fmul_rn(d1, a.c2w[r * 4 + 1])), __fmul_rn(d2, a.c2w[r * 4 + 2]));
    float* out = a.rays + idx * a.ray_stride;
    out[0] = a.c2w[3]; out[1] = a.c2w[7]; out[2] = a.c2w[11];
    out[3] = d[0]; out[4] = d[1]; out[5] = d[2];
    out[6] = a.near; out[7] = a.far;
    if (a.ray_stride >= 11) {
        const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
        out[8] = __fdiv_rn(d[0], nrm); out[9] = __fdiv_rn(d[1], nrm); out[10] = __fdiv_rn(d[2], nrm);
    }
}

hipError_t launch_raygen(const RayGenArgs& a, hipStream_t stream) {
    const long long n = (long long)a.H * a.W;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(raygen_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// weight re-packing on the device (see RepackArgs); conversions round to nearest even like the host packer
__global__ void __launch_bounds__(256) repack_kernel(const RepackArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const int s = a.src[i];
    const float w = s >= 0 ? a.flat[s] : 0.0f;
    const int f = a.fmt ? a.fmt[i] : 0;
    if (f == 0) ((float*)a.dst)[i] = w;
    else if (f == 1) ((__bf16*)a.dst)[i] = (__bf16)w;
    else if (f == 2) ((_Float16*)a.dst)[i] = (_Float16)w;
    else ((_Float16*)a.dst)[i] = (_Float16)((w - (float)(_Float16)w) * 2048.0f);
}
// operands of trunk_wgrad (nrnerf_train.h) that no kernel has written yet: Embedder.embed (run_nerf_helpers.py:120-150) of the
// trunk's input points and the gradient wrt the head's outputs, as [block][row][32 samples] bf16 tiles
__global__ void __launch_bounds__(256) wgrad_operands_kernel(const WgradOperandArgs a) {
    const int bpr = (a.S + 31) >> 5;
    const long long nblocks = (long long)a.n_rays * bpr;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;          // one thread per (block, row, sample)
    if (t >= nblocks * 64 * 32) return;
    const int j = (int)(t & 31), r = (int)((t >> 5) & 63);
    const long long blk = t >> 11;
    const int ray = (int)(blk / bpr), sidx = (int)(blk % bpr) * 32 + j;
    float e = 0.0f, g = 0.0f;
    if (sidx < a.S) {
        const size_t so = (size_t)ray * a.S + sidx;
        if (r < 3) {
            e = a.pts4[so * 4 + r];
        } else if (r < 3 + 6 * a.L) {
            const int k = (r - 3) / 6, w = (r - 3) % 6;
            const float x = a.pts4[so * 4 + (w % 3)] * (float)(1 << k);
            e = (w < 3) ? sinf(x) : cosf(x);
        }
        if (r < 4) g = a.d_raw4[so * 4 + r];
    }
    ((__bf16*)a.enc)[t] = (__bf16)e;
    ((__bf16*)a.g_head)[t] = (__bf16)g;
}
hipError_t launch_wgrad_operands(const WgradOperandArgs& a, hipStream_t stream) {
    const long long total = (long long)a.n_rays * ((a.S + 31) / 32) * 64 * 32;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(wgrad_operands_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_repack(const RepackArgs& a, hipStream_t stream) {
    if (a.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace nrn
