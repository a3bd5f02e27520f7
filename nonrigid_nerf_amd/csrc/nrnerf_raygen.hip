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
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// weight re-packing on the device (see RepackBatchArgs); conversions round to nearest even like the host packer

__device__ __forceinline__ void repack_one(const float* flat, const int32_t* src, const uint8_t* fmt, void* dst, long long i) {
    const int s = src[i];
    const float w = s >= 0 ? flat[s] : 0.0f;
    const int f = fmt ? fmt[i] : 0;
    if (f == 0) ((float*)dst)[i] = w;
    else if (f == 1) ((__bf16*)dst)[i] = (__bf16)w;
    else if (f == 2) ((_Float16*)dst)[i] = (_Float16)w;
    else ((_Float16*)dst)[i] = (_Float16)((w - (float)(_Float16)w) * 2048.0f);
}
__global__ void __launch_bounds__(256) repack_batch_kernel(const RepackBatchArgs a) {
    int k = 0;
    for (int q = 1; q < a.n_segments; ++q)
        if (blockIdx.x >= a.block0[q]) k = q;               // segments are listed in grid order (uniform per workgroup)
    const long long i = (long long)(blockIdx.x - a.block0[k]) * 256 + threadIdx.x;
    if (i < a.n[k]) repack_one(a.flat, a.src[k], a.fmt[k], a.dst[k], i);
}
// operands of trunk_wgrad (nrnerf_train.h) that no kernel has written yet: Embedder.embed (run_nerf_helpers.py:120-150) of the
// trunk's input points and the gradient wrt the head's outputs, as [block][row][32 samples] bf16 tiles.
// 12 x 32 threads per block of 32 samples: thread (q, j) writes, for sample j, the six rows of frequency q (one sincos per
// coordinate), or (q = L) the identity rows and the zero padding of the encoding tile, or (q = L + 1) the head-gradient tile
// (4 rows + 60 zero rows, the zeros as 16-byte stores).  Neighbouring lanes exchange values so that every encoding store is
// a dword holding two samples.  (One thread per element with sinf / cosf and 2-byte stores: 330 us per 2 M samples.)
__device__ __forceinline__ unsigned bf16_bits(float v) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v); }
__global__ void __launch_bounds__(256) wgrad_operands_kernel(const WgradOperandArgs a) {
    const int bpr = (a.S + 31) >> 5, Q = a.L + 2;
    const long long nblocks = (long long)a.n_rays * bpr;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= nblocks * Q * 32) return;                      // (32 | 256: the lanes of a pair are both in or both out)
    const int j = (int)(t & 31), q = (int)((t >> 5) % Q);
    const long long blk = (t >> 5) / Q;
    const int ray = (int)(blk / bpr), sidx = (int)(blk % bpr) * 32 + j;
    const bool in = sidx < a.S;
    const size_t so = (size_t)ray * a.S + (in ? sidx : 0);
    unsigned* enc = (unsigned*)a.enc + (size_t)blk * (64 * 16);              // dwords: [row][16 sample pairs]
    unsigned* gh = (unsigned*)a.g_head + (size_t)blk * (64 * 16);
    const bool odd = j & 1;
    // v[r] for rows base + r of this thread's sample -> dword stores of (even sample, odd sample): even lanes take rows 0, 2, 4, odd lanes 1, 3, 5
    auto store_rows = [&](unsigned* tile, int base, const float (&v)[6], int n) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const unsigned mine = bf16_bits(v[r]), other = __shfl_xor(mine, 1);
            if ((r & 1) == (int)odd && r < n) tile[(base + r) * 16 + (j >> 1)] = odd ? (other | (mine << 16)) : (mine | (other << 16));
        }
    };
    const f32x4 p = in ? *(const f32x4*)(a.pts4 + so * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (q < a.L) {
        const float sc = (float)(1 << q);
        float v[6];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sn, cs;
            sincosf(p[c] * sc, &sn, &cs);              // (the hardware sin / cos gains 10 %: the stores bound this kernel)
            v[c] = in ? sn : 0.0f;
            v[3 + c] = in ? cs : 0.0f;
        }
        store_rows(enc, 3 + 6 * q, v, 6);
    } else if (q == a.L) {
        const float v[6] = {p[0], p[1], p[2], 0.0f, 0.0f, 0.0f};
        store_rows(enc, 0, v, 3);
        for (int r = 3 + 6 * a.L + (odd ? 1 : 0); r < 64; r += 2) enc[r * 16 + (j >> 1)] = 0u;     // padding rows (row 63 for L = 10)
    } else {
        const f32x4 g = in ? *(const f32x4*)(a.d_raw4 + so * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const float v[6] = {g[0], g[1], g[2], g[3], 0.0f, 0.0f};
        store_rows(gh, 0, v, 4);
        if (a.head_sums) {          // the head's bias gradient = column sums of d raw: this block's share (fp32, lanes added in a fixed tree)
            f32x4 sum = g;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1)
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) sum[ch] += __shfl_xor(sum[ch], o);
            if (j == 0) *(f32x4*)(a.head_sums + (size_t)blk * 4) = sum;
        }
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        for (int o = 64 + 4 * j; o < 64 * 16; o += 128) *(u32x4*)(gh + o) = u32x4{0u, 0u, 0u, 0u};      // rows 4 .. 63
    }
}
// view-dependent head: Embedder.embed of the samples' view directions as a third operand tile, [block][64 rows][32 samples]
// bf16, rows in the reference's column order [d, sin(2^f d), cos(2^f d)] (f < LV), the rest zero.  Thread (block, row, sample
// pair) writes one dword.
__global__ void __launch_bounds__(256) wgrad_operand_dirs_kernel(const WgradOperandArgs a) {
    const int bpr = (a.S + 31) >> 5;
    const long long nblocks = (long long)a.n_rays * bpr;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= nblocks * 1024) return;
    const int jp = (int)(t & 15), row = (int)((t >> 4) & 63);
    const long long blk = t >> 10;
    const int ray = (int)(blk / bpr), s0 = (int)(blk % bpr) * 32 + 2 * jp;
    unsigned w = 0;
    if (row < 3 + 6 * a.LV) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int sidx = s0 + e;
            float v = 0.0f;
            if (sidx < a.S) {
                const float* d = a.dirs + ((size_t)ray * a.S + sidx) * 3;
                if (row < 3) v = d[row];
                else {
                    const int kk = row - 3, f = kk / 6, r = kk % 6;
                    const float arg = d[r % 3] * (float)(1 << f);
                    v = r < 3 ? sinf(arg) : cosf(arg);
                }
            }
            w |= bf16_bits(v) << (16 * e);
        }
    }
    ((unsigned*)a.encv)[(size_t)blk * 1024 + row * 16 + jp] = w;
}
hipError_t launch_wgrad_operands(const WgradOperandArgs& a, hipStream_t stream) {
    const long long total = (long long)a.n_rays * ((a.S + 31) / 32) * (a.L + 2) * 32;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(wgrad_operands_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    if (a.dirs && a.encv) {
        const long long tv = (long long)a.n_rays * ((a.S + 31) / 32) * 1024;
        hipLaunchKernelGGL(wgrad_operand_dirs_kernel, dim3((unsigned)((tv + 255) / 256)), dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

// fp32 mode (trunk_wgrad_f32 reads rows): thread (sample, q) writes columns 4 q .. 4 q + 3 of the sample's encoding row and of
// its head-gradient row ([M][64] fp32 each), sincosf as in the fp32 forward kernel.
__global__ void __launch_bounds__(256) wgrad_operands_f32_kernel(const WgradOperandArgs a) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long M = (long long)a.n_rays * a.S;
    if (t >= M * 16) return;
    const long long m = t >> 4;
    const int q = (int)(t & 15);
    const f32x4 p = *(const f32x4*)(a.pts4 + (size_t)m * 4);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int col = 4 * q + e;
        if (col < 3) v[e] = p[col];
        else if (col < 3 + 6 * a.L) {
            const int kk = col - 3, f = kk / 6, r = kk % 6;
            float sn, cs;
            sincosf(p[r % 3] * (float)(1 << f), &sn, &cs);
            v[e] = r < 3 ? sn : cs;
        } else v[e] = 0.0f;
    }
    *(f32x4*)((float*)a.enc + (size_t)m * 64 + 4 * q) = f32x4{v[0], v[1], v[2], v[3]};
    const f32x4 g = (q == 0) ? *(const f32x4*)(a.d_raw4 + (size_t)m * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    *(f32x4*)((float*)a.g_head + (size_t)m * 64 + 4 * q) = g;
    if (a.dirs && a.encv) {             // view-dependent head: the direction's encoding as a third row
        const float* d = a.dirs + (size_t)m * 3;
        float u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = 4 * q + e;
            if (col < 3) u[e] = d[col];
            else if (col < 3 + 6 * a.LV) {
                const int kk = col - 3, f = kk / 6, r = kk % 6;
                float sn, cs;
                sincosf(d[r % 3] * (float)(1 << f), &sn, &cs);
                u[e] = r < 3 ? sn : cs;
            } else u[e] = 0.0f;
        }
        *(f32x4*)((float*)a.encv + (size_t)m * 64 + 4 * q) = f32x4{u[0], u[1], u[2], u[3]};
    }
}
hipError_t launch_wgrad_operands_f32(const WgradOperandArgs& a, hipStream_t stream) {
    const long long total = (long long)a.n_rays * a.S * 16;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(wgrad_operands_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_repack_batch(const RepackBatchArgs& a, hipStream_t stream) {
    if (a.n_segments <= 0 || a.n_segments > REPACK_MAX_SEGMENTS || a.block0[a.n_segments] == 0) return a.n_segments == 0 ? hipSuccess : hipErrorInvalidValue;
    hipLaunchKernelGGL(repack_batch_kernel, dim3(a.block0[a.n_segments]), dim3(256), 0, stream, a);
    return hipGetLastError();
}
}  // namespace nrn
