// nrnerf_composite.hip -- per-ray kernels: alpha compositing (reference raw2outputs,
// train.py:724-789), inverse-CDF hierarchical sampling (sample_pdf, run_nerf_helpers.py:651-698),
// the sort-merge of coarse and importance depths (train.py:920) and z_std (train.py:959).
//
// One wavefront per ray.  Lane l owns the EPL consecutive samples l*EPL .. l*EPL+EPL-1, so the
// exclusive transmittance product and the CDF are a short in-lane serial scan followed by one
// 64-lane prefix scan (DPP/ds_swizzle shuffles); importance samples are drawn by a per-lane
// binary search of the CDF held in LDS; the merge ranks every depth against all others (stable,
// so it equals torch.sort's value output for any input order).  All fp32, HBM traffic is the
// 16 B/sample raw read plus the per-ray outputs.
#include <hip/hip_runtime.h>

#include "nrnerf_kernels.h"
#include "nrnerf_aux.h"
#include "nrnerf_composite_ray.h"

namespace nrn {

static constexpr int RAYS_PER_WG = 4;
// Samples per ray and pass: a lane owns EPL <= 16 consecutive samples, so S and S + I go up to 1024 (the reference has no cap,
// train.py:1090-1094; its configs use 64 + 64 / 64 + 128), forward and backward.  The split-bender paths (8-bit ranks of the new
// samples among the merged depths) and the fused compositing of the network kernels stay at 256: beyond it a render takes the
// fused-bender fine pass and this kernel, a training pass bends all merged samples again.
static constexpr int MAXS = 1024;

template <int EPL, bool SAMPLE>
__global__ void __launch_bounds__(RAYS_PER_WG * 64) composite_kernel(const CompositeArgs a) {
    constexpr int NLDS = SAMPLE ? 64 * EPL : 1;                     // cdf / bins of the S coarse samples
    __shared__ float s_cdf[RAYS_PER_WG][NLDS];
    __shared__ float s_bins[RAYS_PER_WG][NLDS];
    __shared__ float s_z[RAYS_PER_WG][SAMPLE ? MAXS + 4 : 1];       // the S + I depths to merge

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ray_raw = blockIdx.x * RAYS_PER_WG + wave;
    const bool ray_ok = ray_raw < a.n_rays;
    const int ray = ray_ok ? ray_raw : a.n_rays - 1;
    const int S = a.S;

    // ---- alpha compositing, per-ray maps, detail outputs, surface reduction: shared with the network kernels' fused epilogue
    float z[EPL + 1], w[EPL];
    composite_ray<EPL>(a, ray, ray_ok, lane, [&](int ic) { return *(const f32x4*)(a.raw4 + ((size_t)ray * S + ic) * 4); }, z, w);

    if constexpr (SAMPLE) {
        // ---- sample_pdf, z_std, merge: shared with the coarse epilogue of the 16x16x32 trunk kernel (nrnerf_composite_ray.h)
        sample_merge_ray<EPL, false>(a, ray, ray_ok, lane, z, w, s_cdf[wave], s_bins[wave], s_z[wave]);
    }
}

// Backward of the compositing (training).  Same decomposition as the forward kernel: one wave per ray, lane l owns EPL
// consecutive samples.  With  t_i = 1 - alpha_i + 1e-10,  T_i = prod_{j<i} t_j,  w_i = alpha_i T_i  and
// G_i = dL/dw_i = g_rgb . c_i + g_acc + g_depth z_i + g_w_i:
//     dL/dalpha_i = G_i T_i - (sum_{k>i} G_k w_k) / t_i          (every later T_k carries the factor t_i)
//     dL/dsigma_i = dL/dalpha_i * dist_i * exp(-relu(sigma_i + noise_i) dist_i) * [sigma_i + noise_i > 0]
//     dL/drgb_i   = w_i * g_rgb * c_i (1 - c_i)                  (c = sigmoid)
// which is what autograd derives from train.py:740-784 (cumprod backward = reverse cumsum of grad * output / input).
template <int EPL>
__global__ void __launch_bounds__(RAYS_PER_WG * 64) composite_bwd_kernel(const CompositeBwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ray_raw = blockIdx.x * RAYS_PER_WG + wave;
    const bool ray_ok = ray_raw < a.n_rays;
    const int ray = ray_ok ? ray_raw : a.n_rays - 1;
    const int S = a.S;
    const float* rp = a.rays + (size_t)ray * a.ray_stride;
    const float dx = rp[3], dy = rp[4], dz = rp[5];
    const float near = rp[6], far = rp[7];
    const float dnorm = sqrtf(__fadd_rn(__fadd_rn(rounded(dx * dx), rounded(dy * dy)), rounded(dz * dz)));      // as composite_ray

    float z[EPL + 1], sig[EPL], col[EPL][3];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int i = lane * EPL + k;
        const int ic = i < S ? i : S - 1;
        if (a.z) z[k] = a.z[(size_t)ray * S + ic];
        else {
            const float t = c_lin01(ic, S);
            if (a.lindisp)
                z[k] = __fdiv_rn(1.0f, __fadd_rn(__fmul_rn(__fdiv_rn(1.0f, near), __fsub_rn(1.0f, t)), __fmul_rn(__fdiv_rn(1.0f, far), t)));
            else
                z[k] = __fadd_rn(__fmul_rn(near, __fsub_rn(1.0f, t)), __fmul_rn(far, t));
        }
        const f32x4 r = *(const f32x4*)(a.raw4 + ((size_t)ray * S + ic) * 4);
        col[k][0] = r[0]; col[k][1] = r[1]; col[k][2] = r[2]; sig[k] = r[3];
        if (a.noise) sig[k] = __fadd_rn(sig[k], a.noise[(size_t)ray * S + ic]);
    }
    z[EPL] = __shfl_down(z[0], 1);

    // forward quantities, recomputed exactly as composite_kernel does
    float alpha[EPL], ex[EPL], dist[EPL], tfac[EPL], texcl[EPL], w[EPL];
    float run = 1.0f;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int i = lane * EPL + k;
        float d = (i == S - 1) ? 1e10f : __fsub_rn(z[k + 1], z[k]);
        d = __fmul_rn(d, dnorm);
        dist[k] = d;
        const float s = fmaxf(sig[k], 0.0f);
        ex[k] = (i < S) ? expf(-__fmul_rn(s, d)) : 1.0f;
        alpha[k] = (i < S) ? __fsub_rn(1.0f, ex[k]) : 0.0f;
        tfac[k] = (i < S) ? __fadd_rn(__fsub_rn(1.0f, alpha[k]), 1e-10f) : 1.0f;
        texcl[k] = run;
        run = __fmul_rn(run, tfac[k]);
    }
    const float incl = wave_scan_mul(run, lane);
    float before = __shfl_up(incl, 1);
    if (lane == 0) before = 1.0f;
    float c[EPL][3], sdepth = 0.f, sacc = 0.f;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int i = lane * EPL + k;
        texcl[k] = __fmul_rn(before, texcl[k]);                   // T_i
        w[k] = (i < S) ? __fmul_rn(alpha[k], texcl[k]) : 0.0f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) c[k][ch] = 1.0f / (1.0f + expf(-col[k][ch]));
        sdepth += w[k] * z[k]; sacc += w[k];
    }
    sdepth = wave_sum(sdepth); sacc = wave_sum(sacc);

    // gradients of the per-ray maps
    float gr[3] = {a.g_rgb[(size_t)ray * 3], a.g_rgb[(size_t)ray * 3 + 1], a.g_rgb[(size_t)ray * 3 + 2]};
    float g_acc = a.g_acc ? a.g_acc[ray] : 0.0f;
    float g_depth = 0.0f;
    if (a.white_bkgd) g_acc -= gr[0] + gr[1] + gr[2];             // rgb_map += 1 - acc (train.py:786-787)
    if (a.g_disp) {                                               // disp = 1 / max(1e-10, depth / acc) (train.py:781-784)
        const float q = sdepth / sacc;
        if (q > 1e-10f) {
            const float gd = a.g_disp[ray];
            g_depth = -gd * sacc / (sdepth * sdepth);
            g_acc += gd / sdepth;
        }
    }
    float G[EPL], gw_sum = 0.f;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int i = lane * EPL + k;
        G[k] = gr[0] * c[k][0] + gr[1] * c[k][1] + gr[2] * c[k][2] + g_acc + g_depth * z[k];
        if (a.g_w && i < S) G[k] += a.g_w[(size_t)ray * S + i];
        gw_sum += G[k] * w[k];
    }
    // exclusive suffix sum of G_k w_k over the lanes, then inside the lane from its last sample down
    float suf = gw_sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_down(suf, o); if (lane + o < 64) suf += t; }
    float after = suf - gw_sum;
#pragma unroll
    for (int k = EPL - 1; k >= 0; --k) {
        const int i = lane * EPL + k;
        const float dalpha = G[k] * texcl[k] - after / tfac[k];
        after += G[k] * w[k];
        const float dsig = (sig[k] > 0.0f) ? dalpha * dist[k] * ex[k] : 0.0f;
        if (ray_ok && i < S) {
            const float s0 = w[k] * gr[0] * c[k][0] * (1.0f - c[k][0]);
            const float s1 = w[k] * gr[1] * c[k][1] * (1.0f - c[k][1]);
            const float s2 = w[k] * gr[2] * c[k][2] * (1.0f - c[k][2]);
            *(f32x4*)(a.d_raw4 + ((size_t)ray * S + i) * 4) = f32x4{s0, s1, s2, dsig};
        }
    }
}

template <int EPL>
static hipError_t launch_bwd_epl(const CompositeBwdArgs& a, hipStream_t stream) {
    const int grid = (a.n_rays + RAYS_PER_WG - 1) / RAYS_PER_WG;
    if (grid <= 0) return hipSuccess;
    hipLaunchKernelGGL((composite_bwd_kernel<EPL>), dim3(grid), dim3(RAYS_PER_WG * 64), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_composite_bwd(const CompositeBwdArgs& a, hipStream_t stream) {
    if (a.S < 2 || a.S > MAXS) return hipErrorInvalidValue;
    switch ((a.S + 63) / 64) {
        case 1: return launch_bwd_epl<1>(a, stream);
        case 2: return launch_bwd_epl<2>(a, stream);
        case 3: return launch_bwd_epl<3>(a, stream);
        case 4: return launch_bwd_epl<4>(a, stream);
        case 5: case 6: return launch_bwd_epl<6>(a, stream);
        case 7: case 8: return launch_bwd_epl<8>(a, stream);
        case 9: case 10: case 11: case 12: return launch_bwd_epl<12>(a, stream);
        default: return launch_bwd_epl<16>(a, stream);
    }
}

template <int EPL>
static hipError_t launch_epl(const CompositeArgs& a, hipStream_t stream) {
    const int grid = (a.n_rays + RAYS_PER_WG - 1) / RAYS_PER_WG;
    if (grid <= 0) return hipSuccess;
    if (a.n_importance > 0)
        hipLaunchKernelGGL((composite_kernel<EPL, true>), dim3(grid), dim3(RAYS_PER_WG * 64), 0, stream, a);
    else
        hipLaunchKernelGGL((composite_kernel<EPL, false>), dim3(grid), dim3(RAYS_PER_WG * 64), 0, stream, a);
    return hipGetLastError();
}

// train.py:855-868: one thread per sample
__device__ __forceinline__ float jittered_depth(const float* rp, const float* u, long long idx, int i, int S, int lindisp) {
    const float near = rp[6], far = rp[7];
    auto zat = [&](int k) {
        const float t = c_lin01(k, S);
        if (lindisp)
            return __fdiv_rn(1.0f, __fadd_rn(__fmul_rn(__fdiv_rn(1.0f, near), __fsub_rn(1.0f, t)), __fmul_rn(__fdiv_rn(1.0f, far), t)));
        return __fadd_rn(__fmul_rn(near, __fsub_rn(1.0f, t)), __fmul_rn(far, t));
    };
    const float zi = zat(i);
    const float upper = (i < S - 1) ? __fmul_rn(0.5f, __fadd_rn(zat(i + 1), zi)) : zi;         // :857-858
    const float lower = (i > 0) ? __fmul_rn(0.5f, __fadd_rn(zi, zat(i - 1))) : zi;             // :859
    // :868 (no uniforms: the plain spacing).  (hipcc fuses this product into the sum -- one rounding where torch has two; the depths have been
    // these since round 2 and every stochastic parity figure was measured on them, so it stays)
    return u ? __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), u[idx])) : zi;
}
__global__ void __launch_bounds__(256) zjitter_kernel(const JitterArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.n_rays * a.S) return;
    const int ray = (int)(idx / a.S), i = (int)(idx % a.S);
    a.z_out[idx] = jittered_depth(a.rays + (size_t)ray * a.ray_stride, a.u, idx, i, a.S, a.lindisp);
}
// the same, and the sample's point o + d z (train.py:871-873: a product and a sum, each rounded -- torch's two elementwise launches)
__global__ void __launch_bounds__(256) sample_points_kernel(const SamplePointsArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.n_rays * a.S) return;
    const int ray = (int)(idx / a.S), i = (int)(idx % a.S);
    const float* rp = a.rays + (size_t)ray * a.ray_stride;
    const float z = jittered_depth(rp, a.u, idx, i, a.S, a.lindisp);
    a.z_out[idx] = z;
#pragma unroll
    for (int c = 0; c < 3; ++c) a.pts_out[idx * 3 + c] = __fadd_rn(rp[c], rounded(rp[3 + c] * z));
}
hipError_t launch_sample_points(const SamplePointsArgs& a, hipStream_t stream) {
    const long long n = (long long)a.n_rays * a.S;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(sample_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_zjitter(const JitterArgs& a, hipStream_t stream) {
    const long long n = (long long)a.n_rays * a.S;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(zjitter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_composite(const CompositeArgs& a, hipStream_t stream) {
    if (a.S < 2 || a.S > MAXS || a.S + a.n_importance > MAXS) return hipErrorInvalidValue;
    if (a.rank_new && a.S + a.n_importance > 256) return hipErrorInvalidValue;      // 8-bit ranks of the split-bender path
    const int epl = (a.S + 63) / 64;
    switch (epl) {
        case 1: return launch_epl<1>(a, stream);
        case 2: return launch_epl<2>(a, stream);
        case 3: return launch_epl<3>(a, stream);
        case 4: return launch_epl<4>(a, stream);
        case 5: case 6: return launch_epl<6>(a, stream);
        case 7: case 8: return launch_epl<8>(a, stream);
        case 9: case 10: case 11: case 12: return launch_epl<12>(a, stream);
        default: return launch_epl<16>(a, stream);
    }
}


// ---- helpers of the training path (nrnerf_aux.h) -----------------------------------------------------------------------
// One workgroup per ray, one thread per merged row: the row is importance sample i if rank_new[i] names it, else coarse
// sample (row - number of importance samples before it).  Pure permutation: the inverse direction copies the other way.
__global__ void __launch_bounds__(256) merge_rows_kernel(const MergeRowsArgs a) {
    __shared__ unsigned char rk[256];
    const int n = blockIdx.x, r = threadIdx.x, T = a.S + a.I;
    if (r < a.I) rk[r] = a.rank_new[(size_t)n * a.I + r];
    __syncthreads();
    if (r >= T) return;
    int before = 0, mine = -1;
    for (int i = 0; i < a.I; ++i) {
        const int k = rk[i];
        before += k < r;
        mine = (k == r) ? i : mine;
    }
    const size_t mrow = (size_t)n * T + r;
    const int crow = (r - before < a.S) ? r - before : a.S - 1;       // (in range whatever rank_new holds: ranks that are not a set of I distinct rows only give wrong values)
    const size_t srow = (mine >= 0) ? (size_t)n * a.I + mine : (size_t)n * a.S + crow;
    const f32x4* ca = (const f32x4*)(mine >= 0 ? a.n_a : a.c_a);
    const f32x4* cb = (const f32x4*)(mine >= 0 ? a.n_b : a.c_b);
    if (!a.inverse) {
        ((f32x4*)a.m_a)[mrow] = ca[srow];
        if (a.m_b) ((f32x4*)a.m_b)[mrow] = cb[srow];
    } else {
        ((f32x4*)ca)[srow] = ((const f32x4*)a.m_a)[mrow];
        if (a.m_b) ((f32x4*)cb)[srow] = ((const f32x4*)a.m_b)[mrow];
    }
}
hipError_t launch_merge_rows(const MergeRowsArgs& a, hipStream_t stream) {
    if (a.n_rays <= 0 || a.S < 1 || a.I < 1 || a.S + a.I > 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(merge_rows_kernel, dim3(a.n_rays), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// 32 outputs x 8 groups of records per workgroup: group y adds records y, y + 8, y + 16, ... in order (four loads in flight),
// then the eight group sums are added in order -- a fixed tree, so the result does not depend on timing.  (One thread per
// output walking all records was 166 us for the bender's 1024 records of 41600 floats: 40 k threads, 256 dependent rounds.)
__global__ void __launch_bounds__(256) reduce_partials_kernel(const ReducePartialsArgs a) {
    __shared__ float sh[REDUCE_GROUPS][32];
    if (a.aux && blockIdx.x == gridDim.x - 1) {
        // the extra workgroup: thread (slice q, channel c) adds records q, q + 64, ... in order (eight loads in flight), then the 64 slice
        // sums are added in order
        __shared__ float part[256];
        const int c = threadIdx.x & 3, q = threadIdx.x >> 2;
        float s = 0.0f;
        int r = q;
        for (; r + 7 * 64 < a.n_aux; r += 8 * 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = a.aux[(size_t)(r + 64 * u) * 4 + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s = __fadd_rn(s, v[u]);
        }
        for (; r < a.n_aux; r += 64) s = __fadd_rn(s, a.aux[(size_t)r * 4 + c]);
        part[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < 4 && a.aux_pos[threadIdx.x] >= 0) {
            float t = 0.0f;
            for (int g = 0; g < 64; ++g) t = __fadd_rn(t, part[4 * g + threadIdx.x]);
            a.out[a.aux_pos[threadIdx.x]] = t;
        }
        return;
    }
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
    const long long j = (long long)blockIdx.x * 32 + x;
    float s = 0.0f;
    bool leave = false;
    if (j < a.n_out) {
        const int ix = a.index[j];
        leave = ix == -2;
        if (ix >= 0 && (ix & (REDUCE_SHORT_FLAG - 1)) < a.stride) {          // (a position beyond the record reads nothing)
            const int P = (ix & REDUCE_SHORT_FLAG) ? a.n_short : a.n_partials;
            const float* p = a.parts + (ix & (REDUCE_SHORT_FLAG - 1));
            const size_t G = REDUCE_GROUPS;
            int k = y;
            for (; k + 3 * REDUCE_GROUPS < P; k += 4 * REDUCE_GROUPS) {
                const float v0 = p[(size_t)k * a.stride], v1 = p[(k + G) * a.stride], v2 = p[(k + 2 * G) * a.stride], v3 = p[(k + 3 * G) * a.stride];
                s = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(s, v0), v1), v2), v3);
            }
            for (; k < P; k += REDUCE_GROUPS) s = __fadd_rn(s, p[(size_t)k * a.stride]);
        }
    }
    sh[y][x] = s;
    __syncthreads();
    if (y == 0 && j < a.n_out && !leave) {
        float t = sh[0][x];
#pragma unroll
        for (int g = 1; g < REDUCE_GROUPS; ++g) t = __fadd_rn(t, sh[g][x]);
        a.out[j] = t;
    }
}
hipError_t launch_reduce_partials(const ReducePartialsArgs& a, hipStream_t stream) {
    if (a.n_out <= 0 || a.n_partials < 1 || a.n_short < 0 || a.n_short > a.n_partials || a.stride < 1 || a.stride >= REDUCE_SHORT_FLAG) return hipErrorInvalidValue;
    if (a.aux && a.n_aux < 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((a.n_out + 31) / 32) + (a.aux ? 1u : 0u)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// one thread per row of 32 bf16 (64 bytes, four 16-byte loads), added in order in fp32
__global__ void __launch_bounds__(256) tile_row_sums_kernel(const void* tiles, long long n_rows, float* out) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* p = (const u32x4*)tiles + r * 4;
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32x4 w = p[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s = __fadd_rn(s, __builtin_bit_cast(float, w[k] << 16));
            s = __fadd_rn(s, __builtin_bit_cast(float, w[k] & 0xffff0000u));
        }
    }
    out[r] = s;
}
hipError_t launch_tile_row_sums(const void* tiles, long long n_rows, float* out, hipStream_t stream) {
    if (n_rows <= 0) return hipSuccess;
    if (n_rows >= (1ll << 39)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tile_row_sums_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, stream, tiles, n_rows, out);
    return hipGetLastError();
}

// one workgroup per tile: 16-byte loads of the tile into LDS, then per (sample, 8 features) eight 2-byte LDS reads down a
// column (the 32 lanes of a half wave read 32 consecutive samples: conflict-free) and one 16-byte store into the sample's row
__global__ void __launch_bounds__(256) tiles_to_rows_kernel(const void* tiles, int n_rays, int S, int W, void* rows) {
    __shared__ unsigned short t[256 * 32];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int bpr = (S + 31) >> 5;
    const long long blk = blockIdx.x;
    const int ray = (int)(blk / bpr), s0 = (int)(blk % bpr) * 32;
    const u32x4* src = (const u32x4*)tiles + blk * (W * 4);
    for (int c = threadIdx.x; c < W * 4; c += 256) ((u32x4*)t)[c] = src[c];
    __syncthreads();
    const int s = threadIdx.x & 31;
    if (s0 + s >= S) return;
    unsigned short* dst = (unsigned short*)rows + ((size_t)ray * S + s0 + s) * W;
    for (int g = threadIdx.x >> 5; g < W / 8; g += 8) {
        u32x4 w;
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (unsigned)t[(8 * g + 2 * k) * 32 + s] | ((unsigned)t[(8 * g + 2 * k + 1) * 32 + s] << 16);
        *(u32x4*)(dst + 8 * g) = w;
    }
}
hipError_t launch_tiles_to_rows(const void* tiles, int n_rays, int S, int W, void* rows, hipStream_t stream) {
    if (n_rays <= 0 || S < 1 || (W != 256 && W != 128)) return hipErrorInvalidValue;
    const long long nblk = (long long)n_rays * ((S + 31) / 32);
    if (nblk >= (1ll << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tiles_to_rows_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, tiles, n_rays, S, W, rows);
    return hipGetLastError();
}

// finite-difference view directions + their encoding, one thread per sample
namespace {
struct Dir { float d[3]; float r, n; };        // direction, |diff|, |diff| + eps
__device__ __forceinline__ Dir dir_at(const float* bent4, size_t ray0, int m) {      // direction of sample m >= 1 (p_m - p_{m-1})
    const f32x4 a = *(const f32x4*)(bent4 + (ray0 + m) * 4), b = *(const f32x4*)(bent4 + (ray0 + m - 1) * 4);
    Dir o;
    const float x = __fsub_rn(a[0], b[0]), y = __fsub_rn(a[1], b[1]), z = __fsub_rn(a[2], b[2]);
    o.r = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
    o.n = __fadd_rn(o.r, 0.000001f);
    o.d[0] = __fdiv_rn(x, o.n); o.d[1] = __fdiv_rn(y, o.n); o.d[2] = __fdiv_rn(z, o.n);
    return o;
}
template <bool B16>
__device__ __forceinline__ float enc_get(const void* p, size_t i) {
    if constexpr (B16) return __builtin_bit_cast(float, (unsigned)((const unsigned short*)p)[i] << 16);
    else return ((const float*)p)[i];
}
// gradient wrt the direction of row `row` (whose direction is d) from that row's encoding gradient
template <bool B16>
__device__ __forceinline__ void dir_grad(const DirEncodingArgs& a, size_t row, const float (&d)[3], float (&g)[3]) {
    const int C = 3 + 6 * a.L;
    const size_t o = row * C;
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] = enc_get<B16>(a.enc, o + c);
    for (int k = 0; k < a.L; ++k) {
        const float sc = (float)(1 << k);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sn, cs;
            sincosf(__fmul_rn(d[c], sc), &sn, &cs);
            g[c] += sc * (cs * enc_get<B16>(a.enc, o + 3 + 6 * k + c) - sn * enc_get<B16>(a.enc, o + 3 + 6 * k + 3 + c));
        }
    }
}
// gradient wrt diff_m = p_m - p_{m-1} (m >= 1): row m's direction gradient (+ row 0's for m == 1) through the normalisation
template <bool B16>
__device__ __forceinline__ void diff_grad(const DirEncodingArgs& a, size_t ray0, int m, float (&dd)[3]) {
    const Dir D = dir_at(a.bent4, ray0, m);
    float G[3];
    dir_grad<B16>(a, ray0 + m, D.d, G);
    if (m == 1) {
        float G0[3];
        dir_grad<B16>(a, ray0, D.d, G0);
#pragma unroll
        for (int c = 0; c < 3; ++c) G[c] += G0[c];
    }
    const float dot = D.d[0] * G[0] + D.d[1] * G[1] + D.d[2] * G[2];
    const float k2 = D.r > 0.0f ? dot / D.r : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) dd[c] = G[c] / D.n - D.d[c] * k2;
}
}  // namespace
template <bool B16>
__global__ void __launch_bounds__(256) dir_encoding_fwd_kernel(const DirEncodingArgs a) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)a.n_rays * a.S) return;
    const int j = (int)(t % a.S);
    const size_t ray0 = (size_t)(t - j);
    const Dir D = dir_at(a.bent4, ray0, j > 0 ? j : 1);
    const int C = 3 + 6 * a.L;
    auto put = [&](int i, float v) {
        if constexpr (B16) ((__bf16*)a.enc)[(size_t)t * C + i] = (__bf16)v;
        else ((float*)a.enc)[(size_t)t * C + i] = v;
    };
#pragma unroll
    for (int c = 0; c < 3; ++c) put(c, D.d[c]);
    for (int k = 0; k < a.L; ++k) {
        const float sc = (float)(1 << k);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sn, cs;
            sincosf(__fmul_rn(D.d[c], sc), &sn, &cs);
            put(3 + 6 * k + c, sn);
            put(3 + 6 * k + 3 + c, cs);
        }
    }
}
template <bool B16>
__global__ void __launch_bounds__(256) dir_encoding_bwd_kernel(const DirEncodingArgs a) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)a.n_rays * a.S) return;
    const int j = (int)(t % a.S);
    const size_t ray0 = (size_t)(t - j);
    float g[3] = {0.0f, 0.0f, 0.0f}, dd[3];
    if (j >= 1) {                       // p_j is the minuend of diff_j ...
        diff_grad<B16>(a, ray0, j, dd);
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] += dd[c];
    }
    if (j + 1 < a.S) {                  // ... and the subtrahend of diff_{j+1}
        diff_grad<B16>(a, ray0, j + 1, dd);
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] -= dd[c];
    }
    *(f32x4*)(a.g_bent4 + (size_t)t * 4) = f32x4{g[0], g[1], g[2], 0.0f};
}
hipError_t launch_dir_encoding(const DirEncodingArgs& a, bool backward, hipStream_t stream) {
    if (a.n_rays <= 0 || a.S < 2 || a.L < 0 || a.L > 10) return hipErrorInvalidValue;
    const long long total = (long long)a.n_rays * a.S;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (!backward) {
        if (a.enc_bf16) hipLaunchKernelGGL(dir_encoding_fwd_kernel<true>, grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(dir_encoding_fwd_kernel<false>, grid, dim3(256), 0, stream, a);
    } else {
        if (a.enc_bf16) hipLaunchKernelGGL(dir_encoding_bwd_kernel<true>, grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(dir_encoding_bwd_kernel<false>, grid, dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

}  // namespace nrn
