// nrnerf_loss.hip -- the training iteration's loss (reference training_wrapper_class.forward, train.py:207-287) as ONE forward and ONE
// backward kernel over the outputs of render_rays:
//   loss[r] = mean((rgb_map - target)^2) (+ mean((rgb0 - target)^2))                                             train.py:207-218, rnh:10-13
//           + offsets_weight * ( mean_s( w * |off|^(2 - rig) ) + rigidity_weight * mean_s( w * rig ) )            train.py:221-242
//           + divergence_weight * mean_s( (1 - exp(-relu(alpha))) * |div|^2 )                                     train.py:245-287, rnh:61-69
// with w = the coarse pass' visibility weights and (1 - exp(-relu(alpha))) DETACHED as in the reference (:223, rnh:65-66), off / rig
// the coarse pass' unmasked offsets and rigidity mask, div the per-sample divergence of compute_divergence_loss; the schedule factor
// (:240, 285) is folded into the two weights by the caller, or handed in as a device scalar (a replayed HIP graph changes it between steps).  As eager torch ops this is ~30 launches forward and ~50 backward of
// 2-5 us each on a 1024-ray step (profiles/r04_train_step_kernel_sequence_1024.txt).  One wave per ray, lanes stride over samples.
// Gradients follow torch's conventions at the singular points: d|x|/dx = 0 at x = 0, d(n^e)/de = 0 at n = 0 (e >= 1 here).
#include <hip/hip_runtime.h>

#include "nrnerf_loss.h"

namespace nrn {
namespace {
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

constexpr int RAYS_PER_WG = 4;

template <bool BWD>
__global__ void __launch_bounds__(RAYS_PER_WG * 64) loss_kernel(const LossArgs a) {
    const int lane = threadIdx.x & 63, ray = blockIdx.x * RAYS_PER_WG + (threadIdx.x >> 6);
    const int S = ray < a.n_rays ? a.n_samples : 0;         // (a wave beyond the last ray does nothing, but still counts its workgroup done)
    const float inv_s = S > 0 ? 1.0f / (float)S : 0.0f;
    float g = 0.0f;
    if (BWD && ray < a.n_rays) g = (a.g_loss ? a.g_loss[ray] : 0.0f) + (a.g_mean ? *a.g_mean / (float)a.n_rays : 0.0f);
    const int so = a.offsets_stride, sr = a.rigidity_stride;
    const float sched = a.schedule ? *a.schedule : 1.0f;
    const float w_off = a.offsets_weight * sched, w_div = a.divergence_weight * sched;
    float s_off = 0.f, s_rig = 0.f, s_div = 0.f;
    for (int s = lane; s < S; s += 64) {
        const size_t i = (size_t)ray * S + s;
        if (a.weights) {
            const float w = a.weights[i], rig = a.rigidity[sr * i];
            const float ox = a.offsets[so * i], oy = a.offsets[so * i + 1], oz = a.offsets[so * i + 2];
            const float nrm = sqrtf(ox * ox + oy * oy + oz * oz);                     // torch.norm(dim=-1)              :228
            const float e = 2.0f - rig;
            const float pw = powf(nrm, e);                                           // torch.pow(norm, 2 - rigidity)   :229
            if (!BWD) {
                s_off += w * pw;
                s_rig += w * rig;                                                    // :233
            } else {
                const float c = g * w_off * inv_s * w;
                // d/d off = c * e * nrm^(e - 1) * off / nrm (0 at off = 0);  d/d rig = c * (-pw * ln nrm) (0 at nrm = 0) + c * rigidity_weight
                const float dn = nrm > 0.0f ? c * e * powf(nrm, e - 1.0f) / nrm : 0.0f;
                a.g_offsets[3 * i] = dn * ox; a.g_offsets[3 * i + 1] = dn * oy; a.g_offsets[3 * i + 2] = dn * oz;
                a.g_rigidity[i] = (nrm > 0.0f ? -c * pw * logf(nrm) : 0.0f) + c * a.rigidity_weight;
            }
        }
        if (a.divergence) {
            const float al = a.alpha[i];
            const float wd = 1.0f - expf(-(al > 0.0f ? al : 0.0f));                  // train.py:264, detached (rnh:65-66)
            const float d = a.divergence[i];
            if (!BWD) s_div += wd * d * d;                                           // |div|^2 (rnh:61-62)
            else a.g_divergence[i] = g * w_div * inv_s * wd * 2.0f * d;
        }
    }
    if (!BWD) {
        s_off = wsum(s_off); s_rig = wsum(s_rig); s_div = wsum(s_div);
        if (lane == 0 && ray < a.n_rays) {
            float l = 0.f;
            const float* t = a.target + 3 * (size_t)ray;
            for (int k = 0; k < 2; ++k) {
                const float* m = k == 0 ? a.rgb_map : a.rgb0;
                if (!m) continue;
                m += 3 * (size_t)ray;
                const float d0 = m[0] - t[0], d1 = m[1] - t[1], d2 = m[2] - t[2];
                l += (d0 * d0 + d1 * d1 + d2 * d2) * (1.0f / 3.0f);                  // img2mse per ray (rnh:10-13)
            }
            if (a.weights) l += w_off * (s_off * inv_s + a.rigidity_weight * (s_rig * inv_s));
            if (a.divergence) l += w_div * (s_div * inv_s);
            a.loss[ray] = l;
        }
    } else if (lane < 6 && ray < a.n_rays) {
        const int k = lane / 3, c = lane % 3;
        const float* m = k == 0 ? a.rgb_map : a.rgb0;
        float* gm = k == 0 ? a.g_rgb_map : a.g_rgb0;
        if (m && gm) gm[3 * (size_t)ray + c] = g * (2.0f / 3.0f) * (m[3 * (size_t)ray + c] - a.target[3 * (size_t)ray + c]);
    }
}
// one workgroup per code; thread (slice q, column c): the rays q, q + Q, ... in order, then the Q slices in order.  Four rays per trip: the
// index loads of a trip are independent (one at a time the loop is a chain of memory latencies: 37 us for 1024 rays)
constexpr int CG_THREADS = 1024;
__global__ void __launch_bounds__(CG_THREADS) code_gradients_kernel(const CodeGradArgs a) {
    __shared__ float part[CG_THREADS];
    const int k = blockIdx.x, L = a.latent;
    const int Q = CG_THREADS / L;                           // (L <= 256, checked by the caller)
    const int c = threadIdx.x % L, q = threadIdx.x / L;
    float acc = 0.0f;
    if (q < Q)
        for (int r0 = q; r0 < a.n_rays; r0 += 4 * Q) {
            bool hit[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int r = r0 + u * Q; hit[u] = r < a.n_rays && a.index[r < a.n_rays ? r : 0] == (long long)k; }
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = hit[u] ? a.g[(size_t)(r0 + u * Q) * L + c] : 0.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u];
        }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (q == 0 && (int)threadIdx.x < L) {
        float sum = 0.0f;
        for (int j = 0; j < Q; ++j) sum += part[j * L + c];
        a.out[(size_t)k * L + c] = sum;
    }
}
}  // namespace

hipError_t launch_code_gradients(const CodeGradArgs& a, hipStream_t stream) {
    if (a.n_codes <= 0) return hipSuccess;
    hipLaunchKernelGGL(code_gradients_kernel, dim3(a.n_codes), dim3(CG_THREADS), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_loss(const LossArgs& a, bool backward, hipStream_t stream) {
    if (a.n_rays <= 0) return hipSuccess;
    const int grid = (a.n_rays + RAYS_PER_WG - 1) / RAYS_PER_WG;
    if (backward) hipLaunchKernelGGL(loss_kernel<true>, dim3(grid), dim3(RAYS_PER_WG * 64), 0, stream, a);
    else hipLaunchKernelGGL(loss_kernel<false>, dim3(grid), dim3(RAYS_PER_WG * 64), 0, stream, a);
    return hipGetLastError();
}
}  // namespace nrn
