// nrnerf_optim.hip -- the optimiser step of a training iteration, followed by the weight re-pack the next forward needs
// (reference: torch.optim.Adam over grad_vars, train.py:655-658, stepped at train.py:1606-1610; the re-pack is what
// nrnerf_model_update_device does after it).
//
// Why: at the reference's batch size (N_rand = 1024) the step is a chain of short launches; torch's fused Adam is three
// multi_tensor_apply launches (89 us for 1.2 M parameters in 50 tensors), the copy of the parameters into the library's flat vector a
// fourth (29 us) and the re-pack a fifth (27 us) -- 8 % of a 1.8 ms step for 34 MB of traffic.  Here: ONE Adam launch over all
// (parameter, gradient, exp_avg, exp_avg_sq) runs laid end to end -- the parameters ARE the library's flat vector (training.FusedAdam
// re-homes them as views of it), so nothing is copied -- and the re-pack launch right behind it.
// (Both as ONE kernel with a grid barrier between the phases was built first and measured slower: on this multi-die part a device-scope
// barrier writes the dirty L2 lines of every die back before anyone may pass -- 116-167 us per step against 40 for the two launches.)
#include <hip/hip_runtime.h>

#include "nrnerf_kernels.h"
#include "nrnerf_optim.h"

namespace nrn {

// torch.optim.Adam (no weight decay, no amsgrad, maximize = False), fp32, the arithmetic of its fused kernel:
//   m = lerp(m, g, 1 - beta1);  v = beta2 v + (1 - beta2) g g;  p -= (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
__global__ void __launch_bounds__(1024) adam_kernel(const AdamKernelArgs a) {
    const float t = *a.step + 1.0f;
    const float lr = a.lr_device ? *a.lr_device : a.lr;
    const float bc1 = 1.0f - powf(a.beta1, t), bc2_sqrt = sqrtf(1.0f - powf(a.beta2, t));
    const float step_size = lr / bc1;
    const long long gtid = (long long)blockIdx.x * 1024 + threadIdx.x, gsize = (long long)gridDim.x * 1024;
    // ONE loop over the granules (4 elements) of all runs laid end to end -- a loop per run costs a memory round trip per run, and a
    // step has ~50 of them when autograd hands every parameter its own gradient tensor (181 us measured that way; torch's fused Adam: 99)
    const long long ngran = a.gran0[a.n_segments];
    for (long long q = gtid; q < ngran; q += gsize) {
        int k = 0;
        for (int j = 1; j < a.n_segments; ++j)
            if (q >= a.gran0[j]) k = j;
        const AdamSegment s = a.seg[k];
        const long long i0 = (q - a.gran0[k]) * 4;
        const bool vec = ((((size_t)s.p | (size_t)s.g | (size_t)s.m | (size_t)s.v) & 15) == 0) && i0 + 4 <= (long long)s.n;
        typedef float f4 __attribute__((ext_vector_type(4)));
        if (vec) {
            const f4 g = *(const f4*)(s.g + i0);
            f4 p = *(f4*)(s.p + i0), m = *(f4*)(s.m + i0), v = *(f4*)(s.v + i0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                m[e] = m[e] + (1.0f - a.beta1) * (g[e] - m[e]);
                v[e] = a.beta2 * v[e] + (1.0f - a.beta2) * g[e] * g[e];
                p[e] -= step_size * m[e] / (sqrtf(v[e]) / bc2_sqrt + a.eps);
            }
            *(f4*)(s.p + i0) = p; *(f4*)(s.m + i0) = m; *(f4*)(s.v + i0) = v;
        } else {
            for (long long i = i0; i < i0 + 4 && i < (long long)s.n; ++i) {
                const float g = s.g[i];
                const float m = s.m[i] + (1.0f - a.beta1) * (g - s.m[i]);
                const float v = a.beta2 * s.v[i] + (1.0f - a.beta2) * g * g;
                s.m[i] = m; s.v[i] = v;
                s.p[i] -= step_size * m / (sqrtf(v) / bc2_sqrt + a.eps);
            }
        }
    }
    // the step count: every workgroup has read it (above) before it counts itself done; the LAST one to finish writes the new count and
    // re-arms the counter -- no workgroup waits for another
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = atomicAdd(&a.barrier[0], 1u);
        if (done == gridDim.x - 1) {
            *a.step = t;
            __hip_atomic_store(&a.barrier[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

hipError_t launch_adam(const AdamKernelArgs& a, int num_cus, hipStream_t stream) {
    if (a.n_segments < 0 || a.n_segments > ADAM_MAX_SEGMENTS || !a.step || !a.barrier) return hipErrorInvalidValue;
    const long long work = (a.gran0[a.n_segments] + 1023) / 1024;
    long long grid = 2ll * num_cus;
    if (work < grid) grid = work > 0 ? work : 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)grid), dim3(1024), 0, stream, a);
    return hipGetLastError();
}

}  // namespace nrn
