// nrnerf_loss.h -- launcher of the fused training-loss kernels (nrnerf_loss.hip); seen by nrnerf_api.cpp only.
#pragma once
#include <hip/hip_runtime.h>

namespace nrn {
struct LossArgs {
    int n_rays, n_samples;
    const float* rgb_map; const float* rgb0; const float* target;          // [N,3]; rgb0 may be null
    const float* weights; const float* offsets; const float* rigidity;     // [N,S], [N,S,3], [N,S]; weights null: no offsets term
    const float* alpha; const float* divergence;                           // [N,S]; divergence null: no divergence term
    float offsets_weight, rigidity_weight, divergence_weight;
    const float* schedule;                                                  // device scalar multiplied into the two regularisers' weights, or null
    float* loss;                                                            // forward out [N]
    const float* g_loss;                                                    // backward in [N]
    float* g_rgb_map; float* g_rgb0; float* g_offsets; float* g_rigidity; float* g_divergence;     // backward out
    int offsets_stride, rigidity_stride;                                    // floats from one sample's offsets / rigidity to the next (3 / 1 = packed)
    const float* g_mean;                                                    // backward in, or null: device scalar, gradient wrt the mean of loss[] (added to g_loss[r] as g_mean / N)
};
hipError_t launch_loss(const LossArgs&, bool backward, hipStream_t);

// gradient of codes[index] (training_wrapper_class.forward, train.py:173-188): out[k][c] = sum over the rays r with index[r] == k of g[r][c],
// added in ray order (deterministic)
struct CodeGradArgs {
    const long long* index;     // [N]
    const float* g;             // [N][L]
    int n_rays, latent, n_codes;
    float* out;                 // [n_codes][L]
};
hipError_t launch_code_gradients(const CodeGradArgs&, hipStream_t);
}  // namespace nrn
