// nrnerf_optim.h -- host-side entry point of the fused optimiser step (nrnerf_optim.hip), seen by the API layer only
#pragma once
#include "nrnerf_kernels.h"

namespace nrn {
constexpr int ADAM_MAX_SEGMENTS = 40;
struct AdamSegment { float* p; const float* g; float* m; float* v; unsigned long long n; };
struct AdamKernelArgs {
    AdamSegment seg[ADAM_MAX_SEGMENTS];
    long long gran0[ADAM_MAX_SEGMENTS + 1];   // first granule (4 elements) of every run when all runs are laid end to end
    int n_segments;
    float lr, beta1, beta2, eps;
    const float* lr_device;     // overrides lr when not null (a learning-rate schedule inside a captured step)
    float* step;                // device scalar: steps taken so far; incremented by the launch
    unsigned* barrier;          // two words of device memory, zero before the first launch (the model handle owns them): [0] counts finished workgroups
};
hipError_t launch_adam(const AdamKernelArgs& a, int num_cus, hipStream_t stream);
}  // namespace nrn
