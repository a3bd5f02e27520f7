// nrnerf_gx16_bwd_api.h -- arguments of the width-class backward-data kernel (nrnerf_gx16_bwd.h), seen by the API layer
#pragma once
#include <hip/hip_runtime.h>

namespace nrn {
struct GxBwdArgs {
    const float* d_raw4;        // [N, S, 4] gradient wrt rgb, sigma
    const void* relu_bits;      // [depth][n_blocks16][64][WC / 32] bytes (the forward kernel's)
    void* d_pre;                // [depth][save_stride] rows of save_w bf16: gradient wrt every layer's pre-activations (out)
    long long save_stride; int save_w;
    float* d_enc0;              // [N * S][enc_w] gradient of the encoding through pts_linears[0] (out)
    float* d_enc1;              // ... through the layer behind the skip connection (out; null without one)
    int enc_w;                  // 3 + 6 L
    int n_rays, S;
    const void* wstream; const float* bias;      // the backward program's fragment blocks (+ tail copy); bias table: zeros
    int depth, skip, L, n_bias_tiles;
};

hipError_t launch_gx16_bwd(int wc, const GxBwdArgs& a, int num_cus, hipStream_t stream);
}  // namespace nrn
