// nrnerf_gen_train.h -- host-side entry points of nrnerf_gen_train.hip (weight gradients and encoding rows of a non-compiled trunk's
// training step), seen by the API layer only
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nrn {
// one panel (up to 256 x 256) of one product  out[o][k] = sum_m a[m][o] b[m][k]  (+ the column sums of a when k0 == 0 and bias_off >= 0)
struct TnSubJob {
    const void* a; const void* b;       // row-major [n_rows][lda] / [n_rows][ldb], bf16 or fp32 (TnKernelArgs decides)
    int lda, ldb, wo, wi;               // leading dimensions (elements); columns of a / b that take part
    int o0, k0;                         // the panel's first output row / column
    int ldo;                            // leading dimension of the output matrix
    long long out_off, bias_off;        // positions of out[0][0] / of the bias row in a record (-1: no bias)
};
constexpr int TN_MAX_SUBJOBS = 40;
struct TnKernelArgs {
    TnSubJob sub[TN_MAX_SUBJOBS];
    int n_sub, kch;                     // panels of this launch; chunks of samples (= records of partial sums)
    long long n_rows, total;            // samples; floats per record
    float* partials;                    // [kch][total]
};
hipError_t launch_tn_clear(float* partials, long long total, int kch, hipStream_t stream);
hipError_t launch_tn_products(const TnKernelArgs& a, bool f32, hipStream_t stream);
hipError_t launch_tn_reduce(const float* partials, long long total, int kch, float* out, hipStream_t stream);

struct EncodingArgs {
    const float* src; int src_stride;   // [n_rows][src_stride >= 3]
    long long n_rows; int L;
    void* enc; int enc_cols; int enc_bf16;
    const float* codes; int n_lat; int rows_per_code;
    const float* d_enc0; const float* d_enc1; int d_enc_stride;
    float* d_src; int d_src_stride;
};
hipError_t launch_encoding_rows(const EncodingArgs& a, bool backward, hipStream_t stream);
}  // namespace nrn
