// Small device-side helpers of the training path that are neither network nor compositing kernels (nrnerf_composite.hip
// holds them): the row merge of the split fine bender and the reduction of weight-gradient partial sums into the
// parameters' own layouts.  Only nrnerf_api.cpp and nrnerf_composite.hip see this header.
#pragma once
#include <hip/hip_runtime.h>

namespace nrn {

// rows of the S coarse samples and of the I importance samples of a ray <-> the same rows in merged-depth order
// (importance sample i sits at row rank_new[n][i], the coarse samples fill the other rows in order); two [., 4] arrays at once
struct MergeRowsArgs {
    int n_rays, S, I;
    const unsigned char* rank_new;      // [N][I]
    const float* c_a; const float* c_b; // forward in / inverse out   [N][S][4]      (b may be null)
    const float* n_a; const float* n_b; //                            [N][I][4]
    float* m_a; float* m_b;             // forward out / inverse in   [N][S + I][4]
    int inverse;                        // 0: (c, n) -> m;  1: m -> (c, n)
};
hipError_t launch_merge_rows(const MergeRowsArgs&, hipStream_t);

// out[j] = sum over the first P(j) records p of parts[p * stride + (index[j] & OFFSET)],  P(j) = n_short where index[j] has
// the SHORT flag, n_partials otherwise; 0 where index[j] < 0.  Fixed order of additions (deterministic): REDUCE_GROUPS
// interleaved groups of records, each added in order, then the group sums in order.
struct ReducePartialsArgs {
    const float* parts; long long stride; int n_partials, n_short;
    const int* index; long long n_out; float* out;
    // column sums of a second array (one more workgroup): out[aux_pos[c]] = sum over r < n_aux of aux[r][c], c < 4 (aux_pos[c] < 0: skipped);
    // those positions carry index -2 (left alone by the threads above).  The trunk's head bias gradient from wgrad_operands_kernel's
    // per-block sums (torch.sum over d raw: a memset and a reduction, 15 us per network and step)
    const float* aux; int n_aux; long long aux_pos[4];
};
constexpr int REDUCE_SHORT_FLAG = 0x40000000;
constexpr int REDUCE_GROUPS = 8;
hipError_t launch_reduce_partials(const ReducePartialsArgs&, hipStream_t);

// the coarse depths (JitterArgs / zjitter_kernel, train.py:847-868) AND the sample points o + d z (train.py:871-873) in one launch
struct SamplePointsArgs {
    const float* rays; int ray_stride;
    const float* u;          // [N,S] uniforms or nullptr
    int n_rays, S, lindisp;
    float* z_out;            // [N,S]
    float* pts_out;          // [N,S,3]
};
hipError_t launch_sample_points(const SamplePointsArgs&, hipStream_t);

// out[r] = sum of the 32 bf16 values of row r of a [rows][32] array (the sample axis of the training path's block tiles), fp32
hipError_t launch_tile_row_sums(const void* tiles, long long n_rows, float* out, hipStream_t);

// [block][feature][32 samples] bf16 tiles of one layer -> rows [n_rays * S][W] bf16 (the samples a ray really has)
hipError_t launch_tiles_to_rows(const void* tiles, int n_rays, int S, int W, void* rows, hipStream_t);

// The view-dependent head's input under autograd (run_nerf_helpers.py:288-290, 316-356, 120-150): finite-difference view
// directions of the bent points, d_j = (p_j - p_{j-1}) / (|p_j - p_{j-1}| + 1e-6), d_0 = d_1, and their positional encoding
// [d, sin(2^k d), cos(2^k d)] (k < L), one row of 3 + 6 L values per sample; backward: gradient wrt the bent points.
struct DirEncodingArgs {
    const float* bent4;      // [N][S][4] (xyz)
    int n_rays, S, L;
    void* enc;               // forward out / backward in (the gradient): [N * S][3 + 6 L], fp32 or bf16
    int enc_bf16;
    float* g_bent4;          // backward out [N][S][4] (w = 0)
};
hipError_t launch_dir_encoding(const DirEncodingArgs&, bool backward, hipStream_t);

}  // namespace nrn
