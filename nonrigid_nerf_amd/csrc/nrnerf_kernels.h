// nrnerf_kernels.h -- host-visible launch interface of the HIP kernels (internal, C++).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nrn {

// device pointers of one pass' optional per-sample detail outputs (nullable)
struct SampleOut {
    float* vis;        // [N,S]   visibility_weights   (composite kernel)
    float* alpha;      // [N,S]   opacity_alpha        (composite kernel)
    float* init_pts;   // [N,S,3] initial_input_pts    (network kernel)
    float* unmasked;   // [N,S,3]
    float* masked;     // [N,S,3]
    float* in_pts;     // [N,S,3] bent points
    float* rigidity;   // [N,S,1]
};

struct Knobs {
    int has_cutoff;  float cutoff;
    int has_scaling; float scaling;
    int has_removal; float removal;     // only honoured when `detailed` (reference quirk, rnh:308-311)
    int detailed;
};

struct CompositeArgs {
    const float* rays;   int ray_stride;
    const float* raw4;       // [N,S,4]
    const float* z;          // [N,S] or nullptr: coarse linspace
    int lindisp;             // as NetArgs::lindisp
    int white_bkgd;          // rgb += 1 - acc (train.py:786-787)
    const float* noise;      // [N,S] added to sigma before the relu (raw_noise_std * randn, train.py:753,761) or nullptr
    const float* u;          // [N,I] uniforms for sample_pdf (perturb > 0, run_nerf_helpers.py:665) or nullptr: linspace
    int n_rays, S;
    int n_importance;        // I: > 0 -> also run sample_pdf + merge and write z_out [N,S+I], z_std
    float* rgb; float* disp; float* acc;     // [N,3],[N],[N]
    float* z_std;            // [N] or nullptr
    float* z_out;            // [N,S+I] merged sorted depths (workspace, required when I > 0)
    float* z_user;           // optional user copy of the depths of THIS pass ([N,S]) or nullptr
    float* vis; float* alpha;    // [N,S] optional
    // surface reduction (free_viewpoint_rendering.py:621-658): sample whose accumulated visibility is closest to 0.5
    const float* bent4;          // [N,S,4] from the network kernel, or nullptr: no reduction
    float* surf_pts;             // [N,3] bent point at that sample
    float* surf_rig;             // [N]   rigidity mask at that sample
    int* med_idx;                // [N]   its index
    // split-bender path (I > 0): the coarse depths are a subset of the merged depths and the bender is shared by both
    // networks (run_nerf_helpers.py:213-215), so the fine pass re-uses the coarse pass' bent points and only the I new
    // samples go through the bender again.  All nullptr = off.
    const float* split_bent_in;  // [N,S,4]   bent point + rigidity of the coarse samples (written by the coarse network kernel)
    float* split_bent_out;       // [N,S+I,4] the same rows moved to their position among the merged depths
    float* z_new;                // [N,I]     depths of the importance samples, in sample order
    uint8_t* rank_new;           // [N,I]     position of each importance sample among the merged depths (S + I <= 256)
};

// One pass (coarse or fine) of the per-sample network over n_rays * S samples.
struct NetArgs {
    const float* rays;   int ray_stride;
    const float* latents; int lat_stride;
    const float* z;          // [N,S] sample depths, or nullptr: coarse linspace between near and far
    int lindisp;             // coarse spacing linear in inverse depth (train.py:850-852); only read when z == nullptr
    const float* pts4;       // [N,S,4] ready-made network input points (xyz, w unused) instead of o + d z -- the bent points
                             // of the stand-alone bender kernel; only read by the variants without a fused bender; or nullptr
    int n_rays, S;
    const void* wstream;     // packed fragment stream of this pass (whole 16 KiB units, nrnerf_plan.h)
    const float* bias;       // [NTILES*32]
    float* raw4;             // [N,S,4] rgb + sigma workspace consumed by the composite kernel
    float* raw_out;          // [N,S,raw_ch] user-visible raw ("retraw") or nullptr
    int raw_ch;
    float* bent4;            // [N,S,4] bent point xyz + rigidity mask per sample (surface reduction) or nullptr
    SampleOut ex;
    Knobs knobs;
    // Fused compositing of this pass (kernel variants WITHOUT a fused bender; S <= 256): fuse_on != 0 -> every wave owns whole
    // rays, keeps their raw outputs in LDS and composites them itself with `fuse` (nrnerf_composite_ray.h; fuse.raw4 unused,
    // fuse.n_importance must be 0: no sampling follows a fused pass); raw4 is then neither written nor needed (may be nullptr).
    int fuse_on;
    CompositeArgs fuse;
    // net_kernel_x16 only, or nullptr: a device counter, ZERO at launch -- the workgroups then take the next group of rays (fused
    // compositing) / of blocks from it as they finish the previous one, instead of every gridDim-th one (round 6: the eight XCDs run the
    // kernel at rates 3.7 % apart, and with fixed shares the launch lasted as long as the slowest one's)
    unsigned* work_counter;
};


// (The trunk-only pass on v_mfma_f32_16x16x32, nrnerf_net_x16.h, is declared in nrnerf_x16_api.h.)

// Stand-alone bender (ray_bending.forward, run_nerf_helpers.py:507-577) over n_per_ray samples of every ray.
struct BendArgs {
    const float* rays;   int ray_stride;
    const float* latents; int lat_stride;
    const float* z;          // [N, n_per_ray] sample depths, or nullptr: coarse linspace between near and far
    int lindisp;             // as NetArgs::lindisp; only read when z == nullptr
    const uint8_t* rank;     // [N, n_per_ray] row of each sample among the out_stride rows of its ray, or nullptr: identity
    int n_rays, n_per_ray, out_stride;
    const void* wstream;     // packed bender + rigidity fragments (Plan<..., TRUNK = false>)
    const float* bias;
    float* bent4;            // [N, out_stride, 4] bent point xyz + rigidity mask
    Knobs knobs;
    // bend_kernel_x16 only, or nullptr: a device counter, ZERO at launch -- the waves then take chunks of block groups from it as they
    // finish their previous one instead of a fixed share each (two workgroups share a CU and the older one's waves win the issue
    // arbitration: with fixed shares the younger workgroup ran on alone for the last fifth of the launch, measured round 6)
    unsigned* work_counter;
};
hipError_t launch_bend(int precision, int arch_id, const BendArgs& a, int num_cus, hipStream_t stream);

// backward of raw2outputs (train.py:724-789) for one pass: gradients wrt raw4 from the gradients of the per-ray maps
struct CompositeBwdArgs {
    const float* rays;   int ray_stride;
    const float* raw4;       // [N,S,4]
    const float* z;          // [N,S] or nullptr: coarse linspace
    int lindisp, white_bkgd;
    const float* noise;      // [N,S] or nullptr
    int n_rays, S;
    const float* g_rgb;      // [N,3]
    const float* g_disp;     // [N] or nullptr
    const float* g_acc;      // [N] or nullptr
    const float* g_w;        // [N,S] gradient wrt the visibility weights, or nullptr
    float* d_raw4;           // [N,S,4]
};
hipError_t launch_composite_bwd(const CompositeBwdArgs& a, hipStream_t stream);

// training kernels of the trunk (nrnerf_train.h): forward with saved activations, backward-data
struct TrunkArgs {
    const float* pts4;       // [M,4] input points (xyz, pad), M = n_rays * S
    int n_rays, S;
    const void* wstream;     // forward: trunk-only stream (Plan<.., false, false>); backward: PlanB stream
    const float* bias;
    float* raw4;             // forward out [M,4]
    float* raw_out;          // forward out [M,raw_ch] or nullptr
    int raw_ch;
    void* acts;              // hidden activations: fp32 mode [D][M][W] float; bf16 mode [D][nblocks][W][32] bf16 (nrnerf_train.h)
    unsigned short* mask;    // bf16 mode: [D][nblocks][W/32][64] relu masks (forward writes, backward reads); fp32 mode: unused
    const float* d_raw4;     // backward in  [M,4]   (gradient wrt raw4; the 5th raw channel never reaches the loss)
    const float* ray_bias;   // forward in   [n_rays][2][W] fp32 or nullptr: added to the pre-activations of pts_linears[0] and
                             // pts_linears[skip + 1] of every sample of the ray (the latent columns of the time-conditioned baseline)
    void* d_pre;             // backward out [D][M][W] gradient wrt the pre-activations (float or bf16)
    float* d_pts4;           // backward out [M,4]
    // view-dependent head (the *_views kernels): the colour branch behind the trunk
    const float* dirs;       // [M,3] view direction of every sample
    void* hv;                // relu(views layer), W / 2 per sample: fp32 mode [M][W/2] float; bf16 mode [nblocks][W/2][32] bf16 (forward writes)
    unsigned short* hv_mask; // bf16 mode: [nblocks][64 lanes][W/64] relu bits of hv (forward writes, backward reads)
    void* d_pre_v;           // backward out: gradient wrt the views layer's pre-activation, layout of hv
    float* d_dirs;           // backward out [M,3] gradient wrt dirs, or nullptr
};

hipError_t launch_trunk_fwd_train_f32(const TrunkArgs&, int num_cus, hipStream_t);
hipError_t launch_trunk_fwd_train_bf16(const TrunkArgs&, int num_cus, hipStream_t);
hipError_t launch_trunk_bwd_f32(const TrunkArgs&, int num_cus, hipStream_t);
hipError_t launch_trunk_bwd_bf16(const TrunkArgs&, int num_cus, hipStream_t);
hipError_t launch_trunk_fwd_train_f32_views(const TrunkArgs&, int num_cus, hipStream_t);   // width 256 + view-dependent head
hipError_t launch_trunk_fwd_train_bf16_views(const TrunkArgs&, int num_cus, hipStream_t);
hipError_t launch_trunk_bwd_f32_views(const TrunkArgs&, int num_cus, hipStream_t);
hipError_t launch_trunk_bwd_bf16_views(const TrunkArgs&, int num_cus, hipStream_t);
hipError_t launch_trunk_fwd_train_f32_a5(const TrunkArgs&, int num_cus, hipStream_t);      // architecture 5: trunk width 128
hipError_t launch_trunk_fwd_train_bf16_a5(const TrunkArgs&, int num_cus, hipStream_t);
hipError_t launch_trunk_bwd_f32_a5(const TrunkArgs&, int num_cus, hipStream_t);
hipError_t launch_trunk_bwd_bf16_a5(const TrunkArgs&, int num_cus, hipStream_t);

// training kernels of the ray bender (nrnerf_train_bend.h): forward with saved activations, backward-data; always fp32
struct BendTrainArgs {
    const float* rays;   int ray_stride;
    const float* latents; int lat_stride;
    const float* z;          // [N,S] sample depths
    int n_rays, S;
    const void* wstream;     // forward: bender + rigidity stream (Plan<ShapeF32, A, true, false, false>); backward: PlanBB stream
    const float* bias;       // forward only
    Knobs knobs;
    float* bent4;            // [M,4] bent point xyz + rigidity mask (after the cutoff knob): forward writes, backward reads
    float* off4;             // [M,4] unmasked offsets xyz + tanh(rigidity logit):            forward writes, backward reads
    // the saved arrays below are fp32 for an fp32 model and bf16 otherwise (`bf16_arrays` of the launchers): only the
    // weight-gradient kernel reads their values (and rounds them to bf16 for the matrix pipe anyway in that mode), the
    // backward-data chain stays in fp32 registers and takes nothing but the SIGN of an activation from them
    void* acts_b;            // [BD-1][M][BW] hidden activations of the offset MLP
    void* acts_r;            // [RD-1][M][RW] hidden activations of the rigidity MLP
    const float* g_bent4;    // backward in  [M,4] gradient wrt the bent point (w ignored)
    const float* g_bent4_b;  // backward in  [M,4] or nullptr: a second one, added to the first
    const float* g_unmasked; // backward in  [M,3] gradient wrt the unmasked offsets, or nullptr
    const float* g_mask;     // backward in  [M]   gradient wrt the rigidity mask, or nullptr
    void* dz_b;              // backward out [BD-1][M][BW] gradient wrt the hidden pre-activations of the offset MLP
    void* dz_r;              // backward out [RD-1][M][RW] ... of the rigidity MLP
    float* dz_out4;          // backward out [M,4] gradient wrt the offsets (xyz) and the rigidity logit (w)
    float* d_lat;            // backward out [M,LAT] gradient wrt each sample's latent inputs
};
hipError_t launch_bend_fwd_train_a0(const BendTrainArgs&, int num_cus, hipStream_t, bool bf16_arrays);
hipError_t launch_bend_fwd_train_a1(const BendTrainArgs&, int num_cus, hipStream_t, bool bf16_arrays);
hipError_t launch_bend_bwd_a0(const BendTrainArgs&, int num_cus, hipStream_t, bool bf16_arrays);
hipError_t launch_bend_bwd_a1(const BendTrainArgs&, int num_cus, hipStream_t, bool bf16_arrays);

// weight gradients of the bender / rigidity MLPs (bend_wgrad, nrnerf_train_bend.h): products dz^T x over the samples of
// row-major fp32 arrays, at most 64 x 64 each
struct BendWgradJob {
    const void* dz; int ldz, f;       // [M][ldz], the first f <= 64 columns: gradient wrt a layer's pre-activations
    const void* x;  int ldx, g;       // [M][ldx], the first g <= 64 columns: that layer's input; nullptr: the offset MLP's input
                                      // row [point, latent code], formed on the fly from the ray records (BendWgradArgs)
    const void* dz2; const void* x2;  // optional second product of the same shapes and element types, added into the same dW (not
                                      // into db): the tangent chain of the divergence regulariser (bend_div_bwd), or nullptr
    int dz16, x16;                    // 1: the array's elements are bf16 (the saved arrays of a bf16 / f16 model), 0: fp32
};
constexpr int BEND_WGRAD_MAX_JOBS = 16;
constexpr int BEND_WGRAD_SLOT = 64 * 64 + 64;       // floats per (partial, job): dW [64][64] then db [64]
struct BendWgradArgs {
    BendWgradJob job[BEND_WGRAD_MAX_JOBS];
    int njobs, nparts;                // grid = (nparts / 4, njobs), 4 waves per workgroup, one partial per wave
    long long m;
    float* out;                       // [nparts][njobs][BEND_WGRAD_SLOT]
    // for jobs with x == nullptr: point = origin + direction * z, then the ray's latent code
    const float* rays; int ray_stride;
    const float* latents; int lat_stride, lat;
    const float* z; int S;
};
// bf16_operands: round the fp32 rows to bf16 in registers and contract on the bf16 matrix pipe (models in bf16 / f16 mode)
hipError_t launch_bend_wgrad(const BendWgradArgs&, hipStream_t, bool bf16_operands);

// Divergence regulariser of the ray bender (compute_divergence_loss / divergence_approx, run_nerf_helpers.py:22-116):
// d = e^T J e with J = d(masked offsets)/d(point), by ONE forward-mode tangent through both MLPs (the reference takes a
// vector-Jacobian product with create_graph=True and differentiates that graph again), and its backward pass: the
// value chain and the tangent chain share weights and relu masks (bend_div_fwd / bend_div_bwd, nrnerf_train_bend.h).
struct BendDivArgs {
    const float* pts;        // [M,3] points
    const float* latents; int lat_stride;   // one row per POINT, lat_stride floats apart (0: one code for every point)
    const float* e;          // [M,3] probe vectors
    long long m;
    const void* wstream;     // forward: bender + rigidity stream (Plan<ShapeF32, A, true, false, false>); backward: PlanBB stream
    const float* bias;       // forward only
    Knobs knobs;
    float* div;              // forward out [M]
    float* tvec;             // forward out [M,3] or nullptr: the tangent of the masked offsets along e (div = e . tvec)
    const float* g_tvec;     // backward in [M,3] or nullptr: gradient wrt tvec, used INSTEAD of g_div e
    float* off4;             // [M,4] unmasked offsets xyz + tanh(rigidity logit):                 forward writes, backward reads
    float* toff4;            // [M,4] tangent of the offsets xyz + tangent of the rigidity logit:  forward writes, backward reads
    void* acts_b;  void* tacts_b;     // [BD-1][M][BW] hidden activations / their tangents (after the relu mask); fp32 or bf16 as in
    void* acts_r;  void* tacts_r;     // [RD-1][M][RW]                                                            BendTrainArgs
    const float* g_div;      // backward in [M]
    void* dz_b;  void* dtz_b;         // backward out [BD-1][M][BW] gradient wrt the hidden pre-activations / their tangents
    void* dz_r;  void* dtz_r;         // backward out [RD-1][M][RW]
    float* dz_out4;          // backward out [M,4] gradient wrt the offsets (xyz) and the rigidity logit (w)
    float* dtz_out4;         // backward out [M,4] ... wrt their tangents
    float* d_lat;            // backward out [M,LAT] gradient wrt each point's latent inputs
    // RAY MODE of the forward kernel (rays != nullptr; nrnerf_render's exact Jacobian view directions on a non-compiled architecture,
    // rnh:358-385): point i = sample i % S of ray i / S -- o + d z, z from `zr` [M] or the coarse spacing between the ray's near and far --,
    // probe = the ray's unit direction (ray record columns 8..10), latents one row per RAY; pts / e are not read; every saved array,
    // div, off4 and toff4 may be nullptr (nothing is kept for a backward pass)
    const float* rays; int ray_stride;
    const float* zr; int S; int lindisp;
    float* dirs_out;         // [M,3] or nullptr: (J d) / |J d| + 1e-6 with J d = d + tvec's value (rnh:367-378: eps outside the division)
    // backward: the cotangents of a RENDER pass that evaluated the bender at the same points (BendTrainArgs' g_bent4 / g_bent4_b / g_unmasked /
    // g_mask), all optional: added to the value chain's, so that one backward pass + one weight-gradient launch serve the divergence term
    // and the coarse samples' bender evaluation of a training iteration (both chains are linear in their cotangents)
    const float* r_g_bent4; const float* r_g_bent4_b; const float* r_g_unmasked; const float* r_g_mask;
    // forward, optional: the bent point + rigidity mask rows bend_fwd_train writes (BendTrainArgs::bent4) -- with it the divergence forward
    // IS the render pass' bender evaluation of those samples (off4 is common to both)
    float* bent4;
};
hipError_t launch_bend_div_fwd_a0(const BendDivArgs&, int num_cus, hipStream_t, bool bf16_arrays);
hipError_t launch_bend_div_fwd_a1(const BendDivArgs&, int num_cus, hipStream_t, bool bf16_arrays);
hipError_t launch_bend_div_bwd_a0(const BendDivArgs&, int num_cus, hipStream_t, bool bf16_arrays);
hipError_t launch_bend_div_bwd_a1(const BendDivArgs&, int num_cus, hipStream_t, bool bf16_arrays);

// weight gradients of the trunk, bf16 mode (trunk_wgrad, nrnerf_train.h): a list of products  dz^T x  over the samples
struct WgradJob {
    const void* dz;          // [nblocks][W][32] bf16: gradient wrt a layer's pre-activations
    const void* x;           // [nblocks][xw][32] bf16: that layer's input (previous activations, or the encoding)
    int xw;                  // 256-wide trunk: W or 64 (encoding, zero padded); a multiple of 64
    float* dw;               // out [W][xw] fp32 of partial 0; partial c (one per workgroup of the job) at + c * pstride
    float* db;               // out [W] fp32 of partial 0: row sums of dz = bias gradient
    int kch;                 // workgroups (= partial sums) of this job: chosen so that every workgroup of the launch has about
                             // the same work and all of them are resident at once
    int wg0;                 // index of the job's first workgroup in the 1-D grid
    int rows;                // features of dz = rows of the product: W, or W / 2 (the colour branch's hidden layer)
};
constexpr int WGRAD_MAX_JOBS = 14;
struct WgradArgs {
    WgradJob job[WGRAD_MAX_JOBS];
    int njobs, nwg;          // grid = nwg = sum of the jobs' kch
    int sync_every;          // > 0: a workgroup barrier every that many pairs of blocks -- the four waves of a workgroup run free
                             // and the two that share a fragment drift apart until the second request misses L2 (1.58 x the needed
                             // HBM reads at 16 384 rays); a RARE barrier re-aligns them at next to no cost (one per block cost 15-45 %)
    long long nblocks;
    long long pstride;       // floats between consecutive partials (the caller adds the kch partials)
};
// the two small operands of trunk_wgrad in its layout, from the arrays the other kernels already have: the positional
// encoding of the input points and the gradient wrt the head's outputs, bf16 [nblocks][64][32] each (row 63 / rows >= 4 and
// the columns beyond a ray's end zero)
struct WgradOperandArgs {
    const float* pts4;       // [M,4]
    const float* d_raw4;     // [M,4]
    int n_rays, S, L;        // L encoding frequencies (3 + 6 L <= 63)
    void* enc;               // out bf16 [nblocks][64][32]
    void* g_head;            // out bf16 [nblocks][64][32]
    float* head_sums;        // out fp32 [nblocks][4] or nullptr: the sums of d_raw4's four channels over each block's samples (bf16 kernel only)
    // view-dependent head: the encoding of the samples' view directions (reference column order, zero padded to 64) as a third operand
    const float* dirs;       // [M,3] or nullptr
    int LV;
    void* encv;              // out, layout of enc
};
hipError_t launch_wgrad_operands(const WgradOperandArgs&, hipStream_t);
// fp32 mode: the same two operands as rows, fp32 [M][64] each (columns >= 3 + 6 L / >= 4 zero)
hipError_t launch_wgrad_operands_f32(const WgradOperandArgs&, hipStream_t);
hipError_t launch_trunk_wgrad_bf16(const WgradArgs&, hipStream_t);
hipError_t launch_trunk_wgrad_bf16_a5(const WgradArgs&, hipStream_t);
hipError_t launch_trunk_wgrad_bf16_views(const WgradArgs&, hipStream_t);       // + the colour branch's three jobs (rows = W / 2)
// fp32 mode (trunk_wgrad_f32): WgradJob::dz / x are fp32 rows [M][W] / [M][xw], WgradArgs::nblocks = M samples
hipError_t launch_trunk_wgrad_f32(const WgradArgs&, hipStream_t);
hipError_t launch_trunk_wgrad_f32_a5(const WgradArgs&, hipStream_t);
hipError_t launch_trunk_wgrad_f32_views(const WgradArgs&, hipStream_t);

// Re-pack weights on the device (nrnerf_model_update_device): dst[i] = convert(flat[src[i]]) (0 where src[i] < 0).
// fmt[i]: 0 = fp32, 1 = bf16, 2 = f16, 3 = f16((w - f16(w)) * 2^11), the lo part of the bender's split product;
// fmt == nullptr: all fp32 (bias tables).
// All images of a handle in ONE launch: a training step refreshes up to nine images (stream + bias table each) -- eighteen
// launches of a few microseconds, every iteration, when done one by one.  Segment k covers blocks [block0[k], block0[k + 1])
// of 256 elements.
constexpr int REPACK_MAX_SEGMENTS = 32;
struct RepackBatchArgs {
    const float* flat;
    const int32_t* src[REPACK_MAX_SEGMENTS];
    const uint8_t* fmt[REPACK_MAX_SEGMENTS];
    void* dst[REPACK_MAX_SEGMENTS];
    long long n[REPACK_MAX_SEGMENTS];
    unsigned block0[REPACK_MAX_SEGMENTS + 1];
    int n_segments;
};
hipError_t launch_repack_batch(const RepackBatchArgs& a, hipStream_t stream);

// stratified jitter of the coarse depths (train.py:855-868): z = lower + (upper - lower) * u between the mid-points
struct JitterArgs {
    const float* rays;   int ray_stride;
    const float* u;          // [N,S] uniforms in [0,1)
    int n_rays, S, lindisp;
    float* z_out;            // [N,S]
};
hipError_t launch_zjitter(const JitterArgs& a, hipStream_t stream);

struct RayGenArgs {
    float c2w[12];            // camera-to-world [3,4], row-major
    float fx, fy, cx, cy;
    int H, W;
    float near, far;
    float* rays;              // [H*W, ray_stride]
    int ray_stride;           // 8, or 11 to append the unit view direction
};
hipError_t launch_raygen(const RayGenArgs& a, hipStream_t stream);

// ---- run-time-parameterised network kernel (nrnerf_generic.h): any architecture outside the compiled set
constexpr int GEN_MAX_LAYERS = 28;
constexpr int GEN_WAVES = 4;
constexpr int GEN_MAXT = 4;           // output tiles per wave and layer: widths up to 4 * 4 * 32 = 512
constexpr int GEN_MAX_W = 512;
constexpr int GEN_MAX_E = 176;        // padded width of E: 3 + 6 * 16 = 99 encoding columns + 64 latent columns, multiple of 16
constexpr int GEN_MAX_V = 64;         // padded width of V: 3 + 6 * 10 = 63
enum GenBuf : int { GB_E = 0, GB_H = 1, GB_V = 2, GB_O = 3, GB_OUT0 = 4, GB_OUT1 = 5, GB_OUT2 = 6 };     // GB_OUT0 / 1 / 2: straight to GenArgs::gout[.] (training, backward-data)

struct GenLayer {
    int w_frag;          // index of fragment (tile 0, slab 0) in the weight stream; fragment (t, s) = w_frag + t * (ns0 + ns1) + s
    int bias_tile;       // index of tile 0 in the bias table ([tile][lane half][16] floats, as nrnerf_plan.h)
    int nt;              // output tiles of 32 rows
    int src0, ns0;       // first source buffer (GenBuf) and its k-slabs (of KS columns: 16 / 2)
    int src1, ns1;       // second source (ns1 = 0: none)
    int dst;             // GB_H, or GB_O for a head
    int relu;
    int o_col;           // dst == GB_O: output row r of the layer goes to O[sample][o_col + r]  (rows < o_rows)
    int o_rows;
    // training of a non-compiled architecture (round 5): -1 = off
    int save_idx;        // dst == GB_H: after the write-back the tile's H rows (columns < save_w) are copied to GenArgs::save[save_idx]
    int mask_idx;        // dst == GB_H: outputs are zeroed where GenArgs::mask[mask_idx][sample][column] <= 0 (the relu's derivative)
    int boff0, boff1;    // a source that is GB_H is read from column boff (elements, a multiple of 16) on -- the rows of d raw parked beside the activations
};

struct GenArgs {
    int mode;                    // 0: ray bender (bent4 and detail outputs), 1: canonical network (raw4 / raw_out),
                                 // 2: backward-data of the canonical network (training): H starts as the rows of `draw`, no E / V
    const float* rays; int ray_stride;
    const float* latents; int lat_stride, lat;      // lat: latent columns appended to E (bender: always; network: time-conditioned baseline)
    const float* z;              // [N,S] sample depths or nullptr: coarse spacing between near and far
    int lindisp;
    const float* pts4;           // network: [N,S,4] input points (the bender's output) or nullptr: o + d z
    int n_rays, S;
    int L, LV;                   // encoding frequencies of the point / of the view direction (LV < 0: no view-dependent head)
    int dirs_from_pts;           // view directions: 1 = finite differences of pts4 along the ray (rnh:316-356), 0 = the rays' unit directions
    const void* wstream; const float* bias;
    int n_layers;
    GenLayer layer[GEN_MAX_LAYERS];
    int ke, kv, kh;              // padded row widths of E, V and H in elements (multiples of 16; kh = the widest hidden layer)
    int n_bias_tiles;            // tiles of the bias table (all layers)
    int bias_in_lds;             // set by the launcher: the whole table fits next to the activation buffers
    // outputs
    float* raw4; float* raw_out; int raw_ch;        // network
    float* bent4;                // bender: [N,S,4] bent point + rigidity mask (network: read for the removal knob when detailed)
    SampleOut ex;
    Knobs knobs;
    // training of a non-compiled architecture: saved arrays [index][n_rays * S][save_w] in the model's element type (fp32 / 16-bit)
    void* save; const void* mask; long long save_stride; int save_w;
    const float* draw; int draw_ch;      // mode 2: [n_rays * S][draw_ch] gradient of the raw outputs, put into H columns [draw_col, draw_col + 16)
    int draw_col;
    float* gout[3]; int gout_w, gout_w2; // mode 2: fp32 outputs [n_rays * S][gout_w] of the layers with dst == GB_OUT0 / GB_OUT1, [..][gout_w2] of GB_OUT2
    const float* dirs;                   // mode 1, LV >= 0: [n_rays * S][3] one view direction per SAMPLE (training), or nullptr
};

hipError_t launch_generic(int precision, const GenArgs& a, int num_cus, hipStream_t stream);

// precision ids match nrnerf_precision
enum { PREC_F32 = 0, PREC_BF16 = 1, PREC_F16 = 2 };

struct NetLaunchInfo { int grid; int block; size_t lds_bytes; };

// returns hipSuccess or an error; `arch_id` selects a compiled architecture (0 = default 8x256)
hipError_t launch_net(int precision, bool has_bend, bool views, int arch_id, const NetArgs& a, int num_cus,
                      hipStream_t stream);
hipError_t launch_composite(const CompositeArgs& a, hipStream_t stream);

}  // namespace nrn
