/*
 * nrnerf.h -- C ABI of libnrnerf_hip.so: the MI355X-native per-ray renderer that
 * replaces NR-NeRF's render_rays / batchify_rays hot path.
 *
 * Boundary being replaced (reference facebookresearch/nonrigid_nerf):
 *   batchify_rays          train.py:108-137   (chunk loop; subsumed: one nrnerf_render call takes any n_rays)
 *   render_rays            train.py:792-980   (whole per-ray algorithm)
 *     run_network/batchify train.py:57-105, 27-54
 *     Embedder.embed       run_nerf_helpers.py:120-168
 *     NeRF.forward         run_nerf_helpers.py:240-314
 *     ray_bending.forward  run_nerf_helpers.py:507-584
 *     raw2outputs          train.py:724-789
 *     sample_pdf           run_nerf_helpers.py:651-698
 *
 * Conventions
 *   - plain C, no torch types; every pointer in nrnerf_render_args is a DEVICE pointer the
 *     caller owns (inputs, outputs and workspace).  nrnerf_render allocates nothing.
 *   - weights in nrnerf_model_desc are HOST pointers, fp32, PyTorch nn.Linear layout
 *     ([out_features, in_features] row-major); they are packed into MFMA fragment order and
 *     copied to the device once, at nrnerf_model_create.
 *   - every function returns 0 (NRNERF_OK) or a negative nrnerf_status; nothing throws or
 *     aborts across the ABI.  nrnerf_render is asynchronous on the given hipStream_t.
 *   - no global mutable state: safe to call from several host threads on different devices /
 *     models (the reference's DataParallel wrapper does that, train.py:300-323).
 */
#ifndef NRNERF_H
#define NRNERF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRNERF_ABI_VERSION 8
/* samples per ray and pass: nrnerf_render and the training entry points (the split fine bender -- nrnerf_merge_rows,
 * nrnerf_composite_args.rank_new -- up to 256 merged samples: 8-bit ranks) */
#define NRNERF_MAX_SAMPLES 1024

typedef enum nrnerf_status {
    NRNERF_OK = 0,
    NRNERF_ERR_INVALID = -1,      /* bad argument (null pointer, negative size, misaligned buffer) */
    NRNERF_ERR_UNSUPPORTED = -2,  /* architecture / flag combination this build has no kernel for */
    NRNERF_ERR_HIP = -3,          /* a HIP runtime call failed (no device, launch failure, ...) */
    NRNERF_ERR_WORKSPACE = -4,    /* workspace smaller than nrnerf_workspace_bytes() */
    NRNERF_ERR_NOMEM = -5,        /* device memory, or host memory while packing weights */
    NRNERF_ERR_INTERNAL = -6      /* an internal consistency check failed (a C++ exception was caught at the boundary) */
} nrnerf_status;

/* arithmetic type of the MLP contractions (accumulation is always fp32; positional encoding,
 * point generation, compositing and sampling are always fp32) */
typedef enum nrnerf_precision {
    NRNERF_PREC_F32 = 0,   /* v_mfma_f32_32x32x2_f32: exact fp32, the parity mode */
    NRNERF_PREC_BF16 = 1,  /* v_mfma_f32_32x32x16_bf16: the headline throughput mode */
    NRNERF_PREC_F16 = 2    /* v_mfma_f32_32x32x16_f16 */
} nrnerf_precision;

/* one nn.Linear: weight [out_features, in_features] row-major fp32, bias [out_features] or NULL */
typedef struct nrnerf_linear {
    const float* weight;
    const float* bias;
    int32_t out_features;
    int32_t in_features;
} nrnerf_linear;

/* canonical NeRF MLP (reference NeRF.__init__, run_nerf_helpers.py:172-238) */
typedef struct nrnerf_mlp_desc {
    int32_t depth;            /* D: number of pts_linears (8) */
    int32_t width;            /* W (256) */
    int32_t skip;             /* index i after which [input, h] are concatenated (4); -1 = none */
    int32_t output_ch;        /* rows of output_linear (4, or 5 when N_importance > 0; train.py:593) */
    int32_t use_viewdirs;     /* 0: output_linear head; 1: alpha/feature/views/rgb head */
    int32_t time_conditioned; /* naive baseline: latent appended to the net input (rnh:207-209) */
    const nrnerf_linear* pts_linears;   /* [depth] */
    nrnerf_linear output_linear;        /* use_viewdirs == 0 */
    nrnerf_linear alpha_linear;         /* use_viewdirs == 1 ... */
    nrnerf_linear feature_linear;
    nrnerf_linear views_linear;         /* views_linears[0] */
    nrnerf_linear rgb_linear;
} nrnerf_mlp_desc;

/* ray-bending deformation + rigidity networks (reference ray_bending, run_nerf_helpers.py:388-505) */
typedef struct nrnerf_bender_desc {
    int32_t latent_size;      /* 32 */
    int32_t depth;            /* network_depth (5) */
    int32_t hidden;           /* hidden_dimensions (64) */
    int32_t rigidity_depth;   /* 3 */
    int32_t rigidity_hidden;  /* 32 */
    const nrnerf_linear* network;           /* [depth]; last layer has bias == NULL */
    const nrnerf_linear* rigidity_network;  /* [rigidity_depth] */
} nrnerf_bender_desc;

typedef struct nrnerf_model_desc {
    uint32_t struct_size;     /* sizeof(nrnerf_model_desc), for forward compatibility */
    int32_t precision;        /* nrnerf_precision */
    int32_t multires;         /* L of the xyz encoding (10) */
    int32_t multires_views;   /* L of the direction encoding (4); ignored without viewdirs */
    int32_t device;           /* HIP device ordinal the model lives on */
    const nrnerf_bender_desc* bender;   /* NULL: no ray bending (plain NeRF) */
    const nrnerf_mlp_desc* coarse;      /* network_fn */
    const nrnerf_mlp_desc* fine;        /* network_fine; NULL: reuse coarse (train.py:925) */
    int32_t exact_viewdirs;   /* with bender + view-dependent head: 0 = finite-difference directions (the configs' default,
                                 approx_nonrigid_viewdirs=True, rnh:316-356); 1 = normalised J . d from the bender's
                                 Jacobian (exact_nonrigid_viewdirs, rnh:358-385) */
    uint32_t flags;           /* bits of nrnerf_model_flags, since ABI 7; 0 = the defaults */
} nrnerf_model_desc;

/* Kernel-selection switches of a model handle (ABI 7: fields of the call, not process environment -- the library reads no
 * environment variable).  They select among kernels that compute the same function; 0 is what a caller wants. */
typedef enum nrnerf_model_flags {
    NRNERF_MODEL_FORCE_GENERIC = 1u << 0,  /* the run-time-parameterised kernel also for a compiled shape (tests: one route against the other) */
    NRNERF_MODEL_NO_X16_F16 = 1u << 1      /* NRNERF_PREC_F16: no 16x16x32 images; the split path then stays bit-identical to the fused pass */
} nrnerf_model_flags;

typedef struct nrnerf_model nrnerf_model;   /* opaque: packed weights resident in HBM */

/* per-sample detail tensors of one pass ("detailed_output", train.py:960-972); any may be NULL */
typedef struct nrnerf_sample_outputs {
    float* visibility_weights;  /* [N, S]    */
    float* opacity_alpha;       /* [N, S]    */
    float* initial_input_pts;   /* [N, S, 3] */
    float* unmasked_offsets;    /* [N, S, 3] */
    float* masked_offsets;      /* [N, S, 3] */
    float* input_pts;           /* [N, S, 3]  bent points */
    float* rigidity_mask;       /* [N, S, 1] */
} nrnerf_sample_outputs;

typedef struct nrnerf_render_args {
    uint32_t struct_size;       /* sizeof(nrnerf_render_args) */
    int32_t n_rays;             /* N (any size; the chunk loop of batchify_rays is not needed) */
    int32_t n_samples;          /* N_samples  (S, coarse)        <= NRNERF_MAX_SAMPLES */
    int32_t n_importance;       /* N_importance (I); S' = S + I   <= NRNERF_MAX_SAMPLES (the reference has no cap, train.py:1090-1094) */
    /* inputs */
    const float* rays;          /* [N, ray_stride]: o3, d3, near, far (, unit viewdir3)  train.py:397-399 */
    int32_t ray_stride;         /* floats per row: 8 or 11 */
    const float* latents;       /* [N, latent_size] ray_bending_latents; may be NULL without bender */
    int32_t latent_stride;      /* floats between rows; 0 = one code broadcast to every ray */
    /* test-time editing knobs, read per call (free_viewpoint_rendering.py:264-283) */
    int32_t has_rigidity_cutoff;   float rigidity_cutoff;     /* mask[mask <= cutoff] = 0   rnh:563-564 */
    int32_t has_test_time_scaling; float test_time_scaling;   /* masked_offsets *= scaling  rnh:568-569 */
    int32_t has_removal_threshold; float removal_threshold;   /* sigma *= 0 where mask >= thr; applied only
                                                                  when per-sample details are requested, as in
                                                                  the reference (rnh:308-311) */
    int32_t detailed_output;    /* mirrors the reference flag (gates the removal threshold) */
    /* outputs: final pass (fine if I > 0, else coarse) */
    float* rgb_map;             /* [N, 3] */
    float* disp_map;            /* [N]    */
    float* acc_map;             /* [N]    */
    float* raw;                 /* [N, S', C] or NULL ("retraw"); C = output_ch, or 4 with viewdirs */
    /* outputs: coarse pass when I > 0 (NULL allowed) */
    float* rgb0;                /* [N, 3] */
    float* disp0;               /* [N]    */
    float* acc0;                /* [N]    */
    float* z_std;               /* [N]    */
    float* z_vals;              /* [N, S'] merged, sorted sample depths (not a reference key; NULL allowed) */
    /* surface reduction of the final pass (what free_viewpoint_rendering.py:621-658 extracts from the per-sample
     * detail tensors on the host): the sample whose accumulated visibility weight is closest to 0.5.  All NULL = off. */
    float* surface_pts;         /* [N, 3] bent point at that sample   (fine_input_pts[median]) */
    float* surface_rigidity;    /* [N]    rigidity mask at that sample (0 without a bender)    */
    int32_t* median_index;      /* [N]    index of that sample                                 */
    nrnerf_sample_outputs coarse;   /* keys without prefix  */
    nrnerf_sample_outputs fine;     /* keys with "fine_" prefix (I > 0) */
    /* scratch */
    void* workspace;            /* >= nrnerf_workspace_bytes(), 256-byte aligned */
    size_t workspace_bytes;
    /* render_rays' remaining deterministic flags (create_nerf passes False for both, train.py:707,715) */
    int32_t lindisp;            /* coarse depths linear in inverse depth            train.py:850-852 */
    int32_t white_bkgd;         /* rgb_map (and rgb0) += 1 - acc_map                train.py:786-787 */
    /* The stochastic branches of render_rays (perturb > 0, raw_noise_std > 0), made deterministic: the caller draws
     * the random numbers -- in the reference's order: t_rand (train.py:860), coarse noise (:753), u (rnh:665), fine
     * noise (:753) -- and passes them in; NULL = the deterministic branch.  All device pointers. */
    const float* u_coarse;      /* [N, S]   uniforms in [0,1): stratified jitter of the coarse depths  train.py:855-868 */
    const float* noise_coarse;  /* [N, S]   raw_noise_std * randn, added to sigma before the relu       train.py:753,761 */
    const float* u_fine;        /* [N, I]   uniforms for sample_pdf instead of linspace(0,1,I)          rnh:663-665 */
    const float* noise_fine;    /* [N, S']  as noise_coarse, fine pass */
    uint32_t flags;             /* bits of nrnerf_render_flags, since ABI 7; 0 = the defaults */
} nrnerf_render_args;

/* Kernel-selection switches of one nrnerf_render call (ABI 7; they were environment variables read inside the library up to
 * ABI 6).  Every combination renders the same function; the parity tests run the routes against each other. */
typedef enum nrnerf_render_flags {
    NRNERF_RENDER_FUSED_FINE_BENDER = 1u << 0,  /* no split-bender path: the fine pass bends all S + I samples in the network kernel */
    NRNERF_RENDER_UNFUSED_COMPOSITE = 1u << 1,  /* the final pass' compositing as a separate launch instead of the network kernel's epilogue */
    NRNERF_RENDER_SPLIT_COARSE = 1u << 2,       /* split path: stand-alone bender + trunk-only kernel for the coarse pass as well
                                                   (implied when the coarse pass runs on the 16x16x32 kernel) */
    NRNERF_RENDER_NO_X16 = 1u << 3,             /* every pass, and the stand-alone bender, on the 32x32x16 kernels */
    NRNERF_RENDER_X16_FINE_ONLY = 1u << 4,      /* 16x16x32 kernel for the fine pass only (the coarse pass on the fused-bender 32x32x16 kernel) */
    NRNERF_RENDER_BENDER_32X32 = 1u << 5,       /* split path, bf16 mode: the stand-alone bender on the fused kernels' own 32x32x16 tiles
                                                   (bent points then equal the fused-bender pass up to conversion ties) instead of 16x16x32 */
    /* ABI 8: the COARSE pass' compositing + sample_pdf + merge (train.py:889-920) as the epilogue of the coarse trunk kernel (16x16x32
     * kernels of the split path: a wave owns whole rays, their raw outputs stay in LDS) instead of composite_kernel's own launch: same
     * bits either way.  Neither bit = the library's default for the call (DESIGN.md section 3.3 says which and why, from an A/B on one box). */
    NRNERF_RENDER_COARSE_EPILOGUE_ON = 1u << 6,
    NRNERF_RENDER_COARSE_EPILOGUE_OFF = 1u << 7,
    /* Round 6: FIXED shares of the work, as up to round 5 -- every wave of the 16x16x32 stand-alone bender and every workgroup of the
     * 16x16x32 trunk kernels owns a fixed 1 / n of the sample blocks -- instead of taking the next piece from a device counter when it has
     * finished the previous one.  (Measured: the two workgroups of a CU do not run the bender at the same rate, and the eight XCDs do not
     * run the trunk at the same rate -- 3.7 % between the fastest and the slowest --, so with fixed shares a launch ends with idle CUs.)
     * Same bits either way: which wave evaluates a sample does not enter the arithmetic. */
    NRNERF_RENDER_FIXED_SHARES = 1u << 8
} nrnerf_render_flags;

/* per-kernel device time accumulated between nrnerf_profile_begin/_end (HIP events on the render stream) */
#define NRNERF_NUM_KERNELS 6
typedef struct nrnerf_profile {
    /* 0: coarse network, 1: coarse composite+sample_pdf+merge, 2: fine network, 3: fine composite,
     * 4: stand-alone bender over the importance samples, 5: stand-alone bender over the coarse samples (split-bender
     *    path: the network kernels then run without bender layers on ready-made points; see nrnerf_render) */
    double ms[NRNERF_NUM_KERNELS];
    int64_t launches[NRNERF_NUM_KERNELS];
    double flops[NRNERF_NUM_KERNELS];        /* algorithmic 2*MAC, unpadded */
    double mfma_flops[NRNERF_NUM_KERNELS];   /* issued MFMA flops incl. padding */
    char kernel_name[NRNERF_NUM_KERNELS][64]; /* ABI 8: which kernel the slot's LAST launch was (what nrnerf_render actually dispatched:
                                                 "net_kernel_x16 + fused compositing", "gx16_kernel", "gen_kernel", ...); "" = no launch */
} nrnerf_profile;

int nrnerf_abi_version(void);
const char* nrnerf_strerror(int status);

/* Architectures.  Replaces create_nerf's module construction for the render path (train.py:564-630).
 * COMPILED shapes (specialised kernels, activations resident in registers; training entry points available):
 * depth 8, skip after layer 4, 10 encoding frequencies, latent size 32, rigidity MLP 3 x 32, and
 *   width 256, ray bender 5 x 64 or 7 x 64 or none, optional view-dependent head (finite-difference or exact directions);
 *   width 256, time-conditioned baseline (no bender), optional view-dependent head;
 *   width 128 (--netwidth 128 --netwidth_fine 128, train.py:1004-1010), ray bender 5 x 64 or none, no view-dependent head;
 *   coarse and fine network of the same shape.
 * ANY OTHER shape the reference can build (--netdepth / --netwidth and _fine, --multires / --multires_views,
 * --ray_bending_latent_size: train.py:1004-1010, 1060, 1133-1139) gets a run-time-parameterised kernel (nrnerf_render only;
 * the training entry points answer NRNERF_ERR_UNSUPPORTED): depth <= 16, width <= 512 (any value; coarse and fine may differ),
 * <= 16 encoding frequencies, <= 10 direction frequencies, latent size <= 64, bender / rigidity MLPs of any depth (together
 * <= 28 layers) and width <= 512.  Beyond that, and for exact (Jacobian) view directions on a non-compiled shape:
 * NRNERF_ERR_UNSUPPORTED, and the Python boundary defers to the reference.  NRNERF_FORCE_GENERIC=1 in the environment selects
 * the run-time-parameterised kernel for the compiled shapes as well (tests). */
int nrnerf_model_create(const nrnerf_model_desc* desc, nrnerf_model** out);
/* Re-pack new weights of the SAME architecture / precision / device into an existing handle (e.g. after an optimiser
 * step or load_state_dict, train.py:666-682): no allocation; ordered after work already queued on hip_stream, complete
 * on return.  NRNERF_ERR_INVALID if the description is a different model (create a new handle then).  Must not run
 * concurrently with nrnerf_render on the same handle from another stream. */
int nrnerf_model_update(nrnerf_model* model, const nrnerf_model_desc* desc, void* hip_stream);
/* The same refresh from DEVICE memory, without a host round trip (one gather/convert kernel per packed image; what a
 * training loop calls after every optimiser step).  flat_params: device pointer to all parameters as one fp32 vector --
 * every nn.Linear as weight [out, in] row-major followed by its bias [out] (if it has one), in the order
 * bender.network[0..], bender.rigidity_network[0..], then network_fn: pts_linears[0..], output_linear (or alpha_linear,
 * feature_linear, views_linears[0], rgb_linear); then network_fine likewise.  After all of those, per network with the
 * view-dependent head (coarse, then fine), two DERIVED entries the caller computes: the views layer with feature_linear
 * folded in (no nonlinearity sits between them, run_nerf_helpers.py:286-301; the kernels evaluate the pair as ONE layer on
 * the trunk output) -- weight [W/2][W + direction encoding] = [ W_v[:, :W] W_f | W_v[:, W:] ] row-major, then bias [W/2] =
 * W_v[:, :W] b_f + b_v.  n_floats must equal nrnerf_model_flat_size().  Asynchronous on hip_stream; same concurrency
 * rule as nrnerf_model_update. */
int64_t nrnerf_model_flat_size(const nrnerf_model* model);
int nrnerf_model_update_device(nrnerf_model* model, const float* flat_params, int64_t n_floats, void* hip_stream);
/* ABI 8: the optimiser step of a training iteration and the re-pack above as ONE call = two launches (reference: torch.optim.Adam over
 * grad_vars, train.py:655-658, stepped at train.py:1606-1610).  Adam without weight decay / amsgrad, fp32, ONE launch over up to
 * NRNERF_ADAM_MAX_SEGMENTS contiguous runs (parameter, gradient, exp_avg, exp_avg_sq: device pointers, the same length each) laid end to
 * end; then, right behind it on the stream, every packed image of `model` is refreshed from `flat_params` exactly as
 * nrnerf_model_update_device does it (flat_params is normally the storage the segments' `param` pointers point into: training.FusedAdam
 * keeps the networks' parameters as views of one vector in the canonical order, so nothing is copied).  model == NULL or
 * flat_params == NULL: the Adam launch alone.  (Both phases in one kernel behind a grid barrier measured slower on this multi-die part:
 * csrc/nrnerf_optim.hip.)
 * step: device scalar (float) holding the number of steps taken so far; the launch increments it (a captured step replays correctly).
 * lr_device (optional): the learning rate as a device scalar -- a schedule inside a captured step; otherwise `lr`.
 * Asynchronous on hip_stream; same concurrency rule as nrnerf_model_update. */
#define NRNERF_ADAM_MAX_SEGMENTS 40
typedef struct nrnerf_adam_segment {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    uint64_t n;
} nrnerf_adam_segment;
typedef struct nrnerf_adam_args {
    uint32_t struct_size;
    int32_t n_segments;
    nrnerf_adam_segment segments[NRNERF_ADAM_MAX_SEGMENTS];
    float lr, beta1, beta2, eps;
    const float* lr_device;
    float* step;
    const float* flat_params;       /* [nrnerf_model_flat_size()] or NULL */
    int64_t n_floats;
    uint32_t* barrier;              /* model == NULL only: two zero-initialised words of device memory the launch uses to find its last workgroup (a handle owns its own) */
} nrnerf_adam_args;
int nrnerf_adam_step(nrnerf_model* model, const nrnerf_adam_args* args, void* hip_stream);

/* ABI 8: the weight and bias gradients of a NON-COMPILED trunk (any --netdepth / --netwidth, train.py:1004-1010; what autograd derives
 * from nn.Linear in NeRF.forward, rnh:253-258, 284-306) over the two arrays its training kernels save ROW-MAJOR
 * (nrnerf_generic_trunk_forward / _backward: acts and d_pre, [sample][feature], bf16 or fp32) -- a list of products
 *     out[out_offset + o * ldo + k] = sum_m a[m][o] b[m][k]   (o < wo, k < wi),     out[bias_offset + o] = sum_m a[m][o]  (bias_offset >= 0)
 * as one pass over panels of up to 256 x 256 (bf16: v_mfma_f32_16x16x32_bf16 fed by LDS transpose reads; fp32: v_mfma_f32_16x16x4_f32) and one
 * deterministic reduction of the partial sums.  a / b: device pointers, 16-byte aligned, lda / ldb in elements with 16-byte aligned rows that are
 * padded to whole 16-byte pieces (lda >= wo rounded up to 8 bf16 / 4 fp32 values; NRNERF_ERR_INVALID otherwise); `out` [out_floats] is written in full
 * (positions no job covers: zero).  workspace: nrnerf_tn_workspace_bytes() bytes of device memory.  Replaces round 5's chunked library
 * GEMMs (training.py::_chunked_tn_product). */
typedef struct nrnerf_tn_job {
    const void* a; const void* b;
    int32_t lda, ldb, wo, wi;
    int32_t ldo, reserved;
    int64_t out_offset, bias_offset;
} nrnerf_tn_job;
typedef struct nrnerf_tn_args {
    uint32_t struct_size;
    int32_t n_jobs, is_bf16, reserved;
    int64_t n_rows, out_floats;
    const nrnerf_tn_job* jobs;      /* host array [n_jobs] */
    float* out;
    void* workspace; size_t workspace_bytes;
} nrnerf_tn_args;
size_t nrnerf_tn_workspace_bytes(const nrnerf_tn_args* args);
int nrnerf_tn_products(const nrnerf_tn_args* args, void* hip_stream);

/* ABI 8: Embedder.embed (run_nerf_helpers.py:120-150) of 3-vectors as ROWS, and its transposed Jacobian -- the encodings a non-compiled
 * trunk's weight-gradient products read (nrnerf_tn_products: the layers fed by the encoding) and the gradient wrt the points / view
 * directions from the encodings' gradients (nrnerf_generic_trunk_backward's d_enc0 / d_enc1 / d_encv).  Columns in the reference's order:
 * [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)], 3 each.
 *   forward:  enc [n_rows][enc_cols] (fp32 or bf16) = [encoding (3 + 6 L) | codes[row / rows_per_code][0..n_lat) when codes != NULL | zeros]
 *   backward: d_src [n_rows][d_src_stride] = J^T (d_enc0 [+ d_enc1]) in columns 0..2, zeros behind (fp32 rows of stride d_enc_stride) */
typedef struct nrnerf_encoding_args {
    uint32_t struct_size;
    int32_t n_freqs;
    int64_t n_rows;
    const float* src; int32_t src_stride;
    void* enc; int32_t enc_cols; int32_t enc_is_bf16;
    const float* codes; int32_t n_lat; int32_t rows_per_code;
    const float* d_enc0; const float* d_enc1; int32_t d_enc_stride;
    float* d_src; int32_t d_src_stride;
} nrnerf_encoding_args;
int nrnerf_encoding_forward(const nrnerf_encoding_args* args, void* hip_stream);
int nrnerf_encoding_backward(const nrnerf_encoding_args* args, void* hip_stream);
void nrnerf_model_destroy(nrnerf_model* model);
/* the nrnerf_precision the handle was created with (decides the layout of nrnerf_trunk_args.acts / d_pre), or
 * NRNERF_ERR_INVALID for NULL */
int nrnerf_model_precision(const nrnerf_model* model);
/* 1 when the handle renders on the run-time-parameterised kernel (an architecture outside the compiled set, or
 * NRNERF_MODEL_FORCE_GENERIC), 0 on the specialised kernels, NRNERF_ERR_INVALID for NULL -- so that a test or a benchmark line can
 * say which kernels it measured instead of inferring it */
int nrnerf_model_is_generic(const nrnerf_model* model);

size_t nrnerf_workspace_bytes(const nrnerf_model* model, int32_t n_rays, int32_t n_samples,
                              int32_t n_importance);

/* hip_stream: a hipStream_t of the model's device (NULL = its null stream).  Asynchronous.  The calling thread's
 * current device may be any: the launches are issued on the model's device and the previous one is restored.
 *
 * Kernel sequence.  Default: coarse network -> composite + sample_pdf + merge -> fine network -> composite.  With a ray
 * bender (view directions, if any, by finite differences), n_importance > 0 and no per-sample detail outputs requested,
 * the fine pass is split (same results, bit for bit): the coarse network kernel also writes its bent points; the coarse samples keep them in
 * the fine pass (the bender is shared by both networks, run_nerf_helpers.py:213-215, and the coarse depths are a subset
 * of the merged depths, train.py:920); a stand-alone bender kernel handles the n_importance new samples; the fine
 * network kernel runs its trunk on those points (and takes a sample's view direction from the differences of
 * neighbouring points of that array).  Environment variables (read once): NRNERF_FUSED_FINE_BENDER=1 keeps
 * the fused fine pass, NRNERF_SPLIT_COARSE=1 also splits the coarse pass (bender kernel + trunk-only kernel; measured
 * no faster than the fused coarse kernel). */
int nrnerf_render(const nrnerf_model* model, const nrnerf_render_args* args, void* hip_stream);

/* Camera rays of one frame, generated on the device: reference get_rays (run_nerf_helpers.py:588-605) followed by
 * the packing of render() (train.py:380-399).  rays_out [H*W, ray_stride] (device), row j*W+i =
 * [origin3, direction3, near, far (, unit direction3 when ray_stride == 11)].  c2w: host pointer to the 3x4
 * camera-to-world matrix, row-major.  The kernel runs on the device that owns rays_out (hip_stream must belong to
 * it); the calling thread's current device may be any and is restored. */
typedef struct nrnerf_camera {
    float c2w[12];
    float focal_x, focal_y, center_x, center_y;
    int32_t height, width;
} nrnerf_camera;
int nrnerf_generate_rays(const nrnerf_camera* cam, float near_plane, float far_plane, float* rays_out,
                         int32_t ray_stride, void* hip_stream);

/* The coarse sample depths of render_rays (train.py:847-868) as an array: z_out [n_rays, n_samples] (device) =
 * near (1 - t) + far t, t = linspace(0, 1, n_samples) (or linear in inverse depth, `lindisp`), and -- uniforms != NULL --
 * the stratified jitter z = lower + (upper - lower) * u between the mid-points, u [n_rays, n_samples] drawn by the caller
 * (torch.rand, train.py:860).  The render path computes these inside its kernels; the training path (which hands depths to
 * nrnerf_bender_forward / nrnerf_composite_forward) gets them here in one launch instead of a dozen elementwise ones.
 * Runs on the device that owns z_out. */
int nrnerf_sample_depths(const float* rays, int32_t ray_stride, const float* uniforms, int32_t n_rays, int32_t n_samples,
                         int32_t lindisp, float* z_out, void* hip_stream);
/* The same plus the samples' points in the same launch: points_out [n_rays, n_samples, 3] = o + d z (train.py:871-873), the product and
 * the sum each rounded to fp32 as torch's two elementwise kernels round them (bit-identical to `rays_o[..., None, :] + rays_d[..., None, :] *
 * z_vals[..., :, None]`) -- what render_rays hands out as `initial_input_pts` under detailed_output. */
int nrnerf_sample_depths_points(const float* rays, int32_t ray_stride, const float* uniforms, int32_t n_rays, int32_t n_samples,
                                int32_t lindisp, float* z_out, float* points_out, void* hip_stream);

/* ---- training support ------------------------------------------------------------------------------------------
 * The reference trains through autograd (training_wrapper_class.forward, train.py:152-287; backward + optimiser step,
 * train.py:1594-1610).  These entry points are the pieces a torch.autograd.Function needs (nonrigid_nerf_amd/training.py
 * is the binding): the canonical network's trunk forward with saved activations and its backward-data pass, and the
 * compositing forward / backward, and the ray bender's forward / backward (nrnerf_bender_*, below).  The trunk takes
 * ready-made (bent) points and returns the gradient wrt them.
 * Weight gradients are products over two arrays these calls fill: dW_i = d_pre[i]^T x_i with x_0 = encoding,
 * x_i = acts[i-1] (x_{skip+1} = [encoding, acts[skip]]), db_i = column sums of d_pre[i]; d W_out = d_raw^T acts[D-1].
 * Both modes: nrnerf_trunk_wgrad, below (fp32 mode: row-major arrays [layer][sample][width], bf16 mode: block tiles).
 * Available (else NRNERF_ERR_UNSUPPORTED) for the trunks of width 256 and 128 (time-conditioned baseline: through
 * ray_bias; with the view-dependent head -- width 256 -- both branches of the head, on the directions the caller hands
 * in; not with exact view directions), fp32 or bf16 (a model created with NRNERF_PREC_F16 has no training kernels: unscaled f16 gradients
 * underflow; nonrigid_nerf_amd/training.py trains such a model through a bf16 handle). */
typedef struct nrnerf_trunk_args {
    uint32_t struct_size;       /* sizeof(nrnerf_trunk_args) */
    int32_t which;              /* 0 = network_fn (coarse), 1 = network_fine */
    int32_t n_rays, n_samples;  /* M = n_rays * n_samples points, sample-major per ray */
    const float* pts4;          /* [M,4] network input points: xyz + one pad float */
    void* acts;                 /* relu(W_i x_i + b_i) of every hidden layer, forward writes.  fp32 mode: float
                                   [depth][M][width] (backward reads it).  bf16 mode: bf16 [depth][B][width][32], B =
                                   n_rays * ceil(n_samples / 32) blocks of 32 consecutive samples of a ray, the samples
                                   contiguous (columns beyond the ray's end: finite padding) -- the operand layout of
                                   nrnerf_trunk_wgrad */
    void* relu_mask;            /* bf16 mode only: uint16 [depth][B][64 lanes][width/32], which values passed the relu; forward
                                   writes, backward reads (instead of acts) */
    /* forward */
    float* raw4;                /* out [M,4]  rgb + sigma logits (what nrnerf_composite_* consume) */
    float* raw;                 /* out [M,raw_ch] all output channels ("raw" of render_rays), or NULL */
    int32_t raw_ch;             /* 4 or 5 */
    /* backward */
    const float* d_raw4;        /* [M,4] gradient wrt raw4 */
    void* d_pre;                /* out: gradient wrt every layer's pre-activation, type and layout of acts (zero in the
                                   padded columns) */
    float* d_pts4;              /* out [M,4] gradient wrt the input points (xyz, 0) */
    const float* ray_bias;      /* forward: [n_rays, 2, width] fp32 or NULL: per-ray vectors added to the pre-activations of
                                   pts_linears[0] ([ray][0]) and pts_linears[skip + 1] ([ray][1]) for every sample of the ray.  The
                                   time-conditioned baseline (run_nerf_helpers.py:207-209, 273-282) concatenates the ray's latent
                                   code to those two layers' inputs; constant along the ray, that is W[:, latent columns] . latent,
                                   which the caller forms (and differentiates); the library's images of such a model hold the
                                   remaining columns.  The gradient wrt ray_bias[r][k] is the sum of d_pre[0 | skip + 1] over the
                                   ray's samples */
    /* view-dependent head (run_nerf_helpers.py:284-304; models whose networks have use_viewdirs): the colour branch runs
       behind the trunk in the same kernels -- raw4 = [rgb logits, density logit] -- and these are required */
    const float* dirs;          /* [M,3] view direction of every sample: the normalised finite differences of the bent points
                                   (rnh:316-356; nrnerf_direction_encoding with n_freqs = 0 yields them and takes their gradient
                                   back to the points), or the ray's own direction (train.py:73-76) */
    void* hv;                   /* relu(views_linears[0]([feature_linear(h), enc(dirs)])), width/2 values per sample, forward
                                   writes: float [M][width/2] (fp32 mode) or bf16 [B][width/2][32] */
    void* hv_mask;              /* bf16 mode only: uint16 [B][64 lanes][width/64], relu bits of hv; forward writes, backward reads */
    void* d_pre_v;              /* backward out: gradient wrt the views layer's pre-activation, type and layout of hv */
    float* d_dirs;              /* backward out [M,3]: gradient wrt dirs; NULL when nobody differentiates them (no bender) */
} nrnerf_trunk_args;
int nrnerf_trunk_forward(const nrnerf_model* model, const nrnerf_trunk_args* args, void* hip_stream);
int nrnerf_trunk_backward(const nrnerf_model* model, const nrnerf_trunk_args* args, void* hip_stream);

/* ray_bending.forward (run_nerf_helpers.py:507-577) under autograd, for a model with a bender: the sample points
 * p = origin + direction * z of n_rays x n_samples samples go through the offset and rigidity MLPs in exact fp32 (whatever
 * the model's precision).  Differentiable outputs: the bent point p + mask * offsets (* scaling), the unmasked offsets and
 * the rigidity mask; the masked offsets are mask * offsets (* scaling) of those.  Inputs that receive a gradient: the
 * latent codes (d_latents holds one row per SAMPLE; the caller sums the n_samples rows of a ray) and the layers' weights:
 *   offset MLP       dW_i = dz_i^T x_i,  x_0 = [p, latent], x_i = acts_offsets[i-1], dz_i = dz_offsets[i] for the hidden
 *                    layers and dz_out4[:, 0:3] for the last one (which has no bias); db_i = column sums of dz_i;
 *   rigidity MLP     likewise with x_0 = p, acts_rigidity, dz_rigidity and dz_out4[:, 3].
 * The sample positions carry no gradient (the reference detaches z_samples; rays are data).  First order only: the
 * divergence regulariser (second order in autograd's terms) has its own entry points, nrnerf_bender_divergence_*, below.
 * NRNERF_ERR_UNSUPPORTED unless the trunk entry points are available and the model has a bender. */
typedef struct nrnerf_bender_args {
    uint32_t struct_size;       /* sizeof(nrnerf_bender_args) */
    int32_t n_rays, n_samples;  /* M = n_rays * n_samples, sample-major per ray */
    const float* rays; int32_t ray_stride;          /* [N, ray_stride >= 6]: origin3, direction3 */
    const float* latents; int32_t latent_stride;    /* [N, latent_stride >= latent_size] */
    const float* z;             /* [N, n_samples] sample depths */
    int32_t has_rigidity_cutoff;   float rigidity_cutoff;     /* as nrnerf_render_args */
    int32_t has_test_time_scaling; float test_time_scaling;
    /* forward writes, backward reads */
    float* bent4;               /* [M,4] bent point + rigidity mask (what nrnerf_trunk_args.pts4 takes; w ignored there) */
    float* off4;                /* [M,4] unmasked offsets + tanh of the rigidity logit */
    void* acts_offsets;         /* [bender depth - 1][M][bender hidden]; element type of this and the other SAVED arrays below */
    void* acts_rigidity;        /* [rigidity depth - 1][M][rigidity hidden]  (acts_*, dz_offsets, dz_rigidity): fp32 for a model
                                   created with NRNERF_PREC_F32, bf16 otherwise -- only nrnerf_bender_wgrad reads their values
                                   (rounded to bf16 for the matrix pipe in that mode anyway); the backward-data chain runs in
                                   fp32 registers and takes only the sign of an activation from them */
    /* backward */
    const float* g_bent4;       /* [M,4] gradient wrt the bent point (w ignored) */
    const float* g_unmasked_offsets;   /* [M,3] or NULL */
    const float* g_rigidity_mask;      /* [M] or NULL */
    void* dz_offsets;           /* out, shape and element type of acts_offsets: gradient wrt the hidden pre-activations */
    void* dz_rigidity;          /* out, shape and element type of acts_rigidity */
    float* dz_out4;             /* out [M,4] gradient wrt the offsets (xyz) and the rigidity logit (w) */
    float* d_latents;           /* out [M, latent_size] gradient wrt each sample's latent inputs */
    const float* g_bent4_b;     /* [M,4] or NULL: a SECOND gradient wrt the bent point, added to g_bent4 in the kernel -- a training graph
                                   reads the coarse samples' bent points twice (coarse trunk; rows of the merged samples), and adding the
                                   two gradients beforehand was three launches */
} nrnerf_bender_args;
int nrnerf_bender_forward(const nrnerf_model* model, const nrnerf_bender_args* args, void* hip_stream);
int nrnerf_bender_backward(const nrnerf_model* model, const nrnerf_bender_args* args, void* hip_stream);

/* The weight and bias gradients of the bender's two MLPs from the arrays nrnerf_bender_forward / _backward filled, in one
 * launch: job j = layer j in the order network[0 .. depth-1], rigidity_network[0 .. rigidity_depth-1]; every job yields
 * n_partials partial sums (one per wave; the caller adds them) of  dW [64][64] (rows = the layer's outputs, columns = its
 * inputs; only [out_features][in_features] is meaningful) followed by db [64].  The first layers' input rows (point =
 * origin + direction * z, latent code) are formed from rays / latents / z as in nrnerf_bender_args.  Contraction: exact
 * fp32 for a model created with NRNERF_PREC_F32 (fp32 saved arrays); otherwise the saved arrays are bf16 and are contracted
 * on the bf16 matrix pipe with fp32 accumulation, the fp32 rows (dz_out4, points, latent codes) rounded to bf16 in registers
 * (the gradients entering come out of a bf16 trunk there).  The 16-bit route addresses every array by 32-bit byte offsets:
 * n_rays * n_samples * 256 bytes must stay below 4 GiB (NRNERF_ERR_INVALID beyond, also from nrnerf_bender_divergence_backward,
 * which forms its weight gradients the same way). */
#define NRNERF_BENDER_WGRAD_SLOT (64 * 64 + 64)
typedef struct nrnerf_bender_wgrad_args {
    uint32_t struct_size;       /* sizeof(nrnerf_bender_wgrad_args) */
    int32_t n_rays, n_samples;
    const float* rays; int32_t ray_stride;                                       /* as nrnerf_bender_args */
    const float* latents; int32_t latent_stride;
    const float* z;
    const void* acts_offsets; const void* acts_rigidity;                         /* as nrnerf_bender_args: fp32 or bf16 */
    const void* dz_offsets; const void* dz_rigidity; const float* dz_out4;       /* as nrnerf_bender_args */
    int32_t n_partials;         /* a multiple of 4, <= 4096 */
    float* partials;            /* out [n_partials][depth + rigidity_depth][NRNERF_BENDER_WGRAD_SLOT] */
} nrnerf_bender_wgrad_args;
int nrnerf_bender_wgrad(const nrnerf_model* model, const nrnerf_bender_wgrad_args* args, void* hip_stream);

/* The divergence regulariser of the ray bender -- reference compute_divergence_loss / divergence_approx / divergence_exact
 * (run_nerf_helpers.py:22-116), called from training_wrapper_class.forward (train.py:244-287) -- for n_points independent
 * points: divergence[k] = e_k^T J_k e_k, J = d(masked offsets)/d(point) of ray_bending.forward with
 * special_loss_return (run_nerf_helpers.py:507-577), e = probe.  The reference forms J^T e by a vector-Jacobian product
 * with create_graph=True and back-propagates through that graph; here the value is one forward-mode tangent through the
 * offset and rigidity MLPs and the backward pass runs the value chain and the tangent chain together (exact fp32 whatever
 * the model's precision).  With the three unit vectors as probes and the three results added it is the exact divergence
 * (divergence_exact).  The editing knobs act as in nrnerf_bender_args: the cutoff assignment has no derivative, scaling scales.
 * Forward writes `divergence` and the saved arrays; backward reads them and g_divergence and writes the dz / dtz arrays,
 * d_latents (one row per point: the caller reduces rows that share a code) and, in the same call, the weight and bias
 * gradients as n_partials partial sums (one per wave; the caller adds them) of NRNERF_BENDER_WGRAD_SLOT floats per job
 * (dW [64][64], rows = the layer's outputs, then db [64]), jobs in the order
 *   0                 network[0], columns of the point       (dW[:, 0:3])
 *   1                 network[0], columns of the latent code (dW[:, 0:latent_size]  ->  weight[:, 3:]); its db is network[0]'s too
 *   2 .. depth        network[1 .. depth-1]
 *   depth+1 ..        rigidity_network[0 .. rigidity_depth-1]
 * The points themselves receive no gradient (the reference makes them a leaf that nobody reads).
 * NRNERF_ERR_UNSUPPORTED unless nrnerf_bender_forward is available for the model. */
typedef struct nrnerf_divergence_args {
    uint32_t struct_size;       /* sizeof(nrnerf_divergence_args) */
    int64_t n_points;           /* M */
    const float* points;        /* [M,3]  input_points (initial_input_pts of the coarse pass, train.py:248-250) */
    const float* latents; int32_t latent_stride;   /* one row per point, latent_stride >= latent_size floats apart; 0 = one code */
    const float* probe;         /* [M,3]  e (torch.randn_like(offsets), run_nerf_helpers.py:106) */
    int32_t has_rigidity_cutoff;   float rigidity_cutoff;     /* as nrnerf_render_args */
    int32_t has_test_time_scaling; float test_time_scaling;
    /* forward writes, backward reads */
    float* divergence;          /* [M] */
    float* off4;                /* [M,4] unmasked offsets + tanh of the rigidity logit */
    float* toff4;               /* [M,4] their tangents (w: tangent of the logit) */
    void* acts_offsets;  void* tacts_offsets;       /* [bender depth - 1][M][bender hidden]: activations / tangents; these saved */
    void* acts_rigidity; void* tacts_rigidity;      /* [rigidity depth - 1][M][rigidity hidden]    arrays and dz_* / dtz_* below: fp32
                                                       or bf16 as in nrnerf_bender_args */
    /* backward */
    const float* g_divergence;  /* [M] */
    void* dz_offsets;  void* dtz_offsets;           /* out, shapes of acts_offsets */
    void* dz_rigidity; void* dtz_rigidity;          /* out, shapes of acts_rigidity */
    float* dz_out4;  float* dtz_out4;               /* out [M,4] each */
    float* d_latents;           /* out [M, latent_size] */
    int32_t n_partials;         /* a multiple of 4, <= 4096 */
    float* partials;            /* out [n_partials][depth + rigidity_depth + 1][NRNERF_BENDER_WGRAD_SLOT] */
    /* the tangent itself, for callers that need J . e rather than e^T J e -- NeRF.exact_nonrigid_viewdirs
       (run_nerf_helpers.py:358-385: the bent point's Jacobian applied to the ray direction = probe + tangent) under autograd */
    float* tangent;             /* forward out [M,3] or NULL: d(masked offsets)/d(point) . probe  (divergence = probe . tangent) */
    const float* g_tangent;     /* backward in [M,3] or NULL: gradient wrt `tangent`; when given it is used INSTEAD of
                                   g_divergence (which may then be NULL) */
    /* backward, all optional (ABI 8, round 6): the gradients a RENDER pass hands to nrnerf_bender_backward for an evaluation of the bender
       at the SAME points and latent codes (the coarse samples of a training iteration, whose points the divergence term is taken at,
       train.py:248-262): wrt the bent points (two of them, added), the unmasked offsets, the rigidity mask -- nrnerf_bender_args' g_bent4,
       g_bent4_b, g_unmasked_offsets, g_rigidity_mask.  Both backward chains are linear in their cotangents, so with these the ONE call
       yields d_latents and every weight / bias gradient of both uses (no nrnerf_bender_backward / _wgrad for those samples). */
    const float* render_g_bent4;
    const float* render_g_bent4_b;
    const float* render_g_unmasked_offsets;
    const float* render_g_rigidity_mask;
    /* forward, optional: the rows nrnerf_bender_forward writes as bent4 ([M,4]: bent point xyz + rigidity mask after the cutoff knob) --
       with it this call IS the render pass' bender evaluation of those samples as well (off4 has the same meaning in both calls), and
       with render_g_* above neither nrnerf_bender_forward nor _backward / _wgrad runs for them */
    float* bent4;
} nrnerf_divergence_args;
int nrnerf_bender_divergence_forward(const nrnerf_model* model, const nrnerf_divergence_args* args, void* hip_stream);
int nrnerf_bender_divergence_backward(const nrnerf_model* model, const nrnerf_divergence_args* args, void* hip_stream);

/* The split fine bender of the training path (the fine pass bends only its n_importance new samples; the bender is shared
 * by both networks, run_nerf_helpers.py:213-215, and the coarse depths are a subset of the merged depths, train.py:920):
 * per-sample rows of the n_samples coarse samples [N][n_samples][4] and of the new samples [N][n_importance][4] <-> the same
 * rows in merged-depth order [N][n_samples + n_importance][4].  New sample i of a ray sits at row rank_new[ray][i]
 * (nrnerf_composite_args.rank_new), the coarse samples fill the other rows in order.  inverse == 0 writes merged_* from
 * coarse_* and new_*; inverse != 0 writes coarse_* and new_* from merged_* (a permutation: this is also the gradient of the
 * forward direction).  Two arrays per call (bent point + rigidity mask rows, offset rows); the *_b pointers may all be NULL.
 * n_samples + n_importance <= 256.  Runs on the device that owns merged_a. */
int nrnerf_merge_rows(const uint8_t* rank_new, int32_t n_rays, int32_t n_samples, int32_t n_importance, float* coarse_a, float* coarse_b,
                      float* new_a, float* new_b, float* merged_a, float* merged_b, int32_t inverse, void* hip_stream);

/* The partial sums the weight-gradient entry points below write, added up INTO THE PARAMETERS' OWN LAYOUTS in one launch:
 *   out[j] = sum over the first P(j) records p of partials[p * record_stride + (index[j] & (NRNERF_REDUCE_SHORT - 1))]
 * with P(j) = n_short where index[j] carries the NRNERF_REDUCE_SHORT flag (the 64-column products of nrnerf_trunk_wgrad fill
 * only NRNERF_WGRAD_SHORT_PARTIALS records -- no zero-filling of the others needed on this route) and n_partials otherwise;
 * out[j] = 0 where index[j] < 0 (padding the caller wants zeroed, e.g. a parameter row no kernel produces).  `index`
 * (device, int32 [n_out]) is the caller's map from its flat gradient buffer -- every weight and bias of a network back to
 * back, each in its own shape -- to positions in one record.  Fixed order of additions (deterministic): eight interleaved groups
 * of records (p = g, g + 8, ...), each added in order, then the eight group sums in order.  Runs on the device that owns `out`. */
#define NRNERF_REDUCE_SHORT 0x40000000
int nrnerf_reduce_partials(const float* partials, int64_t record_stride, int32_t n_partials, int32_t n_short, const int32_t* index,
                           int64_t n_out, float* out, void* hip_stream);
/* The same, plus the column sums of a second array in the same launch: out[aux_pos[c]] = sum over r < n_aux of aux[r][c] for c < 4
 * (aux [n_aux][4] fp32, device; aux_pos: four HOST integers, a negative one skips its channel), records added in a fixed order.  Those
 * positions must carry index -2 ("written by somebody else": the regular reduction leaves them alone).  For nrnerf_wgrad_args.head_sums. */
int nrnerf_reduce_partials_aux(const float* partials, int64_t record_stride, int32_t n_partials, int32_t n_short, const int32_t* index,
                               int64_t n_out, float* out, const float* aux, int32_t n_aux, const int64_t* aux_pos, void* hip_stream);

/* Sums over the sample axis of the bf16 block tiles nrnerf_trunk_forward / _backward write ([block][feature][32 samples]):
 * out[r] = the 32 values of row r added in order in fp32, r < n_rows = blocks * features of the layer(s) handed in.  The
 * time-conditioned baseline's per-ray bias gradient (nrnerf_trunk_args.ray_bias) is the sum of d_pre over a ray's samples:
 * these row sums, then the ray's blocks (the library's reduction over both axes took 1.4 ms per layer and pass at 16 384 rays,
 * this 0.3).  Runs on the device that owns `out`. */
int nrnerf_tile_row_sums(const void* tiles, int64_t n_rows, float* out, void* hip_stream);

/* One layer of the bf16 block tiles ([block][feature][32 samples], e.g. the last hidden activation acts[depth - 1]) as
 * rows [n_rays * n_samples][width] bf16 (only the samples a ray has): for a caller that wants a saved activation in the
 * reference's layout (an LDS transpose per tile; a strided library copy took 0.9 ms per pass at 16 384 rays).  Not used by the
 * library's own training path any more (round 3 handed the last hidden activation to library GEMMs this way).
 * Runs on the device that owns `rows`. */
int nrnerf_tiles_to_rows(const void* tiles, int32_t n_rays, int32_t n_samples, int32_t width, void* rows, void* hip_stream);

/* The view-dependent head's direction input under autograd, for models with a ray bender (run_nerf_helpers.py:288-290,
 * viewdirs_via_finite_differences :316-356, Embedder :120-150): per sample j of a ray the direction
 * d_j = (p_j - p_{j-1}) / (|p_j - p_{j-1}| + 1e-6) of the BENT points (d_0 = d_1) and its encoding
 * [d, sin(2^k d), cos(2^k d)], k < n_freqs -- one row of 3 + 6 n_freqs values, fp32 or bf16 (`enc_is_bf16`).
 * g_bent4 == NULL: forward, writes `enc` [n_rays * n_samples][3 + 6 n_freqs] from bent4 [n_rays][n_samples][4].
 * g_bent4 != NULL: backward, reads the gradient wrt those rows from `enc` and writes the gradient wrt the bent points
 * ([.,4] rows, w = 0): through the encoding, the normalisation and both differences a point takes part in.
 * n_samples >= 2.  Runs on the device that owns `enc`. */
int nrnerf_direction_encoding(const float* bent4, int32_t n_rays, int32_t n_samples, int32_t n_freqs, void* enc, int32_t enc_is_bf16,
                              float* g_bent4, void* hip_stream);

/* The weight and bias gradients of the trunk from the two arrays nrnerf_trunk_forward / _backward filled, in one launch
 * (the contraction runs over samples; no transposes).  bf16 mode: over their [block][feature][32 samples] tiles on the bf16
 * matrix pipe; fp32 mode: over their rows [sample][feature] on v_mfma_f32_32x32x2_f32 (exact fp32 products and sums).
 *   dw_hidden[i-1] = d_pre[i]^T acts[i-1]  (i = 1 .. depth-1; the skip layer's columns for its encoding input are in dw_enc)
 *   dw_enc[0] = d_pre[0]^T enc,  dw_enc[1] = d_pre[skip+1]^T enc     (64 columns: the 63 encoding columns + one of padding)
 *   dw_head^T = acts[depth-1]^T g                                    (64 columns: the output channels, zero padded)
 *   db[i] = row sums of d_pre[i]
 * each as n_partials partial sums (one per workgroup) the caller adds up: partials[c] is one record of
 * NRNERF_WGRAD_STRIDE(depth, width) floats = dw_hidden [depth-1][width][width], dw_enc [2][width][64], dw_head^T [width][64],
 * db [depth+1][width] (row `depth` is scratch), so one sum over the first axis yields them all.  The 64-column products
 * (dw_enc, dw_head^T, db[0]) are cut into fewer, longer partial sums (they cost less per block: all workgroups of the
 * launch then finish together) and only fill the first NRNERF_WGRAD_SHORT_PARTIALS(n_partials, width) records: the caller
 * zero-fills at least dw_enc, dw_head^T, db[0] and db[depth] of the records beyond that (or simply all of `partials`), or
 * adds them up with nrnerf_reduce_partials, which never reads those.
 * enc / g_head: the encoding of the input points and the gradient wrt the head's outputs in the layout of the mode -- bf16
 * [B][64][32] tiles, or fp32 rows [M][64]: scratch the caller allocates, filled by this call from pts4 and d_raw4 (the
 * arrays given to nrnerf_trunk_backward). */
typedef struct nrnerf_wgrad_args {
    uint32_t struct_size;       /* sizeof(nrnerf_wgrad_args) */
    int32_t n_rays, n_samples;
    const void* acts; const void* d_pre;       /* as nrnerf_trunk_args, in the model's mode */
    const float* pts4; const float* d_raw4;    /* [M,4] each, as nrnerf_trunk_args */
    void* enc;                  /* scratch, bf16 [B][64][32] (fp32 mode: float [M][64]) */
    void* g_head;               /* scratch, bf16 [B][64][32] (fp32 mode: float [M][64]) */
    int32_t n_partials;         /* 1 .. 4096 records; the launch has about 8.9 * n_partials workgroups (width 256): 28 fills an
                                   MI355X with one workgroup per CU (view-dependent head: about 11.1 * n_partials, 23) */
    float* partials;            /* out [n_partials][NRNERF_WGRAD_STRIDE(depth, width)]; view-dependent head:
                                   [n_partials][NRNERF_WGRAD_STRIDE_VIEWS(depth, width)] */
    /* view-dependent head: the arrays of nrnerf_trunk_args, and one more scratch operand.  The record then continues with
         dw_fold [width/2][width]  = d_pre_v^T acts[depth-1]   gradient wrt the FOLDED views layer's hidden columns
                                                               (views_linears[0].weight[:, :width] . feature_linear.weight)
         dw_dirs [width/2][64]     = d_pre_v^T enc(dirs)       wrt views_linears[0].weight[:, width:] (27 columns + padding)
         dw_rgb^T [width/2][64]    = hv^T d_raw4               wrt rgb_linear.weight, transposed (columns 0..2)
         db_views [width/2]        = row sums of d_pre_v       wrt the folded bias (views . feature bias + views bias)
       (dw_fold, db_views: n_partials records; dw_dirs, dw_rgb^T: NRNERF_WGRAD_SHORT_PARTIALS), and dw_head^T's column 3 is
       alpha_linear's gradient.  The chain rule through the fold is the caller's (two small products in parameter space). */
    const float* dirs;          /* [M,3] */
    const void* hv; const void* d_pre_v;
    void* encv;                 /* scratch, layout of enc */
    float* head_sums;           /* out [n_rays * ceil(n_samples / 32)][4] fp32 or NULL (bf16 / f16 handles only; ignored by fp32 ones): the
                                   sums of d_raw4's four channels over each block of 32 samples -- the head's bias gradient once added up,
                                   which nrnerf_reduce_partials_aux does in the launch that adds up `partials` */
} nrnerf_wgrad_args;
#define NRNERF_WGRAD_STRIDE_VIEWS(depth, width) (NRNERF_WGRAD_STRIDE(depth, width) + ((width) / 2) * (width) + 2 * ((width) / 2) * 64 + (width) / 2)
#define NRNERF_WGRAD_SHORT_PARTIALS(n_partials, width) \
    ((((n_partials) * (2 * ((width) / 64) + 2) + (2 * ((width) / 64) + (width) / 32) / 2) / (2 * ((width) / 64) + (width) / 32)) < 1 ? 1 : \
     (((n_partials) * (2 * ((width) / 64) + 2) + (2 * ((width) / 64) + (width) / 32) / 2) / (2 * ((width) / 64) + (width) / 32)))
#define NRNERF_WGRAD_STRIDE(depth, width) (((depth) - 1) * (width) * (width) + 3 * (width) * 64 + ((depth) + 1) * (width))
int nrnerf_trunk_wgrad(const nrnerf_model* model, const nrnerf_wgrad_args* args, void* hip_stream);

/* Training of an architecture OUTSIDE the compiled set (round 5; fp32 or bf16 handles, width % 4 == 0): the
 * canonical network on ready-made points with every hidden activation saved, and its backward-data pass -- the run-time-parameterised
 * kernel's layer programs (forward: NeRF.forward, rnh:240-314; backward: the same layers in reverse with transposed weights).  The weight
 * gradients are products of the two saved arrays, dW_i = d_pre_i^T x_i (x_0 = the encoding, x_{skip+1} = [encoding, activation], else
 * the previous activation; db_i = column sums of d_pre_i) and the gradient wrt the points follows from the encoding's: both are left to
 * the caller (nonrigid_nerf_amd/training.py forms them with library GEMMs -- the one place on the training path where it does).
 *   With the view-dependent head (rnh:284-304; width <= 480) the saved arrays have two more slots: [depth] = feature_linear's outputs
 *   (forward) / their gradient (backward), [depth + 1] = the colour branch's activations / pre-activation gradients in the first
 *   views_linears[0].out_features columns (the rest of those rows is scratch); d sigma and d rgb reach h through the kernel.
 *   acts / d_pre: [depth (+ 2)][n_rays * n_samples][width] in the handle's element type (fp32 for NRNERF_PREC_F32, bf16 for NRNERF_PREC_BF16);
 *   d_enc0 / d_enc1: [n_rays * n_samples][3 + 6 multires] fp32, the gradient of the encoding through pts_linears[0] and through the layer
 *   behind the skip connection (d_enc1 may be NULL for a network without one).  nrnerf_model_trains_generic: 1 when the handle has these. */
typedef struct nrnerf_generic_trunk_args {
    uint32_t struct_size;       /* sizeof(nrnerf_generic_trunk_args) */
    int32_t which;              /* 0 = network_fn (coarse), 1 = network_fine */
    int32_t n_rays, n_samples;
    const float* pts4;          /* forward in: [N,S,4] points (xyz + one ignored float) */
    void* acts;                 /* forward out / backward in */
    float* raw4;                /* forward out: [N,S,4] rgb, sigma */
    float* raw; int32_t raw_ch; /* forward out (may be NULL): [N,S,raw_ch] the reference's "raw" */
    const float* d_raw4;        /* backward in: [N,S,4] */
    void* d_pre;                /* backward out */
    float* d_enc0;              /* backward out */
    float* d_enc1;              /* backward out (skip connection) */
    const float* dirs;          /* forward in, view-dependent head: [N,S,3] one direction per sample */
    float* d_encv;              /* backward out, view-dependent head: [N*S][3 + 6 multires_views] gradient of the direction encoding */
    const float* latents;       /* forward in, time-conditioned baseline (no bender): [N][latent size] one code per ray; d_enc0 / d_enc1 then
                                 * have 3 + 6 multires + latent size columns, the code's gradient (per sample) in the last ones */
    void* relu_bits;            /* ABI 8, optional: nrnerf_generic_trunk_bits_bytes() bytes of device memory.  Forward writes which activations
                                 * passed the relu (one byte per lane and tile pair of the 16x16x32 kernels); a backward call that is handed
                                 * them runs on those kernels' dataflow (csrc/nrnerf_gx16_bwd.h) instead of the run-time-parameterised one
                                 * (bf16, plain head: 0 bytes = not available for this handle, pass NULL) */
} nrnerf_generic_trunk_args;
size_t nrnerf_generic_trunk_bits_bytes(const nrnerf_model* model, int32_t which, int32_t n_rays, int32_t n_samples);
int nrnerf_generic_trunk_forward(const nrnerf_model* model, const nrnerf_generic_trunk_args* args, void* hip_stream);
int nrnerf_generic_trunk_backward(const nrnerf_model* model, const nrnerf_generic_trunk_args* args, void* hip_stream);
int nrnerf_model_trains_generic(const nrnerf_model* model);
/* 1 when the handle has the ray bender's training kernels (nrnerf_bender_*, nrnerf_divergence_*): a compiled architecture with a bender, or a
 * generic handle whose BENDER has a compiled shape (5 or 7 x 64 offsets, 3 x 32 rigidity, latent 32) in fp32 / bf16 */
int nrnerf_model_trains_bender(const nrnerf_model* model);

/* The loss of one training iteration over the outputs of render_rays (reference training_wrapper_class.forward, train.py:207-287), per ray:
 *   loss[r] = mean((rgb_map - target)^2) + [rgb0] mean((rgb0 - target)^2)                                    train.py:207-218, rnh:10-13
 *           + offsets_weight * ( mean_s( w |off|^(2 - rig) ) + rigidity_weight * mean_s( w rig ) )          train.py:221-242
 *           + divergence_weight * mean_s( (1 - exp(-relu(alpha))) |div|^2 )                                 train.py:245-287, rnh:61-69
 * w (visibility weights) and the opacity factor are treated as constants, as the reference detaches them (train.py:223, rnh:65-66); the
 * increasing schedule (train.py:240, 285) is folded into offsets_weight / divergence_weight by the caller.  One launch forward, one
 * backward (g_* = gradient of sum_r g_loss[r] loss[r]); all device pointers; runs on the device that owns `loss` / `g_loss`.
 * Eager torch ops take ~80 launches of 2-5 us for the same on a 1024-ray step. */
typedef struct nrnerf_loss_args {
    uint32_t struct_size;       /* sizeof(nrnerf_loss_args) */
    int32_t n_rays, n_samples;  /* N; S of the per-sample tensors (the coarse pass' detail outputs) */
    const float* rgb_map;       /* [N,3] */
    const float* rgb0;          /* [N,3] or NULL */
    const float* target;        /* [N,3] */
    const float* weights;       /* [N,S] visibility_weights, or NULL: no offsets / rigidity term */
    const float* offsets;       /* [N,S,3] unmasked_offsets */
    const float* rigidity;      /* [N,S] rigidity_mask */
    const float* alpha;         /* [N,S] opacity_alpha (the divergence term's weights before 1 - exp(-relu(.))) */
    const float* divergence;    /* [N,S] per-sample divergence, or NULL: no divergence term */
    float offsets_weight, rigidity_weight, divergence_weight;
    const float* schedule;      /* device scalar multiplied into offsets_weight and divergence_weight (the increasing schedule of
                                   train.py:240, 285 as a value a replayed HIP graph can change between steps), or NULL = 1 */
    float* loss;                /* forward: out [N] */
    const float* g_loss;        /* backward: in [N] */
    float* g_rgb_map;           /* backward outs; g_rgb0 / g_offsets + g_rigidity / g_divergence NULL exactly where the input is */
    float* g_rgb0;
    float* g_offsets;
    float* g_rigidity;
    float* g_divergence;
    /* (ABI 8, second half of round 6) the per-sample inputs where they lie: what render_rays hands out as unmasked_offsets / rigidity_mask
       are the xyz / w parts of [N,S,4] rows */
    int32_t offsets_stride;     /* floats from one sample's offsets to the next; 0 = 3 (packed).  g_offsets is always packed */
    int32_t rigidity_stride;    /* ... rigidity value to the next; 0 = 1 */
    /* the training loop differentiates loss.mean() (train.py:1594): its gradient as the scalar it is, instead of a [N] tensor of g / N */
    const float* g_mean;        /* backward: in, device scalar or NULL: gradient wrt mean(loss); g_loss may then be NULL (both given: both count) */
} nrnerf_loss_args;
int nrnerf_loss_forward(const nrnerf_loss_args* args, void* hip_stream);
int nrnerf_loss_backward(const nrnerf_loss_args* args, void* hip_stream);

/* Gradient of the per-ray latent codes' selection  latents = codes[index]  (training_wrapper_class.forward, train.py:173-188: the stacked
 * per-frame codes indexed by each ray's time step):  out[k][c] = sum over the rays r with index[r] == k of g[r][c], added in ray order
 * (deterministic; autograd's indexing backward sorts, a one-hot GEMM is three launches).  index [n_rays] int64, g [n_rays][latent_size],
 * out [n_codes][latent_size]; latent_size <= 256.  Runs on the device that owns `out`. */
int nrnerf_code_gradients(const int64_t* index, const float* g, int32_t n_rays, int32_t latent_size, int32_t n_codes, float* out, void* hip_stream);

/* raw2outputs (train.py:724-789) of one pass, optionally followed by sample_pdf + merge (run_nerf_helpers.py:651-698,
 * train.py:910-920), and its backward.  Runs on the device that owns raw4. */
typedef struct nrnerf_composite_args {
    uint32_t struct_size;       /* sizeof(nrnerf_composite_args) */
    int32_t n_rays, n_samples, n_importance;   /* n_importance > 0 (forward only): also draw and merge the new depths */
    const float* rays; int32_t ray_stride;     /* as nrnerf_render_args */
    const float* raw4;          /* [N,S,4] */
    const float* z;             /* [N,S] depths of this pass, or NULL: linspace(near, far) (lindisp honoured) */
    int32_t lindisp, white_bkgd;
    const float* noise;         /* [N,S] added to sigma before the relu, or NULL */
    const float* u;             /* [N,I] uniforms for sample_pdf, or NULL: linspace(0,1,I) */
    /* forward outputs */
    float* rgb; float* disp; float* acc;       /* [N,3], [N], [N] */
    float* weights; float* alpha;              /* [N,S] visibility weights / opacities, or NULL */
    float* z_std; float* z_merged;             /* [N], [N,S+I] (n_importance > 0) */
    /* backward: gradients of the forward outputs (NULL = zero), gradient wrt raw4 */
    const float* g_rgb; const float* g_disp; const float* g_acc; const float* g_weights;
    float* d_raw4;              /* out [N,S,4] */
    /* forward, n_importance > 0, both or neither: the importance samples on their own -- depths in sample order and the row
     * each takes among the merged depths (the coarse depths keep their order in the remaining rows).  The bender is shared by
     * both networks and the coarse depths are a subset of the merged ones, so a caller bends only these I samples for the
     * fine pass and re-uses the coarse pass' bent points (as nrnerf_render's split-bender path does). */
    float* z_new;               /* out [N,I] or NULL */
    uint8_t* rank_new;          /* out [N,I] or NULL */
} nrnerf_composite_args;
int nrnerf_composite_forward(const nrnerf_composite_args* args, void* hip_stream);
int nrnerf_composite_backward(const nrnerf_composite_args* args, void* hip_stream);

/* profiling: records hipEvents around each kernel of subsequent nrnerf_render calls on this model */
int nrnerf_profile_begin(nrnerf_model* model);
int nrnerf_profile_end(nrnerf_model* model, nrnerf_profile* out);   /* synchronises the recorded events */

/* Host-only packing (no device needed): writes the MFMA-fragment weight stream + unit table + bias
 * table of one pass exactly as nrnerf_model_create uploads them.  which: 0 = coarse, 1 = fine, 2 = fine without the
 * bender layers, 3 = bender + rigidity layers alone (2, 3: the split-bender path; need a bender and not the exact view directions),
 * 4 / 5 = transposed trunk weights of the coarse / fine network for the backward-data kernel (training),
 * 6 = transposed bender + rigidity weights for nrnerf_bender_backward (always fp32),
 * 7 / 8 / 9 = the layer PROGRAM of the run-time-parameterised kernel for the coarse / fine network / the ray bender (the
 *   bender always fp32): fragment (tile t, k-slab s) of a layer at w_frag + t * (ns0 + ns1) + s; the unit table then holds
 *   n_layers, per layer {w_frag, bias_tile, nt, src0, ns0, src1, ns1, dst, relu, o_col, o_rows} (buffers: 0 = network input,
 *   1 = hidden, 2 = second input, 3 = head outputs), then the padded widths of the three buffers and the latent size
 *   (n_units = that many entries minus one),
 * 10 = the fine network's trunk (+ view-dependent head) for the 16x16x32 kernel (fragments of 16 rows x 32 k; bias table [tile][16]),
 * 11 / 12 = the coarse / fine trunk of ANY plain-headed architecture for the width-class 16x16x32 kernel: the layers' fragment blocks
 *   back to back (first layer, pts_linears[1 ..], output_linear), each padded to a whole number of 4-unit ring periods, + a copy of the
 *   first two units behind the last; width padded to a multiple of 64 with zeros.
 * 13 / 14 = the backward-data PROGRAM of the run-time-parameterised kernel for the coarse / fine network (training of a non-compiled
 *   architecture, nrnerf_generic_trunk_backward): transposed layers in reverse order; the unit table holds n_layers, per layer the 11
 *   integers of 7 / 8 plus {save slot, mask slot, column offsets of the two sources in the hidden buffer} (dst 4 / 5 / 6: straight to
 *   d_enc0 / d_enc1 / d_encv), then the three padded widths, the latent size and the hidden-buffer column where the rows of d raw start.
 * Any output pointer may be NULL to query sizes only.  Used by the CPU-side packing tests. */
typedef struct nrnerf_packed_info {
    uint64_t stream_bytes;     /* fragment stream */
    uint32_t n_units;          /* unit table has n_units + 1 uint32 entries (offsets in 16-byte words) */
    uint32_t n_bias_tiles;     /* bias table: n_bias_tiles * 32 floats */
    uint32_t frag_bytes;       /* 1024 (bf16/f16) or 256 (f32) */
    uint32_t slot_bytes;       /* LDS ring slot size the kernel uses */
    uint32_t mfma_per_block;   /* MFMA instructions per 32-sample block */
} nrnerf_packed_info;
int nrnerf_pack_host(const nrnerf_model_desc* desc, int which, nrnerf_packed_info* info,
                     void* stream_out, size_t stream_cap, uint32_t* unit_table_out,
                     float* bias_table_out);

#ifdef __cplusplus
}
#endif
#endif /* NRNERF_H */
