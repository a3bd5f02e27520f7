"""ctypes binding of ``lib/libnrnerf_hip.so`` (C ABI: ``include/nrnerf.h``).

The structures below mirror the header field for field.  There is no fallback: if the
shared library is missing or does not load, importing the render path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NRNERF_LIB selects an alternative build of the same library (tuning experiments, see csrc/Makefile)
LIB_PATH = os.environ.get("NRNERF_LIB") or os.path.join(_HERE, "lib", "libnrnerf_hip.so")

ABI_VERSION = 8
MAX_SAMPLES = 1024        # NRNERF_MAX_SAMPLES (include/nrnerf.h): per ray and pass of nrnerf_render; training: 256
OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_WORKSPACE, ERR_NOMEM, ERR_INTERNAL = 0, -1, -2, -3, -4, -5, -6
PRECISIONS = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "f16": 2, "fp16": 2, "float16": 2}
_CANONICAL = {0: "f32", 1: "bf16", 2: "f16"}


def canonical_precision(p: str) -> str:
    """"fp32" / "float32" -> "f32", "bfloat16" -> "bf16", "fp16" / "float16" -> "f16": the three names the rest of the
    package compares against (the aliases are accepted at every entry point and normalised there, once)."""
    try:
        return _CANONICAL[PRECISIONS[p]]
    except KeyError:
        raise ValueError(f"unknown precision {p!r} (one of {sorted(PRECISIONS)})") from None
NUM_KERNELS = 6
KERNEL_NAMES = ("net_coarse", "composite_sample_coarse", "net_fine", "composite_fine", "bend_fine", "bend_coarse")

_fp = C.POINTER(C.c_float)


class Linear(C.Structure):
    _fields_ = [("weight", _fp), ("bias", _fp), ("out_features", C.c_int32), ("in_features", C.c_int32)]


class MlpDesc(C.Structure):
    _fields_ = [("depth", C.c_int32), ("width", C.c_int32), ("skip", C.c_int32), ("output_ch", C.c_int32),
                ("use_viewdirs", C.c_int32), ("time_conditioned", C.c_int32),
                ("pts_linears", C.POINTER(Linear)),
                ("output_linear", Linear), ("alpha_linear", Linear), ("feature_linear", Linear),
                ("views_linear", Linear), ("rgb_linear", Linear)]


class BenderDesc(C.Structure):
    _fields_ = [("latent_size", C.c_int32), ("depth", C.c_int32), ("hidden", C.c_int32),
                ("rigidity_depth", C.c_int32), ("rigidity_hidden", C.c_int32),
                ("network", C.POINTER(Linear)), ("rigidity_network", C.POINTER(Linear))]


class ModelDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("precision", C.c_int32), ("multires", C.c_int32),
                ("multires_views", C.c_int32), ("device", C.c_int32),
                ("bender", C.POINTER(BenderDesc)), ("coarse", C.POINTER(MlpDesc)), ("fine", C.POINTER(MlpDesc)),
                ("exact_viewdirs", C.c_int32), ("flags", C.c_uint32)]


class SampleOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("visibility_weights", "opacity_alpha", "initial_input_pts",
                                           "unmasked_offsets", "masked_offsets", "input_pts", "rigidity_mask")]


class RenderArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_rays", C.c_int32), ("n_samples", C.c_int32),
                ("n_importance", C.c_int32),
                ("rays", C.c_void_p), ("ray_stride", C.c_int32),
                ("latents", C.c_void_p), ("latent_stride", C.c_int32),
                ("has_rigidity_cutoff", C.c_int32), ("rigidity_cutoff", C.c_float),
                ("has_test_time_scaling", C.c_int32), ("test_time_scaling", C.c_float),
                ("has_removal_threshold", C.c_int32), ("removal_threshold", C.c_float),
                ("detailed_output", C.c_int32),
                ("rgb_map", C.c_void_p), ("disp_map", C.c_void_p), ("acc_map", C.c_void_p), ("raw", C.c_void_p),
                ("rgb0", C.c_void_p), ("disp0", C.c_void_p), ("acc0", C.c_void_p), ("z_std", C.c_void_p),
                ("z_vals", C.c_void_p),
                ("surface_pts", C.c_void_p), ("surface_rigidity", C.c_void_p), ("median_index", C.c_void_p),
                ("coarse", SampleOutputs), ("fine", SampleOutputs),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("lindisp", C.c_int32), ("white_bkgd", C.c_int32),
                ("u_coarse", C.c_void_p), ("noise_coarse", C.c_void_p), ("u_fine", C.c_void_p), ("noise_fine", C.c_void_p),
                ("flags", C.c_uint32)]


# include/nrnerf.h: nrnerf_model_flags / nrnerf_render_flags (ABI 7: kernel-selection switches are fields of the call; the library
# reads no environment variable).  The test switches NRNERF_* of the environment are mapped to them HERE, on the Python side.
MODEL_FORCE_GENERIC, MODEL_NO_X16_F16 = 1 << 0, 1 << 1
# Python-side only (never reaches the library): a handle for the TRAINING entry points, whose callers compute the view directions themselves --
# described to the library without exact_viewdirs, which the run-time-parameterised RENDER kernel does not do (training.render_rays_train)
MODEL_PY_TRAINING_HANDLE = 1 << 30
RENDER_FUSED_FINE_BENDER, RENDER_UNFUSED_COMPOSITE, RENDER_SPLIT_COARSE, RENDER_NO_X16, RENDER_X16_FINE_ONLY, RENDER_BENDER_32X32, \
    RENDER_COARSE_EPILOGUE_ON, RENDER_COARSE_EPILOGUE_OFF, RENDER_FIXED_SHARES = (1 << i for i in range(9))


def model_flags_from_env() -> int:
    """NRNERF_FORCE_GENERIC=1 / NRNERF_X16_F16=0 -> nrnerf_model_desc.flags (read when a handle is created)."""
    f = 0
    if os.environ.get("NRNERF_FORCE_GENERIC") == "1":
        f |= MODEL_FORCE_GENERIC
    if os.environ.get("NRNERF_X16_F16", "1") == "0":
        f |= MODEL_NO_X16_F16
    return f


_RENDER_ENV = ("NRNERF_FUSED_FINE_BENDER", "NRNERF_UNFUSED_COMPOSITE", "NRNERF_SPLIT_COARSE", "NRNERF_X16_BENDER", "NRNERF_X16",
               "NRNERF_FUSED_COARSE_EPILOGUE", "NRNERF_FIXED_SHARES")
_render_flags_cache = (None, 0)


def render_flags_from_env() -> int:
    """NRNERF_FUSED_FINE_BENDER=1, NRNERF_UNFUSED_COMPOSITE=1, NRNERF_SPLIT_COARSE=1, NRNERF_X16=0|1|2, NRNERF_X16_BENDER=0,
    NRNERF_FUSED_COARSE_EPILOGUE=0|1 -> nrnerf_render_args.flags (read per call: the parity tests render one scene through several kernel
    routes in one process; parsed once per distinct set of values -- a small-batch render should not pay for six string comparisons)."""
    global _render_flags_cache
    key = tuple(os.environ.get(k) for k in _RENDER_ENV)
    if _render_flags_cache[0] == key:
        return _render_flags_cache[1]
    f = 0
    if key[0] == "1":
        f |= RENDER_FUSED_FINE_BENDER
    if key[1] == "1":
        f |= RENDER_UNFUSED_COMPOSITE
    if key[2] == "1":
        f |= RENDER_SPLIT_COARSE
    if key[3] == "0":
        f |= RENDER_BENDER_32X32
    x16 = key[4]
    if x16 is not None and x16.strip() != "":
        if x16.strip() not in ("0", "1", "2"):
            raise ValueError(f"NRNERF_X16={x16!r}: one of 0 (32x32x16 kernels), 1 (16x16x32 fine pass only), 2 (both passes, the default)")
        f |= {0: RENDER_NO_X16, 1: RENDER_X16_FINE_ONLY}.get(int(x16), 0)
    if key[5] is not None and key[5].strip() != "":
        if key[5].strip() not in ("0", "1"):
            raise ValueError(f"NRNERF_FUSED_COARSE_EPILOGUE={key[5]!r}: 0 (composite / sample_pdf / merge as their own launch) or 1 (inside the coarse trunk kernel)")
        f |= RENDER_COARSE_EPILOGUE_ON if key[5].strip() == "1" else RENDER_COARSE_EPILOGUE_OFF
    if key[6] == "1":
        f |= RENDER_FIXED_SHARES
    _render_flags_cache = (key, f)
    return f


class GenericTrunkArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("which", C.c_int32), ("n_rays", C.c_int32), ("n_samples", C.c_int32),
                ("pts4", C.c_void_p), ("acts", C.c_void_p), ("raw4", C.c_void_p), ("raw", C.c_void_p), ("raw_ch", C.c_int32),
                ("d_raw4", C.c_void_p), ("d_pre", C.c_void_p), ("d_enc0", C.c_void_p), ("d_enc1", C.c_void_p),
                ("dirs", C.c_void_p), ("d_encv", C.c_void_p), ("latents", C.c_void_p), ("relu_bits", C.c_void_p)]


class LossArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_rays", C.c_int32), ("n_samples", C.c_int32),
                ("rgb_map", C.c_void_p), ("rgb0", C.c_void_p), ("target", C.c_void_p),
                ("weights", C.c_void_p), ("offsets", C.c_void_p), ("rigidity", C.c_void_p), ("alpha", C.c_void_p), ("divergence", C.c_void_p),
                ("offsets_weight", C.c_float), ("rigidity_weight", C.c_float), ("divergence_weight", C.c_float), ("schedule", C.c_void_p),
                ("loss", C.c_void_p), ("g_loss", C.c_void_p),
                ("g_rgb_map", C.c_void_p), ("g_rgb0", C.c_void_p), ("g_offsets", C.c_void_p), ("g_rigidity", C.c_void_p), ("g_divergence", C.c_void_p),
                ("offsets_stride", C.c_int32), ("rigidity_stride", C.c_int32), ("g_mean", C.c_void_p)]


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * NUM_KERNELS), ("launches", C.c_int64 * NUM_KERNELS),
                ("flops", C.c_double * NUM_KERNELS), ("mfma_flops", C.c_double * NUM_KERNELS),
                ("kernel_name", (C.c_char * 64) * NUM_KERNELS)]


class PackedInfo(C.Structure):
    _fields_ = [("stream_bytes", C.c_uint64), ("n_units", C.c_uint32), ("n_bias_tiles", C.c_uint32),
                ("frag_bytes", C.c_uint32), ("slot_bytes", C.c_uint32), ("mfma_per_block", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("c2w", C.c_float * 12), ("focal_x", C.c_float), ("focal_y", C.c_float), ("center_x", C.c_float),
                ("center_y", C.c_float), ("height", C.c_int32), ("width", C.c_int32)]


class TrunkArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("which", C.c_int32), ("n_rays", C.c_int32), ("n_samples", C.c_int32),
                ("pts4", C.c_void_p), ("acts", C.c_void_p), ("relu_mask", C.c_void_p),
                ("raw4", C.c_void_p), ("raw", C.c_void_p), ("raw_ch", C.c_int32),
                ("d_raw4", C.c_void_p), ("d_pre", C.c_void_p), ("d_pts4", C.c_void_p), ("ray_bias", C.c_void_p),
                ("dirs", C.c_void_p), ("hv", C.c_void_p), ("hv_mask", C.c_void_p), ("d_pre_v", C.c_void_p), ("d_dirs", C.c_void_p)]


class WgradArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_rays", C.c_int32), ("n_samples", C.c_int32),
                ("acts", C.c_void_p), ("d_pre", C.c_void_p), ("pts4", C.c_void_p), ("d_raw4", C.c_void_p),
                ("enc", C.c_void_p), ("g_head", C.c_void_p),
                ("n_partials", C.c_int32), ("partials", C.c_void_p),
                ("dirs", C.c_void_p), ("hv", C.c_void_p), ("d_pre_v", C.c_void_p), ("encv", C.c_void_p), ("head_sums", C.c_void_p)]


REDUCE_SHORT = 0x40000000        # NRNERF_REDUCE_SHORT of include/nrnerf.h


def wgrad_stride(depth: int, width: int) -> int:
    """NRNERF_WGRAD_STRIDE of include/nrnerf.h"""
    return (depth - 1) * width * width + 3 * width * 64 + (depth + 1) * width


def wgrad_stride_views(depth: int, width: int) -> int:
    """NRNERF_WGRAD_STRIDE_VIEWS of include/nrnerf.h"""
    return wgrad_stride(depth, width) + (width // 2) * width + 2 * (width // 2) * 64 + width // 2


def wgrad_short_partials(n_partials: int, width: int) -> int:
    """NRNERF_WGRAD_SHORT_PARTIALS of include/nrnerf.h"""
    trw = width // 64
    return min(n_partials, max(1, (n_partials * (2 * trw + 2) + (2 * trw + width // 32) // 2) // (2 * trw + width // 32)))


class BenderArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_rays", C.c_int32), ("n_samples", C.c_int32),
                ("rays", C.c_void_p), ("ray_stride", C.c_int32),
                ("latents", C.c_void_p), ("latent_stride", C.c_int32),
                ("z", C.c_void_p),
                ("has_rigidity_cutoff", C.c_int32), ("rigidity_cutoff", C.c_float),
                ("has_test_time_scaling", C.c_int32), ("test_time_scaling", C.c_float),
                ("bent4", C.c_void_p), ("off4", C.c_void_p), ("acts_offsets", C.c_void_p), ("acts_rigidity", C.c_void_p),
                ("g_bent4", C.c_void_p), ("g_unmasked_offsets", C.c_void_p), ("g_rigidity_mask", C.c_void_p),
                ("dz_offsets", C.c_void_p), ("dz_rigidity", C.c_void_p), ("dz_out4", C.c_void_p), ("d_latents", C.c_void_p),
                ("g_bent4_b", C.c_void_p)]


class BenderWgradArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_rays", C.c_int32), ("n_samples", C.c_int32),
                ("rays", C.c_void_p), ("ray_stride", C.c_int32), ("latents", C.c_void_p), ("latent_stride", C.c_int32), ("z", C.c_void_p),
                ("acts_offsets", C.c_void_p), ("acts_rigidity", C.c_void_p),
                ("dz_offsets", C.c_void_p), ("dz_rigidity", C.c_void_p), ("dz_out4", C.c_void_p),
                ("n_partials", C.c_int32), ("partials", C.c_void_p)]


BENDER_WGRAD_SLOT = 64 * 64 + 64


class DivergenceArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_points", C.c_int64),
                ("points", C.c_void_p), ("latents", C.c_void_p), ("latent_stride", C.c_int32), ("probe", C.c_void_p),
                ("has_rigidity_cutoff", C.c_int32), ("rigidity_cutoff", C.c_float),
                ("has_test_time_scaling", C.c_int32), ("test_time_scaling", C.c_float),
                ("divergence", C.c_void_p), ("off4", C.c_void_p), ("toff4", C.c_void_p),
                ("acts_offsets", C.c_void_p), ("tacts_offsets", C.c_void_p), ("acts_rigidity", C.c_void_p), ("tacts_rigidity", C.c_void_p),
                ("g_divergence", C.c_void_p),
                ("dz_offsets", C.c_void_p), ("dtz_offsets", C.c_void_p), ("dz_rigidity", C.c_void_p), ("dtz_rigidity", C.c_void_p),
                ("dz_out4", C.c_void_p), ("dtz_out4", C.c_void_p), ("d_latents", C.c_void_p),
                ("n_partials", C.c_int32), ("partials", C.c_void_p),
                ("tangent", C.c_void_p), ("g_tangent", C.c_void_p),
                ("render_g_bent4", C.c_void_p), ("render_g_bent4_b", C.c_void_p), ("render_g_unmasked_offsets", C.c_void_p),
                ("render_g_rigidity_mask", C.c_void_p), ("bent4", C.c_void_p)]


ADAM_MAX_SEGMENTS = 40       # NRNERF_ADAM_MAX_SEGMENTS of include/nrnerf.h


class AdamSegment(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_uint64)]


class AdamArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_segments", C.c_int32), ("segments", AdamSegment * ADAM_MAX_SEGMENTS),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("lr_device", C.c_void_p), ("step", C.c_void_p), ("flat_params", C.c_void_p), ("n_floats", C.c_int64), ("barrier", C.c_void_p)]


class TnJob(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("lda", C.c_int32), ("ldb", C.c_int32), ("wo", C.c_int32), ("wi", C.c_int32),
                ("ldo", C.c_int32), ("reserved", C.c_int32), ("out_offset", C.c_int64), ("bias_offset", C.c_int64)]


class TnArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_jobs", C.c_int32), ("is_bf16", C.c_int32), ("reserved", C.c_int32),
                ("n_rows", C.c_int64), ("out_floats", C.c_int64), ("jobs", C.POINTER(TnJob)), ("out", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class EncodingArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_freqs", C.c_int32), ("n_rows", C.c_int64),
                ("src", C.c_void_p), ("src_stride", C.c_int32),
                ("enc", C.c_void_p), ("enc_cols", C.c_int32), ("enc_is_bf16", C.c_int32),
                ("codes", C.c_void_p), ("n_lat", C.c_int32), ("rows_per_code", C.c_int32),
                ("d_enc0", C.c_void_p), ("d_enc1", C.c_void_p), ("d_enc_stride", C.c_int32),
                ("d_src", C.c_void_p), ("d_src_stride", C.c_int32)]


class CompositeArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_rays", C.c_int32), ("n_samples", C.c_int32), ("n_importance", C.c_int32),
                ("rays", C.c_void_p), ("ray_stride", C.c_int32),
                ("raw4", C.c_void_p), ("z", C.c_void_p), ("lindisp", C.c_int32), ("white_bkgd", C.c_int32),
                ("noise", C.c_void_p), ("u", C.c_void_p),
                ("rgb", C.c_void_p), ("disp", C.c_void_p), ("acc", C.c_void_p), ("weights", C.c_void_p), ("alpha", C.c_void_p),
                ("z_std", C.c_void_p), ("z_merged", C.c_void_p),
                ("g_rgb", C.c_void_p), ("g_disp", C.c_void_p), ("g_acc", C.c_void_p), ("g_weights", C.c_void_p),
                ("d_raw4", C.c_void_p), ("z_new", C.c_void_p), ("rank_new", C.c_void_p)]


EXPORTS = {
    "nrnerf_abi_version": (C.c_int, []),
    "nrnerf_strerror": (C.c_char_p, [C.c_int]),
    "nrnerf_model_create": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]),
    "nrnerf_model_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrnerf_model_flat_size": (C.c_int64, [C.c_void_p]),
    "nrnerf_model_update_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "nrnerf_adam_step": (C.c_int, [C.c_void_p, C.POINTER(AdamArgs), C.c_void_p]),
    "nrnerf_tn_workspace_bytes": (C.c_size_t, [C.POINTER(TnArgs)]),
    "nrnerf_tn_products": (C.c_int, [C.POINTER(TnArgs), C.c_void_p]),
    "nrnerf_encoding_forward": (C.c_int, [C.POINTER(EncodingArgs), C.c_void_p]),
    "nrnerf_encoding_backward": (C.c_int, [C.POINTER(EncodingArgs), C.c_void_p]),
    "nrnerf_model_destroy": (None, [C.c_void_p]),
    "nrnerf_model_precision": (C.c_int, [C.c_void_p]),
    "nrnerf_model_is_generic": (C.c_int, [C.c_void_p]),
    "nrnerf_generic_trunk_forward": (C.c_int, [C.c_void_p, C.POINTER(GenericTrunkArgs), C.c_void_p]),
    "nrnerf_generic_trunk_backward": (C.c_int, [C.c_void_p, C.POINTER(GenericTrunkArgs), C.c_void_p]),
    "nrnerf_model_trains_generic": (C.c_int, [C.c_void_p]),
    "nrnerf_generic_trunk_bits_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "nrnerf_model_trains_bender": (C.c_int, [C.c_void_p]),
    "nrnerf_loss_forward": (C.c_int, [C.POINTER(LossArgs), C.c_void_p]),
    "nrnerf_loss_backward": (C.c_int, [C.POINTER(LossArgs), C.c_void_p]),
    "nrnerf_code_gradients": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "nrnerf_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "nrnerf_render": (C.c_int, [C.c_void_p, C.POINTER(RenderArgs), C.c_void_p]),
    "nrnerf_generate_rays": (C.c_int, [C.POINTER(Camera), C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_void_p]),
    "nrnerf_sample_depths": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "nrnerf_sample_depths_points": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrnerf_merge_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int32, C.c_void_p]),
    "nrnerf_reduce_partials": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "nrnerf_reduce_partials_aux": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                             C.POINTER(C.c_int64), C.c_void_p]),
    "nrnerf_tile_row_sums": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "nrnerf_tiles_to_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "nrnerf_direction_encoding": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "nrnerf_trunk_forward": (C.c_int, [C.c_void_p, C.POINTER(TrunkArgs), C.c_void_p]),
    "nrnerf_trunk_backward": (C.c_int, [C.c_void_p, C.POINTER(TrunkArgs), C.c_void_p]),
    "nrnerf_trunk_wgrad": (C.c_int, [C.c_void_p, C.POINTER(WgradArgs), C.c_void_p]),
    "nrnerf_bender_forward": (C.c_int, [C.c_void_p, C.POINTER(BenderArgs), C.c_void_p]),
    "nrnerf_bender_backward": (C.c_int, [C.c_void_p, C.POINTER(BenderArgs), C.c_void_p]),
    "nrnerf_bender_wgrad": (C.c_int, [C.c_void_p, C.POINTER(BenderWgradArgs), C.c_void_p]),
    "nrnerf_bender_divergence_forward": (C.c_int, [C.c_void_p, C.POINTER(DivergenceArgs), C.c_void_p]),
    "nrnerf_bender_divergence_backward": (C.c_int, [C.c_void_p, C.POINTER(DivergenceArgs), C.c_void_p]),
    "nrnerf_composite_forward": (C.c_int, [C.POINTER(CompositeArgs), C.c_void_p]),
    "nrnerf_composite_backward": (C.c_int, [C.POINTER(CompositeArgs), C.c_void_p]),
    "nrnerf_profile_begin": (C.c_int, [C.c_void_p]),
    "nrnerf_profile_end": (C.c_int, [C.c_void_p, C.POINTER(Profile)]),
    "nrnerf_pack_host": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.POINTER(PackedInfo), C.c_void_p, C.c_size_t,
                                   C.POINTER(C.c_uint32), _fp]),
}

_lib = None


class NrnerfError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        super().__init__(f"{where}: {strerror(status)} (status {status})")


def load() -> C.CDLL:
    """Load the shared library (once).  Raises if it is missing: there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C nonrigid_nerf_amd/csrc -j8`). nonrigid_nerf_amd has no CPU fallback.")
        # PyTorch-ROCm first: the wheel ships its own HIP runtime (torch/lib/libamdhip64.so), this library is linked against the
        # system's.  Loaded after torch, it binds to the runtime already in the process; loaded BEFORE torch, the process ends up
        # with two runtimes and torch's device memory is foreign to this library's (seen on the MI355X as NRNERF_ERR_HIP from
        # nrnerf_model_create when __graft_entry__.build() and smoke() ran in one process).
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(lib, name)          # AttributeError if the .so does not export what nrnerf.h declares
            fn.restype, fn.argtypes = res, args
        v = lib.nrnerf_abi_version()
        if v != ABI_VERSION:
            raise ImportError(f"libnrnerf_hip.so ABI version {v}, binding expects {ABI_VERSION}")
        _lib = lib
    return _lib


def strerror(status: int) -> str:
    return load().nrnerf_strerror(status).decode()


def check(status: int, where: str):
    if status != OK:
        raise NrnerfError(status, where)
