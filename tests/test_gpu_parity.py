"""GPU parity tests proper: the HIP path (through the Python boundary -> C ABI -> kernels) against
(a) the committed reference outputs in tests/golden, (b) the CPU oracle on the same seeded inputs,
(c) size-independent properties at BASELINE.json's full sizes.

fp32 MFMA mode is held to the fp32 tolerances of SURVEY.md section 8c (rgb/acc atol 1e-4, disp rtol 1e-3,
equal_nan); bf16/f16 are judged by PSNR against the fp32 reference render (>= 40 dB on all rays) as BASELINE.md states
-- on the fitted checkpoint, tests/test_fitted_checkpoint.py -- and characterised here on the synthetic stress scene
(network-output SNR, flip-aware PSNR; see test_16bit_modes_psnr_full_pipeline).

The one discrete decision of the algorithm whose outcome legitimately depends on fp32 rounding -- sample_pdf's
`denom < 1e-5` branch (run_nerf_helpers.py:694; see tests/test_oracle_golden.py) -- is handled by checking the
fine pass against the oracle evaluated AT THE DEPTHS THE GPU CHOSE (tight), and the depths themselves
statistically (almost all equal, the rest inside one coarse bin).
"""
import contextlib
import os

import numpy as np
import pytest
import torch

from nonrigid_nerf_amd import render as R
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene
from oracle import nrnerf_oracle as O
from tests.helpers import check_pinned, compare_dict, load_golden, out_of_tolerance_fraction, psnr, split_knobs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@contextlib.contextmanager
def _setenv(name, value):
    """A library switch that nrnerf_render reads per call (NRNERF_X16, NRNERF_UNFUSED_COMPOSITE, ...) for the duration of a block."""
    old = os.environ.get(name)
    os.environ[name] = value
    try:
        yield
    finally:
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old


def _same_bender(precision):
    """Round 6: "f16" mode's default stand-alone bender is the single-product 16x16x32 one as well (a third of the three-product bender's
    MFMAs; its own accuracy statement: test_f16_mode_with_the_single_product_bender_is_at_least_as_accurate_as_bf16_mode and the fitted
    checkpoints).  The "equally good rounding" statements below compare two TRUNK kernels on the same bent points: for "f16" they keep
    the 32x32x16 three-product bender on both sides (NRNERF_X16_BENDER=0 -> NRNERF_RENDER_BENDER_32X32)."""
    return _setenv("NRNERF_X16_BENDER", "0") if precision == "f16" else contextlib.nullcontext()


def _assert_x16_is_an_equally_good_rounding(x16, k32, ref32):
    """The 16-bit modes' split-bender path runs its trunk-only fine pass on the 16x16x32 MFMA kernel (nrnerf_net_x16.h) by default
    and on the 32x32x16 kernel (nrnerf_net_mb.h) with NRNERF_X16=0: the same 16-bit products of the same 16-bit-rounded operands,
    summed in fp32 in another order (k-slices of 32 instead of 16, another feature permutation) -- so hidden activations round
    to the neighbouring 16-bit value now and then and the two renders are two roundings of one fp32 network, not one bit
    pattern.  What must hold: everything upstream of the fine network is bit-identical, and against the fp32-MFMA render of
    the same call the 16x16x32 kernel's error is the 32x32x16 kernel's (mean within 30 %, maximum within 3x)."""
    for k in ("rgb0", "disp0", "acc0", "z_std", "_z_vals"):
        assert torch.equal(torch.nan_to_num(x16[k]), torch.nan_to_num(k32[k])), k
    for k in ("rgb_map", "acc_map"):
        e16, e32 = (x16[k].float() - ref32[k].float()).abs(), (k32[k].float() - ref32[k].float()).abs()
        assert e16.mean().item() <= 1.3 * e32.mean().item() + 1e-6, (k, e16.mean().item(), e32.mean().item())
        assert e16.max().item() <= 3.0 * e32.max().item() + 1e-4, (k, e16.max().item(), e32.max().item())


def _assert_x16_coarse_is_an_equally_good_rounding(x16, k32, ref32):
    """NRNERF_X16=2 (the default of round 5): the COARSE pass of the split path leaves the fused-bender 32x32x16 kernel too -- stand-alone
    bender over the coarse samples (the fused kernel's own building blocks) + the 16x16x32 trunk.  The coarse maps are then another
    rounding of the same fp32 network, and the importance samples follow them: against the fp32-MFMA render the coarse maps' error
    must be the 32x32x16 kernel's (mean within 30 %, maximum within 3x), no more depths may move than with that kernel (+ 30 %), and
    the final maps are held to the same bound."""
    for k in ("rgb0", "acc0", "rgb_map", "acc_map"):
        e16, e32 = (x16[k].float() - ref32[k].float()).abs(), (k32[k].float() - ref32[k].float()).abs()
        assert e16.mean().item() <= 1.3 * e32.mean().item() + 1e-6, (k, e16.mean().item(), e32.mean().item())
        assert e16.max().item() <= 3.0 * e32.max().item() + 1e-4, (k, e16.max().item(), e32.max().item())
    span = float(ref32["_z_vals"].max() - ref32["_z_vals"].min())
    m16 = ((x16["_z_vals"] - ref32["_z_vals"]).abs() > 1e-5 * span).float().mean().item()
    m32 = ((k32["_z_vals"] - ref32["_z_vals"]).abs() > 1e-5 * span).float().mean().item()
    assert m16 <= 1.3 * m32 + 2e-3, ("moved depths", m16, m32)


def hip_render(scene, rays, latents, precision, chunk=1 << 20, retraw=False, detailed=False, knobs=None,
               want_z=True, use_batchify=True, flags=None):
    cfg = scene.cfg
    rb, coarse, fine = build_modules(scene, device=DEV)
    knobs = knobs or {}
    if rb is not None:
        rb.rigidity_test_time_cutoff = knobs.get("rigidity_test_time_cutoff")
        rb.test_time_scaling = knobs.get("test_time_scaling")
    for m in (coarse, fine):
        if m is not None:
            m.test_time_nonrigid_object_removal_threshold = knobs.get("removal_threshold")
    R.set_precision(precision)
    kw = dict(network_fn=coarse, network_fine=fine, network_query_fn=None, N_samples=cfg.N_samples,
              N_importance=cfg.N_importance, perturb=0.0, raw_noise_std=0.0, white_bkgd=False, lindisp=False,
              retraw=retraw, ray_bender=rb, _want_z_vals=want_z)
    kw.update(flags or {})                              # render_rays flags: lindisp, white_bkgd
    api = {"ray_bending_latents": latents.to(DEV)}
    with torch.no_grad():
        if use_batchify:
            out = R.batchify_rays(rays.to(DEV), api, chunk=chunk, detailed_output=detailed, **kw)
        else:
            out = R.render_rays(rays.to(DEV), additional_pixel_information=api, detailed_output=detailed, **kw)
    torch.cuda.synchronize()
    return {k: v.cpu() for k, v in out.items()}


def oracle_fine_given_z(scene, rays, latents, z_vals, knobs=None, detailed=False, white_bkgd=False):
    """The oracle's fine pass evaluated at given merged depths (train.py:921-950)."""
    cfg = scene.cfg
    rays_o, rays_d = rays[:, 0:3], rays[:, 3:6]
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[:, :, None]
    net = scene.fine if scene.fine is not None else scene.coarse
    viewdirs = rays[:, -3:] if rays.shape[-1] > 8 else None
    out = O.query_network(pts, viewdirs, latents, net, scene.bender, cfg, knobs, detailed)
    raw, det = out if detailed else (out, {})
    rgb, disp, acc, alpha, w, _ = O.composite(raw, z_vals, rays_d, white_bkgd)
    res = dict(rgb_map=rgb, disp_map=disp, acc_map=acc, raw=raw)
    if detailed:
        res.update(fine_visibility_weights=w, fine_opacity_alpha=alpha, **{"fine_" + k: v for k, v in det.items()})
    return res


# Finite-difference view directions (run_nerf_helpers.py:339-351) divide by |p_j - p_{j-1}| + 1e-6.  Where an
# importance sample lands within ~1e-6 of its neighbour the difference is pure fp32 cancellation noise, in the
# reference as much as here, and the colour logits of that one sample are arbitrary (its weight is ~0, so no map
# changes).  Allow that for < 0.05 % of the raw entries; everything else stays at the fp32 tolerance.
FD_DIRS_RAW = dict(frac_ok=5e-4, outlier_atol=5.0)

COARSE_KEYS = ["rgb0", "disp0", "acc0", "visibility_weights", "opacity_alpha", "initial_input_pts",
               "unmasked_offsets", "masked_offsets", "input_pts", "rigidity_mask"]


@pytest.mark.parametrize("name", ["coarse_only_1k", "headline_64_128", "detailed_64_128", "ragged_chunks",
                                  "knobs_64_64", "no_bender_64_64", "viewdirs_64_64", "config4_deep_bender_viewdirs",
                                  "time_conditioned_64_64", "lindisp_white_bkgd_64_64", "exact_viewdirs_64_64",
                                  "exact_viewdirs_knobs", "config4_exact_viewdirs", "narrow_128_64_64", "narrow_128_no_bender",
                                  # architectures outside the compiled set: the run-time-parameterised kernel (csrc/nrnerf_generic.h)
                                  "generic_192_320_detailed", "generic_viewdirs_96_160", "generic_shallow_no_bender",
                                  "generic_time_conditioned_448", "generic_exact_viewdirs_192"])
def test_fp32_mode_matches_reference_golden(name):
    meta, cfg, scene, rays, latents, ref = load_golden(name)
    meta["knobs"], flags = split_knobs(meta["knobs"])
    got = hip_render(scene, rays, latents, "f32", chunk=meta["chunk"], retraw=bool(meta["retraw"]),
                     detailed=bool(meta["detailed"]), knobs=meta["knobs"], flags=flags)
    assert set(k for k in got if not k.startswith("_")) == set(ref.keys()), \
        (sorted(set(got) ^ set(ref.keys())))
    for k in ref:
        assert got[k].shape == ref[k].shape and got[k].dtype == torch.float32, k
    fails = []
    if cfg.N_importance == 0:
        fails += compare_dict(got, ref)
    else:
        # 1. coarse pass: no discrete decisions -> fp32 tolerance on everything the reference returns for it
        fails += compare_dict(got, ref, keys=[k for k in COARSE_KEYS if k in ref])
        # 2. merged depths: identical up to rounding for almost every sample
        zo = O.batchify_rays(rays, latents, scene, chunk=meta["chunk"], knobs=O.Knobs(**meta["knobs"]),
                             detailed_output=bool(meta["detailed"]), **flags)["_z_vals"]   # the removal knob only acts when detailed
        zg = got["_z_vals"]
        assert (zg[:, 1:] >= zg[:, :-1]).all(), "merged depths are not sorted"
        moved = ((zg - zo).abs() > 2e-5).float().mean().item()
        assert moved < 0.01, f"{moved:.4f} of merged depths differ from the oracle"
        widest_bin = (cfg.far - cfg.near) if flags.get("lindisp") else 1.1 / (cfg.N_samples - 1)   # inverse-depth bins are uneven
        assert float((zg - zo).abs().max()) < widest_bin, "a depth moved by more than one coarse bin"
        # inverse-depth spacing makes the far bins up to 0.9 wide, so the same few branch decisions of sample_pdf move
        # a sample -- and the ray's colour -- much further (the reference's own fp32 vs an fp64 evaluation of this
        # case: 2 of 48 rays differ, by up to 0.098): wider end-to-end allowance there, steps 1-3 stay tight
        loose = dict(frac_ok=0.25, outlier_atol=0.25) if flags.get("lindisp") else dict(frac_ok=0.10, outlier_atol=2e-2)
        fails += compare_dict(got, ref, keys=["z_std"], frac_ok=loose["frac_ok"], outlier_atol=1e-2)
        # 3. fine pass at the depths the GPU chose: tight
        fine = oracle_fine_given_z(scene, rays, latents, zg, O.Knobs(**meta["knobs"]), bool(meta["detailed"]),
                                   white_bkgd=bool(flags.get("white_bkgd", False)))
        if cfg.use_viewdirs and cfg.ray_bending and cfg.approx_nonrigid_viewdirs:
            fails += compare_dict(got, fine, keys=[k for k in fine if k in got and k != "raw"])
            fails += compare_dict(got, fine, keys=["raw"], **FD_DIRS_RAW)
        else:
            fails += compare_dict(got, fine, keys=[k for k in fine if k in got])
        # 4. end to end against the reference outputs, allowing the few rays whose sample moved ...
        fails += compare_dict(got, ref, keys=["rgb_map", "acc_map"], **loose)
        # ... and the measured fractions pinned at <= 2 x the committed values (not at the allowances above)
        n = rays.shape[0]
        check_pinned("golden/" + name, dict(moved_depths=moved, rgb_map=out_of_tolerance_fraction(got, ref, "rgb_map"),
                                            acc_map=out_of_tolerance_fraction(got, ref, "acc_map")),
                     dict(moved_depths=zg.numel(), rgb_map=3 * n, acc_map=n))
    assert not fails, "\n".join(fails)


def _last_sample_flips(raw_got, raw_ref):
    """Rays whose background decision flipped: the last sample has distance 1e10 (train.py:745), so its alpha is
    exactly 0 or 1 depending on the SIGN of sigma there; a sign change under reduced precision rewrites the whole
    ray (acc jumps by the remaining transmittance).  Measure-zero for exact arithmetic, a few per thousand rays when
    sigma carries a 16-bit rounding error -- and it dominates a PSNR over few thousand rays."""
    return (raw_got[:, -1, 3] > 0) != (raw_ref[:, -1, 3] > 0)


# thresholds on the synthetic STRESS scene (regression guards at the measured level, not the acceptance bar -- that is
# enforced on the fitted checkpoint, tests/test_fitted_checkpoint.py): (network-output SNR dB, PSNR dB over rays whose
# background decision did not flip, max flipped fraction).  "bf16" is the fastest mode (bf16 trunk, single-product f16
# bender); "f16" the accurate 16-bit mode (f16 trunk, fp32-equivalent split-product bender).  The stress scene's
# deformation field is violent (last bender layer ~ N(0, 0.15^2): offsets up to 0.3 of the scene) and its density random,
# which is what makes the single-product bender visible here (-4 dB) and invisible on a fitted model.
PRECISION_BARS = {"bf16": (32.0, 40.0, 0.02), "f16": (50.0, 52.0, 0.004)}


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_16bit_modes_network_precision_coarse_only(precision):
    """Coarse-only render (BASELINE config 1 shape): ours and the fp32 oracle evaluate the networks at identical
    points, so `raw` is directly comparable.  BASELINE.md's bar for reduced precision is PSNR(ours, reference
    render) >= 40 dB; it is met on every ray whose last-sample sign did not flip (see _last_sample_flips)."""
    snr_bar, psnr_bar, flip_bar = PRECISION_BARS[precision]
    cfg = SceneConfig(N_importance=0)
    scene = make_scene(cfg, 0)
    rays, latents = make_rays(4096, 7, cfg)
    ref = O.batchify_rays(rays, latents, scene, chunk=1024, retraw=True)
    got = hip_render(scene, rays, latents, precision, retraw=True)
    err = got["raw"] - ref["raw"]
    snr = [float(20 * torch.log10(ref["raw"][..., c].std() / err[..., c].pow(2).mean().sqrt())) for c in range(4)]
    flips = _last_sample_flips(got["raw"], ref["raw"])
    keep = ~flips
    p_all = psnr(got["rgb_map"], ref["rgb_map"])
    p_keep = psnr(got["rgb_map"][keep], ref["rgb_map"][keep])
    p_acc = psnr(got["acc_map"][keep], ref["acc_map"][keep])
    print(f"[{precision}] coarse-only: raw SNR r,g,b,sigma = {[round(x, 1) for x in snr]} dB; flipped rays "
          f"{int(flips.sum())}/{flips.numel()}; PSNR rgb all rays {p_all:.1f} dB, non-flipped {p_keep:.1f} dB, acc {p_acc:.1f} dB")
    assert min(snr) >= snr_bar, snr
    assert flips.float().mean().item() <= flip_bar
    assert p_keep >= psnr_bar and p_acc >= psnr_bar - 6.0, (p_keep, p_acc)


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_16bit_modes_psnr_full_pipeline(precision):
    """64 + 128 samples (BASELINE config 2 shape) on the SYNTHETIC STRESS SCENE against the fp32 oracle.

    This is a characterisation of the worst case, not the acceptance bar: random He-initialised weights with sigma
    logits ~ N(-2, 6^2) at every sample -- including the last one, whose distance is 1e10 (train.py:745) -- make
    every ray's background decision hinge on the sign of a large random number, so any rounding error at all flips a
    few rays per thousand.  The stated bar (PSNR(ours, reference render) >= 40 dB on ALL rays, |PSNR vs GT difference|
    <= 0.1 dB) is enforced on a realistic model in tests/test_fitted_checkpoint.py (bf16: 65.8 dB on a full 512x384
    frame) and in __graft_entry__.smoke(); the thresholds below are regression guards at the measured level."""
    snr_bar, psnr_bar, flip_bar = PRECISION_BARS[precision]
    cfg = SceneConfig()
    scene = make_scene(cfg, 0)
    rays, latents = make_rays(4096, 7, cfg)
    ref = O.batchify_rays(rays, latents, scene, chunk=1024, retraw=True, detailed_output=True)
    got = hip_render(scene, rays, latents, precision, retraw=True, detailed=True)
    rms = lambda k: float((got[k] - ref[k]).pow(2).mean().sqrt())
    # "f16": the deformation is evaluated with the fp32-equivalent split product, bent points agree to fp32 rounding;
    # "bf16": single f16 product, relative error ~1e-3 of the offsets
    print(f"[{precision}] rmse bent pts {rms('input_pts'):.2e}, offsets {rms('unmasked_offsets'):.2e}, rigidity {rms('rigidity_mask'):.2e}")
    if precision == "f16":
        assert rms("input_pts") < 2e-6 and rms("unmasked_offsets") < 2e-6 and rms("rigidity_mask") < 2e-5
    else:
        assert rms("input_pts") < 1e-4 and rms("unmasked_offsets") < 2e-4 and rms("rigidity_mask") < 2e-3
    flips = _last_sample_flips(got["raw"], ref["raw"])      # last merged depth is always `far`: same point in both
    keep = ~flips
    p_all, p_keep = psnr(got["rgb_map"], ref["rgb_map"]), psnr(got["rgb_map"][keep], ref["rgb_map"][keep])
    p0 = psnr(got["rgb0"], ref["rgb0"])
    print(f"[{precision}] 64+128: PSNR rgb_map all rays {p_all:.1f} dB, non-flipped {p_keep:.1f} dB "
          f"({int(flips.sum())}/{flips.numel()} flipped), rgb0 {p0:.1f} dB; rmse bent pts {rms('input_pts'):.1e}, "
          f"coarse weights {rms('visibility_weights'):.1e}")
    assert flips.float().mean().item() <= flip_bar
    # rays whose COARSE last sample flipped are still in `keep` (the coarse raw is not an output when I > 0); they get
    # differently placed fine samples, which costs a few dB here relative to the coarse-only test
    assert p_keep >= {"bf16": 36.0, "f16": 44.0}[precision], p_keep
    assert p_all >= 32.0, p_all                    # regression guard on the raw number


def test_fp32_mode_vs_oracle_4k_rays():
    """Same comparison as the golden test on a fresh seed and 4096 rays (oracle finishes in seconds)."""
    cfg = SceneConfig()
    scene = make_scene(cfg, 2)
    rays, latents = make_rays(4096, 11, cfg)
    ref = O.batchify_rays(rays, latents, scene, chunk=1024)
    got = hip_render(scene, rays, latents, "f32")
    fails = compare_dict(got, ref, keys=["rgb0", "disp0", "acc0"])
    fine = oracle_fine_given_z(scene, rays, latents, got["_z_vals"])
    fails += compare_dict(got, fine, keys=["rgb_map", "disp_map", "acc_map"])
    fails += compare_dict(got, ref, keys=["rgb_map", "acc_map"], frac_ok=0.10, outlier_atol=2e-2)
    moved = ((got["_z_vals"] - ref["_z_vals"]).abs() > 2e-5).float().mean().item()
    check_pinned("oracle_4k_rays", dict(moved_depths=moved, rgb_map=out_of_tolerance_fraction(got, ref, "rgb_map"),
                                        acc_map=out_of_tolerance_fraction(got, ref, "acc_map")),
                 dict(moved_depths=got["_z_vals"].numel(), rgb_map=3 * 4096, acc_map=4096))
    assert not fails, "\n".join(fails)


def test_fp32_mode_full_frame_vs_gpu_eager_oracle():
    """BASELINE config 2's size in fp32 mode: 196 608 rays (one 512x384 frame), 64+128, against the oracle evaluated as
    eager fp32 torch ops ON THIS GPU (the reference's own arithmetic on this device: the CPU oracle would need minutes).
    Coarse maps at the fp32 tolerance on every ray; merged depths, final maps: measured fractions, pinned."""
    cfg = SceneConfig()
    scene = make_scene(cfg, 1)
    n = 196608
    rays, latents = make_rays(n, 17, cfg)
    with torch.no_grad():
        ref = O.batchify_rays(rays.to(DEV), latents.to(DEV), O.scene_on(scene, DEV), chunk=16384)
    ref = {k: v.cpu() for k, v in ref.items()}
    got = hip_render(scene, rays, latents, "f32")
    # the GPU's eager GEMMs and this library's MFMA chains round differently: the coarse maps agree to the fp32 tolerance
    # on all but a few rays in 10^5 whose sigma sits on the relu kink of a sample with a huge distance
    fails = compare_dict(got, ref, keys=["rgb0", "disp0", "acc0"], frac_ok=1e-4, outlier_atol=2e-2)
    # (2e-2 bounds what one moved sample does to a ray on a few thousand rays; the tail of 196 608 rays reaches 3.7e-2)
    fails += compare_dict(got, ref, keys=["rgb_map", "acc_map"], frac_ok=0.10, outlier_atol=5e-2)
    zg, zo = got["_z_vals"], ref["_z_vals"]
    assert (zg[:, 1:] >= zg[:, :-1]).all()
    moved = ((zg - zo).abs() > 2e-5).float().mean().item()
    assert float((zg - zo).abs().max()) < 1.1 / (cfg.N_samples - 1), "a depth moved by more than one coarse bin"
    check_pinned("full_frame_196608", dict(moved_depths=moved, rgb0=out_of_tolerance_fraction(got, ref, "rgb0"),
                                           rgb_map=out_of_tolerance_fraction(got, ref, "rgb_map"),
                                           acc_map=out_of_tolerance_fraction(got, ref, "acc_map")),
                 dict(moved_depths=zg.numel(), rgb0=3 * n, rgb_map=3 * n, acc_map=n))
    print(f"[fp32, 196 608 rays vs the eager oracle on this GPU] PSNR rgb_map {psnr(got['rgb_map'], ref['rgb_map']):.1f} dB, "
          f"rgb0 {psnr(got['rgb0'], ref['rgb0']):.1f} dB")
    # The yardstick beside the pinned fractions: the reference's OWN arithmetic against itself at another precision -- the
    # oracle in fp64 (same ops, fp64 weights / rays / latents) against the oracle in fp32 on the same frame.  Its moved depths and
    # out-of-tolerance values are what sample_pdf's `denom < 1e-5` branch (rnh:694) does to ANY fp32 evaluation, these kernels' included.
    with torch.no_grad():
        ref64 = O.batchify_rays(rays.to(DEV), latents.to(DEV), O.scene_on(scene, DEV), chunk=16384, dtype=torch.float64)
    ref64 = {k: v.float().cpu() for k, v in ref64.items()}
    moved64 = ((ref["_z_vals"] - ref64["_z_vals"]).abs() > 2e-5).float().mean().item()
    yard = {k: out_of_tolerance_fraction(ref, ref64, k) for k in ("rgb0", "rgb_map", "acc_map")}
    print(f"[yardstick: the oracle in fp32 vs the oracle in fp64, same frame] moved depths {moved64:.4%} (HIP fp32 vs oracle fp32: {moved:.4%}); "
          f"values outside the fp32 tolerance: rgb0 {yard['rgb0']:.4%}, rgb_map {yard['rgb_map']:.4%} (HIP: {out_of_tolerance_fraction(got, ref, 'rgb_map'):.4%}), "
          f"acc_map {yard['acc_map']:.4%}; PSNR rgb_map {psnr(ref['rgb_map'], ref64['rgb_map']):.1f} dB")
    # the kernels must not be further from the reference's fp32 arithmetic than that arithmetic is from its own fp64 evaluation, give or take 2x
    assert moved <= 2.0 * moved64 + 1e-3, (moved, moved64)
    assert out_of_tolerance_fraction(got, ref, "rgb_map") <= 2.0 * yard["rgb_map"] + 1e-2, (out_of_tolerance_fraction(got, ref, "rgb_map"), yard["rgb_map"])
    assert not fails, "\n".join(fails)


def test_split_bender_path_at_full_chunk_size_equals_the_fused_pass():
    """One reference chunk (32 768 rays, 64+128): the split-bender path a plain render takes against the fused fine pass a
    detailed render takes -- every common output bit-identical at BASELINE config 2's size as well (f16 mode; bf16 mode
    with the trunk-only pass on the same 32x32x16 tiles, up to the conversion ties described in
    _assert_split_equals_fused_up_to_conversion_ties; bf16 mode's default 16x16x32 trunk-only kernel as
    _assert_x16_is_an_equally_good_rounding states)."""
    cfg = SceneConfig()
    scene = make_scene(cfg, 0)
    rays, latents = make_rays(32768, 3, cfg)
    with _setenv("NRNERF_X16", "0"):
        a = hip_render(scene, rays, latents, "f16", retraw=True)                 # split-bender path, trunk-only pass on 32x32x16 tiles
    b = hip_render(scene, rays, latents, "f16", retraw=True, detailed=True)      # fused fine pass
    for k in a:
        assert torch.equal(torch.nan_to_num(a[k]), torch.nan_to_num(b[k])), k
    ref32 = hip_render(scene, rays, latents, "f32", retraw=True)
    with _same_bender("f16"):
        with _setenv("NRNERF_X16", "1"):                                         # fine pass on the 16x16x32 kernel, coarse pass as above
            _assert_x16_is_an_equally_good_rounding(hip_render(scene, rays, latents, "f16", retraw=True), a, ref32)
        _assert_x16_coarse_is_an_equally_good_rounding(hip_render(scene, rays, latents, "f16", retraw=True), a, ref32)
    with _setenv("NRNERF_X16", "0"):                                             # trunk-only pass on the 32x32x16 kernel
        a = hip_render(scene, rays, latents, "bf16", retraw=True)
    b = hip_render(scene, rays, latents, "bf16", retraw=True, detailed=True)
    assert torch.equal(a["rgb0"], b["rgb0"]) and torch.equal(a["_z_vals"], b["_z_vals"])
    assert (a["raw"] != b["raw"]).any(-1).float().mean().item() < 5e-4 and (a["rgb_map"] - b["rgb_map"]).abs().max().item() < 2e-3
    # the default trunk-only pass (16x16x32 kernel): another summation order, judged against the fp32-MFMA render
    with _setenv("NRNERF_X16", "1"):
        x16 = hip_render(scene, rays, latents, "bf16", retraw=True)
    _assert_x16_is_an_equally_good_rounding(x16, a, ref32)
    # the default: the coarse pass on the 16x16x32 kernel as well (stand-alone bender + trunk)
    _assert_x16_coarse_is_an_equally_good_rounding(hip_render(scene, rays, latents, "bf16", retraw=True), a, ref32)


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_full_size_properties(precision):
    """BASELINE config 2 size (32768-ray chunk, 64+128): properties that need no oracle."""
    cfg = SceneConfig()
    scene = make_scene(cfg, 0)
    n = 32768 if precision == "bf16" else 8192
    rays, latents = make_rays(n, 3, cfg)
    a = hip_render(scene, rays, latents, precision, detailed=True)
    # determinism / chunk invariance (train.py:344-345): split launch == single launch, bit for bit
    rb, coarse, fine = build_modules(scene, device=DEV)
    R.set_precision(precision)
    model = R.get_model(coarse, fine)
    # (detailed renders take the fused fine pass, plain ones the split-bender path: each compared with its own kind)
    whole_plain = hip_render(scene, rays, latents, precision)
    for detailed, whole in ((True, a), (False, whole_plain)):
        parts = [model.render(rays[i:i + 5000].to(DEV), latents[i:i + 5000].to(DEV), 64, 128, want_z_vals=True, detailed_output=detailed)
                 for i in range(0, n, 5000)]
        torch.cuda.synchronize()
        for k in ("rgb_map", "disp_map", "acc_map", "rgb0", "z_std", "_z_vals"):
            cat = torch.cat([p[k] for p in parts], 0).cpu()
            assert torch.equal(torch.nan_to_num(cat), torch.nan_to_num(whole[k])), f"{k} depends on the launch split (detailed={detailed})"
    # compositing identities
    w = a["fine_visibility_weights"]
    assert torch.allclose(w.sum(-1), a["acc_map"], atol=2e-5)
    assert (w >= 0).all() and (a["acc_map"] <= 1 + 1e-4).all()
    assert (a["fine_opacity_alpha"] >= 0).all() and (a["fine_opacity_alpha"] <= 1).all()
    z = a["_z_vals"]
    assert (z[:, 1:] >= z[:, :-1]).all() and (z >= cfg.near - 1e-6).all() and (z <= cfg.far + 1e-6).all()
    # the coarse depths are a subset of the merged depths (train.py:920)
    t = torch.linspace(0, 1, 64)
    zc = cfg.near * (1 - t) + cfg.far * t
    assert (torch.isclose(z[:, :, None], zc[None, None, :], atol=1e-6, rtol=0).any(1)).all()
    # bending identities (run_nerf_helpers.py:567-570)
    assert torch.allclose(a["fine_masked_offsets"], a["fine_rigidity_mask"] * a["fine_unmasked_offsets"], atol=1e-6)
    assert torch.allclose(a["fine_input_pts"], a["fine_initial_input_pts"] + a["fine_masked_offsets"], atol=1e-6)
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]
    assert torch.allclose(a["fine_initial_input_pts"], pts, atol=1e-6)
    assert not torch.isnan(a["rgb_map"]).any() and not torch.isinf(a["rgb_map"]).any()


def test_broadcast_latent_equals_expanded():
    """render_path hands one frame code expanded to all rays (train.py:464-466): stride-0 path == materialised."""
    cfg = SceneConfig()
    scene = make_scene(cfg, 0)
    rays, latents = make_rays(777, 5, cfg)
    one = latents[:1]
    a = hip_render(scene, rays, one.expand(777, -1), "f32", use_batchify=False)
    b = hip_render(scene, rays, one.expand(777, -1).contiguous(), "f32", use_batchify=False)
    for k in a:
        assert torch.equal(torch.nan_to_num(a[k]), torch.nan_to_num(b[k])), k


@pytest.mark.parametrize("bend", [True, False])
def test_viewdirs_fp32_vs_oracle(bend):
    """View-dependent head: finite-difference directions of the bent points (run_nerf_helpers.py:316-356) with a
    bender, the rays' own unit directions without (train.py:73-76).  513 rays so that workgroup ranges are ragged."""
    cfg = SceneConfig(use_viewdirs=True, N_importance=128, ray_bending=bend)
    scene = make_scene(cfg, 1)
    rays, latents = make_rays(513, 9, cfg)
    ref = O.batchify_rays(rays, latents, scene, chunk=257, retraw=True)
    got = hip_render(scene, rays, latents, "f32", retraw=True)
    assert got["raw"].shape == ref["raw"].shape == (513, 192, 4)
    fails = compare_dict(got, ref, keys=["rgb0", "disp0", "acc0"])
    fine = oracle_fine_given_z(scene, rays, latents, got["_z_vals"])
    fails += compare_dict(got, fine, keys=["rgb_map", "disp_map", "acc_map"])
    fails += compare_dict(got, fine, keys=["raw"], **(FD_DIRS_RAW if bend else {}))
    fails += compare_dict(got, ref, keys=["rgb_map", "acc_map"], frac_ok=0.10, outlier_atol=2e-2)
    assert not fails, "\n".join(fails)
    got16 = hip_render(scene, rays, latents, "bf16")
    assert psnr(got16["rgb0"], ref["rgb0"]) > 30.0


@pytest.mark.parametrize("S,I,n,views", [(48, 37, 5, False), (33, 0, 1, False), (64, 192, 3, False), (2, 0, 7, False),
                                         (40, 24, 9, True), (256, 0, 2, False)])
def test_ragged_sample_counts_vs_oracle(S, I, n, views):
    """Sample counts that are not multiples of the 32-sample block or the 64-lane wave, one ray, the 256-sample cap."""
    cfg = SceneConfig(N_samples=S, N_importance=I, use_viewdirs=views)
    scene = make_scene(cfg, 4)
    rays, latents = make_rays(n, 13, cfg)
    ref = O.batchify_rays(rays, latents, scene, retraw=True)
    got = hip_render(scene, rays, latents, "f32", retraw=True)
    assert got["raw"].shape == ref["raw"].shape
    if I == 0:
        fails = compare_dict(got, ref, keys=[k for k in ref if not k.startswith("_")])
    else:
        fails = compare_dict(got, ref, keys=["rgb0", "disp0", "acc0"])
        fine = oracle_fine_given_z(scene, rays, latents, got["_z_vals"])
        fails += compare_dict(got, fine, keys=["rgb_map", "disp_map", "acc_map"])
        fails += compare_dict(got, fine, keys=["raw"], **(FD_DIRS_RAW if views else {}))
    assert not fails, "\n".join(fails)


def test_million_ray_launch_bf16():
    """2^20 + 3 rays in one call (one launch of 2^20, one of 3): 288 GB sizing, no chunk loop needed (train.py:108-137)."""
    cfg = SceneConfig(N_importance=0)
    scene = make_scene(cfg, 0)
    n = (1 << 20) + 3
    rays, latents = make_rays(4099, 21, cfg)
    reps = (n + 4098) // 4099
    rays_big, lat_big = rays.repeat(reps, 1)[:n], latents.repeat(reps, 1)[:n]
    got = hip_render(scene, rays_big, lat_big, "bf16", want_z=False)
    assert got["rgb_map"].shape == (n, 3)
    # periodic input -> periodic output, bit for bit, across the launch split as well
    for k in ("rgb_map", "acc_map"):
        a = got[k]
        assert torch.equal(torch.nan_to_num(a[:4099]), torch.nan_to_num(a[4099 * 254:4099 * 255]))
        tail = a[(1 << 20):]
        assert torch.equal(torch.nan_to_num(tail), torch.nan_to_num(a[(1 << 20) % 4099:(1 << 20) % 4099 + 3]))


def test_boundary_contract_errors_and_fallback():
    cfg = SceneConfig(netwidth=640, N_importance=64)        # beyond the run-time-parameterised kernel's 512 columns (round 4: W = 192 etc. render natively)
    scene = make_scene(cfg, 0)
    rays, latents = make_rays(8, 0, cfg)
    with pytest.raises(R.Unsupported):
        hip_render(scene, rays, latents, "f32")
    # detailed_output with N_importance == 0 raises like the reference (train.py:900-908 vs 967-970)
    cfg0 = SceneConfig(N_importance=0)
    scene0 = make_scene(cfg0, 0)
    rays0, lat0 = make_rays(8, 0, cfg0)
    with pytest.raises(UnboundLocalError):
        hip_render(scene0, rays0, lat0, "f32", detailed=True)

    # install(): unsupported calls go to the saved reference functions -- the reference's own batchify_rays WITH THE
    # CALLER'S chunk (its memory bound, train.py:344-345), whose module-global render_rays lookup then lands in the saved
    # reference render_rays -- supported ones to the HIP path
    class FakeTrain:
        calls = []

        @staticmethod
        def render_rays(ray_batch, *a, **k):
            FakeTrain.calls.append(("render_rays", ray_batch.shape[0]))
            return {"rgb_map": torch.zeros(ray_batch.shape[0], 3)}

        @staticmethod
        def batchify_rays(rays_flat, api, chunk=1024 * 32, detailed_output=False, **k):      # train.py:108-137
            FakeTrain.calls.append(("batchify_rays", chunk))
            parts = [FakeTrain.render_rays(rays_flat[i:i + chunk], additional_pixel_information=api,
                                           detailed_output=detailed_output, **k)["rgb_map"] for i in range(0, rays_flat.shape[0], chunk)]
            return {"rgb_map": torch.cat(parts, 0)}

    undo = R.install(FakeTrain, precision="f32")
    try:
        rb, coarse, fine = build_modules(scene0, device=DEV)
        api = {"ray_bending_latents": lat0.to(DEV)}
        with torch.no_grad():
            out = FakeTrain.batchify_rays(rays0.to(DEV), api, chunk=5, network_fn=coarse, network_query_fn=None, N_samples=64,
                                          pytest=True)          # numpy-seeded debug randoms (train.py:863-867) -> reference
            assert FakeTrain.calls == [("batchify_rays", 5), ("render_rays", 5), ("render_rays", 3)], FakeTrain.calls
            assert out["rgb_map"].shape == (8, 3)
            FakeTrain.calls.clear()
            out = FakeTrain.batchify_rays(rays0.to(DEV), api, network_fn=coarse, network_query_fn=None, N_samples=64,
                                          perturb=0.0)          # supported -> HIP
            assert FakeTrain.calls == [] and out["rgb_map"].is_cuda
            # a width the library has no kernel for: decided once per batchify call, verdict cached per module
            rbw, cw, fw = build_modules(scene, device=DEV)
            out = FakeTrain.batchify_rays(rays.to(DEV), {"ray_bending_latents": latents.to(DEV)}, chunk=3, network_fn=cw,
                                          network_fine=fw, network_query_fn=None, N_samples=64, N_importance=64)
            assert FakeTrain.calls[0] == ("batchify_rays", 3) and len(FakeTrain.calls) == 4
            FakeTrain.calls.clear()
        # a frozen coarse net with a trainable bender under autograd is a training call as well (the bender is not a
        # submodule of the NeRF modules, run_nerf_helpers.py:213-215)
        for p_ in list(coarse.parameters()) + list(fine.parameters() if fine is not None else []):
            p_.requires_grad_(False)
        # ... and is taken by the native training path (nonrigid_nerf_amd/training.py), not handed to the reference
        out = FakeTrain.batchify_rays(rays0.to(DEV), api, network_fn=coarse, network_query_fn=None, N_samples=64)
        assert FakeTrain.calls == [] and out["rgb_map"].requires_grad
        # exact Jacobian view directions under autograd: native since the end of round 4 (the tangent of the divergence kernels)
        cfgv = SceneConfig(N_importance=0, use_viewdirs=True, approx_nonrigid_viewdirs=False)
        rbv, cv, _ = build_modules(make_scene(cfgv, 0), device=DEV)
        rv, lv = make_rays(8, 0, cfgv)
        out = FakeTrain.batchify_rays(rv.to(DEV), {"ray_bending_latents": lv.to(DEV)}, network_fn=cv, network_query_fn=None, N_samples=64)
        assert FakeTrain.calls == [] and out["rgb_map"].requires_grad
        # a trunk width outside the compiled set with the plain head: trained natively since the end of round 5 (training._GenericTrunk) ...
        cfgw = SceneConfig(N_importance=0, netwidth=192)
        rbw2, cw2, _ = build_modules(make_scene(cfgw, 0), device=DEV)
        rw2, lw2 = make_rays(8, 0, cfgw)
        out = FakeTrain.batchify_rays(rw2.to(DEV), {"ray_bending_latents": lw2.to(DEV)}, network_fn=cw2, network_query_fn=None, N_samples=64)
        assert FakeTrain.calls == [] and out["rgb_map"].requires_grad
        out["rgb_map"].sum().backward()
        assert all(p_.grad is not None and bool(torch.isfinite(p_.grad).all()) and float(p_.grad.abs().max()) > 0
                   for p_ in list(cw2.pts_linears.parameters()) + list(cw2.output_linear.parameters()))
        # ... a training call the native path has no kernels for (the same width with exact Jacobian directions and a bender of another
        # shape than the two the bender's training kernels are compiled for: rendered by the generic kernel, but not trained) still goes to the reference
        cfgw = SceneConfig(N_importance=0, netwidth=192, use_viewdirs=True, approx_nonrigid_viewdirs=False, bend_hidden=96)
        rbw3, cw3, _ = build_modules(make_scene(cfgw, 0), device=DEV)
        rw3, lw3 = make_rays(8, 0, cfgw)
        out = FakeTrain.batchify_rays(rw3.to(DEV), {"ray_bending_latents": lw3.to(DEV)}, network_fn=cw3, network_query_fn=None, N_samples=64)
        assert FakeTrain.calls and FakeTrain.calls[0][0] == "batchify_rays"
        # wrong latent shape: the reference raises in expand/split; here a ValueError, never an out-of-bounds read
        with torch.no_grad(), pytest.raises(ValueError):
            R.render_rays(rays0.to(DEV), coarse, N_samples=64, additional_pixel_information={"ray_bending_latents": lat0[:, :16].to(DEV)})
        with torch.no_grad(), pytest.raises(ValueError):
            R.render_rays(rays0.to(DEV), coarse, N_samples=64, additional_pixel_information={"ray_bending_latents": lat0[:4].to(DEV)})
    finally:
        undo()
    assert FakeTrain.render_rays.__name__ == "render_rays" and FakeTrain.render_rays is not R.render_rays


def test_raygen_matches_reference_golden():
    """nrnerf_generate_rays vs reference get_rays outputs (tests/golden/raygen.npz) and render()'s packing."""
    import os
    import numpy as np
    from nonrigid_nerf_amd.driver import generate_rays
    from tests.helpers import GOLDEN_DIR, synthetic_camera
    z = np.load(os.path.join(GOLDEN_DIR, "raygen.npz"))
    for k in range(3):
        c2w, intrin = synthetic_camera(k)
        rays = generate_rays(c2w, intrin, 0.0022, 1.0024, True, DEV).cpu()
        ro, rd = torch.from_numpy(z[f"out__rays_o_{k}"]).reshape(-1, 3), torch.from_numpy(z[f"out__rays_d_{k}"]).reshape(-1, 3)
        assert rays.shape == (24 * 32, 11)
        assert torch.equal(rays[:, 0:3], ro)
        assert torch.allclose(rays[:, 3:6], rd, rtol=0, atol=2e-7)
        assert torch.equal(rays[:, 6], torch.full((768,), 0.0022)) and torch.equal(rays[:, 7], torch.full((768,), 1.0024))
        assert torch.allclose(rays[:, 8:11], rd / rd.norm(dim=-1, keepdim=True), rtol=0, atol=2e-7)
        assert generate_rays(c2w, intrin, 0.0022, 1.0024, False, DEV).shape == (768, 8)


def test_render_path_driver_vs_oracle():
    """Three 32x24 frames through the frame driver (device ray generation, broadcast frame code, async copies)
    against the oracle's render_path restatement (train.py:419-553)."""
    from nonrigid_nerf_amd.driver import render_path
    from tests.helpers import synthetic_camera
    cfg = SceneConfig(N_importance=0)
    scene = make_scene(cfg, 0)
    cams = [synthetic_camera(k) for k in range(3)]
    poses, intrins = [c for c, _ in cams], [i for _, i in cams]
    codes = torch.randn(3, 32, generator=torch.Generator().manual_seed(3)) * 0.1
    ref_rgb, ref_disp = O.render_path(poses, intrins, scene, codes)
    rb, coarse, fine = build_modules(scene, device=DEV)
    R.set_precision("f32")
    kw = dict(network_fn=coarse, network_fine=fine, network_query_fn=None, N_samples=cfg.N_samples, N_importance=0,
              perturb=False, raw_noise_std=0.0, white_bkgd=False, lindisp=False, ndc=False, use_viewdirs=False,
              ray_bender=rb, near=cfg.near, far=cfg.far)
    rgbs, disps = render_path([p.to(DEV) for p in poses], intrins, 1024 * 32, kw, codes.to(DEV))
    assert rgbs.shape == (3, 24, 32, 3) and disps.shape == (3, 24, 32) and rgbs.dtype == np.float32
    assert torch.allclose(torch.from_numpy(rgbs), ref_rgb, atol=1e-4, rtol=0)
    d, rd_ = torch.from_numpy(disps), ref_disp
    both_nan = torch.isnan(d) & torch.isnan(rd_)
    assert (both_nan | ((d - rd_).abs() <= 1e-4 + 1e-3 * rd_.abs())).all()
    half_rgb, half_disp = render_path([p.to(DEV) for p in poses], intrins, 1024 * 32, kw, codes.to(DEV), render_factor=2)
    assert half_rgb.shape == (3, 12, 16, 3) and half_disp.shape == (3, 12, 16)       # train.py:434-446


@pytest.mark.parametrize("cfg_kw", [dict(), dict(netdepth=6, netwidth=192, netwidth_fine=320, latent_size=16), dict(netwidth=96)],
                         ids=["compiled", "generic", "generic_compiled_bender"])
def test_surface_reduction_matches_host_side_reduction(cfg_kw):
    """free_viewpoint_rendering.py:621-658 picks, per pixel, the sample whose accumulated visibility is closest to 0.5
    and reads the bent point and rigidity there.  The in-kernel reduction must agree with doing exactly that on the
    detailed outputs of the same render (bit-identical weights), and with the oracle up to near-ties.  Also for the
    architectures on the run-time-parameterised kernel (whose bender pass writes the points the reduction reads)."""
    cfg = SceneConfig(**cfg_kw)
    scene = make_scene(cfg, 0)
    rays, latents = make_rays(2048, 17, cfg)
    rb, coarse, fine = build_modules(scene, device=DEV)
    R.set_precision("f32")
    model = R.get_model(coarse, fine)
    with torch.no_grad():
        out = model.render(rays.to(DEV), latents.to(DEV), 64, 128, detailed_output=True, surface=True)
    torch.cuda.synchronize()
    out = {k: v.cpu() for k, v in out.items()}
    idx, pts, rig = O.surface_from_details(out["fine_visibility_weights"], out["fine_input_pts"], out["fine_rigidity_mask"])
    same = out["median_index"].long() == idx
    assert same.float().mean() > 0.999            # same sequential cumsum: plateaus of zero weight tie exactly, first index wins
    acc = torch.cumsum(out["fine_visibility_weights"], -1)
    d_ours = (acc[torch.arange(2048), out["median_index"].long()] - 0.5).abs()
    d_ref = (acc[torch.arange(2048), idx] - 0.5).abs()
    assert torch.allclose(d_ours, d_ref, atol=1e-6)
    assert torch.equal(out["surface_pts"][same], pts[same]) and torch.equal(out["surface_rigidity"][same], rig[same])
    assert torch.equal(out["surface_pts"], out["fine_input_pts"][torch.arange(2048), out["median_index"].long()])
    # the same reduction WITHOUT detail outputs (the route a frame render takes: split-bender path + fused compositing, or the
    # generic path with the compiled bender): same index, same point
    with torch.no_grad():
        plain = model.render(rays.to(DEV), latents.to(DEV), 64, 128, surface=True)
    if not cfg_kw:          # compiled fp32 path: the split-bender route is bit-identical to the fused one
        assert torch.equal(plain["median_index"].cpu(), out["median_index"]) and torch.equal(plain["surface_pts"].cpu(), out["surface_pts"])
    else:                   # generic: the plain route may take the compiled bender kernel (an ulp away from the generic fp32 bender)
        assert (plain["median_index"].cpu() == out["median_index"]).float().mean() > 0.995
        assert torch.allclose(plain["surface_pts"].cpu(), out["surface_pts"], atol=2e-2) and (plain["surface_pts"].cpu() - out["surface_pts"]).abs().median() < 1e-6
    # coarse-only render: reduction over the coarse pass
    cfg0 = SceneConfig(N_importance=0, **cfg_kw)
    scene0 = make_scene(cfg0, 0)
    rb0, c0, _ = build_modules(scene0, device=DEV)
    m0 = R.get_model(c0, None)
    with torch.no_grad():
        o0 = m0.render(rays[:100].to(DEV), latents[:100].to(DEV), 64, 0, surface=True)
    ref0 = O.render_rays(rays[:100], latents[:100], scene0, retraw=True)
    w0 = O.composite(ref0["raw"], ref0["_z_vals"], rays[:100, 3:6])[4]
    assert (o0["median_index"].cpu().long() == O.surface_from_details(w0, torch.zeros(100, 64, 3))[0]).float().mean() > 0.97


def test_concurrent_renders_from_two_threads_on_two_streams():
    """SURVEY.md section 8b (threading / streams): the handle is immutable during render and the call is asynchronous on
    the caller's current stream, so two host threads may render through the same handle on their own streams (the
    DataParallel wrapper, train.py:310-323, runs one thread per replica).  Scratch is kept per stream."""
    import threading
    cfg = SceneConfig(N_importance=64)
    scene = make_scene(cfg, 0)
    rb, coarse, fine = build_modules(scene, device=DEV)
    R.set_precision("bf16")
    model = R.get_model(coarse, fine)
    batches = [tuple(t.to(DEV) for t in make_rays(20000 + 7 * k, 40 + k, cfg)) for k in range(2)]
    with torch.no_grad():
        want = [model.render(r, l, 64, 64) for r, l in batches]
    torch.cuda.synchronize()
    got, errors = [None, None], []

    def work(k):
        try:
            s = torch.cuda.Stream(device=DEV)
            with torch.no_grad(), torch.cuda.stream(s):
                for _ in range(3):                       # overlapping launch sequences on both streams
                    out = model.render(*batches[k], 64, 64)
            s.synchronize()
            got[k] = out
        except Exception as e:                           # surfaced in the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    for k in range(2):
        for key in ("rgb_map", "disp_map", "acc_map", "rgb0", "z_std"):
            assert torch.equal(torch.nan_to_num(got[k][key]), torch.nan_to_num(want[k][key])), (k, key)


def test_changed_weights_are_repacked_into_the_same_handle():
    """Weights mutated in place (optimiser step, load_state_dict: train.py:666-682) bump the parameters' version; the
    next call must render with the new weights, through nrnerf_model_update on the existing handle."""
    cfg = SceneConfig(N_importance=64)
    scene = make_scene(cfg, 0)
    rb, coarse, fine = build_modules(scene, device=DEV)
    rays, latents = make_rays(512, 5, cfg)
    rays, latents = rays.to(DEV), latents.to(DEV)
    R.set_precision("f32")
    kw = dict(network_fn=coarse, network_fine=fine, N_samples=64, N_importance=64,
              additional_pixel_information={"ray_bending_latents": latents})
    with torch.no_grad():
        m0 = R.get_model(coarse, fine, device=DEV)
        before = R.render_rays(rays, **kw)["rgb_map"].clone()
        coarse.pts_linears[3].weight.mul_(1.05)
        fine.output_linear.bias.add_(0.25)
        rb.network[1].weight.mul_(0.9)
        m1 = R.get_model(coarse, fine, device=DEV)
        after = R.render_rays(rays, **kw)["rgb_map"].clone()
        fresh = R.Model(coarse, fine, "f32", DEV)                       # packed from scratch from the modified modules
        want = fresh.render(rays, latents, 64, 64)["rgb_map"]
    torch.cuda.synchronize()
    assert m1 is m0, "same architecture: the handle must be refreshed, not replaced"
    assert (after - before).abs().max() > 1e-3, "the render did not pick up the new weights"
    assert torch.equal(after, want)


def test_report_eager_pytorch_on_the_same_gpu():
    """Context number ("also report the reference eager path on the MI355X"): the oracle's PyTorch
    ops executed eagerly on the GPU -- what running the reference through PyTorch-ROCm amounts to, minus its netchunk
    loop and host syncs -- against the HIP path on the same rays.  Printed with `pytest -s`; the assertion only guards
    the ordering."""
    import time
    cfg = SceneConfig()
    scene = make_scene(cfg, 0)
    n = 32768
    rays, latents = make_rays(n, 100, cfg)
    rays, latents = rays.to(DEV), latents.to(DEV)
    sc = O.scene_on(scene, DEV)
    rb, coarse, fine = build_modules(scene, device=DEV)
    kw = dict(network_fn=coarse, network_fine=fine, N_samples=64, N_importance=128)
    api = {"ray_bending_latents": latents}

    def rate(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return n * reps / (time.perf_counter() - t0)

    with torch.no_grad():
        eager = rate(lambda: O.batchify_rays(rays, latents, sc, chunk=n))
        R.set_precision("f32")
        ours32 = rate(lambda: R.batchify_rays(rays, api, **kw))
        R.set_precision("bf16")
        ours16 = rate(lambda: R.batchify_rays(rays, api, **kw), reps=10)
    print(f"\n[eager torch fp32 on the GPU] {eager / 1e6:.3f} M rays/s; HIP path: f32 mode {ours32 / 1e6:.3f} M rays/s "
          f"({ours32 / eager:.1f}x), bf16 mode {ours16 / 1e6:.3f} M rays/s ({ours16 / eager:.1f}x)")
    assert ours32 > eager and ours16 > 5 * eager


VARIANT_CFGS = {
    "no_bender":            dict(N_importance=0, ray_bending=False),
    "viewdirs_bender":      dict(N_importance=0, use_viewdirs=True),
    "viewdirs_no_bender":   dict(N_importance=0, use_viewdirs=True, ray_bending=False),
    "deep_bender":          dict(N_importance=0, bend_depth=7),
    "deep_bender_viewdirs": dict(N_importance=0, bend_depth=7, use_viewdirs=True),
    "time_conditioned":     dict(N_importance=0, ray_bending=False, time_conditioned_baseline=True),
    "time_conditioned_viewdirs": dict(N_importance=0, ray_bending=False, time_conditioned_baseline=True, use_viewdirs=True),
    "exact_viewdirs":       dict(N_importance=0, use_viewdirs=True, approx_nonrigid_viewdirs=False),
    "deep_bender_exact_viewdirs": dict(N_importance=0, use_viewdirs=True, bend_depth=7, approx_nonrigid_viewdirs=False),
    "narrow_128":           dict(N_importance=0, netwidth=128),
    "narrow_128_no_bender": dict(N_importance=0, netwidth=128, ray_bending=False),
    # outside the compiled set: the run-time-parameterised kernel (hidden activations in the 16-bit type, inputs f16, bender fp32)
    "generic_192_depth6":   dict(N_importance=0, netdepth=6, netwidth=192, multires=8, latent_size=16),
    "generic_viewdirs_320": dict(N_importance=0, netwidth=320, use_viewdirs=True, multires_views=2),
    "generic_512_no_bender": dict(N_importance=0, netwidth=512, netdepth=10, ray_bending=False),
}


@pytest.mark.parametrize("precision", ["bf16", "f16"])
@pytest.mark.parametrize("variant", list(VARIANT_CFGS))
def test_16bit_kernels_of_every_compiled_variant_track_the_fp32_kernel(variant, precision):
    """Every compiled 16-bit kernel family (no bender, view-dependent head with / without bender, 7-layer bender,
    time-conditioned baseline) against the exact-fp32 kernel of the same family on the same rays (coarse-only, so no
    sampling decision sits between the two): network-output SNR and flip-aware PSNR at the bars of the default family."""
    snr_bar, psnr_bar, flip_bar = PRECISION_BARS[precision]
    cfg = SceneConfig(**VARIANT_CFGS[variant])
    scene = make_scene(cfg, 3)
    rays, latents = make_rays(4096, 21, cfg)
    ref = hip_render(scene, rays, latents, "f32", retraw=True)
    with _same_bender(precision):       # ("f16": its tight bars are those of the f16 trunk behind the fp32-equivalent three-product bender)
        got = hip_render(scene, rays, latents, precision, retraw=True)
    # ... and against the ORACLE itself (eager fp32 torch ops on the GPU), so the 16-bit kernels are not only held to
    # another kernel of this library
    with torch.no_grad():
        orc = O.batchify_rays(rays.to(DEV), latents.to(DEV), O.scene_on(scene, DEV), chunk=4096, retraw=True)
    orc = {k: v.cpu() for k, v in orc.items()}
    err = got["raw"] - ref["raw"]
    snr = [float(20 * torch.log10(ref["raw"][..., c].std() / err[..., c].pow(2).mean().sqrt())) for c in range(4)]
    err_o = got["raw"] - orc["raw"]
    snr_o = [float(20 * torch.log10(orc["raw"][..., c].std() / err_o[..., c].pow(2).mean().sqrt())) for c in range(4)]
    flips = _last_sample_flips(got["raw"], ref["raw"])
    keep = ~flips
    p_keep = psnr(got["rgb_map"][keep], ref["rgb_map"][keep])
    keep_o = ~_last_sample_flips(got["raw"], orc["raw"])
    p_keep_o = psnr(got["rgb_map"][keep_o], orc["rgb_map"][keep_o])
    print(f"[{variant} / {precision}] raw SNR {[round(x, 1) for x in snr]} dB (vs oracle {[round(x, 1) for x in snr_o]}), "
          f"flipped {int(flips.sum())}/{flips.numel()}, PSNR non-flipped {p_keep:.1f} dB (vs oracle {p_keep_o:.1f})")
    # the SAME bars for every family (1.5 dB under the default family's network-output bar: other seeds / wider inputs --
    # time-conditioned: 95 instead of 63 columns); no extra allowance for the view-dependent head with finite-difference
    # directions (measured round 3: colour SNR 34.4-38.7 dB, PSNR over non-flipped rays 52-62 dB in bf16 mode)
    assert min(snr) >= snr_bar - 1.5, snr
    assert flips.float().mean().item() <= flip_bar
    assert p_keep >= psnr_bar, p_keep
    assert min(snr_o) >= snr_bar - 1.5, snr_o
    assert (~keep_o).float().mean().item() <= flip_bar and p_keep_o >= psnr_bar, p_keep_o


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("cfg_kw", [dict(N_importance=64), dict(N_importance=64, use_viewdirs=True), dict(N_importance=32, ray_bending=False, time_conditioned_baseline=True),
                                    dict(N_importance=64, netwidth=128, bend_depth=7)],
                         ids=["default", "viewdirs", "time_conditioned", "w128_deep_bender"])
def test_generic_kernel_agrees_with_the_compiled_kernels_on_their_own_architectures(cfg_kw, precision):
    """NRNERF_FORCE_GENERIC=1 sends a compiled architecture through the run-time-parameterised kernel (csrc/nrnerf_generic.h):
    the two implementations of the same network must agree -- fp32: to rounding (different summation order of the same
    products); bf16: both within the 16-bit bars of each other -- with detailed outputs, ragged ray counts and knobs."""
    import os
    cfg = SceneConfig(**cfg_kw)
    scene = make_scene(cfg, 6)
    rays, latents = make_rays(777, 41, cfg)
    knobs = dict(rigidity_test_time_cutoff=0.4, test_time_scaling=0.8) if cfg.ray_bending else {}
    outs = []
    old = os.environ.get("NRNERF_FORCE_GENERIC")
    try:
        for force in ("0", "1"):
            os.environ["NRNERF_FORCE_GENERIC"] = force
            R.invalidate()                       # the handle is per architecture route: build a fresh one
            outs.append(hip_render(scene, rays, latents, precision, retraw=True, detailed=True, knobs=knobs))
    finally:
        R.invalidate()
        if old is None:
            os.environ.pop("NRNERF_FORCE_GENERIC", None)
        else:
            os.environ["NRNERF_FORCE_GENERIC"] = old
    compiled, generic = outs
    assert set(compiled) == set(generic)
    if precision == "f32":
        fails = compare_dict(generic, compiled, keys=[k for k in COARSE_KEYS if k in compiled])
        moved = ((generic["_z_vals"] - compiled["_z_vals"]).abs() > 2e-5).float().mean().item()
        assert moved < 0.01, moved
        fails += compare_dict(generic, compiled, keys=["rgb_map", "acc_map"], frac_ok=0.10, outlier_atol=2e-2)
        assert not fails, "\n".join(fails)
    else:
        assert psnr(generic["rgb0"], compiled["rgb0"]) >= 34.0       # two bf16 evaluations of the stress scene against each other
        for k in ("input_pts", "unmasked_offsets", "rigidity_mask"):
            if k in compiled:       # generic bender: fp32; compiled bf16 mode: single-product f16 (bars of test_16bit_modes_psnr_full_pipeline)
                assert float((generic[k] - compiled[k]).pow(2).mean().sqrt()) < 2e-3, k


@pytest.mark.parametrize("precision", ["bf16", "f16"])
@pytest.mark.parametrize("cfg_kw", [dict(N_importance=64, netdepth=6, netwidth=192, netwidth_fine=320, multires=8),
                                    dict(N_importance=64, netdepth=5, netwidth=96, skips=(1,), use_viewdirs=True, multires_views=3),
                                    dict(N_importance=128, netwidth=448, netdepth=3, skips=())],
                         ids=["w192_w320_l8", "w96_viewdirs_lv3_skip1", "w448_d3_noskip"])
def test_width_class_kernel_is_an_equally_good_rounding_of_the_generic_kernel(cfg_kw, precision):
    """Non-compiled architectures in the 16-bit modes: a plain render runs its trunks on the width-class 16x16x32 kernel
    (csrc/nrnerf_gx16.h) by default and on the run-time-parameterised kernel of nrnerf_generic.h with NRNERF_X16=0.  Two 16-bit
    evaluations of one network, against the exact-fp32 render of the same call: coarse and final maps' errors alike (mean within
    30 %, maximum within 3 x), no more moved depths, sorted merged depths; ragged ray count."""
    cfg = SceneConfig(**cfg_kw)
    scene = make_scene(cfg, 4)
    rays, latents = make_rays(1777, 23, cfg)
    ref32 = hip_render(scene, rays, latents, "f32", retraw=True)
    with _setenv("NRNERF_X16", "0"):
        gen = hip_render(scene, rays, latents, precision, retraw=True)
    with _same_bender(precision):
        gx = hip_render(scene, rays, latents, precision, retraw=True)
    assert set(gx) == set(gen) and gx["raw"].shape == gen["raw"].shape
    assert (gx["_z_vals"][:, 1:] >= gx["_z_vals"][:, :-1]).all() and torch.isfinite(gx["raw"]).all()
    assert not torch.equal(gx["rgb0"], gen["rgb0"]), "both renders took the same kernel"
    _assert_x16_coarse_is_an_equally_good_rounding(gx, gen, ref32)


@pytest.mark.parametrize("n", [1, 7, 33, 257])
def test_tiny_and_ragged_ray_counts_vs_oracle(n):
    """Batches smaller than one workgroup tile (8 blocks) and not a multiple of anything, with detailed outputs."""
    cfg = SceneConfig(N_importance=64)
    scene = make_scene(cfg, 0)
    rays, latents = make_rays(n, 31, cfg)
    got = hip_render(scene, rays, latents, "f32", retraw=True, detailed=True)
    ref = O.batchify_rays(rays, latents, scene, retraw=True, detailed_output=True)
    assert all(got[k].shape == ref[k].shape for k in ref if not k.startswith("_"))
    fails = compare_dict(got, ref, keys=[k for k in COARSE_KEYS if k in ref])
    fine = oracle_fine_given_z(scene, rays, latents, got["_z_vals"], None, True)
    fails += compare_dict(got, fine, keys=[k for k in fine if k in got])
    assert not fails, "\n".join(fails)


@pytest.mark.parametrize("cfg_kw", [dict(N_samples=128, N_importance=128), dict(N_samples=256, N_importance=0),
                                    dict(N_samples=192, N_importance=320), dict(N_samples=64, N_importance=450),
                                    dict(N_samples=600, N_importance=0), dict(N_samples=512, N_importance=512),
                                    dict(N_samples=200, N_importance=300, ray_bending=False)],
                         ids=["128+128", "256+0", "192+320", "64+450", "600+0", "512+512", "200+300_no_bender"])
def test_large_sample_counts_vs_oracle(cfg_kw):
    """The reference has no cap on --N_samples / --N_importance (train.py:1090-1094).  Up to 256 samples per ray a render
    takes the split-bender path and the fused compositing; beyond (up to NRNERF_MAX_SAMPLES = 1024 per pass) the
    fused-bender fine pass and the composite kernel with up to 16 samples per lane."""
    cfg = SceneConfig(**cfg_kw)
    scene = make_scene(cfg, 0)
    rays, latents = make_rays(96, 37, cfg)
    got = hip_render(scene, rays, latents, "f32", retraw=True)
    ref = O.batchify_rays(rays, latents, scene, retraw=True)
    if cfg.N_importance == 0:
        fails = compare_dict(got, ref)
    else:
        fails = compare_dict(got, ref, keys=["rgb0", "disp0", "acc0"])
        zg = got["_z_vals"]
        assert zg.shape == (96, cfg.N_samples + cfg.N_importance) and (zg[:, 1:] >= zg[:, :-1]).all()
        assert ((zg - ref["_z_vals"]).abs() > 2e-5).float().mean().item() < 0.02
        fine = oracle_fine_given_z(scene, rays, latents, zg)
        fails += compare_dict(got, fine, keys=["rgb_map", "disp_map", "acc_map", "raw"])
    assert not fails, "\n".join(fails)


@pytest.mark.parametrize("precision", ["bf16", "f16"])
@pytest.mark.parametrize("cfg_kw", [dict(N_samples=192, N_importance=128), dict(N_samples=64, N_importance=450), dict(N_samples=600, N_importance=300)],
                         ids=["192+128", "64+450", "600+300"])
def test_large_sample_counts_in_the_16_bit_modes(cfg_kw, precision):
    """Above 256 samples per pass the 16-bit modes leave the split-bender path and the fused compositing as well (fused-bender fine
    pass, composite kernel with up to 16 samples per lane): held to the exact-fp32 kernels of the same call -- coarse maps as close as
    at 64 + 128 (colour SNR bar of the per-family test), merged depths sorted and as many as asked for, final maps within the stress
    scene's all-ray bar."""
    cfg = SceneConfig(**cfg_kw)
    scene = make_scene(cfg, 0)
    rays, latents = make_rays(512, 37, cfg)
    ref = hip_render(scene, rays, latents, "f32", retraw=True)
    got = hip_render(scene, rays, latents, precision, retraw=True)
    SF = cfg.N_samples + cfg.N_importance
    assert got["_z_vals"].shape == (512, SF) and (got["_z_vals"][:, 1:] >= got["_z_vals"][:, :-1]).all()
    assert got["raw"].shape == ref["raw"].shape and torch.isfinite(got["raw"]).all()
    for k, bar in (("rgb0", 30.0), ("rgb_map", 30.0), ("acc0", 30.0), ("acc_map", 30.0)):
        p = psnr(got[k], ref[k])
        assert p >= bar, (k, p)
    # the depths follow the coarse weights: a 16-bit rounding of sigma moves importance samples inside their bins, never out of range
    lo, hi = float(ref["_z_vals"].min()), float(ref["_z_vals"].max())
    assert float(got["_z_vals"].min()) >= lo - 1e-6 and float(got["_z_vals"].max()) <= hi + 1e-6


@pytest.mark.parametrize("bend_depth", [5, 7])
def test_x16_bender_is_an_equally_good_rounding_of_the_bender(bend_depth):
    """bf16 mode's stand-alone bender of the split path runs on 16x16x32 MFMAs (csrc/nrnerf_bend_x16.h) by default and on the fused
    kernels' own 32x32x16 tiles with NRNERF_X16_BENDER=0: the same f16 products summed in fp32 in another order.  The bender ALONE,
    through the surface reduction of a plain render (surface_pts = the bent point at the median sample, rigidity = its mask): against
    the exact-fp32 bender's bent points AT THE SAME DEPTHS (nrnerf_bender_forward on the render's own merged depths) both benders
    must be equally far (rmse within 30 %, maximum within 3 x), on the rays whose median sample is the same in both renders; and
    the maps of the whole render are held to the same statement against the fp32-MFMA render."""
    cfg = SceneConfig(bend_depth=bend_depth)
    scene = make_scene(cfg, 0)
    rays, latents = make_rays(4096, 21, cfg)
    rb, coarse, fine = build_modules(scene, device=DEV)
    r, l = rays.to(DEV), latents.to(DEV)

    def run(precision, env):
        model = R.get_model(coarse, fine, precision=precision)
        with contextlib.ExitStack() as st:
            for k, v in env.items():
                st.enter_context(_setenv(k, v))
            with torch.no_grad():
                out = model.render(r, l, cfg.N_samples, cfg.N_importance, want_z_vals=True, surface=True)
        torch.cuda.synchronize()
        return out

    # (NRNERF_X16=1: the coarse pass on the fused-bender kernel in both, so the importance samples are the same bits and only
    #  the bender of the new samples and the fine trunk differ)
    new = run("bf16", {"NRNERF_X16": "1"})
    old = run("bf16", {"NRNERF_X16": "1", "NRNERF_X16_BENDER": "0"})
    assert torch.equal(new["_z_vals"], old["_z_vals"]) and torch.equal(new["rgb0"], old["rgb0"])
    # the reference bent points AT THESE DEPTHS: the exact-fp32 bender (nrnerf_bender_forward, the training entry point) on the merged depths
    from nonrigid_nerf_amd import training
    with torch.no_grad():
        bent_ref, det = training.bend_native(R.get_model(coarse, fine, precision="f32"), rb, r, new["_z_vals"], l)
    same = new["median_index"] == old["median_index"]
    assert same.float().mean().item() > 0.9
    idx = new["median_index"].long()[:, None, None]
    want = torch.gather(bent_ref, 1, idx.expand(-1, 1, 3))[:, 0]
    want_rig = torch.gather(det["rigidity_mask"], 1, idx.expand(-1, 1, 1))[:, 0, 0]
    e_new = (new["surface_pts"] - want)[same].norm(dim=-1)
    e_old = (old["surface_pts"] - want)[same].norm(dim=-1)
    print(f"\n[bender {bend_depth} x 64, bf16 mode] |bent point - exact fp32 bender's|: 16x16x32 rmse {e_new.pow(2).mean().sqrt().item():.2e} max {e_new.max().item():.2e}; "
          f"32x32x16 rmse {e_old.pow(2).mean().sqrt().item():.2e} max {e_old.max().item():.2e}")
    assert e_new.pow(2).mean().sqrt().item() <= 1.3 * e_old.pow(2).mean().sqrt().item() + 1e-7
    assert e_new.max().item() <= 3.0 * e_old.max().item() + 1e-6
    er_new = (new["surface_rigidity"] - want_rig)[same].abs()
    er_old = (old["surface_rigidity"] - want_rig)[same].abs()
    assert er_new.mean().item() <= 1.3 * er_old.mean().item() + 1e-6 and er_new.max().item() <= 3.0 * er_old.max().item() + 1e-4
    ref = run("f32", {})
    for k in ("rgb_map", "acc_map"):
        a_, b_ = (new[k] - ref[k]).abs(), (old[k] - ref[k]).abs()
        assert a_.mean().item() <= 1.3 * b_.mean().item() + 1e-6 and a_.max().item() <= 3.0 * b_.max().item() + 1e-4, k


def test_more_samples_than_the_library_takes_are_refused_with_the_documented_status():
    cfg = SceneConfig(N_samples=1000, N_importance=100)
    scene = make_scene(cfg, 0)
    rays, latents = make_rays(8, 37, cfg)
    rb, coarse, fine = build_modules(scene, device=DEV)
    R.set_precision("f32")
    with torch.no_grad(), pytest.raises(R.Unsupported):
        R.batchify_rays(rays.to(DEV), {"ray_bending_latents": latents.to(DEV)}, network_fn=coarse, network_fine=fine,
                        N_samples=cfg.N_samples, N_importance=cfg.N_importance)


@pytest.mark.parametrize("perturb,noise", [(1.0, 0.0), (0.0, 0.7), (1.0, 0.7)])
def test_stochastic_branches_consume_the_generator_like_the_reference(perturb, noise):
    """perturb > 0 (stratified coarse depths, random sample_pdf uniforms) and raw_noise_std > 0: the boundary draws
    the random numbers with the reference's own torch calls in the reference's order (train.py:860, 753; rnh:665), so
    after the same manual_seed the HIP path and the oracle -- whose draw order is pinned against the seeded reference
    golden on CPU -- see the same numbers on this device.  Chunked (3 chunks) so the interleaving is checked too."""
    cfg = SceneConfig(N_importance=64)
    scene = make_scene(cfg, 0)
    rays, latents = make_rays(80, 9, cfg)
    sc = O.scene_on(scene, DEV)
    torch.manual_seed(77)
    ref = O.batchify_rays(rays.to(DEV), latents.to(DEV), sc, chunk=32, retraw=True, perturb=perturb, raw_noise_std=noise)
    ref = {k: v.cpu() for k, v in ref.items()}
    torch.manual_seed(77)
    got = hip_render(scene, rays, latents, "f32", chunk=32, retraw=True, flags=dict(perturb=perturb, raw_noise_std=noise))
    if perturb > 0:
        assert (got["_z_vals"][:, 1:] >= got["_z_vals"][:, :-1]).all()
        det = hip_render(scene, rays, latents, "f32", chunk=32)
        assert (det["_z_vals"] - got["_z_vals"]).abs().max() > 1e-4, "depths were not jittered"
    fails = compare_dict(got, ref, keys=["rgb0", "disp0", "acc0"])
    moved = ((got["_z_vals"] - ref["_z_vals"]).abs() > 2e-5).float().mean().item()
    assert moved < 0.02, f"{moved:.4f} of merged depths differ from the oracle"
    fails += compare_dict(got, ref, keys=["rgb_map", "acc_map", "z_std"], frac_ok=0.10, outlier_atol=5e-2)
    assert not fails, "\n".join(fails)
    # after both runs the generator must be in the same state: nothing drawn in a different amount
    torch.manual_seed(77)
    O.batchify_rays(rays.to(DEV), latents.to(DEV), sc, chunk=32, perturb=perturb, raw_noise_std=noise)
    a = torch.rand(4, device=DEV)
    torch.manual_seed(77)
    hip_render(scene, rays, latents, "f32", chunk=32, flags=dict(perturb=perturb, raw_noise_std=noise))
    b = torch.rand(4, device=DEV)
    assert torch.equal(a, b)


@pytest.mark.parametrize("n,flags", [(1, {}), (33, dict(lindisp=True, white_bkgd=True)), (80, dict(perturb=1.0, raw_noise_std=0.7)),
                                     (257, dict(perturb=1.0))], ids=["one_ray", "lindisp_white", "stochastic", "perturb_ragged"])
@pytest.mark.parametrize("cfg_kw", [dict(N_importance=48, N_samples=40, netdepth=6, netwidth=192, netwidth_fine=320, multires=8, latent_size=16),
                                    dict(N_importance=64, netwidth=96, use_viewdirs=True, multires_views=2)], ids=["generic_192_320", "generic_viewdirs_96"])
def test_generic_kernel_edge_cases_vs_oracle(cfg_kw, n, flags):
    """The run-time-parameterised kernel under everything the boundary passes on: tiny and ragged ray counts (tiles of 32 / 64
    samples that straddle rays), inverse-depth spacing + white background, and the stochastic branches with the reference's own
    random draws (seeded: same numbers as the oracle on this device), chunked like batchify_rays."""
    cfg = SceneConfig(**cfg_kw)
    scene = make_scene(cfg, 3)
    rays, latents = make_rays(n, 19, cfg)
    sc = O.scene_on(scene, DEV)
    rflags = {k: v for k, v in flags.items() if k in ("perturb", "raw_noise_std", "lindisp", "white_bkgd")}
    torch.manual_seed(91)
    with torch.no_grad():
        ref = O.batchify_rays(rays.to(DEV), latents.to(DEV), sc, chunk=32, retraw=True, **rflags)
    ref = {k: v.cpu() for k, v in ref.items()}
    torch.manual_seed(91)
    got = hip_render(scene, rays, latents, "f32", chunk=32, retraw=True, flags=rflags)
    assert set(k for k in got if not k.startswith("_")) == set(k for k in ref if not k.startswith("_"))
    loose = dict(frac_ok=0.25, outlier_atol=0.25) if flags.get("lindisp") else dict(frac_ok=0.10, outlier_atol=5e-2)
    fails = compare_dict(got, ref, keys=["rgb0", "disp0", "acc0"])
    zg, zo = got["_z_vals"], ref["_z_vals"]
    assert (zg[:, 1:] >= zg[:, :-1]).all()
    assert ((zg - zo).abs() > 2e-5).float().mean().item() < 0.03
    fails += compare_dict(got, ref, keys=["rgb_map", "acc_map"], **loose)
    fine = oracle_fine_given_z(scene, rays, latents, zg, white_bkgd=bool(flags.get("white_bkgd", False)))
    if not flags.get("raw_noise_std"):        # (the noise on sigma is part of the compositing: with it only the end-to-end comparison above)
        if cfg.use_viewdirs:
            fails += compare_dict(got, fine, keys=["rgb_map", "disp_map", "acc_map"])
            fails += compare_dict(got, fine, keys=["raw"], **FD_DIRS_RAW)
        else:
            fails += compare_dict(got, fine, keys=["rgb_map", "disp_map", "acc_map", "raw"])
    assert not fails, "\n".join(fails)


@pytest.mark.parametrize("cfg_kw", [dict(), dict(use_viewdirs=True, N_importance=64),
                                    dict(use_viewdirs=True, N_importance=64, approx_nonrigid_viewdirs=False)],
                         ids=["default", "viewdirs", "viewdirs_exact"])
def test_repeated_launches_are_bit_identical(cfg_kw):
    """Race detector: the LDS ring hand-offs, the counted waits and the view-direction mailbox use no atomics, so the
    same launch must reproduce itself bit for bit; a race would show up as a run-to-run difference
    (tools/soak_determinism.py is the long version over every variant and precision)."""
    cfg = SceneConfig(**cfg_kw)
    scene = make_scene(cfg, 0)
    rb, coarse, fine = build_modules(scene, device=DEV)
    rays, lat = make_rays(30011, 5, cfg)
    rays, lat = rays.to(DEV), lat.to(DEV)
    R.set_precision("bf16")
    model = R.get_model(coarse, fine)
    with torch.no_grad():
        first = model.render(rays, lat, cfg.N_samples, cfg.N_importance, retraw=True)
        for _ in range(12):
            out = model.render(rays, lat, cfg.N_samples, cfg.N_importance, retraw=True)
            for k in first:
                assert torch.equal(torch.nan_to_num(out[k]), torch.nan_to_num(first[k])), k


def test_render_path_and_surface_reduction_match_reference_golden():
    """The frame driver with the in-kernel surface reduction against outputs of the REFERENCE's ``train.render_path``
    (train.py:419-553) and of the reduction free_viewpoint_rendering.py:621-658 applied to the reference's own detail
    tensors (tests/golden/render_path_2frames.npz, oracle/make_golden.py::run_render_path)."""
    import os
    from nonrigid_nerf_amd.driver import render_path
    from tests.helpers import GOLDEN_DIR, synthetic_camera
    z = np.load(os.path.join(GOLDEN_DIR, "render_path_2frames.npz"))
    cams = [synthetic_camera(k, H=8, W=12) for k in range(2)]
    poses, intrins = [c for c, _ in cams], [i for _, i in cams]
    codes = torch.from_numpy(z["in__codes"])
    cfg = SceneConfig(N_importance=64)
    scene = make_scene(cfg, 0)
    rb, coarse, fine = build_modules(scene, device=DEV)
    R.set_precision("f32")
    kw = dict(network_fn=coarse, network_fine=fine, network_query_fn=None, N_samples=64, N_importance=64, perturb=False,
              raw_noise_std=0.0, white_bkgd=False, lindisp=False, ndc=False, use_viewdirs=False, ray_bender=rb,
              near=cfg.near, far=cfg.far)
    rgbs, disps, extra = render_path([p.to(DEV) for p in poses], intrins, 32768, kw, codes.to(DEV), surface_outputs=True)
    got = {"rgb_map": torch.from_numpy(rgbs).reshape(-1, 3)}
    ref = {"rgb_map": torch.from_numpy(z["out__rgbs"]).reshape(-1, 3)}
    assert not compare_dict(got, ref, frac_ok=0.10, outlier_atol=2e-2)        # a few rays whose fine sample moved (sample_pdf branch)
    for f in range(2):
        same = torch.from_numpy(extra[f]["median_index"]).int() == torch.from_numpy(z[f"out__median_indices_{f}"])
        assert same.float().mean() >= 0.95, float(same.float().mean())
        assert torch.allclose(torch.from_numpy(extra[f]["surface_pts"])[same], torch.from_numpy(z[f"out__surface_pixels_{f}"])[same], atol=1e-4)
        assert torch.allclose(torch.from_numpy(extra[f]["surface_rigidity"])[same], torch.from_numpy(z[f"out__rigidity_{f}"])[same], atol=1e-4)


@pytest.mark.parametrize("precision", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("cfg_kw,knobs,flags", [
    (dict(), {}, {}),
    (dict(N_importance=64, bend_depth=7), {}, {}),
    (dict(N_samples=48, N_importance=37), dict(rigidity_test_time_cutoff=0.45, test_time_scaling=0.5), {}),
    (dict(N_importance=64), {}, dict(perturb=1.0, raw_noise_std=0.5)),
    (dict(N_samples=128, N_importance=128), {}, dict(lindisp=True, white_bkgd=True)),
    (dict(N_importance=64, netwidth=128), {}, {}),
    (dict(N_importance=64, use_viewdirs=True), {}, {}),
    (dict(N_samples=48, N_importance=37, use_viewdirs=True, bend_depth=7), dict(rigidity_test_time_cutoff=0.45, test_time_scaling=0.5), {}),
], ids=["headline", "deep_bender", "ragged_knobs", "stochastic", "max_samples_flags", "narrow_128", "viewdirs", "config4_ragged_knobs"])
def test_split_bender_path_equals_the_fused_fine_pass_bit_for_bit(precision, cfg_kw, knobs, flags):
    """nrnerf_render's split-bender path (coarse bent points carried over, stand-alone bender kernel for the importance
    samples, trunk-only fine kernel; nrnerf_bend.h) against the fused fine pass, which a request for per-sample detail
    outputs selects: same arithmetic, so every common output must be bit-identical -- maps, raw logits, merged depths
    and the surface reduction."""
    cfg = SceneConfig(**cfg_kw)
    scene = make_scene(cfg, 2)
    rays, latents = make_rays(3001, 23, cfg)
    rb, coarse, fine = build_modules(scene, device=DEV)
    rb.rigidity_test_time_cutoff = knobs.get("rigidity_test_time_cutoff")
    rb.test_time_scaling = knobs.get("test_time_scaling")
    R.set_precision(precision)
    model = R.get_model(coarse, fine)
    r, l = rays.to(DEV), latents.to(DEV)
    lind, wb = bool(flags.get("lindisp")), bool(flags.get("white_bkgd"))

    def run(mdl, detailed):
        torch.manual_seed(5)
        randoms = R._draw_randoms(r, cfg.N_samples, cfg.N_importance, flags.get("perturb", 0.0), flags.get("raw_noise_std", 0.0))
        with torch.no_grad():
            out = mdl.render(r, l, cfg.N_samples, cfg.N_importance, retraw=True, detailed_output=detailed,
                             rigidity_cutoff=rb.rigidity_test_time_cutoff, test_time_scaling=rb.test_time_scaling,
                             want_z_vals=True, surface=True, lindisp=lind, white_bkgd=wb, randoms=randoms)
        torch.cuda.synchronize()
        return out

    with _same_bender(precision), _setenv("NRNERF_X16", "1"):      # (the fine pass on the 16x16x32 kernel; 2, the default, moves the coarse pass there as well)
        split16 = run(model, False)
    fused = run(model, True)
    assert "fine_input_pts" in fused and "fine_input_pts" not in split16
    # the trunk-only pass on the fused pass's own 32x32x16 tiles (NRNERF_X16=0) is what the bit-for-bit statement is about; the default
    # trunk-only pass of the 16-bit modes (16x16x32 kernel; width 256, no view branch) is as good a rounding of the fp32 network
    with _setenv("NRNERF_X16", "0"):
        split = run(model, False)
    if precision != "f32":
        ref32 = run(R.get_model(coarse, fine, precision="f32"), False)
        _assert_x16_is_an_equally_good_rounding(split16, split, ref32)
        if cfg.N_importance > 0:
            with _same_bender(precision):
                _assert_x16_coarse_is_an_equally_good_rounding(run(model, False), split, ref32)
    if precision == "bf16":
        _assert_split_equals_fused_up_to_conversion_ties(split, fused, views=cfg.use_viewdirs)
        return
    if cfg.use_viewdirs:
        # View-dependent head: the trunk-only kernel takes a sample's direction from the same bent points (read back from
        # the array instead of shuffled between lanes), but it is another template instantiation and hipcc rounds the
        # direction encoding's fp32 arithmetic differently in the two (the same effect as between the one- and the
        # two-blocks-per-wave kernels): sigma, depths and the coarse maps stay bit-identical, colour
        # logits differ by an ulp on a few per cent of the samples (measured, tools/experiments/debug_split_views.py: f32 1.7 % of the
        # samples by <= 2.4e-7, rgb_map <= 6e-8; f16 0.1 % by <= 1.1e-4, rgb_map <= 1.2e-6).
        for k in ("rgb0", "disp0", "acc0", "z_std", "_z_vals", "disp_map", "acc_map", "median_index", "surface_pts", "surface_rigidity"):
            assert torch.equal(torch.nan_to_num(split[k].float()), torch.nan_to_num(fused[k].float())), k
        assert torch.equal(split["raw"][..., 3], fused["raw"][..., 3])
        d = (split["raw"][..., :3] - fused["raw"][..., :3]).abs()
        tol, frac = (1e-6, 0.04) if precision == "f32" else (5e-4, 0.01)
        assert d.max().item() <= tol and (d.amax(-1) > 0).float().mean().item() <= frac, (d.max().item(), (d.amax(-1) > 0).float().mean().item())
        assert (split["rgb_map"] - fused["rgb_map"]).abs().max().item() <= tol
        return
    for k in split:
        assert torch.equal(torch.nan_to_num(split[k].float()), torch.nan_to_num(fused[k].float())), k
    # and the carried-over / separately bent points are the fused kernel's: the surface point is one of them
    idx = split["median_index"].long()
    assert torch.equal(split["surface_pts"], fused["fine_input_pts"][torch.arange(3001, device=DEV), idx])


@pytest.mark.parametrize("precision", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("cfg_kw,knobs,flags", [
    (dict(), {}, {}),                                                                   # headline: 64 + 128, split-bender path
    (dict(N_samples=48, N_importance=37), dict(rigidity_test_time_cutoff=0.45), {}),    # ragged: 85 samples = 3 blocks per ray (odd)
    (dict(N_samples=33, N_importance=64), {}, dict(perturb=1.0, raw_noise_std=1.0)),    # 97 samples = 4 blocks, noise on sigma
    (dict(N_samples=128, N_importance=128), {}, dict(white_bkgd=True)),                 # 256 samples: 8 blocks, 4 samples per lane
    (dict(N_samples=64, N_importance=96, ray_bending=False), {}, {}),                   # no bender: 160 samples = 5 blocks (odd)
    (dict(N_importance=0, ray_bending=False), {}, {}),                                  # no bender, coarse only: K1 inside K0
    (dict(N_importance=64, use_viewdirs=True), {}, {}),                                 # view-dependent head, split path
    (dict(N_importance=64, use_viewdirs=True, ray_bending=False), {}, {}),              # view-dependent head on ray directions
    (dict(N_importance=64, netwidth=128), {}, {}),                                      # width 128: one block per wave in every mode
    (dict(N_importance=32, ray_bending=False, time_conditioned_baseline=True), {}, {}), # time-conditioned baseline
], ids=["headline", "ragged_knobs", "noise", "max_samples_white", "no_bender_odd", "no_bender_coarse_only", "viewdirs", "viewdirs_no_bender",
        "narrow_128", "time_conditioned"])
def test_compositing_fused_into_the_network_kernel_equals_the_composite_kernel_bit_for_bit(precision, cfg_kw, knobs, flags):
    """north_star: "compositing fused into the ray loop".  The final pass' network kernel (every variant without a fused
    bender: the trunk-only fine pass of the split-bender path, any pass of a model without bender) keeps each ray's raw
    outputs in LDS and composites them itself with the composite kernel's own code (nrnerf_composite_ray.h); the route with
    the separate composite launch (NRNERF_UNFUSED_COMPOSITE=1: raw [N,S,4] through HBM) must give the same bits for every
    output -- maps, raw logits, depths, surface reduction -- in every precision, for even and odd block counts per ray, a
    ray count that leaves the last group of rays half empty, noise, white background."""
    import os
    cfg = SceneConfig(**cfg_kw)
    scene = make_scene(cfg, 4)
    n = 3001
    rays, latents = make_rays(n, 29, cfg)
    rb, coarse, fine = build_modules(scene, device=DEV)
    if rb is not None:
        rb.rigidity_test_time_cutoff = knobs.get("rigidity_test_time_cutoff")
    R.set_precision(precision)
    model = R.get_model(coarse, fine)
    r, l = rays.to(DEV), latents.to(DEV)
    outs = []
    old = os.environ.get("NRNERF_UNFUSED_COMPOSITE")
    try:
        for unfused in ("0", "1"):
            os.environ["NRNERF_UNFUSED_COMPOSITE"] = unfused
            torch.manual_seed(5)
            randoms = R._draw_randoms(r, cfg.N_samples, cfg.N_importance, flags.get("perturb", 0.0), flags.get("raw_noise_std", 0.0))
            with torch.no_grad():
                outs.append(model.render(r, l, cfg.N_samples, cfg.N_importance, retraw=True, detailed_output=False,
                                         rigidity_cutoff=knobs.get("rigidity_test_time_cutoff"), test_time_scaling=None,
                                         want_z_vals=True, surface=True, white_bkgd=bool(flags.get("white_bkgd")), randoms=randoms))
            torch.cuda.synchronize()
    finally:
        if old is None:
            os.environ.pop("NRNERF_UNFUSED_COMPOSITE", None)
        else:
            os.environ["NRNERF_UNFUSED_COMPOSITE"] = old
    fused, unfused = outs
    assert set(fused) == set(unfused)
    for k in fused:
        assert torch.equal(torch.nan_to_num(fused[k].float()), torch.nan_to_num(unfused[k].float())), k
    assert torch.isfinite(fused["rgb_map"]).all() and float(fused["acc_map"].max()) > 0.5


def _assert_split_equals_fused_up_to_conversion_ties(split, fused, views=False):
    """bf16 mode (single-product f16 bender): the stand-alone bender kernel and the fused kernel are the same arithmetic,
    but hipcc pairs the f32 -> f16 conversions of the first-layer inputs differently in the two kernels
    (v_cvt_pk_f16_f32 vs v_cvt_f16_f32), and the two instructions disagree on rare inputs -- measured: 21 of 384 126 new
    samples (5e-5, the rate of round-to-nearest ties) get a bent point that differs by <= 5e-6 (tools/experiments/debug_split_vs_fused.py).
    Everything that does not pass through a new sample's bender is still bit-identical."""
    for k in ("rgb0", "disp0", "acc0", "z_std", "_z_vals"):
        assert torch.equal(torch.nan_to_num(split[k]), torch.nan_to_num(fused[k])), k
    differs = (split["raw"] != fused["raw"]).any(-1)
    # (view-dependent head: on top of that, the one-ulp flips of the f16-rounded direction encoding between the two template
    #  instantiations -- 0.1 % of the samples in f16 mode, see the f16 / f32 branch of the caller -- now reach the views layer's
    #  pre-activation directly, feature_linear being folded into it: 5.6e-4 of the samples measured)
    assert differs.float().mean().item() < (2e-3 if views else 5e-4), differs.float().mean().item()
    assert (split["rgb_map"] - fused["rgb_map"]).abs().max().item() < 2e-3
    assert (split["rgb_map"] != fused["rgb_map"]).any(-1).float().mean().item() < 0.05
    same_idx = split["median_index"] == fused["median_index"]
    assert same_idx.float().mean().item() > 0.99
    n = split["surface_pts"].shape[0]
    ref = fused["fine_input_pts"][torch.arange(n, device=split["surface_pts"].device), split["median_index"].long()]
    assert (split["surface_pts"] - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("cfg_kw", [dict(N_importance=64, use_viewdirs=True), dict(N_importance=64, bend_depth=7),
                                    dict(N_importance=64, ray_bending=False, time_conditioned_baseline=True),
                                    dict(N_importance=0, ray_bending=False), dict(N_importance=64, use_viewdirs=True, approx_nonrigid_viewdirs=False)],
                         ids=["viewdirs", "deep_bender", "time_conditioned", "coarse_only_no_bender", "exact_viewdirs"])
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_device_side_weight_refresh_equals_a_fresh_pack(cfg_kw, precision):
    """nrnerf_model_update_device on every architecture family: after in-place weight changes the cached handle -- refreshed
    on the device from the flat parameter vector, in the library's canonical order -- must render exactly what a handle
    packed on the host from the modified modules renders (every packed image: fused passes, split-bender images, heads)."""
    cfg = SceneConfig(**cfg_kw)
    scene = make_scene(cfg, 0)
    rb, coarse, fine = build_modules(scene, device=DEV)
    rays, latents = make_rays(700, 5, cfg)
    rays, latents = rays.to(DEV), latents.to(DEV)
    R.set_precision(precision)
    I = cfg.N_importance
    with torch.no_grad():
        m0 = R.get_model(coarse, fine if I > 0 else None, device=DEV)
        before = m0.render(rays, latents, 64, I, retraw=True)["rgb_map"].clone()
        for mod in (rb, coarse, fine):
            if mod is not None:
                for p_ in mod.parameters():
                    p_.mul_(1.03).add_(0.001)
        used = {"dev": 0}
        orig = R.Model.update_from_device

        def counting(self, *a, **k):
            ok = orig(self, *a, **k)
            used["dev"] += int(ok)
            return ok

        R.Model.update_from_device = counting
        try:
            m1 = R.get_model(coarse, fine if I > 0 else None, device=DEV)
        finally:
            R.Model.update_from_device = orig
        after = m1.render(rays, latents, 64, I, retraw=True)
        fresh = R.Model(coarse, fine if I > 0 else None, precision, DEV).render(rays, latents, 64, I, retraw=True)
    torch.cuda.synchronize()
    assert m1 is m0 and used["dev"] == 1, "the handle must be refreshed in place, on the device"
    assert (after["rgb_map"] - before).abs().max() > 1e-4
    for k in fresh:
        a_, f_ = torch.nan_to_num(after[k]).float(), torch.nan_to_num(fresh[k]).float()
        if cfg.use_viewdirs:
            # view-dependent head: the views layer is packed with feature_linear folded in (nrnerf_plan.h); the host packer forms
            # W_v1 W_f in fp64 and rounds once, the device-side refresh takes it from an fp32 GEMM -- the same weights to an ulp
            # (bf16 mode: a packed weight can land on the other side of a rounding boundary), not bit for bit
            tol = 2e-5 if precision == "f32" else 2e-2
            assert float((a_ - f_).abs().max()) <= tol * max(1.0, float(f_.abs().max())), (k, float((a_ - f_).abs().max()))
        else:
            assert torch.equal(a_, f_), k


@pytest.mark.parametrize("n_rays,cfg_kw,stochastic", [(1024, {}, False), (128, dict(use_viewdirs=True), False), (777, dict(ray_bending=False), False),
                                                      (1024, {}, True)],
                         ids=["1024_default", "128_viewdirs", "777_no_bender", "1024_stochastic"])
def test_graphed_render_equals_the_eager_call(n_rays, cfg_kw, stochastic):
    """render.GraphedRender (small batches from a HIP graph): a replay returns what the eager call returns
    -- bit for bit, every output key -- for new rays and new latent codes copied into its static buffers; after an in-place
    weight change (an optimiser step bumps the parameters' versions) the next replay renders with the NEW weights (refreshed
    into the buffers the graph reads); a stochastic call draws fresh numbers on every replay and consumes torch's generator
    like the eager call; other shapes are refused."""
    cfg = SceneConfig(N_importance=64, **cfg_kw)
    scene = make_scene(cfg, 3)
    rb, coarse, fine = build_modules(scene, device=DEV)
    R.set_precision("bf16")
    kw = dict(network_fine=fine, N_samples=cfg.N_samples, N_importance=cfg.N_importance, perturb=1.0 if stochastic else 0.0,
              raw_noise_std=0.5 if stochastic else 0.0, retraw=True)
    rays0, lat0 = make_rays(n_rays, 1, cfg)
    rays1, lat1 = make_rays(n_rays, 2, cfg)
    rays0, lat0, rays1, lat1 = (t.to(DEV) for t in (rays0, lat0, rays1, lat1))
    needs_lat = cfg.ray_bending or cfg.time_conditioned_baseline
    g = R.GraphedRender(rays0, coarse, latents=lat0 if needs_lat else None, **kw)

    def eager(rays, lat):
        with torch.no_grad():
            out = R.render_rays(rays, coarse, additional_pixel_information={"ray_bending_latents": lat} if needs_lat else None, **kw)
        return {k: v.clone() for k, v in out.items()}

    def same(a, b):
        assert set(a) == set(b)
        for k in a:
            assert torch.equal(a[k], b[k]), k

    if stochastic:
        torch.manual_seed(5)
        first = {k: v.clone() for k, v in g(rays1, lat1).items()}
        second = {k: v.clone() for k, v in g(rays1, lat1).items()}
        assert not torch.equal(first["rgb_map"], second["rgb_map"]), "every replay must draw new random numbers"
        torch.manual_seed(5)
        same(first, eager(rays1, lat1))                      # same generator state -> the eager call's numbers
        return
    same(g(rays1, lat1 if needs_lat else None), eager(rays1, lat1))
    same(g(rays0, lat0 if needs_lat else None), eager(rays0, lat0))
    before = g(rays1, lat1 if needs_lat else None)["rgb_map"].clone()
    with torch.no_grad():
        fine.pts_linears[3].weight.mul_(1.1)                 # in-place: bumps _version, as an optimiser step does
    after = g(rays1, lat1 if needs_lat else None)
    assert not torch.equal(after["rgb_map"], before)
    same(after, eager(rays1, lat1))
    with pytest.raises(ValueError):
        g(rays1[:-1], lat1[:-1] if needs_lat else None)


@pytest.mark.parametrize("precision", ["bf16", "f16"])
@pytest.mark.parametrize("cfg_kw,flags,n", [
    (dict(), {}, 3001),                                                                  # headline: 64 + 128; last group of rays half empty
    (dict(N_samples=48, N_importance=37), {}, 4200),                                     # ragged: 3 blocks per ray -> groups of 4 rays per wave
    (dict(N_samples=33, N_importance=64), dict(perturb=1.0, raw_noise_std=1.0), 4500),   # stochastic: caller's uniforms + noise on sigma
    (dict(N_samples=128, N_importance=128), dict(white_bkgd=True), 2200),                # 128 coarse samples: two per lane, 256 merged
    (dict(N_importance=64, use_viewdirs=True), {}, 2100),                                # view-dependent head behind the coarse trunk
    (dict(N_importance=64, netwidth=128), {}, 4100),                                     # width 128: eight waves per workgroup
], ids=["headline", "ragged", "stochastic", "coarse_128_white", "viewdirs", "narrow_128"])
def test_coarse_epilogue_inside_the_trunk_kernel_equals_the_composite_launch_bit_for_bit(precision, cfg_kw, flags, n):
    """north_star: "compositing fused into the ray loop" -- the COARSE pass of a hierarchical render (row g1).  With
    NRNERF_FUSED_COARSE_EPILOGUE=1 (nrnerf_render_args.flags: NRNERF_RENDER_COARSE_EPILOGUE_ON) the coarse trunk kernel of the split path
    composites each of its rays, draws the importance depths (sample_pdf), computes z_std and merges the depths as its epilogue
    (train.py:889-920; sample_merge_ray: composite_kernel<EPL, true>'s own code), raw_c never reaching HBM; =0 keeps composite_kernel's own
    launch.  Every output must have the same bits either way: the coarse maps, z_std, the merged depths, and -- through the new samples'
    list and the carried-over bent points -- everything the fine pass makes of them."""
    cfg = SceneConfig(**cfg_kw)
    scene = make_scene(cfg, 4)
    rays, latents = make_rays(n, 31, cfg)
    rb, coarse, fine = build_modules(scene, device=DEV)
    R.set_precision(precision)
    model = R.get_model(coarse, fine)
    r, l = rays.to(DEV), latents.to(DEV)
    outs, kernels = [], []
    for on in ("1", "0"):
        with _setenv("NRNERF_FUSED_COARSE_EPILOGUE", on):
            torch.manual_seed(5)
            randoms = R._draw_randoms(r, cfg.N_samples, cfg.N_importance, flags.get("perturb", 0.0), flags.get("raw_noise_std", 0.0))
            model.profile_begin()
            with torch.no_grad():
                outs.append(model.render(r, l, cfg.N_samples, cfg.N_importance, retraw=True, want_z_vals=True, surface=True,
                                         white_bkgd=bool(flags.get("white_bkgd")), randoms=randoms))
            torch.cuda.synchronize()
            kernels.append(model.profile_end())
    fused, separate = outs
    # the routes really differ: the fused one launched no coarse composite kernel, and the library says so
    assert kernels[0]["composite_sample_coarse"]["launches"] == 0 and "sample_pdf" in kernels[0]["net_coarse"]["kernel"], kernels[0]
    assert kernels[1]["composite_sample_coarse"]["launches"] == 1 and kernels[1]["net_coarse"]["kernel"] == "net_kernel_x16", kernels[1]
    assert set(fused) == set(separate)
    for k in fused:
        assert torch.equal(torch.nan_to_num(fused[k].float()), torch.nan_to_num(separate[k].float())), k
    assert torch.isfinite(fused["rgb_map"]).all() and float(fused["acc_map"].max()) > 0.5 and float(fused["z_std"].max()) > 0


@pytest.mark.parametrize("precision", ["bf16", "f16"])
@pytest.mark.parametrize("cfg_kw,n,one_code,env", [
    (dict(), 70001, True, {}),                                            # headline shape, one frame code: many more groups than workgroups
    (dict(N_samples=48, N_importance=37), 3001, False, {}),               # ragged counts (groups of four rays per wave), a code per ray
    (dict(N_importance=64, bend_depth=7), 257, False, {}),                # the 7-layer bender; fewer groups than workgroups: most get none
    (dict(N_importance=128, netdepth=6, netwidth=192), 5000, True, {}),   # a non-compiled trunk: the generic route's bender passes
    (dict(N_importance=64, netwidth=128), 20011, False, {}),              # width 128: eight waves per workgroup
    (dict(N_importance=64, use_viewdirs=True), 9001, True, {}),           # the view-dependent head behind both trunks
    (dict(), 9001, True, {"NRNERF_FUSED_COARSE_EPILOGUE": "1"}),          # the coarse pass with its epilogue: a group per iteration, fused
    (dict(), 9001, True, {"NRNERF_UNFUSED_COMPOSITE": "1"}),              # both passes write raw rows: a group per iteration, not fused
    (dict(N_samples=128, N_importance=128), 6001, True, {}),              # two iterations per group in the coarse pass' numbering
], ids=["headline_frame_code", "ragged_per_ray_codes", "deep_bender_tiny", "generic_w192", "narrow_128", "viewdirs", "coarse_epilogue",
        "unfused_composite", "128_plus_128"])
def test_dynamic_shares_of_the_work_equal_fixed_shares_bit_for_bit(precision, cfg_kw, n, one_code, env):
    """Round 6: the waves of the 16x16x32 stand-alone bender and the workgroups of the 16x16x32 trunk kernels take their next piece of work
    from device counters (BendArgs / NetArgs::work_counter, zeroed by a memset node of the call) instead of owning a fixed share;
    NRNERF_FIXED_SHARES=1 (nrnerf_render_args.flags: NRNERF_RENDER_FIXED_SHARES) keeps the fixed shares.  Which wave evaluates a sample
    must not matter: every output the same bits, on repeated calls too (the counters are re-armed per call)."""
    import contextlib
    cfg = SceneConfig(**cfg_kw)
    scene = make_scene(cfg, 4)
    rays, latents = make_rays(n, 31, cfg)
    rb, coarse, fine = build_modules(scene, device=DEV)
    R.set_precision(precision)
    model = R.get_model(coarse, fine)
    r = rays.to(DEV)
    l = latents[:1].to(DEV).expand(n, -1) if one_code else latents.to(DEV)
    outs = []
    with contextlib.ExitStack() as stack:
        for k, v in env.items():
            stack.enter_context(_setenv(k, v))
        for fixed in ("1", "0", "0"):
            with _setenv("NRNERF_FIXED_SHARES", fixed):
                with torch.no_grad():
                    outs.append({k: v.clone() for k, v in model.render(r, l, cfg.N_samples, cfg.N_importance, want_z_vals=True, surface=True).items()})
    torch.cuda.synchronize()
    for other in outs[1:]:
        assert set(other) == set(outs[0])
        for k in other:
            assert torch.equal(torch.nan_to_num(other[k].float()), torch.nan_to_num(outs[0][k].float())), k
    assert torch.isfinite(outs[0]["rgb_map"]).all() and float(outs[0]["acc_map"].max()) > 0.5


@pytest.mark.parametrize("precision", ["bf16", "f16"])
@pytest.mark.parametrize("cfg_kw,flags,n", [
    (dict(N_importance=128, netdepth=6, netwidth=192, netwidth_fine=320, multires=8), {}, 3001),        # 192 merged samples: three per lane
    (dict(N_samples=48, N_importance=37, netwidth=96), dict(raw_noise_std=1.0), 2049),                  # ragged, noise on sigma
    (dict(N_samples=128, N_importance=128, netwidth=448), dict(white_bkgd=True), 1100),                 # 256 samples; two blocks per wave (width class 448)
    (dict(N_importance=0, netwidth=192), {}, 2100),                                                     # coarse only: the coarse pass is the final one
    (dict(N_importance=64, netwidth=160, use_viewdirs=True), {}, 2100),                                 # view-dependent head
], ids=["w192_320", "w96_ragged_noise", "w448_256_white", "w192_coarse_only", "w160_viewdirs"])
def test_width_class_kernel_with_fused_compositing_equals_the_composite_launch_bit_for_bit(precision, cfg_kw, flags, n):
    """Row g1 for architectures outside the compiled set: the FINAL pass of the width-class trunk kernel (gx16_kernel<.., FUSE>) composites
    its rays itself (composite_ray, the composite kernel's own code) instead of writing raw [N, S, 4] for a composite_kernel launch
    (NRNERF_UNFUSED_COMPOSITE=1 keeps that route): same bits for every output, and the library reports which kernels ran."""
    cfg = SceneConfig(**cfg_kw)
    scene = make_scene(cfg, 4)
    rays, latents = make_rays(n, 37, cfg)
    rb, coarse, fine = build_modules(scene, device=DEV)
    R.set_precision(precision)
    model = R.get_model(coarse, fine if cfg.N_importance > 0 else None)
    assert model.generic
    r, l = rays.to(DEV), latents.to(DEV)
    final = "net_fine" if cfg.N_importance > 0 else "net_coarse"
    final_composite = "composite_fine" if cfg.N_importance > 0 else "composite_sample_coarse"
    outs, kernels = [], []
    for unfused in ("0", "1"):
        with _setenv("NRNERF_UNFUSED_COMPOSITE", unfused):
            torch.manual_seed(5)
            randoms = R._draw_randoms(r, cfg.N_samples, cfg.N_importance, 0.0, flags.get("raw_noise_std", 0.0))
            model.profile_begin()
            with torch.no_grad():
                outs.append(model.render(r, l, cfg.N_samples, cfg.N_importance, retraw=True, want_z_vals=True, surface=True,
                                         white_bkgd=bool(flags.get("white_bkgd")), randoms=randoms))
            torch.cuda.synchronize()
            kernels.append(model.profile_end())
    assert kernels[0][final]["kernel"] == "gx16_kernel + fused compositing" and kernels[0][final_composite]["launches"] == 0, kernels[0]
    assert kernels[1][final]["kernel"] == "gx16_kernel" and kernels[1][final_composite]["launches"] == 1, kernels[1]
    fused, separate = outs
    assert set(fused) == set(separate)
    for k in fused:
        assert torch.equal(torch.nan_to_num(fused[k].float()), torch.nan_to_num(separate[k].float())), k
    assert torch.isfinite(fused["rgb_map"]).all() and float(fused["acc_map"].max()) > 0.5


@pytest.mark.parametrize("cfg_kw", [dict(), dict(N_importance=64, bend_depth=7), dict(N_importance=64, netdepth=6, netwidth=192, netwidth_fine=320, multires=8)],
                         ids=["headline", "deep_bender", "generic_w192_320"])
def test_f16_mode_with_the_single_product_bender_is_at_least_as_accurate_as_bf16_mode(cfg_kw):
    """Round 6 (BASELINE config 5 is an "f16" workload): the stand-alone benders of "f16" mode's split path run on the 16x16x32 single-product
    kernel by default, like "bf16" mode's -- a third of the MFMAs of the three-product bender (11.7 % of a 1080p frame in round 5).  What
    that costs in accuracy, stated against the exact-fp32 render on the synthetic stress scene: the default "f16" render is (a) closer to
    fp32 than the "bf16" render of the same call (same bender arithmetic, three more mantissa bits in the trunk), (b) within 6x of the
    three-product route's error, which stays selectable (NRNERF_X16_BENDER=0 per call, NRNERF_X16_F16=0 per handle), and (c) at least
    "bf16" mode's PSNR.  The fitted checkpoints hold it to the stated bar against the oracle and the ground truth (test_fitted_checkpoint.py)."""
    cfg = SceneConfig(**cfg_kw)
    scene = make_scene(cfg, 0)
    rays, latents = make_rays(8192, 3, cfg)
    ref32 = hip_render(scene, rays, latents, "f32")
    f16 = hip_render(scene, rays, latents, "f16")
    bf16 = hip_render(scene, rays, latents, "bf16")
    with _setenv("NRNERF_X16_BENDER", "0"):
        f16_three = hip_render(scene, rays, latents, "f16")
    assert not torch.equal(f16["rgb0"], f16_three["rgb0"]), "both f16 renders took the same bender"
    for k in ("rgb0", "rgb_map"):
        e = {n: (o[k].float() - ref32[k].float()).abs().mean().item() for n, o in (("f16", f16), ("bf16", bf16), ("f16_three_product", f16_three))}
        print(f"[{k}] mean |error| vs the fp32 kernels: {e}; PSNR f16 {psnr(f16[k], ref32[k]):.1f} dB, bf16 {psnr(bf16[k], ref32[k]):.1f} dB, "
              f"f16 three-product {psnr(f16_three[k], ref32[k]):.1f} dB")
        assert e["f16"] <= e["bf16"], (k, e)
        assert e["f16"] <= 6.0 * e["f16_three_product"] + 1e-6, (k, e)
    # (no absolute bar here: this is the numerical stress scene -- sigma logits ~ N(-2, 6^2) -- on which "bf16" mode sits at 34-37 dB; the
    #  stated >= 40 dB / <= 0.1 dB bar is held on the fitted checkpoints, tests/test_fitted_checkpoint.py)
    assert psnr(f16["rgb0"], ref32["rgb0"]) >= psnr(bf16["rgb0"], ref32["rgb0"])
