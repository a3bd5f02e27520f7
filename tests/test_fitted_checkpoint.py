"""Accuracy on a realistic model: NR-NeRF fitted to the down-sampled example sequence (tests/golden/fitted_latest.tar,
written by oracle/fit_checkpoint.py in the reference's latest.tar layout and read back through
nonrigid_nerf_amd.checkpoint.load_checkpoint).

BASELINE.md section 3 / BASELINE.json north_star: 16-bit modes need PSNR(ours, reference render) >= 40 dB and a PSNR
against ground truth within 0.1 dB of the reference's.  Both are measured here on ALL rays (no exclusions), with the
PSNR definition of free_viewpoint_rendering.py:821-828, the fp32 oracle standing in for the reference render (it is
pinned to the reference's outputs in tests/test_oracle_golden.py) and running on the GPU as eager PyTorch ops."""
import os

import numpy as np
import pytest
import torch

from nonrigid_nerf_amd import render as R
from nonrigid_nerf_amd.checkpoint import load_checkpoint
from nonrigid_nerf_amd.synthetic import Scene, SceneConfig
from tests.helpers import GOLDEN_DIR, psnr

CKPT = os.path.join(GOLDEN_DIR, "fitted_latest.tar")
# One fitted checkpoint per compiled architecture FAMILY (oracle/fit_checkpoint.py --arch ...): the stated bar holds on each.
#   default  reference defaults;  config4  BASELINE config 4 (view-dependent head with finite-difference directions,
#   7-layer bender);  w128  --netwidth 128 --netwidth_fine 128
FAMILIES = {
    "default": ("fitted_latest.tar", dict()),
    "config4": ("fitted_config4.tar", dict(use_viewdirs=True, bend_depth=7)),
    "w128": ("fitted_w128.tar", dict(netwidth=128)),
    # NOT a compiled shape (coarse 192 wide, fine 320): rendered by the run-time-parameterised kernel (csrc/nrnerf_generic.h)
    "w192_320": ("fitted_w192_320.tar", dict(netwidth=192, netwidth_fine=320)),
}
COMPILED = ("default", "config4", "w128")
# (family, route): every family on the kernels the library picks for it, and the three compiled families through the GENERIC kernel as
# well (NRNERF_FORCE_GENERIC=1 -> nrnerf_model_desc.flags): the generic kernel is held to the same stated bar, against the oracle
ROUTES = [(f, "default") for f in FAMILIES] + [(f, "generic") for f in COMPILED]
FIXTURE = os.path.join(GOLDEN_DIR, "example_sequence_96x72.npz")
DEV = "cuda:0"


def _load(family="default"):
    fname, arch = FAMILIES[family]
    path = os.path.join(GOLDEN_DIR, fname)
    # all four families' checkpoints are committed: a dropped or renamed one must fail the accuracy gate, not skip it
    assert os.path.exists(path), f"{path} missing: python oracle/fit_checkpoint.py --arch {family} on a GPU box and commit its output"
    ck = load_checkpoint(path, N_samples=64, N_importance=128)
    z = np.load(FIXTURE)
    near, far = float(z["bds"].min()) * 0.9, float(z["bds"].max())
    cfg = SceneConfig(near=near, far=far, **arch)
    sd = lambda m: {k: v.detach().clone() for k, v in m.state_dict().items()}
    scene = Scene(cfg, sd(ck.ray_bender), sd(ck.network_fn), sd(ck.network_fine))
    return ck, z, cfg, scene


def _intrin(z, width):
    s = width / float(z["hwf"][1])
    h = int(round(float(z["hwf"][0]) * s))
    return dict(height=h, width=width, focal_x=float(z["hwf"][2]) * s, focal_y=float(z["hwf"][2]) * s,
                center_x=width / 2, center_y=h / 2)


def test_checkpoint_fixture_is_a_reference_layout_checkpoint():
    """CPU tier: the committed file has the reference's keys (train.py:1680-1698) and the default architecture."""
    raw = torch.load(CKPT, map_location="cpu", weights_only=False)
    for k in ("global_step", "network_fn_state_dict", "network_fine_state_dict", "ray_bender_state_dict",
              "ray_bending_latent_codes", "intrinsics", "scripts_dict", "dataset_extras"):
        assert k in raw, k
    ck, z, cfg, scene = _load()
    assert ck.arch["W"] == 256 and ck.arch["D"] == 8 and ck.arch["bender"]["depth"] == 5
    assert ck.latents.shape == (z["images"].shape[0], 32) and ck.global_step >= 1000
    # trained, not initialised: the bender's last layer starts at zero (run_nerf_helpers.py:452-455)
    assert float(ck.ray_bender.network[-1].weight.abs().max()) > 1e-4


@pytest.mark.parametrize("family", ["config4", "w128", "w192_320"])
def test_the_other_families_have_fitted_checkpoints_of_their_architecture(family):
    """CPU tier: the committed files of the other families are reference-layout checkpoints of THAT architecture."""
    ck, z, cfg, scene = _load(family)
    assert ck.arch["W"] == cfg.netwidth and ck.arch["bender"]["depth"] == cfg.bend_depth
    assert int(ck.network_fine.W) == (cfg.netwidth_fine or cfg.netwidth)
    assert bool(ck.arch["use_viewdirs"]) == cfg.use_viewdirs and ck.arch["input_ch_views"] == cfg.input_ch_views
    assert ck.global_step >= 1000 and float(ck.ray_bender.network[-1].weight.abs().max()) > 1e-4


def _render_frame(ck, z, cfg, frame, width, precision, route="default"):
    from nonrigid_nerf_amd.driver import generate_rays
    rays = generate_rays(torch.from_numpy(z["poses"][frame]), _intrin(z, width), cfg.near, cfg.far, cfg.use_viewdirs, DEV)
    code = ck.latents[frame].to(DEV).reshape(1, -1)
    R.set_precision(precision)
    old = os.environ.get("NRNERF_FORCE_GENERIC")
    if route == "generic":
        os.environ["NRNERF_FORCE_GENERIC"] = "1"          # -> nrnerf_model_desc.flags & NRNERF_MODEL_FORCE_GENERIC (its own cached handle)
    try:
        with torch.no_grad():
            out = R.batchify_rays(rays, {"ray_bending_latents": code.expand(rays.shape[0], -1)}, network_fn=ck.network_fn,
                                  network_fine=ck.network_fine, N_samples=64, N_importance=128)
        torch.cuda.synchronize()
        # which kernels rendered it: asked of the handle, not inferred
        is_generic = R.get_model(ck.network_fn, ck.network_fine, device=torch.device(DEV)).generic
        assert is_generic == (route == "generic" or cfg.netwidth not in (128, 256) or (cfg.netwidth_fine or cfg.netwidth) != cfg.netwidth)
    finally:
        if route == "generic":
            if old is None:
                os.environ.pop("NRNERF_FORCE_GENERIC", None)
            else:
                os.environ["NRNERF_FORCE_GENERIC"] = old
    return rays, code, out


# regression guards at the measured level (PSNR vs the fp32 oracle, dB): family -> precision -> (rgb_map, rgb0)
GUARDS = {
    "default": {"bf16": (58.0, 62.0), "f16": (66.0, 75.0)},       # round 2: bf16 65.8 / 70.7, f16 74.0 / 85.3
    "config4": {"bf16": (60.0, 66.0), "f16": (70.0, 82.0)},       # round 4: bf16 67.9 / 74.4, f16 77.8 / 91.0
    "w128": {"bf16": (59.0, 64.0), "f16": (68.0, 78.0)},          # round 4: bf16 66.5 / 71.8, f16 76.2 / 86.4
}


@pytest.mark.gpu
@pytest.mark.parametrize("family,route", ROUTES, ids=[f"{f}-{r}" for f, r in ROUTES])
def test_full_frame_psnr_vs_oracle_all_rays(family, route):
    """One full 512x384 frame (196 608 rays, 64+128): every precision of the HIP path against the fp32 oracle render of
    the same rays and weights, for every compiled architecture family, for a fitted NON-compiled shape (coarse 192 / fine 320 wide),
    and for the compiled families sent through the generic kernel.  The bar of the 16-bit modes is the stated one
    (>= 40 dB), on ALL rays, for the final map and the coarse one."""
    from oracle import nrnerf_oracle as O
    ck, z, cfg, scene = _load(family)
    frame = 3
    rays, code, _ = _render_frame(ck, z, cfg, frame, 512, "f32", route)
    assert rays.shape[0] == 196608
    with torch.no_grad():
        ref = O.batchify_rays(rays, code.expand(rays.shape[0], -1).contiguous(), O.scene_on(scene, DEV), chunk=16384)
    res = {}
    for prec in ("f32", "bf16", "f16"):
        _, _, got = _render_frame(ck, z, cfg, frame, 512, prec, route)
        res[prec] = {k: psnr(got[k].cpu(), ref[k].cpu()) for k in ("rgb_map", "rgb0", "acc_map")}
        d, dr = got["disp_map"].cpu(), ref["disp_map"].cpu()
        ok = torch.isfinite(d) & torch.isfinite(dr)
        res[prec]["disp_rel"] = float(((d - dr).abs() / dr.abs().clamp_min(1e-6))[ok].median())
    print(f"\n[fitted checkpoint '{family}', {route} kernels, 512x384, all rays] PSNR vs fp32 oracle: " + "; ".join(
        f"{p}: rgb {r['rgb_map']:.1f} dB, rgb0 {r['rgb0']:.1f} dB, acc {r['acc_map']:.1f} dB, median rel disp err {r['disp_rel']:.1e}"
        for p, r in res.items()))
    assert res["f32"]["rgb0"] >= 80.0 and res["f32"]["rgb_map"] >= 55.0, res["f32"]     # fine pass: a few moved samples (rnh:694)
    for prec in ("bf16", "f16"):
        assert res[prec]["rgb_map"] >= 40.0 and res[prec]["rgb0"] >= 40.0, (prec, res[prec])     # the stated bar
        if route == "default" and family in GUARDS:
            lo_map, lo_0 = GUARDS[family][prec]
            assert res[prec]["rgb_map"] >= lo_map and res[prec]["rgb0"] >= lo_0, (family, prec, res[prec])


@pytest.mark.gpu
@pytest.mark.parametrize("family,route", ROUTES, ids=[f"{f}-{r}" for f, r in ROUTES])
def test_psnr_vs_ground_truth_within_a_tenth_of_a_db(family, route):
    """Held-out frame and two training frames at the fixture's resolution: PSNR against the ground-truth image for the
    fp32 oracle (the reference render) and for every precision of the HIP path, final and coarse maps; north_star: within
    0.1 dB.  The checkpoint itself must reproduce the sequence (>= 25 dB on every frame)."""
    from oracle import nrnerf_oracle as O
    ck, z, cfg, scene = _load(family)
    W = int(z["hwf"][1])
    rows = []
    for frame in (int(z["i_test"]), 0, 30):
        gt = torch.from_numpy(z["images"][frame]).float().reshape(-1, 3) / 255.0
        rays, code, _ = _render_frame(ck, z, cfg, frame, W, "f32", route)
        with torch.no_grad():
            ref = O.batchify_rays(rays, code.expand(rays.shape[0], -1).contiguous(), O.scene_on(scene, DEV), chunk=8192)
        row = {"frame": frame, "oracle": psnr(ref["rgb_map"].cpu(), gt), "oracle0": psnr(ref["rgb0"].cpu(), gt)}
        for prec in ("f32", "bf16", "f16"):
            _, _, got = _render_frame(ck, z, cfg, frame, W, prec, route)
            row[prec] = psnr(got["rgb_map"].cpu(), gt)
            row[prec + "_0"] = psnr(got["rgb0"].cpu(), gt)
        rows.append(row)
    print(f"\n[fitted checkpoint '{family}', {route} kernels] PSNR vs ground truth (rgb_map / rgb0): " + "; ".join(
        f"frame {r['frame']}: oracle {r['oracle']:.3f} / {r['oracle0']:.3f}, f32 {r['f32']:.3f} / {r['f32_0']:.3f}, "
        f"bf16 {r['bf16']:.3f} / {r['bf16_0']:.3f}, f16 {r['f16']:.3f} / {r['f16_0']:.3f} dB" for r in rows))
    for r in rows:
        assert r["oracle"] >= 25.0, "the checkpoint does not reproduce the sequence"
        for prec in ("f32", "bf16", "f16"):
            assert abs(r[prec] - r["oracle"]) <= 0.1, (r["frame"], prec, r[prec], r["oracle"])
            assert abs(r[prec + "_0"] - r["oracle0"]) <= 0.1, (r["frame"], prec, "rgb0", r[prec + "_0"], r["oracle0"])


@pytest.mark.gpu
def test_config5_chunk_f16_vs_oracle():
    """BASELINE config 5 shape: one 65 536-ray chunk with f16 weights against the fp32 oracle (all rays)."""
    from oracle import nrnerf_oracle as O
    ck, z, cfg, scene = _load()
    rays, code, _ = _render_frame(ck, z, cfg, 7, 512, "f32")
    rays = rays[:65536]
    lat = code.expand(65536, -1)
    with torch.no_grad():
        ref = O.batchify_rays(rays, lat.contiguous(), O.scene_on(scene, DEV), chunk=16384)
        R.set_precision("f16")
        got = R.batchify_rays(rays, {"ray_bending_latents": lat}, chunk=65536, network_fn=ck.network_fn,
                              network_fine=ck.network_fine, N_samples=64, N_importance=128)
    p = psnr(got["rgb_map"].cpu(), ref["rgb_map"].cpu())
    print(f"\n[config 5 chunk, f16] PSNR vs fp32 oracle, all 65536 rays: {p:.1f} dB")
    assert p >= 40.0
