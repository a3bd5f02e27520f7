"""Reference checkpoint -> HIP render kwargs (nonrigid_nerf_amd/checkpoint.py; SURVEY.md section 8f #4, loader half).

The test checkpoints are written in the reference's on-disk layout (train.py:1680-1698) from the seeded synthetic
weights.  Their state-dict keys and shapes are pinned against the reference's own modules by
tests/golden/checkpoint_layout.json (oracle/make_golden.py imports the reference to produce it), and the golden
renders of tests/golden/*.npz were produced by the reference after load_state_dict(strict=True) of the very same arrays.
"""
import json
import os

import numpy as np
import pytest
import torch

from nonrigid_nerf_amd.checkpoint import load_checkpoint
from nonrigid_nerf_amd.synthetic import SceneConfig, make_rays, make_scene
from tests.helpers import compare_dict, load_golden

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CONFIGS = {
    "default":          dict(),
    "coarse_only":      dict(N_importance=0),
    "viewdirs":         dict(N_importance=64, use_viewdirs=True),
    "no_bender":        dict(N_importance=64, ray_bending=False),
    "time_conditioned": dict(N_importance=64, ray_bending=False, time_conditioned_baseline=True),
    "deep_bender":      dict(N_importance=64, use_viewdirs=True, bend_depth=7),
}


def write_reference_checkpoint(path, scene, frames=5, seed=0):
    """What train.py:1680-1698 saves (optimizer state and the script/dataset extras are placeholders)."""
    g = torch.Generator().manual_seed(seed)
    ck = {"global_step": 1234,
          "network_fn_state_dict": {k: v.clone() for k, v in scene.coarse.items()},
          "network_fine_state_dict": None if scene.fine is None else {k: v.clone() for k, v in scene.fine.items()},
          "ray_bender_state_dict": None if scene.bender is None else {k: v.clone() for k, v in scene.bender.items()},
          "optimizer_state_dict": {"state": {}, "param_groups": []},
          "ray_bending_latent_codes": torch.randn(frames, scene.cfg.latent_size, generator=g) * 0.1,
          "intrinsics": [{"height": 384, "width": 512, "focal_x": 256.6, "focal_y": 256.6, "center_x": 256.0, "center_y": 192.0}],
          "scripts_dict": {"train.py": "..."}, "dataset_extras": {"imageid_to_timestepid": np.arange(frames)}}
    torch.save(ck, path)
    return ck


@pytest.mark.parametrize("name", list(CONFIGS))
def test_architecture_is_inferred_from_the_state_dicts(tmp_path, name):
    cfg = SceneConfig(**CONFIGS[name])
    scene = make_scene(cfg, 0)
    path = os.path.join(str(tmp_path), "latest.tar")
    ck = write_reference_checkpoint(path, scene)
    got = load_checkpoint(path, N_samples=cfg.N_samples, N_importance=cfg.N_importance)
    a = got.arch
    assert (a["D"], a["W"], a["input_ch"], a["skips"]) == (cfg.netdepth, cfg.netwidth, cfg.input_ch, tuple(cfg.skips))
    assert a["use_viewdirs"] == cfg.use_viewdirs and a["time_conditioned_baseline"] == cfg.time_conditioned_baseline
    assert a["input_ch_views"] == cfg.input_ch_views and a["output_ch"] == cfg.output_ch
    if cfg.ray_bending:
        assert a["bender"] == dict(latent_size=cfg.latent_size, hidden=cfg.bend_hidden, depth=cfg.bend_depth,
                                   rigidity_hidden=cfg.rigidity_hidden, rigidity_depth=cfg.rigidity_depth)
        assert got.network_fn.ray_bender[0] is got.ray_bender and got.render_kwargs_test["ray_bender"] is got.ray_bender
        if got.network_fine is not None:
            assert got.network_fine.ray_bender[0] is got.ray_bender
    else:
        assert a["bender"] is None and got.ray_bender is None and got.network_fn.ray_bender == (None,)
    # weights arrive bit-exactly, nothing requires grad, the holders expose the reference's attribute surface
    for holder, sd in ((got.network_fn, ck["network_fn_state_dict"]), (got.network_fine, ck["network_fine_state_dict"]),
                       (got.ray_bender, ck["ray_bender_state_dict"])):
        if sd is None:
            assert holder is None
            continue
        hs = holder.state_dict()
        assert set(hs) == set(sd)
        assert all(torch.equal(hs[k], sd[k]) for k in sd)
        assert not any(p.requires_grad for p in holder.parameters())
    kw = got.render_kwargs_test                        # the dictionary create_nerf returns (train.py:698-719)
    assert set(kw) == {"network_query_fn", "perturb", "N_importance", "network_fine", "N_samples", "network_fn",
                       "ray_bender", "use_viewdirs", "white_bkgd", "raw_noise_std", "ndc", "lindisp"}
    assert kw["N_samples"] == cfg.N_samples and kw["N_importance"] == cfg.N_importance and not kw["perturb"]
    assert got.network_fn.num_ray_samples == cfg.N_samples
    if got.network_fine is not None:
        assert got.network_fine.num_ray_samples == cfg.N_samples + cfg.N_importance
    assert got.latents.shape == (5, cfg.latent_size) and got.global_step == 1234
    assert got.intrinsics[0]["width"] == 512 and "scripts_dict" in got.raw


def test_layout_matches_the_reference_modules():
    """Keys / shapes of the synthetic state dicts == what the reference's own modules hold (fixture made by importing it)."""
    layout = json.load(open(os.path.join(GOLDEN, "checkpoint_layout.json")))
    for name, want in layout.items():
        cfg = SceneConfig(**CONFIGS[name])
        scene = make_scene(cfg, 0)
        for part, sd in (("network_fn", scene.coarse), ("network_fine", scene.fine), ("ray_bender", scene.bender)):
            if want[part] is None:
                assert sd is None, (name, part)
            else:
                assert {k: list(v.shape) for k, v in sd.items()} == want[part], (name, part)


def test_inconsistent_requests_are_rejected(tmp_path):
    scene = make_scene(SceneConfig(N_importance=0), 0)
    path = os.path.join(str(tmp_path), "c.tar")
    write_reference_checkpoint(path, scene)
    with pytest.raises(ValueError):
        load_checkpoint(path, N_importance=64)                 # no fine network stored
    ck = torch.load(path, weights_only=False)
    ck["network_fn_state_dict"]["pts_linears.3.weight"] = torch.zeros(256, 100)
    with pytest.raises(ValueError):
        load_checkpoint(ck)


@pytest.mark.gpu
@pytest.mark.parametrize("golden", ["headline_64_128", "viewdirs_64_64", "time_conditioned_64_64"])
def test_render_from_checkpoint_matches_reference_golden(tmp_path, golden):
    """latest.tar -> load_checkpoint -> batchify_rays on the GPU == the reference's render of the same weights."""
    from nonrigid_nerf_amd import render as R
    meta, cfg, scene, rays, latents, ref = load_golden(golden)
    path = os.path.join(str(tmp_path), "latest.tar")
    write_reference_checkpoint(path, scene)
    ck = load_checkpoint(path, N_samples=cfg.N_samples, N_importance=cfg.N_importance, device="cuda:0")
    R.set_precision("f32")
    with torch.no_grad():
        out = R.batchify_rays(rays.to("cuda:0"), {"ray_bending_latents": latents.to("cuda:0")}, chunk=meta["chunk"],
                              retraw=bool(meta["retraw"]), **ck.render_kwargs_test)
    out = {k: v.cpu() for k, v in out.items()}
    assert set(out) == set(ref)
    fails = compare_dict(out, ref, keys=["rgb0", "disp0", "acc0"])
    fails += compare_dict(out, ref, keys=["rgb_map", "acc_map"], frac_ok=0.10, outlier_atol=2e-2)
    assert not fails, "\n".join(fails)


@pytest.mark.gpu
def test_free_viewpoint_frames_from_checkpoint():
    """The free-viewpoint use: latest.tar -> load_checkpoint -> render_path (host-resident holders, one code per frame,
    uint8 frames) against the oracle's render_path of the same weights."""
    import tempfile
    from nonrigid_nerf_amd import render as R
    from nonrigid_nerf_amd.driver import render_path
    from oracle import nrnerf_oracle as O
    from tests.helpers import synthetic_camera
    cfg = SceneConfig(N_importance=64)
    scene = make_scene(cfg, 0)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "latest.tar")
        write_reference_checkpoint(path, scene, frames=3)
        ck = load_checkpoint(path, N_samples=cfg.N_samples, N_importance=cfg.N_importance, device="cuda:0")
    cams = [synthetic_camera(k) for k in range(3)]
    poses, intrins = [c for c, _ in cams], [i for _, i in cams]
    ref_rgb, ref_disp = O.render_path(poses, intrins, scene, ck.latents.cpu())
    R.set_precision("f32")
    kw = dict(ck.render_kwargs_test, near=cfg.near, far=cfg.far)
    rgbs, disps = render_path(poses, intrins, 1024 * 32, kw, ck.latents)
    ok = (torch.from_numpy(rgbs) - ref_rgb).abs().amax(-1) <= 1e-4
    assert ok.float().mean() > 0.9 and (torch.from_numpy(rgbs) - ref_rgb).abs().max() < 0.1      # a few rays move a fine sample
    rgb8, _ = render_path(poses, intrins, 1024 * 32, kw, ck.latents, rgb_dtype="uint8")
    assert rgb8.dtype == np.uint8 and rgb8.shape == rgbs.shape
    assert np.array_equal(rgb8, (255 * np.clip(rgbs, 0, 1)).astype(np.uint8))                  # to8b, run_nerf_helpers.py:19
