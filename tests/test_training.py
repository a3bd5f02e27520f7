"""Native training path (nonrigid_nerf_amd/training.py over nrnerf_trunk_* / nrnerf_composite_*): gradients against the
REFERENCE's autograd (tests/golden/gradients_64_64.npz, produced by oracle/make_golden.py::run_gradients from the
unmodified reference) and against the oracle's autograd on larger, stochastic batches; device-side weight refresh; a
short optimisation run."""
import os

import numpy as np
import pytest
import torch

from nonrigid_nerf_amd import render as R
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene
from tests.helpers import GOLDEN_DIR

DEV = "cuda:0"


def _modules(scene, requires_grad=True):
    rb, coarse, fine = build_modules(scene, device=DEV)
    for m in (rb, coarse, fine):
        if m is not None:
            m.requires_grad_(requires_grad)
    return rb, coarse, fine


def _named(rb, coarse, fine):
    out = {}
    for part, mod in (("bender", rb), ("coarse", coarse), ("fine", fine)):
        if mod is not None:
            for k, p in mod.named_parameters():
                out[(part, k)] = p
    return out


@pytest.mark.gpu
def test_gradients_match_reference_autograd_golden():
    """fp32 mode: d(sum rgb_map + sum rgb0) wrt the latent codes and a few parameters of every network, against what the
    reference's own autograd produced (train.render under grad, z_samples detached): within 2e-3 of each tensor's scale."""
    ref = np.load(os.path.join(GOLDEN_DIR, "gradients_64_64.npz"))
    cfg = SceneConfig(N_importance=64)
    scene = make_scene(cfg, 0)
    rays, latents = make_rays(16, 0, cfg)
    rb, coarse, fine = _modules(scene)
    lat = latents.to(DEV).requires_grad_(True)
    R.set_precision("f32")
    out = R.batchify_rays(rays.to(DEV), {"ray_bending_latents": lat}, network_fn=coarse, network_fine=fine, network_query_fn=None,
                          N_samples=64, N_importance=64, perturb=0.0, raw_noise_std=0.0, retraw=True)
    assert out["rgb_map"].requires_grad and out["rgb0"].requires_grad and out["raw"].shape == (16, 128, 5)
    loss = out["rgb_map"].sum() + out["rgb0"].sum()
    loss.backward()
    assert abs(float(loss.detach()) - float(ref["loss"])) < 1e-4 * abs(float(ref["loss"]))
    named = _named(rb, coarse, fine)
    checks = {"grad__latents": lat.grad}
    for key in ref.files:
        if key.startswith("grad__") and key != "grad__latents":
            _, part, name = key.split("__", 2)
            checks[key] = named[(part, name)].grad
    for key, g in checks.items():
        want = torch.from_numpy(ref[key])
        scale = float(want.abs().max()) + 1e-12
        assert g is not None and tuple(g.shape) == tuple(want.shape), key
        err = float((g.cpu() - want).abs().max())
        assert err <= 2e-3 * scale, (key, err, scale)


def _oracle_grads(scene, rays, latents, seed, perturb, noise, detailed_loss):
    from oracle import nrnerf_oracle as O
    sc = O.scene_on(scene, DEV)
    leaves = {}
    for part in ("bender", "coarse", "fine"):
        d = getattr(sc, part)
        if d is None:
            continue
        for k in d:
            d[k] = d[k].clone().requires_grad_(True)
            leaves[(part, k)] = d[k]
    lat = latents.to(DEV).clone().requires_grad_(True)
    torch.manual_seed(seed)
    out = O.render_rays(rays.to(DEV), lat, sc, retraw=True, detailed_output=detailed_loss, perturb=perturb, raw_noise_std=noise)
    loss = _loss(out, detailed_loss)
    loss.backward()
    return float(loss.detach()), lat.grad, {k: v.grad for k, v in leaves.items()}, out


def _loss(out, detailed):
    target = torch.linspace(0.1, 0.9, 3, device=out["rgb_map"].device)
    loss = ((out["rgb_map"] - target) ** 2).mean() + ((out["rgb0"] - target) ** 2).mean() + 0.1 * out["acc_map"].mean()
    loss = loss + 0.05 * (out["disp_map"].clamp(max=50.0)).mean() * 1e-2
    if detailed:        # the offsets regulariser's shape (train.py:219-236): detached weights x offset norms, plus the rigidity term
        w = out["visibility_weights"].detach()
        off = torch.norm(out["unmasked_offsets"], dim=-1)
        rig = out["rigidity_mask"][..., 0]
        loss = loss + 3.0 * (w * torch.pow(off + 1e-12, 2.0 - rig)).mean() + 0.01 * (w * rig).mean()
        loss = loss + 0.02 * out["fine_visibility_weights"].pow(2).mean()      # a gradient through the weights output
    return loss


@pytest.mark.gpu
@pytest.mark.parametrize("perturb,noise,detailed,cfg_kw", [(0.0, 0.0, False, dict(N_importance=64)),
                                                           (1.0, 1.0, True, dict(N_samples=48, N_importance=37)),
                                                           (1.0, 0.5, False, dict(N_importance=128, ray_bending=False))],
                         ids=["deterministic", "stochastic_detailed_ragged", "no_bender_64_128"])
def test_fp32_gradients_vs_oracle_autograd(perturb, noise, detailed, cfg_kw):
    """Every parameter of every network + the latent codes, fp32 mode, against the oracle's autograd (eager torch on the
    GPU, same seeded random numbers): 2e-3 of each tensor's scale; loss with data, acc, disp, weights and regulariser terms."""
    cfg = SceneConfig(**cfg_kw)
    scene = make_scene(cfg, 1)
    rays, latents = make_rays(131, 3, cfg)
    l_ref, glat_ref, g_ref, out_ref = _oracle_grads(scene, rays, latents, 99, perturb, noise, detailed)
    rb, coarse, fine = _modules(scene)
    lat = latents.to(DEV).requires_grad_(True)
    R.set_precision("f32")
    torch.manual_seed(99)
    out = R.render_rays(rays.to(DEV), coarse, None, cfg.N_samples, retraw=True, perturb=perturb, N_importance=cfg.N_importance,
                        network_fine=fine, raw_noise_std=noise, additional_pixel_information={"ray_bending_latents": lat},
                        detailed_output=detailed)
    assert set(out) == set(k for k in out_ref if not k.startswith("_"))
    for k in ("rgb_map", "rgb0", "acc_map"):
        assert torch.allclose(out[k], out_ref[k].detach(), atol=2e-3), k          # (a few fine samples may move: sample_pdf branch)
    loss = _loss(out, detailed)
    loss.backward()
    assert abs(float(loss.detach()) - l_ref) <= 2e-3 * abs(l_ref)
    named = _named(rb, coarse, fine)
    fails = []
    if cfg.ray_bending:
        scale = float(glat_ref.abs().max()) + 1e-12
        if float((lat.grad - glat_ref).abs().max()) > 5e-3 * scale:
            fails.append(("latents", float((lat.grad - glat_ref).abs().max()), scale))
    for (part, name), gr in g_ref.items():
        g = named[(part, name)].grad
        if gr is None:
            continue
        scale = float(gr.abs().max()) + 1e-12
        err = float((g - gr).abs().max())
        if err > 5e-3 * scale:          # a moved fine sample changes a few rays' contributions: slightly wider than the golden test
            fails.append((part, name, err, scale))
    assert not fails, fails


@pytest.mark.gpu
def test_bf16_gradients_point_the_same_way():
    """bf16 training mode (bf16 activations, d z and weight-gradient GEMMs): gradient direction and size against fp32."""
    cfg = SceneConfig(N_importance=64)
    scene = make_scene(cfg, 1)
    rays, latents = make_rays(512, 3, cfg)
    grads = {}
    for prec in ("f32", "bf16"):
        rb, coarse, fine = _modules(scene)
        lat = latents.to(DEV).requires_grad_(True)
        R.set_precision(prec)
        out = R.render_rays(rays.to(DEV), coarse, None, 64, N_importance=64, network_fine=fine,
                            additional_pixel_information={"ray_bending_latents": lat})
        _loss(out, False).backward()
        g = {k: p.grad.flatten().float() for k, p in _named(rb, coarse, fine).items()}
        g[("latents", "")] = lat.grad.flatten()
        grads[prec] = g
    for k, g32 in grads["f32"].items():
        g16 = grads["bf16"][k]
        if float(g32.norm()) < 1e-10:
            continue
        cos = float(torch.dot(g32, g16) / (g32.norm() * g16.norm() + 1e-30))
        ratio = float(g16.norm() / g32.norm())
        assert cos > 0.97 and 0.9 < ratio < 1.1, (k, cos, ratio)


@pytest.mark.gpu
def test_training_steps_fit_a_target_and_refresh_weights_on_the_device():
    """A short optimisation run through the drop-in boundary (Adam, 1024 rays, perturb + raw noise as in training): the
    loss falls, and every step's weight refresh goes through nrnerf_model_update_device (no host round trip)."""
    from tests.test_fitted_checkpoint import FIXTURE
    z = np.load(FIXTURE)
    cfg = SceneConfig(N_importance=64)
    scene = make_scene(cfg, 0)
    rb, coarse, fine = _modules(scene)
    lat_codes = torch.zeros(4, 32, device=DEV, requires_grad=True)
    params = list(rb.parameters()) + list(coarse.parameters()) + list(fine.parameters()) + [lat_codes]
    opt = torch.optim.Adam(params, lr=5e-4)
    rays, _ = make_rays(1024, 7, cfg)
    rays = rays.to(DEV)
    frame = torch.randint(0, 4, (1024,), device=DEV)
    target = torch.from_numpy(z["images"][0]).float().reshape(-1, 3)[:1024].to(DEV) / 255.0
    R.set_precision("bf16")
    calls = {"dev": 0}
    orig = R.Model.update_from_device

    def counting(self, *a, **k):
        ok = orig(self, *a, **k)
        calls["dev"] += int(ok)
        return ok

    R.Model.update_from_device = counting
    try:
        losses = []
        torch.manual_seed(0)
        for step in range(30):
            opt.zero_grad(set_to_none=True)
            out = R.batchify_rays(rays, {"ray_bending_latents": lat_codes[frame]}, chunk=32768, network_fn=coarse, network_fine=fine,
                                  network_query_fn=None, N_samples=64, N_importance=64, perturb=1.0, raw_noise_std=1.0, retraw=True)
            loss = ((out["rgb_map"] - target) ** 2).mean() + ((out["rgb0"] - target) ** 2).mean()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    finally:
        R.Model.update_from_device = orig
    assert all(np.isfinite(losses)), losses
    assert np.mean(losses[-5:]) < 0.7 * np.mean(losses[:5]), losses
    assert calls["dev"] >= 25, calls
    assert lat_codes.grad is not None and float(lat_codes.grad.abs().max()) > 0
