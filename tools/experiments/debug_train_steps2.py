#!/usr/bin/env python
"""GPU box: the scenario of tests/test_training.py::test_training_steps..., printing |grad| of the latent codes per step."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd import render as R  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene  # noqa: E402

DEV = "cuda:0"
z = np.load(os.path.join(REPO, "tests", "golden", "example_sequence_96x72.npz"))
cfg = SceneConfig(N_importance=64)
scene = make_scene(cfg, 0)
rb, coarse, fine = build_modules(scene, device=DEV)
for m in (rb, coarse, fine):
    m.requires_grad_(True)
lat_codes = torch.zeros(4, 32, device=DEV, requires_grad=True)
params = list(rb.parameters()) + list(coarse.parameters()) + list(fine.parameters()) + [lat_codes]
opt = torch.optim.Adam(params, lr=5e-4)
rays, _ = make_rays(1024, 7, cfg)
rays = rays.to(DEV)
frame = torch.randint(0, 4, (1024,), device=DEV)
target = torch.from_numpy(z["images"][0]).float().reshape(-1, 3)[::6][:1024].to(DEV) / 255.0
R.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
torch.manual_seed(0)
for step in range(30):
    opt.zero_grad(set_to_none=True)
    out = R.batchify_rays(rays, {"ray_bending_latents": lat_codes[frame]}, chunk=32768, network_fn=coarse, network_fine=fine,
                          network_query_fn=None, N_samples=64, N_importance=64, perturb=1.0, raw_noise_std=1.0, retraw=True)
    loss = ((out["rgb_map"] - target) ** 2).mean() + ((out["rgb0"] - target) ** 2).mean()
    loss.backward()
    g = lat_codes.grad
    print(f"step {step:2d} loss {float(loss.detach()):.5f} |g codes| {float(g.norm()):.3e} |g rb.net0.w| {float(rb.network[0].weight.grad.norm()):.3e} "
          f"|g rb.net4.w| {float(rb.network[4].weight.grad.norm()):.3e} |g rig2.w| {float(rb.rigidity_network[2].weight.grad.norm()):.3e} "
          f"|g c.pts0| {float(coarse.pts_linears[0].weight.grad.norm()):.3e} acc {float(out['acc_map'].detach().mean()):.3f} "
          f"rig mean {0.0:.1f} |net4.w| {float(rb.network[4].weight.detach().norm()):.3f}")
    opt.step()
