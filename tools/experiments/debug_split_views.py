#!/usr/bin/env python
"""GPU box: split-bender path vs fused fine pass with the view-dependent head (finite-difference directions): where and by
how much do they differ, per precision?"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd import render as R  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene  # noqa: E402

DEV = "cuda:0"
cfg = SceneConfig(N_importance=64, use_viewdirs=True)
scene = make_scene(cfg, 2)
rays, latents = make_rays(3001, 23, cfg)
rb, coarse, fine = build_modules(scene, device=DEV)
for prec in ("f32", "f16", "bf16"):
    R.set_precision(prec)
    model = R.get_model(coarse, fine)
    r, l = rays.to(DEV), latents.to(DEV)
    with torch.no_grad():
        split = model.render(r, l, 64, 64, retraw=True, want_z_vals=True, surface=True)
        fused = model.render(r, l, 64, 64, retraw=True, detailed_output=True, want_z_vals=True, surface=True)
    torch.cuda.synchronize()
    dr = (split["raw"] - fused["raw"]).abs()          # [N, 128, 4]
    bad = dr.amax(-1) > 0
    pos = torch.arange(128, device=DEV)[None, :].expand_as(bad)
    print(f"[{prec}] z equal {torch.equal(split['_z_vals'], fused['_z_vals'])}; rgb0 equal {torch.equal(split['rgb0'], fused['rgb0'])}; "
          f"raw differs on {int(bad.sum())} of {bad.numel()} samples, max |diff| rgb logits {float(dr[..., :3].max()):.3e} sigma {float(dr[..., 3].max()):.3e}; "
          f"rgb_map max |diff| {float((split['rgb_map'] - fused['rgb_map']).abs().max()):.3e}")
    if bad.any():
        hist = torch.bincount(pos[bad], minlength=128)
        print("      differing samples by position along the ray (first 40):", hist[:40].tolist())
        print("      positions mod 32:", torch.bincount(pos[bad] % 32, minlength=32).tolist())
