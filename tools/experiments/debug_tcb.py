import sys, numpy as np, torch
sys.path.insert(0, '.')
from nonrigid_nerf_amd import render as R, training as T
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene
from oracle import nrnerf_oracle as O
DEV = "cuda:0"
cfg = SceneConfig(N_importance=64, ray_bending=False, time_conditioned_baseline=True)
scene = make_scene(cfg, 0)
rays, latents = make_rays(16, 0, cfg)
rb, coarse, fine = build_modules(scene, device=DEV)
for m in (coarse, fine): m.requires_grad_(True)
lat = latents.to(DEV).requires_grad_(True)
R.set_precision("f32")
for trial in range(3):
    out = R.batchify_rays(rays.to(DEV), {"ray_bending_latents": lat}, network_fn=coarse, network_fine=fine, network_query_fn=None,
                          N_samples=64, N_importance=64, perturb=0.0, raw_noise_std=0.0, retraw=True, _want_z_vals=True)
    print(trial, float(out["rgb_map"].sum() + out["rgb0"].sum()), float(out["rgb0"].sum()))
with torch.no_grad():
    o2 = R.batchify_rays(rays.to(DEV), {"ray_bending_latents": lat.detach()}, network_fn=coarse, network_fine=fine, network_query_fn=None,
                         N_samples=64, N_importance=64, perturb=0.0, raw_noise_std=0.0, retraw=True)
print("inference path:", float(o2["rgb_map"].sum() + o2["rgb0"].sum()), float(o2["rgb0"].sum()))
ref = O.render_rays(rays, latents, scene)
print("oracle:", float(ref["rgb_map"].sum() + ref["rgb0"].sum()), float(ref["rgb0"].sum()))
print("rgb0 err per ray", (out["rgb0"].detach().cpu() - ref["rgb0"]).abs().max(1).values)
