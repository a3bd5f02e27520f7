import sys, torch, time
sys.path.insert(0, ".")
from nonrigid_nerf_amd import render as R
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene
cfg = SceneConfig(N_importance=64)
scene = make_scene(cfg, 0)
rb, coarse, fine = build_modules(scene, device="cuda:0")
rays, lat = make_rays(4096, 1, cfg); rays, lat = rays.cuda(), lat.cuda()
api = {"ray_bending_latents": lat}
kw = dict(network_fn=coarse, network_fine=fine, N_samples=64, N_importance=64)
free0, total = torch.cuda.mem_get_info()
with torch.no_grad():
    for it in range(300):
        if it % 3 == 0:
            coarse.pts_linears[it % 8].weight.mul_(1.0001)          # forces nrnerf_model_update
        if it % 50 == 0:
            R.set_precision(["bf16", "f16", "f32"][(it // 50) % 3])  # new handles
        out = R.batchify_rays(rays, api, chunk=1024 * (1 + it % 5), detailed_output=(it % 7 == 0), perturb=float(it % 2), **kw)
        assert torch.isfinite(out["rgb_map"]).all()
torch.cuda.synchronize()
free1, _ = torch.cuda.mem_get_info()
print("device memory delta MiB:", (free0 - free1) / 2**20, "torch allocated MiB:", torch.cuda.memory_allocated() / 2**20)
