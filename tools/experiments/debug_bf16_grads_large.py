import sys, torch
sys.path.insert(0, "/root/repo")
from nonrigid_nerf_amd import render as R, training
from nonrigid_nerf_amd.synthetic import SceneConfig, make_rays, make_scene
from tests.test_training import _modules, _named, _loss
DEV = "cuda:0"
def run(S, I, nrays, split=True, seed=3):
    cfg = SceneConfig(N_samples=S, N_importance=I)
    scene = make_scene(cfg, 1)
    rays, latents = make_rays(nrays, seed, cfg)
    grads = {}
    old = training.SPLIT_FINE_BENDER
    training.SPLIT_FINE_BENDER = split
    for prec in ("f32", "bf16"):
        rb, coarse, fine = _modules(scene)
        lat = latents.to(DEV).requires_grad_(True)
        R.set_precision(prec)
        out = R.render_rays(rays.to(DEV), coarse, None, S, N_importance=I, network_fine=fine,
                            additional_pixel_information={"ray_bending_latents": lat}, detailed_output=True)
        _loss(out, True).backward()
        g = {k: p.grad.flatten().float() for k, p in _named(rb, coarse, fine).items() if p.grad is not None}
        g[("latents", "")] = lat.grad.flatten()
        grads[prec] = g
    training.SPLIT_FINE_BENDER = old
    rows = []
    for k, g32 in grads["f32"].items():
        g16 = grads["bf16"][k]
        if float(g32.norm()) < 1e-10: continue
        cos = float(torch.dot(g32, g16) / (g32.norm() * g16.norm() + 1e-30)); ratio = float(g16.norm() / g32.norm())
        if k[0] in ("bender", "latents") and (cos < 0.95 or not 0.8 < ratio < 1.25): rows.append((k[1], round(cos, 3), round(ratio, 3)))
    print(f"S={S} I={I} rays={nrays} split={split}: loose-tensor outliers: {rows}")
for args in ((64, 64, 96), (64, 64, 512), (64, 64, 96, False), (300, 400, 96), (300, 400, 512), (192, 128, 96), (64, 192, 96), (64, 193, 96), (128, 128, 96), (200, 57, 96, False)):
    run(*args)
