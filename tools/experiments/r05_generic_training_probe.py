"""One training iteration (training.training_loss with the shipped regulariser weights) on shapes that train on the run-time-parameterised
kernel: finite loss and gradients on every parameter (no reference function is installed: every term ran on the library).  GPU box: python tools/experiments/r05_generic_training_probe.py"""
import torch
from nonrigid_nerf_amd import render as R, training as T
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene

DEV = "cuda:0"
for name, kw in (("w128_viewdirs (forced generic handle)", dict(netwidth=128, use_viewdirs=True)), ("w192", dict(netwidth=192)),
                 ("w320_d5_viewdirs", dict(netwidth=320, netdepth=5, skips=(2,), use_viewdirs=True))):
    for prec in ("f32", "bf16"):
        cfg = SceneConfig(N_importance=64, **kw)
        scene = make_scene(cfg, 1)
        rays, lat = make_rays(512, 3, cfg)
        rb, c, f = build_modules(scene, device=DEV)
        for m in (rb, c, f):
            m.requires_grad_(True)
        R.set_precision(prec)
        lat = lat.to(DEV).requires_grad_(True)
        target = torch.rand(512, 3, device=DEV)
        rk = dict(network_fn=c, network_fine=f, network_query_fn=None, N_samples=64, N_importance=64, perturb=1.0, raw_noise_std=1.0)
        loss, extras = T.training_loss(rays.to(DEV), lat, target, rk, offsets_loss_weight=60.0, divergence_loss_weight=3.0, rigidity_loss_weight=5e-4,
                                       global_step=1000)
        loss.mean().backward()
        model = T.R.model_of_bender(rb, DEV)     # (the handle the iteration ran on)
        ps = [p for m in (rb, c, f) for p in m.parameters()]
        with_grad = [p for p in ps if p.grad is not None]
        ok = all(bool(torch.isfinite(p.grad).all()) for p in with_grad) and bool(torch.isfinite(lat.grad).all())
        print(f"{name} {prec}: loss {float(loss.mean()):.5f}, {len(with_grad)}/{len(ps)} parameters with a gradient, finite {ok}; generic handle {model.generic}, "
              f"bender training kernels on it {model.trains_bender}")
