#!/usr/bin/env python
"""Race detector for the training kernels: the same seeded training iteration (the reference's shipped loss through the
drop-in entry points: trunk / bender / divergence forward, backward and weight-gradient kernels, compositing) repeated
many times from the same state must give bit-identical gradients -- none of the kernels uses atomics, trunk_wgrad's
re-alignment barrier, the mask records and the split fine bender must not change a bit from run to run.
    python tools/soak_training_determinism.py [repeats]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nonrigid_nerf_amd import render as R, training
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
CASES = {"default 64+64": dict(N_importance=64), "deep bender": dict(N_importance=64, bend_depth=7),
         "viewdirs": dict(N_importance=64, use_viewdirs=True), "narrow 128": dict(N_importance=64, netwidth=128),
         "time-conditioned": dict(N_importance=64, ray_bending=False, time_conditioned_baseline=True)}
bad = 0
for name, kw in CASES.items():
    cfg = SceneConfig(**kw)
    scene = make_scene(cfg, 0)
    for prec in ("f32", "bf16"):
        for n in (333, 2048):
            rays, lat = make_rays(n, 3, cfg)
            rb, coarse, fine = build_modules(scene, device=dev)
            mods = [m for m in (rb, coarse, fine) if m is not None]
            for m in mods:
                m.requires_grad_(True)
            lat_d = lat.to(dev).requires_grad_(True)
            target = torch.rand(n, 3, generator=torch.Generator().manual_seed(1)).to(dev)
            rkw = dict(network_fn=coarse, network_fine=fine, network_query_fn=None, N_samples=64, N_importance=64, perturb=1.0, raw_noise_std=1.0)
            w = dict(offsets_loss_weight=60.0, divergence_loss_weight=3.0, rigidity_loss_weight=0.0005) if rb is not None else {}
            R.set_precision(prec)
            first, differing = None, 0
            for r in range(reps):
                for m in mods:
                    m.zero_grad(set_to_none=True)
                lat_d.grad = None
                torch.manual_seed(5)
                loss, _ = training.training_loss(rays.to(dev), lat_d, target, rkw, global_step=1000, N_iters=200000, chunk=32768, **w)
                loss.mean().backward()
                g = torch.cat([p.grad.reshape(-1) for m in mods for p in m.parameters() if p.grad is not None] + [lat_d.grad.reshape(-1)])
                if first is None:
                    first = g.clone()
                elif not torch.equal(first, g):
                    differing += 1
            bad += differing
            print(f"{name:18s} {prec:4s} {n:5d} rays: {reps} repeated iterations, {differing} with different gradients ({g.numel()} values)")
print("TOTAL differing:", bad)
sys.exit(1 if bad else 0)
