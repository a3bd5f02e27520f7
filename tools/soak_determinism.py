#!/usr/bin/env python
"""Race detector for the hand-synchronised kernels: the same launch repeated many times must give bit-identical
outputs (the LDS ring hand-offs, counted waits and the view-direction mailbox have no atomics; any race shows up as a
run-to-run difference).  python tools/soak_determinism.py [repeats]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nonrigid_nerf_amd import render as R
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
CASES = {
    "default 64+128":        dict(),
    "no bender":             dict(ray_bending=False, N_importance=64),
    "viewdirs (finite diff)": dict(use_viewdirs=True, N_importance=64),
    "viewdirs (exact)":      dict(use_viewdirs=True, N_importance=64, approx_nonrigid_viewdirs=False),
    "deep bender + viewdirs": dict(use_viewdirs=True, N_importance=64, bend_depth=7),
    "time-conditioned":      dict(ray_bending=False, time_conditioned_baseline=True, N_importance=64),
    "width 128":             dict(netwidth=128, N_importance=64),
    # round 4: the fused compositing epilogue (LDS stages, prefetched inputs) is on every plain render above; odd block counts per ray and
    # the run-time-parameterised kernel (LDS activation buffers, two barriers per layer) get their own rows
    "odd blocks 64+96":      dict(N_importance=96),
    "generic 192/320":       dict(netdepth=6, netwidth=192, netwidth_fine=320, multires=8, latent_size=16, N_importance=64),
    "generic viewdirs 96":   dict(netwidth=96, use_viewdirs=True, multires_views=2, N_importance=64),
    # round 5: the 16-bit rows above run on the 16x16x32 kernels (both passes; stand-alone bender on 16x16x32 in bf16 mode), the generic
    # rows on the width-class kernel (run-time layer loop on a run-time-pointer ring); two more width classes, one of them above 256
    # (two blocks per wave), one without skip connection and with the reference's bender (the compiled bender kernels)
    "generic 448 d3 noskip": dict(netwidth=448, netdepth=3, skips=(), N_importance=64),
    "generic 320 d10":       dict(netwidth=320, netdepth=10, N_importance=32),
}
bad = 0
for name, kw in CASES.items():
    cfg = SceneConfig(**kw)
    scene = make_scene(cfg, 0)
    rb, coarse, fine = build_modules(scene, device="cuda:0")
    for n in (50000, 3333):
        rays, lat = make_rays(n, 5, cfg)
        rays, lat = rays.cuda(), lat.cuda()
        for prec in ("bf16", "f16", "f32"):
            R.set_precision(prec)
            model = R.get_model(coarse, fine)
            with torch.no_grad():
                first = model.render(rays, lat, cfg.N_samples, cfg.N_importance, retraw=True)
                diffs = 0
                for _ in range(reps if prec != "f32" else max(reps // 8, 2)):
                    out = model.render(rays, lat, cfg.N_samples, cfg.N_importance, retraw=True)
                    diffs += sum(int(not torch.equal(torch.nan_to_num(out[k]), torch.nan_to_num(first[k]))) for k in first)
            torch.cuda.synchronize()
            bad += diffs
            print(f"{name:24s} n={n:6d} {prec}: {'identical' if diffs == 0 else str(diffs) + ' DIFFERENCES'}", flush=True)
print("TOTAL differing tensors:", bad)
sys.exit(1 if bad else 0)
