#!/usr/bin/env python
"""GPU box: small render_rays calls, eager vs render.GraphedRender (one HIP graph replay per call).
Two numbers per batch size: THROUGHPUT (200 calls queued back to back, one synchronisation) and LATENCY (the caller waits for
every result, as determine_nerf_volume_extent's chunk loop and a training loop's validation probes do).
    python tools/small_batch_bench.py [precision]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd import render as R  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = "cuda:0"
cfg = SceneConfig()
scene = make_scene(cfg, 0)
rb, coarse, fine = build_modules(scene, device=dev)
R.set_precision(prec)
kw = dict(network_fine=fine, N_samples=cfg.N_samples, N_importance=cfg.N_importance, perturb=0.0, raw_noise_std=0.0)


def timed(fn, calls, sync_each):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        fn()
        if sync_each:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / calls


print(f"# {prec}, {cfg.N_samples} + {cfg.N_importance} samples, default architecture; ms per call (M rays/s)")
print(f"# {'rays':>6} | {'eager, queued':>20} | {'graph, queued':>20} | {'eager, waited for':>20} | {'graph, waited for':>20}")
for n in (128, 512, 1024, 2048, 4096):
    rays, lat = make_rays(n, 1, cfg)
    rays, lat = rays.to(dev), lat.to(dev)
    api = {"ray_bending_latents": lat}
    with torch.no_grad():
        eager = lambda: R.render_rays(rays, coarse, additional_pixel_information=api, **kw)
        g = R.GraphedRender(rays, coarse, latents=lat, check_weights=False, **kw)
        graph = lambda: g(rays, lat)
        cells = []
        for fn in (eager, graph):
            cells.append(timed(fn, 200, False))
        for fn in (eager, graph):
            cells.append(timed(fn, 200, True))
    print(f"  {n:>6} | " + " | ".join(f"{c * 1e3:8.3f} ({n / c / 1e6:6.2f})   " for c in cells))
