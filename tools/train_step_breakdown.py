#!/usr/bin/env python
"""GPU box: where a native training step (argv: precision, rays per step; default bf16, 1024; 64+128 samples) spends its time -- wall time per phase with a device
synchronisation after each (so launch overheads of the eager pieces are included), plus the top GPU kernels by time."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd import render as R  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene  # noqa: E402

DEV = torch.device("cuda:0")


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    NR = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    cfg = SceneConfig()
    scene = make_scene(cfg, 0)
    rb, coarse, fine = build_modules(scene, device=DEV)
    params = []
    for m in (rb, coarse, fine):
        m.requires_grad_(True)
        params += list(m.parameters())
    codes = torch.zeros(8, 32, device=DEV, requires_grad=True)
    opt = torch.optim.Adam(params + [codes], lr=5e-4)
    rays, _ = make_rays(NR, 5, cfg)
    rays = rays.to(DEV)
    frame = torch.randint(0, 8, (NR,), device=DEV)
    target = torch.rand(NR, 3, device=DEV)
    R.set_precision(prec)
    kw = dict(network_fn=coarse, network_fine=fine, network_query_fn=None, N_samples=64, N_importance=128, perturb=1.0,
              raw_noise_std=1.0, retraw=True)
    acc = {}

    def tick(name, t0):
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0

    def step(measure):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        R.get_model(coarse, fine, device=DEV)                       # weight refresh (device re-pack after the previous opt.step)
        if measure: tick("refresh (get_model -> update_from_device)", t0)
        t0 = time.perf_counter()
        out = R.batchify_rays(rays, {"ray_bending_latents": codes[frame]}, chunk=32768, **kw)
        loss = ((out["rgb_map"] - target) ** 2).mean() + ((out["rgb0"] - target) ** 2).mean()
        if measure: tick("forward", t0)
        t0 = time.perf_counter()
        loss.backward()
        if measure: tick("backward", t0)
        t0 = time.perf_counter()
        opt.step()
        if measure: tick("optimizer", t0)

    for _ in range(5):
        step(False)
    torch.cuda.synchronize()
    n = 20
    for _ in range(n):
        step(True)
    print(f"[{prec}] per step, phases synchronised: " + ", ".join(f"{k} {v / n * 1e3:.2f} ms" for k, v in acc.items()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step(False)
    torch.cuda.synchronize()
    print(f"[{prec}] free-running: {(time.perf_counter() - t0) / n * 1e3:.2f} ms/step")
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(5):
            step(False)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70))


if __name__ == "__main__":
    main()
