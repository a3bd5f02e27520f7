#!/usr/bin/env python
"""GPU box: gradient norms over a few optimiser steps, device-repacked handle vs a freshly host-packed one."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd import render as R  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene  # noqa: E402

DEV = "cuda:0"


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    cfg = SceneConfig(N_importance=64)
    scene = make_scene(cfg, 0)
    rb, coarse, fine = build_modules(scene, device=DEV)
    for m in (rb, coarse, fine):
        m.requires_grad_(True)
    codes = torch.zeros(4, 32, device=DEV, requires_grad=True)
    params = list(rb.parameters()) + list(coarse.parameters()) + list(fine.parameters()) + [codes]
    opt = torch.optim.Adam(params, lr=5e-4)
    rays, _ = make_rays(1024, 7, cfg)
    rays = rays.to(DEV)
    frame = torch.randint(0, 4, (1024,), device=DEV)
    target = torch.rand(1024, 3, device=DEV)
    R.set_precision(prec)
    kw = dict(network_fn=coarse, network_fine=fine, network_query_fn=None, N_samples=64, N_importance=64, perturb=1.0,
              raw_noise_std=1.0, retraw=True)
    watch = {"codes": codes, "rb.net0.w": rb.network[0].weight, "rb.net4.w": rb.network[4].weight, "c.pts0.w": coarse.pts_linears[0].weight,
             "c.pts5.w": coarse.pts_linears[5].weight, "c.pts7.w": coarse.pts_linears[7].weight, "c.out.w": coarse.output_linear.weight,
             "f.pts0.w": fine.pts_linears[0].weight, "f.pts7.w": fine.pts_linears[7].weight, "f.out.w": fine.output_linear.weight}

    def grads(seed):
        for p in params:
            p.grad = None
        torch.manual_seed(seed)
        out = R.batchify_rays(rays, {"ray_bending_latents": codes[frame]}, chunk=32768, **kw)
        loss = ((out["rgb_map"] - target) ** 2).mean() + ((out["rgb0"] - target) ** 2).mean()
        loss.backward()
        return float(loss.detach()), {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in watch.items()}

    for step in range(3):
        l_dev, g_dev = grads(step)                  # handle refreshed on the device (after step 0: the previous opt.step)
        R.invalidate()                               # forget the handle: next call packs on the host from scratch
        l_host, g_host = grads(step)
        print(f"step {step}: loss device-refreshed {l_dev:.6f} host-packed {l_host:.6f}")
        for k in watch:
            a, b = g_dev[k], g_host[k]
            na = float(a.norm()) if a is not None else float("nan")
            nb = float(b.norm()) if b is not None else float("nan")
            d = float((a - b).norm()) if a is not None and b is not None else float("nan")
            print(f"    {k:10s} |g| device-refreshed {na:.4e}  host-packed {nb:.4e}  |diff| {d:.2e}")
        grads(step)
        opt.step()
        R.get_model(coarse, fine, device=DEV)        # device refresh happens here


if __name__ == "__main__":
    main()
