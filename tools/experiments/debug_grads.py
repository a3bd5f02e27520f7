#!/usr/bin/env python
"""GPU box: per-tensor gradient errors of the native training path (fp32 mode) against (a) the reference-autograd golden,
(b) the oracle's autograd on the GPU, free-running and (c) with its fine pass forced to the depths the HIP path chose."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd import render as R  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene  # noqa: E402
from oracle import nrnerf_oracle as O  # noqa: E402

DEV = "cuda:0"


def ours(scene, cfg, rays, latents, prec="f32"):
    rb, coarse, fine = build_modules(scene, device=DEV)
    for m in (rb, coarse, fine):
        if m is not None:
            m.requires_grad_(True)
    lat = latents.to(DEV).requires_grad_(True)
    R.set_precision(prec)
    out = R.render_rays(rays.to(DEV), coarse, None, cfg.N_samples, retraw=True, N_importance=cfg.N_importance, network_fine=fine,
                        additional_pixel_information={"ray_bending_latents": lat}, _want_z_vals=True)
    loss = out["rgb_map"].sum() + out["rgb0"].sum()
    loss.backward()
    g = {"latents": lat.grad}
    for part, mod in (("bender", rb), ("coarse", coarse), ("fine", fine)):
        if mod is not None:
            for k, p in mod.named_parameters():
                g[f"{part}__{k}"] = p.grad
    return float(loss.detach()), g, out


def oracle(scene, cfg, rays, latents, z_override=None, dtype=torch.float32):
    sc = O.scene_on(scene, DEV)
    leaves = {}
    for part in ("bender", "coarse", "fine"):
        d = getattr(sc, part)
        if d is None:
            continue
        for k in d:
            d[k] = d[k].clone().to(dtype).requires_grad_(True)
            leaves[f"{part}__{k}"] = d[k]
    lat = latents.to(DEV).clone().to(dtype).requires_grad_(True)
    out = O.render_rays(rays.to(DEV), lat, sc, retraw=True, z_fine_override=z_override, dtype=dtype)
    loss = out["rgb_map"].sum() + out["rgb0"].sum()
    loss.backward()
    g = {"latents": lat.grad}
    g.update({k: v.grad for k, v in leaves.items()})
    return float(loss.detach()), g, out


def table(title, a, b):
    print(f"--- {title}")
    for k in a:
        if k not in b or a[k] is None or b[k] is None:
            continue
        x, y = a[k].double().cpu(), b[k].double().cpu()
        scale = float(y.abs().max()) + 1e-30
        print(f"  {k:40s} max|err|/scale {float((x - y).abs().max()) / scale:.2e}   rel l2 {float((x - y).norm() / (y.norm() + 1e-30)):.2e}")


def main():
    cfg = SceneConfig(N_importance=64)
    scene = make_scene(cfg, 0)
    rays, latents = make_rays(16, 0, cfg)
    l_o, g_o, out_o = ours(scene, cfg, rays, latents)
    zg = out_o["_z_vals"].detach()
    l_r, g_r, out_r = oracle(scene, cfg, rays, latents)
    l_z, g_z, out_z = oracle(scene, cfg, rays, latents, z_override=zg)
    l_z64, g_z64, _ = oracle(scene, cfg, rays, latents, z_override=zg, dtype=torch.float64)
    moved = (zg - out_r["_z_vals"].detach()).abs() > 2e-5
    print(f"loss ours {l_o:.6f}  oracle {l_r:.6f}  oracle@our depths {l_z:.6f}; merged depths that differ: {int(moved.sum())} in rays {moved.any(1).nonzero().flatten().tolist()}")
    ref = np.load(os.path.join(REPO, "tests", "golden", "gradients_64_64.npz"))
    g_gold = {"latents": torch.from_numpy(ref["grad__latents"])}
    for key in ref.files:
        if key.startswith("grad__") and key != "grad__latents":
            g_gold[key[len("grad__"):]] = torch.from_numpy(ref[key])
    table("ours vs reference-autograd golden (CPU reference, its own sample depths)", {k: g_o[k] for k in g_gold}, g_gold)
    table("oracle on this GPU vs golden", {k: g_r[k] for k in g_gold}, g_gold)
    table("ours vs oracle fp32 evaluated at OUR merged depths (all tensors)", g_o, g_z)
    table("oracle fp32 vs oracle fp64, both at our depths (the yardstick's own rounding)", g_z, g_z64)
    table("ours vs oracle fp64 at our depths", g_o, g_z64)
    # raw pieces: forward agreement
    for k in ("rgb_map", "rgb0", "raw"):
        print(f"  forward {k}: max|ours - oracle@our depths| = {float((out_o[k].detach() - out_z[k].detach()).abs().max()):.2e}")


if __name__ == "__main__":
    main()
