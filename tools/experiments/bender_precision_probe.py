#!/usr/bin/env python
"""GPU box, decision experiment (not product code): how much image accuracy would a SINGLE-product f16 bender (instead of
the fp32-equivalent 3-term split product) cost on the fitted checkpoint?  Emulated with the oracle's torch ops: bender
weights and layer inputs rounded to f16, fp32 accumulation, everything else exact fp32."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd.checkpoint import load_checkpoint  # noqa: E402
from nonrigid_nerf_amd.driver import generate_rays  # noqa: E402
from nonrigid_nerf_amd.synthetic import Scene, SceneConfig  # noqa: E402
from oracle import nrnerf_oracle as O  # noqa: E402

DEV = "cuda:0"
MODE = {"round": None, "exact_input": False, "scale": 1.0}


def r16(x):
    m = MODE["round"]
    return x if m is None else x.to(m).to(torch.float32)


_orig = O.bend_points


def bend_emulated(pts, latents, bender, knobs=None):
    knobs = knobs or O.Knobs()
    n_off = 1 + max(int(k.split(".")[1]) for k in bender if k.startswith("network."))
    h = torch.cat([pts, latents.to(pts.dtype)], -1)
    for i in range(n_off):
        b = bender.get(f"network.{i}.bias")
        hin = h if (i == 0 and MODE["exact_input"]) else r16(h)       # exact_input: layer-0 input carried as hi + lo
        h = F.linear(hin, r16(bender[f"network.{i}.weight"].to(h.device)), None if b is None else b.to(h.device))
        if i != n_off - 1:
            h = F.relu(h)
    unmasked = h * MODE["scale"]
    n_rig = 1 + max(int(k.split(".")[1]) for k in bender if k.startswith("rigidity_network."))
    h = pts
    for i in range(n_rig):
        hin = h if (i == 0 and MODE["exact_input"]) else r16(h)
        h = F.linear(hin, r16(bender[f"rigidity_network.{i}.weight"].to(h.device)), bender[f"rigidity_network.{i}.bias"].to(h.device))
        if i != n_rig - 1:
            h = F.relu(h)
    mask = (torch.tanh(h) + 1) / 2
    masked = mask * unmasked
    return pts + masked, dict(unmasked_offsets=unmasked, rigidity_mask=mask, masked_offsets=masked)


def main():
    gold = os.path.join(REPO, "tests", "golden")
    ck = load_checkpoint(os.path.join(gold, "fitted_latest.tar"), N_samples=64, N_importance=128)
    z = np.load(os.path.join(gold, "example_sequence_96x72.npz"))
    near, far = float(z["bds"].min()) * 0.9, float(z["bds"].max())
    sd = lambda m: {k: v.detach().clone() for k, v in m.state_dict().items()}
    scene = O.scene_on(Scene(SceneConfig(near=near, far=far), sd(ck.ray_bender), sd(ck.network_fn), sd(ck.network_fine)), DEV)
    s = 256.0 / float(z["hwf"][1])
    intrin = dict(height=192, width=256, focal_x=float(z["hwf"][2]) * s, focal_y=float(z["hwf"][2]) * s, center_x=128.0, center_y=96.0)
    for frame in (3, 10, 30):
        rays = generate_rays(torch.from_numpy(z["poses"][frame]), intrin, near, far, False, DEV)
        lat = ck.latents[frame].to(DEV).reshape(1, -1).expand(rays.shape[0], -1).contiguous()
        with torch.no_grad():
            for scale in (1.0, 5.0, 20.0):
                MODE.update(round=None, exact_input=False, scale=scale)
                O.bend_points = bend_emulated          # exact arithmetic, offsets scaled (a more non-rigid scene)
                ref = O.batchify_rays(rays, lat, scene, chunk=16384)
                out = {}
                for name, dt, ex in (("f16 single", torch.float16, False), ("f16 single + exact layer-0 input", torch.float16, True)):
                    MODE.update(round=dt, exact_input=ex)
                    got = O.batchify_rays(rays, lat, scene, chunk=16384)
                    mse = float(((got["rgb_map"] - ref["rgb_map"]) ** 2).mean())
                    out[name] = -10 * np.log10(max(mse, 1e-30))
                print(f"frame {frame} offsets x{scale:g}: " + "; ".join(f"{k}: rgb_map {a:.1f} dB" for k, a in out.items()))
            O.bend_points = _orig
    # offsets' size on this model, for scale
    with torch.no_grad():
        pts = rays[:4096, 0:3] + rays[:4096, 3:6] * 0.5
        bent, d = _orig(pts, lat[:4096], scene.bender)
        print(f"|masked offset| mean {float(d['masked_offsets'].norm(dim=-1).mean()):.3e}, max {float(d['masked_offsets'].norm(dim=-1).max()):.3e}")


if __name__ == "__main__":
    main()
