#!/usr/bin/env python
"""GPU box: the device kernels of ONE forward + backward of a non-compiled architecture (W 192, 2048 rays x (64 + 128), bf16), in order."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from nonrigid_nerf_amd import render as R  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene  # noqa: E402

DEV = "cuda:0"
W = int(sys.argv[1]) if len(sys.argv) > 1 else 192
bend = "--bender" in sys.argv
cfg = SceneConfig(N_samples=64, N_importance=128, netwidth=W, ray_bending=bend)
scene = make_scene(cfg, 1)
rays, latents = make_rays(2048, 3, cfg)
rays = rays.to(DEV)
rb, coarse, fine = build_modules(scene, device=DEV)
for m in (rb, coarse, fine):
    if m is not None:
        m.requires_grad_(True)
R.set_precision("bf16")
target = torch.linspace(0.1, 0.9, 3, device=DEV)
api = {"ray_bending_latents": latents.to(DEV)} if bend else None


def ours():
    out = R.render_rays(rays, coarse, None, 64, N_importance=128, network_fine=fine, perturb=1.0, raw_noise_std=1.0, additional_pixel_information=api)
    (((out["rgb_map"] - target) ** 2).mean() + ((out["rgb0"] - target) ** 2).mean()).backward()


for _ in range(3):
    ours()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    ours()
    torch.cuda.synchronize()
ev = sorted([e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
t0 = ev[0].time_range.start
busy = 0.0
print(f"# {len(ev)} device kernels / copies of one forward + backward, W {W}, 2048 rays x (64 + 128), bf16{', ray bender' if bend else ''}")
for e in ev:
    d = e.time_range.end - e.time_range.start
    busy += d
    print(f"{e.time_range.start - t0:9.1f} {d:8.1f}  {e.name[:100]}")
print(f"# busy {busy:.0f} us of {ev[-1].time_range.end - t0:.0f} us")
