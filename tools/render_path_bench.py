#!/usr/bin/env python
"""End-to-end free-viewpoint loop (driver.render_path: device ray generation, one frame code per frame, async D2H)
on synthetic weights: frames/s and rays/s including the copies to pinned host memory.
    python tools/render_path_bench.py [frames] [H] [W] [config4]"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nonrigid_nerf_amd import render as R
from nonrigid_nerf_amd.driver import render_path
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_scene

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 120
H = int(sys.argv[2]) if len(sys.argv) > 2 else 384
W = int(sys.argv[3]) if len(sys.argv) > 3 else 512
# fourth argument "config4": BASELINE config 4 -- use_viewdirs=True and the deeper (7-layer) ray-bending MLP, e.g. `300 384 512 config4`
config4 = len(sys.argv) > 4 and sys.argv[4] == "config4"
cfg = SceneConfig(use_viewdirs=True, bend_depth=7) if config4 else SceneConfig()
scene = make_scene(cfg, 0)
rb, coarse, fine = build_modules(scene, device="cuda:0")
R.set_precision("bf16")
poses, intr = [], []
for k in range(frames):
    a = 0.02 * k - 0.4
    c2w = torch.tensor([[math.cos(a), 0.0, math.sin(a), 0.1 * math.sin(a)], [0.0, 1.0, 0.0, 0.0],
                        [-math.sin(a), 0.0, math.cos(a), 0.15]])
    poses.append(c2w)
    intr.append(dict(height=H, width=W, focal_x=256.6 * W / 512, focal_y=256.6 * H / 384, center_x=W / 2, center_y=H / 2))
codes = torch.randn(frames, 32, generator=torch.Generator().manual_seed(1)) * 0.1
kw = dict(network_fn=coarse, network_fine=fine, network_query_fn=None, N_samples=64, N_importance=128, perturb=False,
          raw_noise_std=0.0, white_bkgd=False, lindisp=False, ndc=False, use_viewdirs=cfg.use_viewdirs, ray_bender=rb, near=cfg.near, far=cfg.far)
for dtype in ("float32", "uint8"):
    render_path(poses[:3], intr[:3], 32768, kw, codes[:3].cuda(), rgb_dtype=dtype)          # warm-up
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    t0 = time.perf_counter()
    rgbs, disps = render_path(poses, intr, 32768, kw, codes.cuda(), rgb_dtype=dtype)
    dt = time.perf_counter() - t0
    free1, _ = torch.cuda.mem_get_info()
    print(f"render_path {frames} frames {W}x{H}, 64+128, bf16{', use_viewdirs + 7-layer bender (config 4)' if config4 else ''}, rgb as {dtype}: {frames / dt:.2f} frames/s = "
          f"{frames * H * W / dt / 1e6:.2f} M rays/s end to end ({dt / frames * 1e3:.1f} ms/frame); device memory delta "
          f"{(free0 - free1) / 2**20:.0f} MiB; host arrays {rgbs.nbytes / 2**20:.0f} + {disps.nbytes / 2**20:.0f} MiB")
