#!/usr/bin/env python
"""Per-phase cycle breakdown of bend_kernel_x16 from its NRN_TIMING build (tools/build_timing_bender.sh):
    NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing_bend.so python tools/timing_probe_bender.py
Workgroup 0's eight waves; cycles per iteration (NB blocks of 16 samples per wave)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nonrigid_nerf_amd import _lib, render as R
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene

cfg = SceneConfig()
scene = make_scene(cfg, 0)
rb, coarse, fine = build_modules(scene, device="cuda:0")
R.set_precision("bf16")
rays, lat = make_rays(196608, 1, cfg)
rays, lat = rays.cuda(), lat[:1].cuda().expand(rays.shape[0], -1)      # one frame code for every ray (latent_stride = 0)
model = R.get_model(coarse, fine)
lib = _lib.load()
fn = lib.nrnerf_debug_timing_bend_x16
fn.argtypes, fn.restype = [C.POINTER(C.c_ulonglong)], C.c_int
buf = (C.c_ulonglong * 64)()
names = ["iteration", "operands", "offset MLP (5 layers)", "rigidity MLP (3 layers)", "tail + stores", "100 MHz ticks", "-", "iterations"]
with torch.no_grad():
    model.render(rays, lat, 64, 128); torch.cuda.synchronize(); fn(buf)
    for rep in range(2):
        model.render(rays, lat, 64, 128); torch.cuda.synchronize(); fn(buf)
        print("== both bender launches of one 196608-ray frame (64 coarse + 128 new samples per ray), workgroup 0: cycles per iteration and wave")
        t0 = buf[6 * 8]
        span = lambda r, k: f"{(buf[r * 8 + k] - t0) / 100.0:.1f}..{(buf[r * 8 + k + 1] - t0) / 100.0:.1f} us"
        print(f"   (LAST launch) wave 0 of workgroup 0: {span(6, 0)}; 128: {span(7, 2)}; 255: {span(6, 2)}; 256: {span(6, 4)}; 384: {span(6, 6)}; last: {span(7, 0)}")
        for w in range(6):
            row = [buf[w * 8 + k] for k in range(8)]
            n = max(1, row[7])
            ghz = (row[0] / n) / (row[5] / n * 10.0) if row[5] else 0.0
            print(f"wave {w}: " + ", ".join(f"{names[k]} {row[k] / n:.0f}" for k in (0, 1, 2, 3, 4)) + f"; {n} iterations; shader clock {ghz:.2f} GHz")
