#!/usr/bin/env python
"""Per-phase cycle breakdown of net_kernel (bf16, bender, no views) from the NRN_TIMING build.

    make -C nonrigid_nerf_amd/csrc -j8 TUNE=-DNRN_TIMING SUFFIX=_timing
    NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing.so python tools/timing_probe.py

With `--x16`: the 16x16x32 trunk-only kernel (nrnerf_net_x16.h; the object of the 129..192-sample case is enough: nrnerf_net_x16.hip
with -DNRN_X16_EPL=3 -DNRN_TIMING, linked with the shipped objects -- tools/build_timing_x16.sh does that), whose counters also hold the iteration time in 100 MHz ticks,
i.e. the shader clock the workgroup actually ran at.  `--raw`: all eight slots per iteration.
"""
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
rays, lat = rays.cuda(), lat.cuda()
model = R.get_model(coarse, fine)
lib = _lib.load()
if "--x16" in sys.argv:
    fn = lib.nrnerf_debug_timing_launch_net_x16
    fn.argtypes, fn.restype = [C.POINTER(C.c_ulonglong)], C.c_int
    buf = (C.c_ulonglong * 64)()
    names = ["iteration", "points+encoding", "layers+head", "outputs+ring tail", "compositing", "100 MHz ticks", "ring wait+barrier", "iterations"]
    rays, lat = make_rays(262144, 1, cfg)
    rays, lat = rays.cuda(), lat.cuda()
    with torch.no_grad():
        model.render(rays, lat, 64, 128); torch.cuda.synchronize(); fn(buf)
        for rep in range(2):
            model.render(rays, lat, 64, 128); torch.cuda.synchronize(); fn(buf)
            print("== fine pass, 262144 rays x 192 samples: cycles per iteration (4 blocks of 16 samples per wave), workgroup 0")
            for w in range(4):
                row = [buf[w * 8 + i] for i in range(8)]
                n = max(row[7], 1)
                mhz = 100.0 * row[0] / max(row[5], 1)
                if "--raw" in sys.argv:
                    print(f"  wave {w}: slots per iteration " + "  ".join(f"[{i}] {row[i] / n:8.0f}" for i in range(7)) + f"  iterations {row[7]}")
                    continue
                print(f"  wave {w}: " + "  ".join(f"{names[i]} {row[i] / n:8.0f}" for i in (0, 1, 2, 3, 4, 6)) + f"  iterations {row[7]}  clock {mhz:6.0f} MHz")
            t0 = buf[6 * 8]           # (row 6 slot 0: workgroup 0's start; rows 4, 5: when sixteen workgroups spread over the grid left their loops)
            ends = [(buf[4 * 8 + k] - t0) / 100.0 for k in range(16) if k != 8 or True]
            print("  end of the loop of workgroups 0 .. 15 (XCD = index % 8) after workgroup 0's start (us): " + " ".join(f"{e:.0f}" for e in ends))
    sys.exit(0)
fn = lib.nrnerf_debug_timing_launch_net_a0_bf16_bend
fn.argtypes, fn.restype = [C.POINTER(C.c_ulonglong)], C.c_int
buf = (C.c_ulonglong * 64)()
names = ["total", "front(z,pts)", "bender+rigidity", "mask,dirs,encoding", "trunk+head", "stores+pad", "ring wait+barrier", "passes"]
with torch.no_grad():
    model.render(rays, lat, 64, 128); torch.cuda.synchronize(); fn(buf)          # warm-up, clear
    for label, I in (("coarse-only launch (64 samples)", 0), ("64 + 128 (coarse + fine launches)", 128)):
        model.render(rays, lat, 64, I); torch.cuda.synchronize(); fn(buf)
        print(f"== {label}: cycles per 32-sample block pass, workgroup 0")
        for w in range(8):
            row = [buf[w * 8 + i] for i in range(8)]
            n = max(row[7], 1)
            print(f"  wave {w}: " + "  ".join(f"{names[i]} {row[i] / n:9.0f}" for i in range(7)) + f"  passes {row[7]}")
