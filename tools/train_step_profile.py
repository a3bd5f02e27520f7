#!/usr/bin/env python
"""GPU box, under `rocprofv3 --kernel-trace --stats`: native training steps with the reference's shipped recipe
(configs/example_sequence.txt: 64 + 64 samples, detailed outputs, data + offsets / rigidity + divergence terms; bf16 mode) at
one batch size.    python tools/train_step_profile.py [rays] [precision] [--table] [--views | --exact]  (--table: torch profiler's kernel table)"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd import training  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 1024
prec = args[1] if len(args) > 1 else "bf16"
dev = torch.device("cuda:0")
CFG = SceneConfig(use_viewdirs=True, approx_nonrigid_viewdirs=False) if "--exact" in sys.argv else SceneConfig(use_viewdirs=True) if "--views" in sys.argv else (SceneConfig(ray_bending=False, time_conditioned_baseline=True) if "--tcb" in sys.argv else SceneConfig())   # --views: view-dependent head; --tcb: time-conditioned baseline
_SceneConfig, SceneConfig = SceneConfig, (lambda: CFG)
if "--table" in sys.argv:
    from torch.profiler import ProfilerActivity, profile
    training._time_training(SceneConfig(), dev, prec, n, 64, 3, 3, regularised=True)
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        dt, _ = training._time_training(SceneConfig(), dev, prec, n, 64, 10, 2, regularised=True)
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=80))
else:
    dt, loss = training._time_training(SceneConfig(), dev, prec, n, 64, 20, 3, regularised=True)
    print(f"[{prec}] shipped recipe, {n} rays/step: {dt * 1e3:.3f} ms/step (loss {loss:.5f})")
