#!/usr/bin/env python
"""GPU box: the native training step (shipped recipe where the configuration has a ray bender, data term otherwise) of the
other compiled configurations, bf16 mode, 1024 and 16384 rays per step: view-dependent head (finite-difference directions),
7-layer bender, trunk width 128, time-conditioned baseline."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd import training  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig  # noqa: E402

dev = torch.device("cuda:0")
CONFIGS = [("default (5-layer bender, 8 x 256)", dict()),
           ("view-dependent head", dict(use_viewdirs=True)),
           ("7-layer bender + view-dependent head (BASELINE config 4)", dict(use_viewdirs=True, bend_depth=7)),
           ("trunk width 128", dict(netwidth=128)),
           ("time-conditioned baseline (no bender)", dict(ray_bending=False, time_conditioned_baseline=True))]
for name, kw in CONFIGS:
    cfg = SceneConfig(**kw)
    for n in (1024, 16384):
        dt, loss = training._time_training(cfg, dev, "bf16", n, 64, 10 if n > 4096 else 30, 3, regularised=True, repeats=2)
        print(f"{name:58s} {n:6d} rays/step: {dt * 1e3:8.3f} ms/step = {n / dt / 1e3:7.1f} k rays/s   (final loss {loss:.4f})", flush=True)
