#!/usr/bin/env python
"""GPU box, under `rocprofv3 --kernel-trace --stats`: native training steps (bf16 mode, 64+128 samples) at one batch size."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd import training  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
r = training.bench_train_step(None, SceneConfig(), torch.device("cuda:0"), precision="bf16", n_rays=n, steps=20, warmup=3)
print(f"[bf16] {n} rays/step: {r['ms_per_step']:.3f} ms/step under the profiler")
