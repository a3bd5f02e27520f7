#!/usr/bin/env python
"""GPU box: the native training step at the reference's batch size and at larger ones (rays/s, share of the MFMA peak)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd import training  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig  # noqa: E402

cfg = SceneConfig()
for prec in ("bf16", "f32"):
    for n in (1024, 4096, 16384):
        r = training.bench_train_step(None, cfg, torch.device("cuda:0"), precision=prec, n_rays=n, steps=10 if n > 4096 else 30, warmup=3)
        rf = r["roofline"]
        print(f"[{prec}] {n:6d} rays/step (shipped recipe): {r['ms_per_step']:8.3f} ms/step, {r['rays_per_s'] / 1e3:8.1f} k rays/s; saved-array traffic "
              f"{rf['achieved']:7.1f} GB/s ({rf['frac']:.3f} of HBM), {rf['mfma']['achieved']:7.1f} TFLOP/s ({rf['mfma']['frac']:.3f} of the {prec} MFMA peak); "
              f"HIP graph {r['hip_graph'].get('ms_per_step')} ms; data-term-only step {r['data_term_only']['ms_per_step']} ms")
