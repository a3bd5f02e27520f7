#!/usr/bin/env python
"""GPU box: where the HOST time of an eager native training step goes (cProfile over 60 steps at 1024 rays)."""
import cProfile
import os
import pstats
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd import training  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig  # noqa: E402

dev = torch.device("cuda:0")
training._time_training(SceneConfig(), dev, "bf16", 1024, 64, 5, 3, regularised=True)
pr = cProfile.Profile()
pr.enable()
dt, _ = training._time_training(SceneConfig(), dev, "bf16", 1024, 64, 60, 3, regularised=True)
pr.disable()
print(f"{dt * 1e3:.3f} ms/step under cProfile")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(40)
