#!/usr/bin/env python
"""GPU box: the device kernels / copies of ONE eager training step with the shipped recipe, in launch order (start us, duration us, name)
-- what profiles/rNN_train_step_kernel_sequence_1024.txt holds.    python tools/train_step_sequence.py [rays] [precision] [--torch-adam]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from nonrigid_nerf_amd import training  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 1024
prec = args[1] if len(args) > 1 else "bf16"
dev = torch.device("cuda:0")
torch_adam = "--torch-adam" in sys.argv
# warm everything (kernels loaded, allocator pools grown), then ONE more call of the same driver under the profiler with a single step
training._time_training(SceneConfig(), dev, prec, n, 64, 3, 3, regularised=True, torch_adam=torch_adam)
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    training._time_training(SceneConfig(), dev, prec, n, 64, 1, 3, regularised=True, torch_adam=torch_adam)
ev = sorted([e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
# the last step = the kernels after the last optimiser launch but one; simpler: take the trailing events from the last `vectorized_gather` / first kernel of a step
starts = [i for i, e in enumerate(ev) if "gather" in e.name.lower() and "scatter" not in e.name.lower()]
first = starts[-1] if starts else 0
step = ev[first:]
t0 = step[0].time_range.start
print(f"# {len(step)} device kernels / copies in one {n}-ray step ({prec}, {'torch.optim.Adam(fused)' if torch_adam else 'training.FusedAdam'}); start us, duration us, name")
busy = 0.0
for e in step:
    d = e.time_range.end - e.time_range.start
    busy += d
    print(f"{e.time_range.start - t0:9.1f} {d:8.1f}  {e.name[:110]}")
print(f"# busy {busy:.0f} us of {step[-1].time_range.end - t0:.0f} us")
