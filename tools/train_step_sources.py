#!/usr/bin/env python
"""GPU box: which torch op -- and which autograd Function around it -- issues each device launch of ONE eager training step (shipped recipe) -- the work list
for folding the step's small launches.    python tools/train_step_sources.py [rays] [precision]"""
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
training._time_training(SceneConfig(), dev, prec, n, 64, 3, 3, regularised=True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    training._time_training(SceneConfig(), dev, prec, n, 64, 1, 0, regularised=True)
rows = []
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    # only the innermost op that owns the kernels (children repeat them)
    if any(c.kernels for c in (e.cpu_children or [])):
        continue
    chain, q = [], e.cpu_parent
    while q is not None and len(chain) < 6:
        if not q.name.startswith(("autograd::engine", "ProfilerStep")):
            chain.append(q.name[:48])
        q = q.cpu_parent
    where = " < ".join(chain) or "(top level)"
    for k in e.kernels:
        rows.append((e.time_range.start, k.duration, k.name[:60], e.name[:40], where[:110]))
rows.sort()
print(f"# {len(rows)} launches owned by a CPU op with a stack (library launches through ctypes appear under the autograd Function that made them, if at all)")
for t, d, k, op, where in rows:
    print(f"{d:7.1f}  {k:60s}  {op:40s}  {where}")
