#!/usr/bin/env python
"""GPU box: where the small launches of a native training step come from -- per (source line in this package, aten op) the
number of device kernels per step, from torch.profiler with stacks.    python tools/experiments/op_inventory.py [rays]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd import training  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
STEPS = 4
training._time_training(SceneConfig(), dev, "bf16", n, 64, 2, 2, regularised=True)
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    training._time_training(SceneConfig(), dev, "bf16", n, 64, STEPS, 0, regularised=True)
ev = prof.events()
# kernels are children (by correlation) of the CPU op that launched them: walk CPU ops that have device kernels
by = collections.Counter()
t_by = collections.Counter()
for e in ev:
    if e.device_type.name != "CPU" or not e.kernels:
        continue
    if e.cpu_children:          # count at the leaf op only
        if any(c.kernels for c in e.cpu_children):
            continue
    where = "?"
    for fr in (e.stack or []):
        if "nonrigid_nerf_amd/" in fr:
            where = fr.split("nonrigid_nerf_amd/")[-1].strip()
            break
    if where == "?" and e.stack:
        where = "| " + e.stack[0].strip()[-60:]
    by[(where, e.name)] += len(e.kernels)
    t_by[(where, e.name)] += sum(k.duration for k in e.kernels)
tot = sum(by.values())
print(f"{n} rays: {tot / STEPS:.1f} device kernels per step (incl. warm-up-free steps), {sum(t_by.values()) / STEPS:.1f} us")
for (where, name), c in sorted(by.items(), key=lambda kv: -t_by[kv[0]]):
    print(f"{c / STEPS:7.2f} /step {t_by[(where, name)] / STEPS:9.1f} us  {name:40s} {where}")
if "--fills" in sys.argv:
    shown = 0
    for e in ev:
        if e.device_type.name == "CPU" and e.kernels and e.name in ("aten::fill_", "aten::add", "aten::add_") and shown < 14:
            st = [fr.strip()[-70:] for fr in (e.stack or [])][:12]
            if any("step" in fr for fr in st) and not any("query" in fr or "training_loss" in fr for fr in st):
                print("----", e.name, [tuple(s) for s in (e.input_shapes or [])][:3])
                print("\n".join("      " + fr for fr in st))
                shown += 1
