#!/usr/bin/env python
"""GPU box: autograd node histogram of one native training iteration's loss, and forward ops by source line."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd import render as R, training  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig, make_rays  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
cfg = SceneConfig()
rec = training.SHIPPED_RECIPE
rb, coarse, fine = training._fresh_training_modules(cfg, dev, 64)
for m in (rb, coarse, fine):
    m.requires_grad_(True)
codes = torch.zeros(8, cfg.latent_size, device=dev, requires_grad=True)
rays, _ = make_rays(n, 5, cfg)
rays = rays.to(dev)
frame = torch.randint(0, 8, (n,), device=dev)
target = 0.5 + 0.4 * torch.sin(3.0 * rays[:, 3:6])
kw = dict(network_fn=coarse, network_fine=fine, network_query_fn=None, N_samples=cfg.N_samples, N_importance=64, perturb=rec["perturb"], raw_noise_std=rec["raw_noise_std"])
w = dict(offsets_loss_weight=rec["offsets_loss_weight"], divergence_loss_weight=rec["divergence_loss_weight"], rigidity_loss_weight=rec["rigidity_loss_weight"])
R.set_precision("bf16")


def fwd():
    loss, _ = training.training_loss(rays, codes[frame], target, kw, global_step=1000, N_iters=rec["N_iters"], chunk=rec["chunk"], **w)
    return loss.mean()


fwd().backward()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof:
    loss = fwd()
    torch.cuda.synchronize()
seen, hist = set(), collections.Counter()
stack = [loss.grad_fn]
while stack:
    f = stack.pop()
    if f is None or f in seen:
        continue
    seen.add(f)
    hist[type(f).__name__] += 1
    stack += [g for g, _ in f.next_functions]
print("autograd nodes:", sum(hist.values()))
for k, v in hist.most_common():
    print(f"  {v:4d} {k}")
by, t_by = collections.Counter(), collections.Counter()
for e in prof.events():
    if e.device_type.name != "CPU" or not e.kernels:
        continue
    if any(c.kernels for c in (e.cpu_children or [])):
        continue
    where = "?"
    for fr in (e.stack or []):
        if "nonrigid_nerf_amd/" in fr:
            where = fr.split("nonrigid_nerf_amd/")[-1].strip()
            break
    by[(where, e.name)] += len(e.kernels)
    t_by[(where, e.name)] += sum(k.duration for k in e.kernels)
print(f"forward: {sum(by.values())} device kernels, {sum(t_by.values()):.1f} us")
for (where, name), c in sorted(by.items(), key=lambda kv: kv[0]):
    print(f"{c:4d} {t_by[(where, name)]:8.1f} us  {name:28s} {where}")
