#!/usr/bin/env python
"""GPU box: the device kernels of ONE native training step (shipped recipe), in launch order, with their durations --
what runs between the big kernels.    python tools/experiments/step_kernel_sequence.py [rays] [precision] [--views]"""
import os
import re
import sys

import torch
from torch.profiler import ProfilerActivity, profile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd import render as R, training  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig, make_rays  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 1024
prec = args[1] if len(args) > 1 else "bf16"
dev = torch.device("cuda:0")
cfg = SceneConfig(use_viewdirs=True) if "--views" in sys.argv else SceneConfig()
rec = training.SHIPPED_RECIPE
rb, coarse, fine = training._fresh_training_modules(cfg, dev, rec["N_importance"])
params = []
for m in (rb, coarse, fine):
    m.requires_grad_(True)
    params += list(m.parameters())
codes = torch.zeros(8, cfg.latent_size, device=dev, requires_grad=True)
opt = torch.optim.Adam(params + [codes], lr=5e-4, fused=True)
rays, _ = make_rays(n, 5, cfg)
rays = rays.to(dev)
frame = torch.randint(0, 8, (n,), device=dev)
target = 0.5 + 0.4 * torch.sin(3.0 * rays[:, 3:6])
kw = dict(network_fn=coarse, network_fine=fine, network_query_fn=None, N_samples=cfg.N_samples, N_importance=rec["N_importance"],
          perturb=rec["perturb"], raw_noise_std=rec["raw_noise_std"])
w = dict(offsets_loss_weight=rec["offsets_loss_weight"], divergence_loss_weight=rec["divergence_loss_weight"], rigidity_loss_weight=rec["rigidity_loss_weight"])
R.set_precision(prec)


def step(i):
    opt.zero_grad(set_to_none=True)
    loss, _ = training.training_loss(rays, training.select_codes(codes, frame), target, kw, global_step=i, N_iters=rec["N_iters"], chunk=rec["chunk"], **w)
    loss.mean().backward()
    opt.step()


for i in range(4):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    step(4)
    torch.cuda.synchronize()
ks = sorted((e for e in prof.events() if e.device_type.name == "CUDA"), key=lambda e: e.time_range.start)
t0 = ks[0].time_range.start
print(f"# {len(ks)} device kernels / copies in one {n}-ray step ({prec}{', use_viewdirs' if cfg.use_viewdirs else ''}); start us, duration us, name")
for e in ks:
    name = re.sub(r"\(anonymous namespace\)::|at::native::|void |<.*", "", e.name)[:70]
    print(f"{e.time_range.start - t0:9.1f} {e.time_range.end - e.time_range.start:8.1f}  {name}")
busy = sum(e.time_range.end - e.time_range.start for e in ks)
print(f"# busy {busy:.0f} us of {ks[-1].time_range.end - t0:.0f} us")
