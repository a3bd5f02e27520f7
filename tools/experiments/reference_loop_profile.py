#!/usr/bin/env python
"""GPU box, reference staged (tools/with_reference.sh): kernel / operator table of the reference's own training loop after
install(train, precision="bf16") -- training_wrapper_class + backward + torch.optim.Adam on its real modules, 1024 rays."""
import argparse
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
import make_golden as G  # noqa: E402
from nonrigid_nerf_amd import render as R  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig, make_rays, make_scene  # noqa: E402

H, T = G.import_reference()
dev = torch.device("cuda:0")
T.device = dev
ts = G.TRAIN_STEP
n_rays = 1024
cfg = SceneConfig(N_importance=ts["N_importance"])
scene = make_scene(cfg, ts["seed"])
rays, _ = make_rays(n_rays, ts["seed"], cfg)
g = torch.Generator().manual_seed(11)
codes0 = torch.randn(ts["n_frames"], cfg.latent_size, generator=g) * 0.1
image_ids = torch.randint(0, ts["n_frames"], (n_rays,), generator=g)
target = torch.rand(n_rays, 3, generator=g).to(dev)
args = argparse.Namespace(offsets_loss_weight=ts["offsets_loss_weight"], divergence_loss_weight=ts["divergence_loss_weight"],
                          rigidity_loss_weight=ts["rigidity_loss_weight"], chunk=ts["chunk"], N_iters=ts["N_iters"],
                          N_samples=ts["N_samples"], ray_bending_latent_size=cfg.latent_size)
bpi = torch.stack([image_ids, torch.zeros_like(image_ids), torch.zeros_like(image_ids)], 1)
ro, rd = rays[:, 0:3].to(dev), rays[:, 3:6].to(dev)
kw, rb, coarse, fine = G.reference_kwargs(H, T, scene)
for m in (rb, coarse, fine):
    m.to(dev)
kw.update(perturb=ts["perturb"], raw_noise_std=ts["raw_noise_std"])
codes = [c.clone().to(dev).requires_grad_(True) for c in codes0]
wrapper = T.training_wrapper_class(coarse, codes, fine_model=fine, ray_bender=rb)
opt = torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()) + list(rb.parameters()) + codes, lr=5e-4)
R.install(T, precision="bf16")


def it(i):
    opt.zero_grad()
    loss = wrapper(args, ro, rd, 100, dict(kw), target, ts["global_step"] + i, 0, {"imageid_to_timestepid": list(range(ts["n_frames"]))}, bpi)
    loss.mean().backward()
    opt.step()


for i in range(5):
    it(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for i in range(10):
        it(5 + i)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=70))
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
