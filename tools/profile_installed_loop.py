#!/usr/bin/env python
"""GPU box with the reference staged (tools/with_reference.sh): where the HOST time of the reference's own training loop goes after
install() -- cProfile of `training_wrapper_class` forward + backward + torch.optim.Adam at N_rand = 1024 (the loop of
tests/test_install_reference.py::test_reference_training_loop_is_faster_after_install).    python tools/profile_installed_loop.py [precision] [iters]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
import make_golden as G  # noqa: E402
from nonrigid_nerf_amd import render as R  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig, make_rays, make_scene  # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
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
params = list(coarse.parameters()) + list(fine.parameters()) + list(rb.parameters()) + codes
opt = torch.optim.Adam(params, lr=5e-4, betas=(0.9, 0.999))
R.install(T, precision=precision)
phase = {"fwd": 0.0, "bwd": 0.0, "opt": 0.0}


def it(i, timed=False):
    t0 = time.perf_counter()
    opt.zero_grad()
    loss = wrapper(args, ro, rd, 100, dict(kw), target, ts["global_step"] + i, 0, {"imageid_to_timestepid": list(range(ts["n_frames"]))}, bpi)
    t1 = time.perf_counter()
    loss.mean().backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    if timed:
        phase["fwd"] += t1 - t0; phase["bwd"] += t2 - t1; phase["opt"] += t3 - t2


for i in range(5):
    it(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(iters):
    it(5 + i, True)
torch.cuda.synchronize()
print(f"[{precision}] {(time.perf_counter() - t0) / iters * 1e3:.2f} ms / iteration; host time per phase (ms): " + ", ".join(f"{k} {v / iters * 1e3:.2f}" for k, v in phase.items()))
pr = cProfile.Profile()
pr.enable()
for i in range(iters):
    it(100 + i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
