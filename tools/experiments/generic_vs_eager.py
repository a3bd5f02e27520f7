"""A non-compiled architecture (--netdepth 6 --netwidth 192 / --netwidth_fine 320, 8 frequencies, latent 16): the eager PyTorch path the
reference takes on this GPU (the oracle's ops, fp32) against the run-time-parameterised kernel.  python tools/experiments/generic_vs_eager.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nonrigid_nerf_amd import render as R
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene
from oracle import nrnerf_oracle as O

dev = "cuda:0"
cfg = SceneConfig(N_importance=128, netdepth=6, netwidth=192, netdepth_fine=10, netwidth_fine=320, multires=8, latent_size=16)
scene = make_scene(cfg, 0)
n = 32768
rays, lat = make_rays(n, 3, cfg)
rays, lat = rays.to(dev), lat.to(dev)
sc = O.scene_on(scene, dev)
rb, coarse, fine = build_modules(scene, device=dev)


def rate(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return n * reps / (time.perf_counter() - t0)


with torch.no_grad():
    eager = rate(lambda: O.batchify_rays(rays, lat, sc, chunk=32768), 3)
    out = {}
    for prec in ("f32", "bf16"):
        R.set_precision(prec)
        out[prec] = rate(lambda: R.batchify_rays(rays, {"ray_bending_latents": lat}, network_fn=coarse, network_fine=fine, N_samples=64, N_importance=128), 10)
print(f"[D 6 / W 192 coarse, D 10 / W 320 fine, 64 + 128 samples, {n} rays] eager fp32 torch on this GPU {eager / 1e6:.3f} M rays/s; "
      f"generic kernel f32 {out['f32'] / 1e6:.3f} M ({out['f32'] / eager:.1f} x), bf16 {out['bf16'] / 1e6:.3f} M ({out['bf16'] / eager:.1f} x)")
