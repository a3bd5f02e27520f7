#!/usr/bin/env python
"""Build container only: time the UNMODIFIED reference (train.render, `kind: "reference"`) and the oracle port
(oracle/nrnerf_oracle.py, `kind: "port"`) side by side on the same CPU, rays and weights, so that the ratio behind
bench.py's `cpu_baseline.kind = "port"` (the GPU box has no /root/reference) is on record in BASELINE.md."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
import make_golden as G  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig, make_rays, make_scene  # noqa: E402
from oracle import nrnerf_oracle as O  # noqa: E402


def best(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    H, T = G.import_reference()
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count()
    torch.set_num_threads(threads)
    for n in (1024, 4096):
        cfg = SceneConfig()
        scene = make_scene(cfg, 0)
        rays, lat = make_rays(n, 100, cfg)
        kw, rb, coarse, fine = G.reference_kwargs(H, T, scene)
        with torch.no_grad():
            t_ref = best(lambda: T.render(rays[:, 0:3], rays[:, 3:6], chunk=1024,
                                          additional_pixel_information={"ray_bending_latents": lat}, **kw))
            t_port = best(lambda: O.batchify_rays(rays, lat, scene, chunk=1024))
        print(f"{n} rays, 64+128, chunk 1024, {threads} threads, torch {torch.__version__}: reference {n / t_ref:.0f} rays/s, "
              f"port {n / t_port:.0f} rays/s, port/reference = {t_ref / t_port:.3f}")


if __name__ == "__main__":
    main()
