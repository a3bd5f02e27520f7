#!/usr/bin/env python
"""Headline benchmark: rendered rays/s of the NR-NeRF per-ray hot path at 64+128 samples/ray.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one ``batchify_rays`` pass (reference train.py:108-137) over one 512x384 frame worth of
synthetic rays per GPU (196 608 rays = the reference's 6 chunks of 32 768; BASELINE.json config 2:
64 coarse + 128 importance samples, 8x256 canonical MLPs + ray bender, bf16 contractions with fp32
accumulation) with inputs already resident in HBM, plus -- for N > 1 -- the all-gather of the rendered
pixels over RCCL/xGMI (config 3).  Weak scaling: every rank renders its own frame-sized shard.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  "roofline":     the dominant kernel (fine-pass network kernel) against the dense bf16 MFMA peak, from
                  HIP events recorded around that kernel on the render stream during the timed steps;
  "cpu_baseline": the CPU oracle (a PyTorch-CPU port of the reference path, oracle/nrnerf_oracle.py)
                  timed on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}       # dense MFMA, MI355X_MICROARCH.md
ALGO_MFLOP_PER_RAY = 260.18                                        # SURVEY.md section 8d (64 + 192 evaluations)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=196608, help="rays per GPU per step (default: one 512x384 frame)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--cpu-rays", type=int, default=8192, help="rays of the same workload timed on the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--use-viewdirs", action="store_true", help="BASELINE config 4: view-dependent head (not the headline config)")
    ap.add_argument("--bend-depth", type=int, default=5, help="BASELINE config 4: deeper ray-bending MLP (5 or 7)")
    ap.add_argument("--exact-viewdirs", action="store_true", help="with --use-viewdirs: Jacobian instead of finite-difference directions")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --rays per GPU; strong: --rays in total, sharded contiguously over the ranks "
                         "(BASELINE config 3: one frame over 8 GPUs)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # NRNERF_BENCH_ONE_GPU=1 (testing aid for a 1-GPU box): every rank renders on cuda:0 and the pixels are gathered over
    # gloo, so the multi-process path of this script can be exercised without a multi-GPU node.  Never a benchmark.
    one_gpu = os.environ.get("NRNERF_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:       # N processes building their synthetic scene on the host at once: do not oversubscribe the cores
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 8) // (2 * world))))

    from nonrigid_nerf_amd import render as R
    from nonrigid_nerf_amd.distributed import gather_pixels
    from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene

    if args.scaling == "strong":                          # one frame for the whole job: ceil(n / G) rays per rank
        args.rays = (args.rays + world - 1) // world
    cfg = SceneConfig(use_viewdirs=args.use_viewdirs, bend_depth=args.bend_depth,      # default: 64 + 128, W = 256, bender on, latent 32
                      approx_nonrigid_viewdirs=not args.exact_viewdirs)
    scene = make_scene(cfg, 0)
    rb, coarse, fine = build_modules(scene, device=dev)
    R.set_precision(args.precision)
    rays, latents = make_rays(args.rays, seed=100 + rank, cfg=cfg)
    rays, latents = rays.to(dev), latents.to(dev)
    api = {"ray_bending_latents": latents}
    kw = dict(network_fn=coarse, network_fine=fine, network_query_fn=None, N_samples=cfg.N_samples,
              N_importance=cfg.N_importance, perturb=0.0, raw_noise_std=0.0)
    model = R.get_model(coarse, fine, device=dev)          # weights packed once, outside the timed region

    def step():
        out = R.batchify_rays(rays, api, chunk=1024 * 32, **kw)
        packed = torch.cat([out["rgb_map"], out["disp_map"][:, None], out["acc_map"][:, None]], -1)
        if world > 1 and one_gpu:
            return gather_pixels(packed.cpu())          # gloo has no all_gather for device tensors
        return gather_pixels(packed) if world > 1 else packed

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        barrier()
        model.profile_begin()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            img = step()
        barrier()
        dt = time.perf_counter() - t0
        prof = model.profile_end()
    assert img.shape == (world * args.rays, 5)

    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_rays = world * args.rays * args.steps
    value = total_rays / dt

    if rank == 0:
        k = prof["net_fine"]
        ach = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
        peak = PEAK_TFLOPS[args.precision]
        roofline = {"bound": "mfma", "kernel": "net_kernel (fine pass, 192 samples/ray)",
                    "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "avg_launch_ms": round(k["ms"] / max(k["launches"], 1), 4),
                    "issued_mfma_tflops": round(k["mfma_flops"] / (k["ms"] * 1e-3) / 1e12, 2) if k["ms"] > 0 else 0.0,
                    "traffic": pmc_traffic(args),
                    "traffic_unit": "bytes per launch (PMC, profiles/r01_pmc_summary.txt); algorithmic 7.86e8",
                    "other_kernels_ms_per_step": {n: round(v["ms"] / args.steps, 4) for n, v in prof.items()},
                    # context, not the peak: what a plain hipBLASLt GEMM sustains on this box right now (the chip
                    # clocks down under MFMA load)
                    "library_gemm_tflops_same_box": library_gemm_tflops(dev, args.precision) if world == 1 else None}
        res = {"metric": "rendered rays/sec (64+128 samples/ray)", "value": round(value, 1), "unit": "rays/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
               "vs_baseline": None, "dtype": args.precision, "data": "synthetic" + (" (NOT A BENCHMARK: all ranks on one GPU, gloo)" if one_gpu else ""),
               "config": {"workload": "BASELINE config 2: example_sequence-shaped frame (512x384 = 196608 rays/GPU/step), "
                                      "64 coarse + 128 importance samples, netwidth 256, ray bender on, latent 32"
                                      + (f"; NON-HEADLINE VARIANT: use_viewdirs={args.use_viewdirs}, bend_depth={args.bend_depth}, "
                                         f"exact_viewdirs={args.exact_viewdirs}" if (args.use_viewdirs or args.bend_depth != 5) else ""),
                          "rays_per_gpu_per_step": args.rays, "N_samples": 64, "N_importance": 128,
                          "parallelism": f"rays sharded over {world} rank(s)" + (", all-gather of [rgb,disp,acc] over RCCL" if world > 1 else "")},
               "mflop_per_ray_algorithmic": round(sum(v["flops"] for v in prof.values()) / max(args.rays * args.steps, 1) / 1e6, 2),
               "end_to_end_tflops": round(value * sum(v["flops"] for v in prof.values()) / max(args.rays * args.steps, 1) / 1e12, 2),
               "roofline": roofline}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(scene, cfg, args.cpu_rays)
        print(json.dumps(res), flush=True)

    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def library_gemm_tflops(dev, precision):
    """8192^3 GEMM through torch (hipBLASLt) in the kernel's input type, ~30 ms; None if it cannot run."""
    try:
        dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[precision]
        n = 8192 if precision != "f32" else 4096
        a = torch.randn(n, n, device=dev, dtype=dt)
        b = torch.randn(n, n, device=dev, dtype=dt)
        for _ in range(3):
            a @ b
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            a @ b
        torch.cuda.synchronize()
        return round(2.0 * n ** 3 * 20 / (time.perf_counter() - t0) / 1e12, 1)
    except Exception:
        return None


def pmc_traffic(args):
    """HBM bytes per launch of the fine-pass network kernel for THIS workload, from the committed rocprofv3 --pmc passes
    (FETCH_SIZE x 2 + WRITE_SIZE, tools/pmc_summary.py; counters cannot be read from inside the process).  None when
    the run does not match the profiled configuration."""
    path = os.path.join(REPO, "profiles", "r01_pmc_fine.json")
    if args.rays != 196608 or args.precision != "bf16" or not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            return float(json.load(f)["fine"]["hbm_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(scene, cfg, n):
    """The CPU oracle (PyTorch-CPU port of reference render_rays/batchify_rays) on ``n`` rays of the same workload."""
    from nonrigid_nerf_amd.synthetic import make_rays
    from oracle import nrnerf_oracle as O
    rays, latents = make_rays(n, seed=100, cfg=cfg)
    # torch's default (one thread per logical core) oversubscribes a 256-thread host badly (measured 277 rays/s
    # at 128 threads vs ~1300 at 8); probe a few thread counts on a small sample and report the best.
    best_t, best_r = torch.get_num_threads(), 0.0
    with torch.no_grad():
        for t in sorted({8, 16, 32, 64} & set(range(1, (os.cpu_count() or 8) + 1))):
            torch.set_num_threads(t)
            O.batchify_rays(rays[:512], latents[:512], scene, chunk=512)      # warm-up
            t0 = time.perf_counter()
            O.batchify_rays(rays[:1024], latents[:1024], scene, chunk=1024)
            r = 1024 / (time.perf_counter() - t0)
            if r > best_r:
                best_t, best_r = t, r
        torch.set_num_threads(best_t)
        threads = best_t
        t0 = time.perf_counter()
        O.batchify_rays(rays, latents, scene, chunk=1024)
        dt = time.perf_counter() - t0
    return {"value": round(n / dt, 1), "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"{n} rays of the same 64+128 workload, chunk 1024, torch {torch.__version__} CPU, "
                      f"{threads} threads of {os.cpu_count()} host cores, {dt:.1f} s"}


if __name__ == "__main__":
    main()
