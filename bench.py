#!/usr/bin/env python
"""Headline benchmark: rendered rays/s of the NR-NeRF per-ray hot path at 64+128 samples/ray.

  python bench.py                                   # 1 GPU
  python bench.py --gpus 8 --steps 20 --warmup 5    # spawns 8 ranks itself (torch.distributed.run, RCCL)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W     # or under an external launcher (RANK / WORLD_SIZE in the env)

A "step" is one ``batchify_rays`` pass (reference train.py:108-137) over one 512x384 frame worth of rays per GPU
(196 608 rays = the reference's 6 chunks of 32 768; BASELINE.json config 2: 64 coarse + 128 importance samples,
8x256 canonical MLPs + ray bender, bf16 contractions with fp32 accumulation) with inputs already resident in HBM,
plus -- for N > 1 -- the all-gather of the rendered pixels over RCCL/xGMI (config 3), issued on a side stream so
that frame f's gather overlaps frame f+1's render.  Weak scaling: every rank renders its own frame-sized shard
(``--scaling strong``: one frame sharded contiguously over the ranks).

``--scene fitted`` (default): the weights of tests/golden/fitted_latest.tar (NR-NeRF fitted to the down-sampled example
sequence by oracle/fit_checkpoint.py: trained-like weight statistics) and real camera rays of the example sequence at
512x384, one latent code per frame -- the reference's free-viewpoint render of that sequence.
``--scene synthetic``: seeded random weights + random rays (nonrigid_nerf_amd/synthetic.py; the round-1 workload, a
numerical stress scene: sigma logits ~ N(-2, 6^2) everywhere, so 16-bit rounding flips background decisions).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  "roofline":          the dominant kernel (fine-pass network kernel) against the dense MFMA peak of the dtype, from HIP
                       events recorded around that kernel on the render stream during the timed steps;
  "psnr_vs_oracle_db": PSNR (free_viewpoint_rendering.py:821-828) of this run's precision against the fp32 oracle
                       render of the same rays and weights (all rays, no exclusions), checker use of oracle/ only;
  "train_step":        the reference's training iteration at its batch size (1024 rays, perturb + raw noise, forward +
                       backward + Adam step + device-side weight refresh) through the same boundary under autograd, with its
                       own roofline entry; not part of the timed region.  Top-level figures: the iteration replayed from one HIP
                       graph (how the library runs it; also under "hip_graph"); the host-bound eager Python loop: "eager_loop";
  "cpu_baseline":      the CPU oracle (a PyTorch-CPU port of the reference path, oracle/nrnerf_oracle.py) timed on this
                       box's host cores on a bounded sample of the same workload (after the GPU section).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}       # dense MFMA, MI355X_MICROARCH.md
SUSTAINED_MFMA_TFLOPS = {"bf16": (2046.0, 2190.0), "f16": (2046.0, 2190.0)}   # measured: profiles/r04_mfma_shape_power.txt (power-capped clock)
FITTED = os.path.join(REPO, "tests", "golden", "fitted_latest.tar")
FIXTURE = os.path.join(REPO, "tests", "golden", "example_sequence_96x72.npz")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=196608, help="rays per GPU per step (default: one 512x384 frame)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--scene", default="fitted" if os.path.exists(FITTED) else "synthetic", choices=["synthetic", "fitted"])
    ap.add_argument("--cpu-rays", type=int, default=4096, help="rays of the same workload timed on the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-psnr", action="store_true")
    ap.add_argument("--psnr-rays", type=int, default=32768)
    ap.add_argument("--min-gpu-seconds", type=float, default=6.0,
                    help="keep the GPU busy at least this long in total (untimed extra frames after the timed region) so "
                         "that an external utilisation sampler sees the run")
    ap.add_argument("--use-viewdirs", action="store_true", help="BASELINE config 4: view-dependent head (not the headline config)")
    ap.add_argument("--bend-depth", type=int, default=5, help="BASELINE config 4: deeper ray-bending MLP (5 or 7)")
    ap.add_argument("--netwidth", type=int, default=256, help="trunk width of both networks (256 = headline; 128 is the other compiled width)")
    ap.add_argument("--exact-viewdirs", action="store_true", help="with --use-viewdirs: Jacobian instead of finite-difference directions")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --rays per GPU; strong: --rays in total, sharded contiguously over the ranks "
                         "(BASELINE config 3: one frame over 8 GPUs)")
    ap.add_argument("--chunk", type=int, default=1024 * 32, help="batchify_rays' chunk argument (the reference's memory bound; the HIP "
                    "path treats it as a lower bound of its launch size)")
    ap.add_argument("--max-rays-per-launch", type=int, default=0, help="cap the rays of one nrnerf_render launch sequence (default: 2^20; "
                    "BASELINE config 5 words its workload as 65 536-ray chunks: --chunk 65536 --max-rays-per-launch 65536)")
    ap.add_argument("--frames", type=int, default=0,
                    help="frame-sharded sequence mode (NOT the headline line): a step = driver.render_path over F frames of 512x384 "
                         "(example-sequence poses, one latent code per frame), rank r rendering frames r, r + G, ... with its own packed "
                         "weights; no collective while rendering, ONE all-gather of the uint8 frames at the end of each step")
    ap.add_argument("--no-train-step", action="store_true",
                    help="skip the train_step leg (the native training iteration: 1024 rays forward + backward + Adam, untimed part of the run)")
    return ap.parse_args()


def spawn_ranks(args):
    """``--gpus N`` without a launcher: re-exec this script under torch.distributed.run, one rank per GPU."""
    one_gpu = os.environ.get("NRNERF_BENCH_ONE_GPU") == "1"
    have = torch.cuda.device_count()
    if not one_gpu and have < args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible (NRNERF_BENCH_ONE_GPU=1 runs all ranks on "
                 f"GPU 0 over gloo, as a functional test only)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 8) // (2 * args.gpus)))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def build_workload(args, rank, world, dev, frame_rays=None, scaling=None):
    """(scene, cfg, modules, rays, latents-as-passed, description).  Inputs resident in HBM on return.
    weak: every rank gets its own frame of ``frame_rays`` rays; strong: ONE frame of ``frame_rays`` rays, rank r gets the
    contiguous slice [r * ceil(n / G), (r + 1) * ceil(n / G)) of it (DataParallel's scatter rule, distributed.shard_bounds)."""
    from nonrigid_nerf_amd.distributed import shard_bounds
    from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene
    frame_rays = frame_rays or args.rays
    scaling = scaling or args.scaling
    strong = scaling == "strong" and world > 1
    lo, hi, per = shard_bounds(frame_rays, world, rank) if strong else (0, frame_rays, frame_rays)
    n = frame_rays
    if args.scene == "fitted":
        import numpy as np
        from nonrigid_nerf_amd.checkpoint import load_checkpoint
        from nonrigid_nerf_amd.driver import generate_rays
        from nonrigid_nerf_amd.synthetic import Scene
        ck = load_checkpoint(args.fitted_ckpt, N_samples=64, N_importance=128)
        z = np.load(FIXTURE)
        near, far = float(z["bds"].min()) * 0.9, float(z["bds"].max())
        cfg = SceneConfig(near=near, far=far, use_viewdirs=args.use_viewdirs, bend_depth=args.bend_depth, netwidth=args.netwidth)
        sd = lambda m: {k: v.detach().clone() for k, v in m.state_dict().items()}
        scene = Scene(cfg, sd(ck.ray_bender), sd(ck.network_fn), sd(ck.network_fine))
        rb, coarse, fine = ck.ray_bender, ck.network_fn, ck.network_fine
        # one full-resolution frame of the sequence per rank (512x384; the fixture's intrinsics are 96x72)
        frame = (3 + (0 if strong else rank)) % int(z["poses"].shape[0])
        s = 512.0 / float(z["hwf"][1])
        intrin = dict(height=384, width=512, focal_x=float(z["hwf"][2]) * s, focal_y=float(z["hwf"][2]) * s,
                      center_x=256.0, center_y=192.0)
        rays = generate_rays(torch.from_numpy(z["poses"][frame]), intrin, near, far, bool(args.use_viewdirs), dev)
        reps = (n + rays.shape[0] - 1) // rays.shape[0]
        rays = rays.repeat(reps, 1)[:n]
        rays = _pad_rows(rays[lo:hi], per).contiguous()   # strong: this rank's slice (the last rank's padded to equal blocks)
        code = ck.latents[frame].to(dev).reshape(1, -1)
        latents = code.expand(per, -1)                   # stride-0 view, as render_path passes it (train.py:464-466)
        desc = (f"example_sequence frame {int(z['frame_ids'][frame])} camera rays at 512x384, weights fitted to the "
                f"down-sampled sequence ({ck.global_step} oracle iterations, tests/golden/{os.path.basename(args.fitted_ckpt)}), one latent per frame")
        return scene, cfg, (rb, coarse, fine), rays, latents, desc
    cfg = SceneConfig(use_viewdirs=args.use_viewdirs, bend_depth=args.bend_depth,      # default: 64 + 128, W = 256, bender on, latent 32
                      approx_nonrigid_viewdirs=not args.exact_viewdirs, netwidth=args.netwidth)
    scene = make_scene(cfg, 0)
    mods = build_modules(scene, device=dev)
    rays, latents = make_rays(n, seed=100 + (0 if strong else rank), cfg=cfg)
    rays, latents = _pad_rows(rays[lo:hi], per), _pad_rows(latents[lo:hi], per)
    return scene, cfg, mods, rays.to(dev), latents.to(dev), "seeded random weights and rays (nonrigid_nerf_amd/synthetic.py)"


def _pad_rows(t, rows):
    """The all-gather moves equal blocks: a short last shard is padded by repeating its last row (rendered, then ignored)."""
    if t.shape[0] == rows:
        return t
    return torch.cat([t, t[-1:].expand(rows - t.shape[0], -1)], 0)


def fitted_checkpoint_for(args):
    gold = os.path.join(REPO, "tests", "golden")
    if args.exact_viewdirs:
        return None
    if not args.use_viewdirs and args.bend_depth == 5 and args.netwidth == 256:
        name = "fitted_latest.tar"
    elif args.use_viewdirs and args.bend_depth == 7 and args.netwidth == 256:
        name = "fitted_config4.tar"
    elif not args.use_viewdirs and args.bend_depth == 5 and args.netwidth == 128:
        name = "fitted_w128.tar"
    else:
        return None
    path = os.path.join(gold, name)
    return path if os.path.exists(path) else None


def main():
    args = parse_args()
    # a fitted checkpoint exists for each compiled architecture FAMILY (oracle/fit_checkpoint.py --arch ...): the default one,
    # BASELINE config 4 (view-dependent head + 7-layer bender) and width 128; any other variant runs on the synthetic stress scene
    args.fitted_ckpt = fitted_checkpoint_for(args)
    if args.fitted_ckpt is None:
        args.scene = "synthetic"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # NRNERF_BENCH_ONE_GPU=1 (testing aid for a 1-GPU box): every rank renders on cuda:0 and the pixels are gathered over
    # gloo, so the multi-process path of this script can be exercised without a multi-GPU node.  Never a benchmark.
    one_gpu = os.environ.get("NRNERF_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        backend = "gloo" if one_gpu else "nccl"
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:       # N processes building their synthetic scene on the host at once: do not oversubscribe the cores
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 8) // (2 * world))))

    from nonrigid_nerf_amd import render as R

    if args.frames > 0:
        frames_mode(args, rank, world, dev, one_gpu, backend)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    frame_rays = args.rays                                # weak: per rank; strong: for the whole job (ceil(n / G) per rank)
    scene, cfg, (rb, coarse, fine), rays, latents, data_desc = build_workload(args, rank, world, dev)
    args.rays = int(rays.shape[0])
    R.set_precision(args.precision)
    if args.max_rays_per_launch > 0:
        R._MAX_RAYS_PER_LAUNCH = int(args.max_rays_per_launch)
    api = {"ray_bending_latents": latents}
    kw = dict(network_fn=coarse, network_fine=fine, network_query_fn=None, N_samples=cfg.N_samples,
              N_importance=cfg.N_importance, perturb=0.0, raw_noise_std=0.0)
    model = R.get_model(coarse, fine, device=dev)          # weights packed once, outside the timed region

    # ---- multi-GPU: frame f's all-gather runs on a side stream while frame f+1 renders (distributed.OverlappedGather)
    n = args.rays
    if world > 1:
        import torch.distributed as dist

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(rays_, api_, n_, profile):
        """args.warmup untimed + args.steps timed passes over `rays_` (n_ rays on this rank), bracketed by barrier +
        synchronize; returns (max over ranks of the wall time, this rank's own wall time, kernel profile, last image)."""
        gather = None
        if world > 1:
            from nonrigid_nerf_amd.distributed import OverlappedGather
            gather = OverlappedGather(n_, "cpu" if one_gpu else dev)      # gloo (the one-GPU functional mode) gathers host tensors

        def step(i):
            out = R.batchify_rays(rays_, api_, chunk=args.chunk, **kw)
            if gather is None:
                return out
            if one_gpu:
                out = {k: out[k].cpu() for k in ("rgb_map", "disp_map", "acc_map")}
            return gather.submit(i, out)

        with torch.no_grad():
            for i in range(args.warmup):
                step(i)
            if gather is not None:
                gather.drain()
            barrier()
            if profile:
                model.profile_begin()
            t0 = time.perf_counter()
            for i in range(args.steps):
                img = step(i)
            if gather is not None:
                gather.drain()
            barrier()
            dt_own = time.perf_counter() - t0
            prof_ = model.profile_end() if profile else None
        dt_max, per_rank = dt_own, [dt_own]
        if world > 1:
            assert img.shape == (world * n_, 5)
            t = torch.tensor([dt_own], dtype=torch.float64, device="cpu" if one_gpu else dev)
            allt = torch.empty(world, dtype=torch.float64, device=t.device)
            dist.all_gather_into_tensor(allt, t)
            per_rank = [float(x) for x in allt.cpu()]
            dt_max = max(per_rank)
        return dt_max, per_rank, prof_, step

    main_stream = torch.cuda.current_stream(dev)
    t_gpu0 = time.perf_counter()
    dt, per_rank_dt, prof, step = timed_region(rays, api, n, profile=True)
    # the other scaling of the same job, for N > 1 (one JSON line carries both): the primary record is `args.scaling`
    other = None
    if world > 1:
        other_scaling = "strong" if args.scaling == "weak" else "weak"
        _, _, _, rays_o, latents_o, _ = build_workload(args, rank, world, dev, frame_rays=frame_rays, scaling=other_scaling)
        n_o = int(rays_o.shape[0])
        dt_o, per_rank_o, _, _ = timed_region(rays_o, {"ray_bending_latents": latents_o}, n_o, profile=False)
        other = scaling_record(other_scaling, world, n_o, args.steps, dt_o, per_rank_o)
    else:                # N = 1: one frame on one GPU is both the weak and the strong workload
        other = dict(scaling_record("strong" if args.scaling == "weak" else "weak", 1, n, args.steps, dt, per_rank_dt),
                     note="identical to the other scaling at N = 1 (not run twice)")
    # every rank's own kernel time per step (HIP events around each launch on its render stream): lets a scaling record separate
    # kernel time from launch / collective time -- per-rank wall minus this is what the launches and the all-gather cost
    names = list(prof.keys())
    kern_own = torch.tensor([prof[k]["ms"] / args.steps for k in names], dtype=torch.float64)
    per_rank_kernels = [{k: round(float(v), 4) for k, v in zip(names, kern_own) if v > 0}]
    if world > 1:
        t = kern_own.to("cpu" if one_gpu else dev).contiguous()
        allk = torch.empty(world * len(names), dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(allk, t)
        per_rank_kernels = [{k: round(float(v), 4) for k, v in zip(names, row) if v > 0} for row in allk.cpu().view(world, len(names))]
    rccl_ranks = 1
    if world > 1:        # counted by an actual collective on the job's backend, not copied from the environment
        one = torch.ones(1, device="cpu" if one_gpu else dev)
        cnt = torch.empty(world, device=one.device)
        dist.all_gather_into_tensor(cnt, one)
        rccl_ranks = int(cnt.sum().item())
    total_rays = world * n * args.steps
    value = total_rays / dt

    extra = {}
    if rank == 0:
        with torch.no_grad():
            if not args.no_psnr and world == 1:          # (the accuracy / training / CPU legs belong to the N = 1 line: the other
                extra["psnr_vs_oracle_db"] = psnr_vs_oracle(args, scene, cfg, rays, latents, api, kw, dev)     # ranks of a multi-GPU run would wait for rank 0)
        if world == 1 and not args.no_train_step and args.precision != "f16" and not (args.use_viewdirs or args.exact_viewdirs or args.netwidth != 256):
            extra["train_step"] = train_step_leg(args, scene, cfg, dev)          # needs autograd: outside the no_grad block
        with torch.no_grad():
            gemm = library_gemm_tflops(dev, args.precision) if world == 1 else None
            # keep the device visibly busy for an external sampler (the timed region alone is < 1 s)
            extra_frames = 0
            while world == 1 and time.perf_counter() - t_gpu0 < args.min_gpu_seconds:
                step(0)
                torch.cuda.synchronize()
                extra_frames += 1

        k = prof["net_fine"]
        peak = PEAK_TFLOPS[args.precision]

        def roof(kk):
            ach = kk["flops"] / (kk["ms"] * 1e-3) / 1e12 if kk["ms"] > 0 else 0.0
            issued = kk["mfma_flops"] / (kk["ms"] * 1e-3) / 1e12 if kk["ms"] > 0 else 0.0
            # `frac` prices the REFERENCE's flops (2 x MAC, unpadded); `frac_issued_mfma` what the matrix pipe really
            # executes -- more where K is padded (63 -> 64), LESS where the packer folds layers (view-dependent head: feature_linear
            # folded into views_linears[0], 11 % fewer flops): there `frac` overstates the pipe's utilisation, read this one
            return {"achieved": round(ach, 2), "frac": round(ach / peak, 4),
                    "avg_launch_ms": round(kk["ms"] / max(kk["launches"], 1), 4),
                    "issued_mfma_tflops": round(issued, 2), "frac_issued_mfma": round(issued / peak, 4)}
        rf = roof(k)
        traffic, traffic_note, traffic_source = pmc_traffic(args)
        # the coarse pass as ONE figure: with the coarse pass on the 16x16x32 kernel it is two launches (stand-alone bender over the
        # 64 coarse samples + trunk), priced together against the same reference flops as the fused kernel it replaced
        cp = dict(prof["net_coarse"])
        if prof.get("bend_coarse", {}).get("launches"):
            for key in ("ms", "flops", "mfma_flops"):
                cp[key] = prof["net_coarse"][key] + prof["bend_coarse"][key]
        coarse_roof = roof(cp)
        coarse_roof["launches_per_pass"] = 2 if prof.get("bend_coarse", {}).get("launches") else 1
        coarse_roof["trunk_kernel_alone"] = roof(prof["net_coarse"])
        # which kernel: what nrnerf_render dispatched for the fine pass of the timed steps, as the library recorded it (nrnerf_profile.kernel_name)
        roofline = {"bound": "mfma", "kernel": f"{k['kernel']} (fine pass, {cfg.N_samples + cfg.N_importance} samples/ray)",
                    "kernels_launched": {nm: v["kernel"] for nm, v in prof.items() if v["launches"]},
                    "achieved": rf["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": rf["frac"],
                    "avg_launch_ms": rf["avg_launch_ms"], "issued_mfma_tflops": rf["issued_mfma_tflops"], "frac_issued_mfma": rf["frac_issued_mfma"],
                    "traffic": traffic, "traffic_unit": traffic_note, "traffic_source": traffic_source,
                    "coarse_pass_incl_bender": coarse_roof,
                    "kernels_ms_per_step": {nm: round(v["ms"] / args.steps, 4) for nm, v in prof.items() if v["launches"]},
                    # context, not the peak: what a plain hipBLASLt GEMM sustains on this box right now (the chip
                    # clocks down under MFMA load)
                    "library_gemm_tflops_same_box": gemm}
        if traffic is None and "BASELINE config 2" not in workload_label(args, cfg, frame_rays, n, world):
            # a variant run has no PMC pass of its own on file: no traffic keys at all rather than a null that reads like a measurement
            for key in ("traffic", "traffic_unit", "traffic_source"):
                roofline.pop(key)
        if args.precision in SUSTAINED_MFMA_TFLOPS:
            # `frac` is against the paper peak (2.4 GHz x 256 CUs).  Under the socket's power cap a register-only stream of the SAME MFMA
            # (no LDS, no VALU, no memory: tools/probes/mfma_shape_power.hip) sustains less; read `frac` against both.
            lo, hi = SUSTAINED_MFMA_TFLOPS[args.precision]
            roofline["sustained_peak"] = {"tflops_range": [lo, hi], "frac_of_sustained": [round(rf["achieved"] / hi, 4), round(rf["achieved"] / lo, 4)],
                                          "source": "profiles/r04_mfma_shape_power.txt (register-only v_mfma_f32_16x16x32 stream on MI355X under the 1400 W cap), "
                                                    "profiles/r06_power_trace.txt (socket power and shader clock during this kernel)"}
        flops_per_ray = sum(v["flops"] for v in prof.values()) / max(n * args.steps, 1)
        res = {"metric": "rendered rays/sec (64+128 samples/ray)", "value": round(value, 1), "unit": "rays/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
               "vs_baseline": None, "dtype": args.precision,
               "data": ("synthetic: " if args.scene == "synthetic" else "example_sequence: ") + data_desc
                       + (" (NOT A BENCHMARK: all ranks on one GPU, gloo)" if one_gpu else ""),
               "config": {"workload": workload_label(args, cfg, frame_rays, n, world),
                          "scene": args.scene, "rays_per_gpu_per_step": n, "N_samples": cfg.N_samples, "N_importance": cfg.N_importance,
                          "netwidth": args.netwidth, "use_viewdirs": bool(args.use_viewdirs), "bend_depth": args.bend_depth,
                          "chunk": args.chunk, "rays_per_launch": min(n, max(args.chunk, R._MAX_RAYS_PER_LAUNCH)),
                          "parallelism": f"rays sharded over {world} rank(s)"
                                         + (", all-gather of [rgb,disp,acc] on a side stream, overlapped with the next frame" if world > 1 else "")},
               "rccl_ranks": rccl_ranks, "backend": backend,
               # both scalings of an N-GPU job in the one line: weak = a 512x384 frame per rank per step, strong = BASELINE
               # config 3, ONE frame sharded contiguously over the ranks (24 576 rays per rank at N = 8), all-gather overlapped
               **{rec["scaling"]: rec for rec in (scaling_record(args.scaling, world, n, args.steps, dt, per_rank_dt), other) if rec},
               "mflop_per_ray_algorithmic": round(flops_per_ray / 1e6, 2),
               "end_to_end_tflops": round(value * flops_per_ray / 1e12, 2),
               "untimed_extra_frames": extra_frames,
               "per_rank_kernels_ms_per_step": per_rank_kernels,
               "roofline": roofline}
        res.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(scene, cfg, args, rays, latents)
        print(json.dumps(res), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def frames_mode(args, rank, world, dev, one_gpu, backend):
    """``--frames F``: the no-collective half of SURVEY.md section 8(e) -- a free-viewpoint SEQUENCE (what free_viewpoint_rendering.py
    does with config 3's checkpoint: 120-300 poses through render_path, train.py:419-553) sharded by whole frames.  A step = one
    ``driver.render_path(..., group=, rgb_dtype="uint8")`` over F frames; rank r renders frames r, r + G, ...; the only collective is the
    all-gather of the finished uint8 frames at the end of the step (inside the timed region)."""
    import numpy as np
    import torch.distributed as dist
    from nonrigid_nerf_amd import render as R
    from nonrigid_nerf_amd.driver import frame_shard, render_path
    scene, cfg, (rb, coarse, fine), _, _, data_desc = build_workload(args, 0, 1, dev, frame_rays=64)
    R.set_precision(args.precision)
    z = np.load(FIXTURE)
    F, nposes = int(args.frames), int(z["poses"].shape[0])
    s = 512.0 / float(z["hwf"][1])
    intrin = dict(height=384, width=512, focal_x=float(z["hwf"][2]) * s, focal_y=float(z["hwf"][2]) * s, center_x=256.0, center_y=192.0)
    poses = [torch.from_numpy(z["poses"][f % nposes]) for f in range(F)]
    if args.scene == "fitted":
        from nonrigid_nerf_amd.checkpoint import load_checkpoint
        codes = load_checkpoint(args.fitted_ckpt, N_samples=64, N_importance=128).latents
        codes = torch.stack([codes[f % nposes] for f in range(F)], 0)
    else:
        codes = 0.1 * torch.randn(F, cfg.latent_size, generator=torch.Generator().manual_seed(5))
    kw = dict(network_fn=coarse, network_fine=fine, network_query_fn=None, N_samples=cfg.N_samples, N_importance=cfg.N_importance,
              perturb=0.0, raw_noise_std=0.0, near=cfg.near, far=cfg.far, use_viewdirs=bool(args.use_viewdirs))
    model = R.get_model(coarse, fine, device=dev)
    grp = True if world > 1 else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        return render_path(poses, [intrin] * F, args.chunk, kw, codes, rgb_dtype="uint8", device=dev, group=grp, gather="all")

    for _ in range(args.warmup):
        step()
    barrier()
    model.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rgbs, disps = step()
    barrier()
    dt_own = time.perf_counter() - t0
    prof = model.profile_end()
    assert rgbs.shape == (F, 384, 512, 3) and rgbs.dtype == np.uint8
    kern_ms = sum(v["ms"] for v in prof.values())
    mine = len(frame_shard(F, world, rank))
    per_rank = [(dt_own, kern_ms, mine)]
    if world > 1:
        t = torch.tensor([dt_own, kern_ms, float(mine)], dtype=torch.float64, device="cpu" if one_gpu else dev)
        allt = torch.empty(world * 3, dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(allt, t)
        per_rank = [tuple(float(x) for x in row) for row in allt.cpu().view(world, 3)]
    dt = max(r[0] for r in per_rank)
    if rank != 0:
        return
    rays_per_frame = 384 * 512
    res = {"metric": "rendered rays/sec (64+128 samples/ray)", "value": round(F * rays_per_frame * args.steps / dt, 1), "unit": "rays/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.precision,
           "data": ("synthetic: " if args.scene == "synthetic" else "example_sequence: ") + data_desc
                   + (" (NOT A BENCHMARK: all ranks on one GPU, gloo)" if one_gpu else ""),
           "config": {"workload": f"NOT a BASELINE.json config line (variant run): frame-sharded free-viewpoint sequence, {F} frames of 512x384 per step through "
                                  f"driver.render_path, rank r renders frames r, r + {world}, ...; {cfg.N_samples} + {cfg.N_importance} samples, netwidth {args.netwidth}, "
                                  f"{args.precision}; ray generation, uint8 conversion, D2H and the final all-gather of the frames inside the timed region",
                      "frames_per_step": F, "rays_per_frame": rays_per_frame, "scene": args.scene,
                      "parallelism": f"frames sharded round-robin over {world} rank(s), no collective while rendering, one all-gather of uint8 frames per step"},
           "backend": backend,
           "frames_per_s": round(F * args.steps / dt, 3),
           "per_rank": [{"frames_per_step": int(m), "frames_per_s": round(m * args.steps / t, 3) if t > 0 else None,
                         "ms_per_step": round(t / args.steps * 1e3, 3), "kernels_ms_per_step": round(k / args.steps, 3)} for t, k, m in per_rank]}
    print(json.dumps(res), flush=True)


def workload_label(args, cfg, frame_rays, rays_per_rank, world):
    """config.workload, built from the arguments of THIS run: which BASELINE.json config it is (only when it is one), else what differs."""
    default_net = not (args.use_viewdirs or args.exact_viewdirs or args.bend_depth != 5 or args.netwidth != 256)
    shape = (f"{frame_rays} rays per {'frame (sharded over the ranks)' if args.scaling == 'strong' and world > 1 else 'GPU and step'}"
             + (" (= one 512x384 frame)" if frame_rays == 196608 else " (= one 1920x1080 frame)" if frame_rays == 2073600 else "")
             + f", {cfg.N_samples} coarse + {cfg.N_importance} importance samples, netwidth {args.netwidth}, "
             + ("view-dependent head (" + ("exact Jacobian" if args.exact_viewdirs else "finite-difference") + " directions), " if args.use_viewdirs or args.exact_viewdirs else "")
             + f"ray bender {args.bend_depth} x 64, latent 32, {args.precision}, chunk {args.chunk}"
             + (f", at most {args.max_rays_per_launch} rays per launch" if args.max_rays_per_launch > 0 else ""))
    if default_net and frame_rays == 196608 and args.precision == "bf16" and args.max_rays_per_launch == 0:
        name = "BASELINE config 3 (one frame sharded over the ranks)" if (args.scaling == "strong" and world > 1) else "BASELINE config 2"
    elif args.use_viewdirs and not args.exact_viewdirs and args.bend_depth == 7 and args.netwidth == 256:
        name = "BASELINE config 4 (view-dependent head, deeper ray-bending MLP; one latent per frame)"
    elif default_net and frame_rays == 2073600 and args.precision == "f16" and args.max_rays_per_launch == 65536:
        name = "BASELINE config 5 (1080p frame in 65 536-ray launches, f16 weights)"
    else:
        name = "NOT a BASELINE.json config (variant run)"
    return f"{name}: {shape}"


def scaling_record(scaling, world, rays_per_rank, steps, dt_max, per_rank_dt):
    """One scaling's figures of an N-rank job: whole-job rays/s from the slowest rank's wall time, every rank's own
    ms/step (a straggler shows), and the shard size.  Pure function of the measurements (unit-tested on the CPU tier)."""
    return {"scaling": scaling, "value": round(world * rays_per_rank * steps / dt_max, 1), "unit": "rays/s",
            "ms_per_step": round(dt_max / steps * 1e3, 3), "rays_per_rank_per_step": int(rays_per_rank),
            "rays_per_step_whole_job": int(world * rays_per_rank),
            "per_rank_ms_per_step": [round(t / steps * 1e3, 3) for t in per_rank_dt],
            "what": ("a frame of that many rays per rank per step (512x384 at the default size)" if scaling == "weak"
                     else "BASELINE config 3: ONE frame (512x384 at the default size) sharded contiguously over the ranks, "
                          "all-gather of the pixels overlapped with the next frame")}


def psnr_vs_oracle(args, scene, cfg, rays, latents, api, kw, dev):
    """PSNR of this run's precision (and of the other precisions, for context) against the fp32 oracle on the first
    ``--psnr-rays`` rays of the benchmark workload.  All rays count.  The oracle runs on the GPU (device-agnostic eager
    PyTorch, the reference's own ops): checker use only."""
    import math

    from nonrigid_nerf_amd import render as R
    from oracle import nrnerf_oracle as O
    m = min(args.psnr_rays, rays.shape[0])
    r, lat = rays[:m], latents[:m]
    ref = O.batchify_rays(r, lat.contiguous(), O.scene_on(scene, dev), chunk=8192)
    out = {}
    for prec in ("f32", "bf16", "f16"):
        R.set_precision(prec)
        got = R.batchify_rays(r, {"ray_bending_latents": lat}, **kw)
        res = {}
        for key in ("rgb_map", "rgb0"):
            mse = float(((got[key].double() - ref[key].double()) ** 2).mean())
            res[key] = round(-10.0 * math.log10(max(mse, 1e-30)), 2)
        out[prec] = res
    R.set_precision(args.precision)
    return {"dtype": args.precision, "rgb_map": out[args.precision]["rgb_map"], "rgb0": out[args.precision]["rgb0"],
            "rays": m, "all_precisions": out, "reference": "fp32 oracle (eager PyTorch ops of the reference) on the same rays and weights"}


def train_step_leg(args, scene, cfg, dev):
    try:
        from nonrigid_nerf_amd import training
    except Exception as e:                                    # not built yet
        return {"error": f"{type(e).__name__}: {e}"}
    return training.bench_train_step(scene, cfg, dev, precision=args.precision)


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


def kernel_source_sha16(train=False):
    """Hash of the device code the rendering kernels OF THE COMPILED ARCHITECTURES are built from (csrc/*.h, *.hip: plans, kernels,
    launchers; excluded: the training-only kernels nrnerf_train* and the run-time-parameterised kernel nrnerf_generic*, which no
    workload that looks a profile up by this hash launches -- pmc_traffic below answers for the default architecture only):
    identifies the profiled kernels independently of the build (a rebuilt .so need not be byte-identical) and of host-only edits
    (nrnerf_api.cpp).  ``train=True``: the same over ALL device sources, training kernels included -- what the training
    profiles (profiles/*_train_*) are stamped with."""
    h = hashlib.sha256()
    csrc = os.path.join(REPO, "nonrigid_nerf_amd", "csrc")
    for name in sorted(f for f in os.listdir(csrc) if f.endswith((".h", ".hip")) and (train or not f.startswith(("nrnerf_train", "nrnerf_generic")))):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def _latest_profile(suffix):
    """profiles/rNN<suffix> of the highest round on file."""
    prof = os.path.join(REPO, "profiles")
    names = sorted(n for n in os.listdir(prof) if n.endswith(suffix) and n[0] == "r" and n[1:3].isdigit())
    return os.path.join(prof, names[-1]) if names else None


def pmc_traffic(args):
    """HBM bytes per launch of the fine-pass network kernel.  Hardware counters cannot be read from inside the process:
    they come from separate ``rocprofv3 --pmc`` passes of this very command (tools/collect_profiles.sh), stored with the
    hash of the kernel sources they profiled.  Reported only when that hash matches the sources of this checkout and the
    workload is the profiled one; otherwise null (a stale number is worse than none)."""
    path = _latest_profile("_pmc_fine.json")
    default_arch = not (args.use_viewdirs or args.exact_viewdirs or args.bend_depth != 5 or args.netwidth != 256)
    if args.rays != 196608 or args.precision != "bf16" or not path or not default_arch or os.environ.get("NRNERF_FORCE_GENERIC") == "1":
        return None, "null: no rocprofv3 --pmc pass of this build and workload on file (profiles/rNN_pmc_fine.json)", None
    name = os.path.relpath(path, REPO)
    try:
        with open(path) as f:
            j = json.load(f)
        if j.get("scene") != args.scene:
            return None, f"null: {name} was collected on the {j.get('scene')} scene", None
        if j.get("kernel_source_sha16") != kernel_source_sha16():
            return None, f"null: {name} was collected from different kernel sources", None
        note = ("bytes per launch, 2 x FETCH_SIZE + WRITE_SIZE from rocprofv3 --pmc passes "
                f"of these kernel sources ({name.replace('_pmc_fine.json', '_pmc_summary.txt')}); by design 0.76e9 since the compositing is fused "
                "into the kernel (16 B/sample of points + 4 B/sample of depths in, 44 B/ray out; was 1.21e9 with raw written out)")
        source = (f"committed profile {name}: separate rocprofv3 --pmc passes of this command over the same kernel sources (hash-checked), "
                  "NOT measured in this run -- hardware counters cannot be read from inside the process")
        return float(j["fine"]["hbm_bytes_per_launch"]), note, source
    except Exception as e:
        return None, f"null: {type(e).__name__}", None


def cpu_baseline(scene, cfg, args, rays_dev, latents_dev):
    """The CPU oracle (PyTorch-CPU port of reference render_rays/batchify_rays) on ``n`` rays of the same workload."""
    from oracle import nrnerf_oracle as O
    n = min(args.cpu_rays, rays_dev.shape[0])
    rays, latents = rays_dev[:n].cpu(), latents_dev[:n].cpu().contiguous()
    # Thread sweep, recorded in the line: torch's default (one thread per logical core) oversubscribes a 256-thread host badly --
    # the oracle's ops are small ([1024 x 64 x 95] rows through 256-wide layers per chunk), so beyond a few dozen threads the
    # intra-op fork/join and the cross-socket traffic cost more than the extra cores give (measured 277 rays/s at 128 threads
    # vs ~1300 at 8 on the build container).  Each candidate: warm-up, then best of 3 on 1024 rays; the winner: best of 3 on
    # the whole sample.
    sweep = {}
    with torch.no_grad():
        for t in sorted({8, 16, 32, 64, 128} & set(range(1, (os.cpu_count() or 8) + 1))):
            torch.set_num_threads(t)
            O.batchify_rays(rays[:512], latents[:512], scene, chunk=512)      # warm-up
            best = 0.0
            for _ in range(3):
                t0 = time.perf_counter()
                O.batchify_rays(rays[:1024], latents[:1024], scene, chunk=1024)
                best = max(best, 1024 / (time.perf_counter() - t0))
            sweep[t] = round(best, 1)
        threads = max(sweep, key=sweep.get)
        torch.set_num_threads(threads)
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.batchify_rays(rays, latents, scene, chunk=1024)
            times.append(time.perf_counter() - t0)
        dt = min(times)
    port = {"value": round(n / dt, 1), "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"{n} rays of the same 64+128 workload, chunk 1024, torch {torch.__version__} CPU, "
                      f"{threads} threads of {os.cpu_count()} host cores, best of 3: {dt:.1f} s (all: {', '.join(f'{x:.1f}' for x in times)} s)",
            "thread_sweep_rays_per_s": {str(k): v for k, v in sweep.items()},
            "why_not_all_cores": "the oracle's per-chunk ops are too small to amortise a 256-way fork/join: throughput peaks at a few dozen threads (sweep above)",
            "port_vs_reference": {"note": "the unmodified reference (train.render) and this port timed side by side, same rays / weights / "
                                          "chunk; NOT measured in this run (the GPU box carries no copy of the reference unless "
                                          "NRNERF_REFERENCE points at one -- then this line has kind 'reference' and port_same_box)",
                                  "gpu_host_256_core_epyc_32_threads": 1.25, "gpu_host_source": "BASELINE.md section 2, round 3, profiles/r03_bench_bf16.json: reference 675 rays/s, port 845",
                                  "build_container_8_vcpu": "0.96-0.99", "build_container_source": "BASELINE.md section 2, tools/cpu_reference_vs_port.py",
                                  "this_host": f"{socket.gethostname()}, {os.cpu_count()} logical cores"}}
    ref = reference_cpu_baseline(scene, rays, latents, threads)
    if ref is None:
        return port
    ref["port_same_box"] = {"value": port["value"], "port_over_reference": round(port["value"] / ref["value"], 3)}
    return ref


def reference_cpu_baseline(scene, rays, latents, threads):
    """The UNMODIFIED reference (train.render -> batchify_rays -> render_rays, CPU) on the same rays and weights -- only when
    the operator points NRNERF_REFERENCE at a checkout of facebookresearch/nonrigid_nerf (the driver's GPU boxes carry none,
    and nothing is read from a default path: then the port above is the baseline).  tools/with_reference.sh stages one."""
    ref_dir = os.environ.get("NRNERF_REFERENCE")
    if not ref_dir or not os.path.isfile(os.path.join(ref_dir, "train.py")):
        return None
    try:
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        import make_golden as G
        old_get_device = torch.Tensor.get_device
        H, T = G.import_reference()
        try:
            kw, rb, coarse, fine = G.reference_kwargs(H, T, scene)
            n = rays.shape[0]
            with torch.no_grad():
                run = lambda m: T.render(rays[:m, 0:3], rays[:m, 3:6], chunk=1024, additional_pixel_information={"ray_bending_latents": latents[:m]}, **kw)
                run(512)
                t0 = time.perf_counter()
                run(n)
                dt = time.perf_counter() - t0
        finally:
            torch.Tensor.get_device = old_get_device
        return {"value": round(n / dt, 1), "unit": "rays/s", "cores": threads, "kind": "reference",
                "sample": f"{n} rays of the same 64+128 workload through the unmodified reference's train.render (chunk 1024, torch "
                          f"{torch.__version__} CPU, {threads} threads of {os.cpu_count()} host cores, {dt:.1f} s); checkout at $NRNERF_REFERENCE"}
    except Exception as e:                                                    # a broken checkout must not take the bench down
        print(f"[bench] reference CPU baseline unavailable: {type(e).__name__}: {e}", file=sys.stderr)
        return None


if __name__ == "__main__":
    main()
