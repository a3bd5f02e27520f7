"""N > 1 path on CPU: world_size 2, gloo.  Rays sharded by rank + all-gather of the rendered pixels must give
every rank exactly the single-process image.  The renderer plugged in here is the CPU
oracle (test infrastructure); on GPUs bench.py plugs in the HIP path and the backend is RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nonrigid_nerf_amd.distributed import gather_pixels, render_sharded, shard_bounds
from nonrigid_nerf_amd.synthetic import SceneConfig, make_rays, make_scene


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import nrnerf_oracle as O
    cfg = SceneConfig(N_importance=0)
    scene = make_scene(cfg, 0)
    rays, lat = make_rays(n, 4, cfg)
    calls = []

    def render_fn(r, l):
        calls.append(r.shape[0])
        return O.render_rays(r, l, scene)

    img = render_sharded(render_fn, rays, lat)
    lo, hi, per = shard_bounds(n, world, rank)
    assert calls == ([hi - lo] if hi > lo else []), (calls, lo, hi)
    blocks = gather_pixels(torch.full((3, 5), float(rank)))
    assert blocks.shape == (3 * world, 5) and all(float(blocks[3 * r, 0]) == r for r in range(world))
    torch.save(img, os.path.join(out_dir, f"img{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [37, 64])
def test_sharded_render_matches_single_process(tmp_path, n):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    from oracle import nrnerf_oracle as O
    cfg = SceneConfig(N_importance=0)
    scene = make_scene(cfg, 0)
    rays, lat = make_rays(n, 4, cfg)
    ref = O.render_rays(rays, lat, scene)
    full = torch.cat([ref["rgb_map"], ref["disp_map"][:, None], ref["acc_map"][:, None]], -1)
    for r in range(world):
        img = torch.load(os.path.join(str(tmp_path), f"img{r}.pt"))
        assert img.shape == (n, 5)
        assert torch.allclose(torch.nan_to_num(img), torch.nan_to_num(full), atol=1e-6), f"rank {r}"


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 8, 9, 196608):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi, per = shard_bounds(n, world, r)
                assert hi - lo <= per
                seen += list(range(lo, hi))
            assert seen == list(range(n))


def _rccl_worker(port, out_path):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from nonrigid_nerf_amd import render as R
    from nonrigid_nerf_amd.synthetic import build_modules
    cfg = SceneConfig(N_importance=64)
    scene = make_scene(cfg, 0)
    rb, coarse, fine = build_modules(scene, device="cuda:0")
    rays, lat = make_rays(1000, 4, cfg)
    rays, lat = rays.cuda(), lat.cuda()
    R.set_precision("bf16")
    kw = dict(network_fn=coarse, network_fine=fine, N_samples=64, N_importance=64)

    def render_fn(r, l):
        return R.batchify_rays(r, {"ray_bending_latents": l}, **kw)

    with torch.no_grad():
        img = render_sharded(render_fn, rays, lat, force_collective=True)         # all_gather_into_tensor over RCCL
        direct = render_fn(rays, lat)
    dist.barrier()
    want = torch.cat([direct["rgb_map"], direct["disp_map"][:, None], direct["acc_map"][:, None]], -1)
    ok = torch.equal(torch.nan_to_num(img), torch.nan_to_num(want))
    # the overlapped form bench.py uses: asynchronous all_gather_into_tensor over RCCL from a side stream, two buffer pairs
    from nonrigid_nerf_amd.distributed import OverlappedGather
    g = OverlappedGather(1000, torch.device("cuda", 0), force_collective=True)
    frames = []
    with torch.no_grad():
        for i in range(4):
            out = render_fn(rays + 0.001 * i, lat)
            full = g.submit(i, out)
            frames.append((full, torch.cat([out["rgb_map"], out["disp_map"][:, None], out["acc_map"][:, None]], -1).clone()))
        g.drain()       # (a buffer pair is reused two frames later: after the loop the pairs hold frames 2 and 3)
    torch.cuda.synchronize()
    ok_overlap = all(torch.equal(torch.nan_to_num(frames[i][0]), torch.nan_to_num(frames[i][1])) for i in (2, 3))   # the two live pairs
    torch.save({"ok": ok, "ok_overlap": ok_overlap, "backend": dist.get_backend()}, out_path)
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_path_with_one_rank_on_the_gpu(tmp_path):
    """The collective of the multi-GPU path over the real backend (torch "nccl" = RCCL on ROCm) with a one-rank group:
    what a 1-GPU box can exercise of SURVEY.md section 8e (the 2/4/8-GPU points are the driver's to measure)."""
    out = os.path.join(str(tmp_path), "r.pt")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), out))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    res = torch.load(out)
    assert res["ok"] and res["ok_overlap"] and res["backend"] == "nccl"


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher must start two ranks by itself (torch.distributed.run) and report
    them.  On a 1-GPU box NRNERF_BENCH_ONE_GPU=1 puts both ranks on GPU 0 and gathers over gloo (functional test of the
    script's N > 1 path, not a measurement)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NRNERF_BENCH_ONE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--rays", "8192", "--no-cpu-baseline", "--no-psnr", "--no-train-step"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["backend"] == "gloo"
    assert res["steps"] == 3 and res["value"] > 0


def _overlap_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nonrigid_nerf_amd.distributed import OverlappedGather
    n = 11
    g = OverlappedGather(n, "cpu")
    results = []
    for i in range(5):                                   # five "frames": buffers are reused from frame 2 on
        base = 100.0 * i + 10.0 * rank
        out = {"rgb_map": torch.full((n, 3), base), "disp_map": torch.full((n,), base + 1), "acc_map": torch.full((n,), base + 2)}
        full = g.submit(i, out)
        if i >= 1:                                       # frame i-1 may be read after its collective has been joined
            g.pending[(i - 1) & 1].wait()
            results.append(g.full[(i - 1) & 1].clone())
    g.drain()
    results.append(full.clone())
    torch.save(results, os.path.join(out_dir, f"ov{rank}.pt"))
    dist.destroy_process_group()


def test_overlapped_gather_double_buffering_world_size_2(tmp_path):
    """The N > 1 path of bench.py (asynchronous all-gather of frame f while frame f+1 is produced, two buffer pairs) on
    CPU with gloo: every rank must see every rank's pixels of every frame, in rank order."""
    world = 2
    mp.spawn(_overlap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), f"ov{r}.pt"))
        assert len(res) == 5
        for i, full in enumerate(res):
            assert full.shape == (22, 5)
            for src in range(world):
                blk = full[11 * src:11 * (src + 1)]
                base = 100.0 * i + 10.0 * src
                assert torch.equal(blk[:, 0:3], torch.full((11, 3), base)) and torch.equal(blk[:, 3], torch.full((11,), base + 1)) \
                    and torch.equal(blk[:, 4], torch.full((11,), base + 2)), (r, i, src)
