"""N > 1 path on CPU: world_size 2 and 8, gloo.  Rays sharded by rank + all-gather of the rendered pixels must give
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


def _worker8(rank, world, port, n, out_dir):
    """One rank of the 8-rank job: `render_sharded` (contiguous ragged shards, some of them EMPTY when n < world) and the
    overlapped all-gather bench.py uses for a sequence of frames, on one process group."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from nonrigid_nerf_amd.distributed import OverlappedGather

    def render_fn(r, l):           # a closed-form "renderer": every pixel names its own ray, so misplaced shards show
        return {"rgb_map": r[:, 0:3] * 2.0 + l[:, 0:3], "disp_map": r[:, 3] - l[:, 3], "acc_map": r[:, 4] * l[:, 4]}

    g = torch.Generator().manual_seed(11)
    rays, lat = torch.randn(n, 8, generator=g), torch.randn(n, 32, generator=g)
    img = render_sharded(render_fn, rays, lat)
    lo, hi, per = shard_bounds(n, world, rank)
    og = OverlappedGather(per, "cpu")
    frames = []
    for i in range(4):
        mine = render_fn(rays[lo:hi] + float(i), lat[lo:hi])
        pad = {k: torch.cat([v, v.new_zeros((per - (hi - lo),) + v.shape[1:])], 0) for k, v in mine.items()}    # equal blocks: short shards padded
        frames.append(og.submit(i, pad))
        if i >= 1:
            og.pending[(i - 1) & 1].wait()
            frames[i - 1] = frames[i - 1].clone()
    og.drain()
    torch.save({"img": img, "frames": [f.clone() for f in frames]}, os.path.join(out_dir, f"w8_{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [5, 61, 200])
def test_eight_ranks_ragged_shards_and_overlapped_gather(tmp_path, n):
    """BASELINE config 3's rank count on the CPU tier (gloo, 8 processes): ragged ceil(n / 8) shards -- with n = 5 three ranks own
    NOTHING, with n = 61 the last rank owns 5 of 8 rows -- through `render_sharded`, and four frames through `OverlappedGather`
    with double buffering; every rank must end up with every ray's pixels at the ray's own row."""
    world = 8
    mp.spawn(_worker8, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(11)
    rays, lat = torch.randn(n, 8, generator=g), torch.randn(n, 32, generator=g)
    want = lambda r: torch.cat([r[:, 0:3] * 2.0 + lat[:, 0:3], (r[:, 3] - lat[:, 3])[:, None], (r[:, 4] * lat[:, 4])[:, None]], -1)
    per = shard_bounds(n, world, 0)[2]
    for rank in range(world):
        res = torch.load(os.path.join(str(tmp_path), f"w8_{rank}.pt"))
        assert res["img"].shape == (n, 5) and torch.equal(res["img"], want(rays)), rank
        for i, full in enumerate(res["frames"]):
            assert full.shape == (world * per, 5)
            rows = torch.cat([full[per * r: per * r + (shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0])] for r in range(world)], 0)
            assert torch.equal(rows, want(rays + float(i))), (rank, i)


def test_bench_workload_label_follows_the_arguments():
    """config.workload of the bench line is built from the run's own arguments: only the headline arguments may call themselves
    BASELINE config 2, a config-4 / config-5 / variant run says what it is (the config-5 record carried config 2's label)."""
    import argparse
    import importlib.util
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module2", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    base = dict(use_viewdirs=False, exact_viewdirs=False, bend_depth=5, netwidth=256, precision="bf16", chunk=32768,
                max_rays_per_launch=0, scaling="weak")
    cfg = SceneConfig()
    lab = lambda frame_rays=196608, world=1, **kw: bench.workload_label(argparse.Namespace(**dict(base, **kw)), cfg, frame_rays, frame_rays, world)
    assert lab().startswith("BASELINE config 2:") and "196608 rays" in lab()
    assert lab(world=8, scaling="strong").startswith("BASELINE config 3")
    assert lab(use_viewdirs=True, bend_depth=7).startswith("BASELINE config 4") and "view-dependent" in lab(use_viewdirs=True, bend_depth=7)
    l5 = lab(frame_rays=2073600, precision="f16", chunk=65536, max_rays_per_launch=65536)
    assert l5.startswith("BASELINE config 5") and "2073600" in l5 and "config 2" not in l5
    for kw in (dict(netwidth=128), dict(precision="f16"), dict(frame_rays=32768), dict(use_viewdirs=True)):
        assert lab(**kw).startswith("NOT a BASELINE.json config"), kw


def test_bench_traffic_lookup_returns_value_note_and_source(tmp_path, monkeypatch):
    """bench.pmc_traffic: (bytes, note, source) on every path -- the committed profile of these kernel sources, a stale one, none."""
    import argparse
    import importlib.util
    import json
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module3", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    args = argparse.Namespace(rays=196608, precision="bf16", use_viewdirs=False, exact_viewdirs=False, bend_depth=5, netwidth=256, scene="fitted")
    prof = tmp_path / "r99_pmc_fine.json"
    monkeypatch.setattr(bench, "_latest_profile", lambda suffix: str(prof))
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    prof.write_text(json.dumps({"scene": "fitted", "kernel_source_sha16": bench.kernel_source_sha16.__wrapped__() if hasattr(bench.kernel_source_sha16, "__wrapped__") else "x", "fine": {"hbm_bytes_per_launch": 7.7e8}}))
    monkeypatch.setattr(bench, "kernel_source_sha16", lambda train=False: "x")
    prof.write_text(json.dumps({"scene": "fitted", "kernel_source_sha16": "x", "fine": {"hbm_bytes_per_launch": 7.7e8}}))
    v, note, source = bench.pmc_traffic(args)
    assert v == 7.7e8 and "FETCH_SIZE" in note and "NOT measured in this run" in source
    prof.write_text(json.dumps({"scene": "fitted", "kernel_source_sha16": "other", "fine": {"hbm_bytes_per_launch": 1.0}}))
    v, note, source = bench.pmc_traffic(args)
    assert v is None and "different kernel sources" in note and source is None
    args.rays = 1000
    v, note, source = bench.pmc_traffic(args)
    assert v is None and source is None


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
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_spawns_its_own_ranks(scaling):
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
                        "--rays", "8192", "--no-cpu-baseline", "--no-psnr", "--no-train-step", "--scaling", scaling],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["backend"] == "gloo"
    assert res["steps"] == 3 and res["value"] > 0
    # both scalings of the job in the one line: weak = --rays per rank, strong = --rays sharded over the ranks
    # (the record named by --scaling is the line's primary one; BASELINE config 3 -- one frame sharded over the ranks -- is "strong")
    assert res["scaling"] == scaling and res[scaling]["value"] == res["value"] and res["weak"]["rays_per_rank_per_step"] == 8192
    assert res["strong"]["rays_per_rank_per_step"] == 4096 and res["strong"]["rays_per_step_whole_job"] == 8192
    assert len(res["strong"]["per_rank_ms_per_step"]) == 2 and res["strong"]["value"] > 0


def test_bench_scaling_records_are_whole_job_figures():
    """bench.py's per-scaling record (pure function of the measurements): whole-job rays/s from the SLOWEST rank's wall
    time, every rank's own ms/step, shard sizes of BASELINE config 3 (one 512x384 frame over 8 ranks = 24 576 rays each)."""
    import importlib.util
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    per_rank = [0.100, 0.104, 0.098, 0.101, 0.100, 0.100, 0.099, 0.102]
    rec = bench.scaling_record("strong", 8, 24576, 20, max(per_rank), per_rank)
    assert rec["rays_per_step_whole_job"] == 196608 and rec["rays_per_rank_per_step"] == 24576
    assert abs(rec["value"] - 8 * 24576 * 20 / 0.104) < 1.0 and rec["ms_per_step"] == 5.2
    assert rec["per_rank_ms_per_step"] == [5.0, 5.2, 4.9, 5.05, 5.0, 5.0, 4.95, 5.1]
    weak = bench.scaling_record("weak", 8, 196608, 20, 0.8, [0.8] * 8)
    assert weak["value"] == 8 * 196608 * 20 / 0.8 and weak["scaling"] == "weak"


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


def _train_setup(n):
    """A tiny differentiable render on the CPU (the oracle, test infrastructure) with its parameters as leaves, bucketed in
    the order backward produces them."""
    from oracle import nrnerf_oracle as O
    from nonrigid_nerf_amd.distributed import GradientBuckets  # noqa: F401
    cfg = SceneConfig(N_samples=16, N_importance=8)
    scene = make_scene(cfg, 0)
    rays, lat = make_rays(n, 4, cfg)
    sc = O.scene_on(scene, "cpu")
    leaves = {}
    for part in ("fine", "coarse", "bender"):
        d = getattr(sc, part)
        for k in d:
            d[k] = d[k].clone().requires_grad_(True)
            leaves[(part, k)] = d[k]
    lat = lat.clone().requires_grad_(True)
    target = torch.rand(n, 3, generator=torch.Generator().manual_seed(3))

    def loss_of(lo, hi):
        out = O.render_rays(rays[lo:hi], lat[lo:hi], sc)
        return ((out["rgb_map"] - target[lo:hi]) ** 2).mean() + ((out["rgb0"] - target[lo:hi]) ** 2).mean()

    loss_of(0, 2).backward()                    # which tensors of the scene take part at all (no view-dependent head here)
    leaves = {k: v for k, v in leaves.items() if v.grad is not None}
    for v in list(leaves.values()) + [lat]:
        v.grad = None
    buckets = [[v for (p, _), v in leaves.items() if p == "fine"], [v for (p, _), v in leaves.items() if p == "coarse"],
               [v for (p, _), v in leaves.items() if p == "bender"] + [lat]]
    return leaves, lat, buckets, loss_of


def _train_worker(rank, world, port, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from nonrigid_nerf_amd.distributed import GradientBuckets
    leaves, lat, buckets, loss_of = _train_setup(n)
    gb = GradientBuckets(buckets)
    lo, hi, _ = shard_bounds(n, world, rank)
    snaps = []
    for step in range(2):                       # two steps: the buffers are reused, zeroing must not detach the views
        gb.zero_grad()
        loss_of(lo, hi).backward()
        assert all(p is not None for p in gb.pending), "every bucket's all-reduce was issued from inside backward"
        gb.finish()
        snaps.append({k: v.grad.clone() for k, v in leaves.items()} | {("latents", ""): lat.grad.clone()})
    gb.close()
    torch.save(snaps, os.path.join(out_dir, f"grads{rank}.pt"))
    dist.destroy_process_group()


def test_data_parallel_training_gradients_world_size_2(tmp_path):
    """GradientBuckets (the replacement of nn.DataParallel over training_wrapper_class, train.py:300-323): two gloo ranks
    render half of the ray batch each; after finish() both hold the gradient of the full-batch mean loss, in both of two
    consecutive steps, and the collectives were launched from inside backward (one per bucket)."""
    world, n = 2, 16
    mp.spawn(_train_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    leaves, lat, _, loss_of = _train_setup(n)
    (0.5 * (loss_of(0, n // 2) + loss_of(n // 2, n))).backward()
    want = {k: v.grad for k, v in leaves.items()} | {("latents", ""): lat.grad}
    for r in range(world):
        snaps = torch.load(os.path.join(str(tmp_path), f"grads{r}.pt"))
        assert len(snaps) == 2
        for got in snaps:
            assert set(got) == set(want)
            for k, g in want.items():
                if k == ("latents", ""):
                    # each rank only back-propagates into its own rays' codes; the mean over ranks halves them
                    assert torch.allclose(got[k], g, rtol=1e-4, atol=1e-7 * float(g.abs().max()) + 1e-12), k
                else:
                    assert torch.allclose(got[k], g, rtol=1e-4, atol=1e-5 * float(g.abs().max()) + 1e-12), k


def _identity_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nonrigid_nerf_amd.distributed import GradientBuckets
    a, b, c = (torch.full((3,), float(k + 1), requires_grad=True) for k in range(3))
    gb = GradientBuckets([[a, b, c]])
    log = []
    # steps 1, 2: only a and b take part -> the first step reduces in finish(), the second launches from inside backward
    for _ in range(2):
        gb.zero_grad()
        ((a * (rank + 1)).sum() + (b * 2).sum()).backward()
        log.append(gb.pending[0] is not None)
        gb.finish()
        assert torch.allclose(a.grad, torch.full((3,), 1.5)) and torch.allclose(b.grad, torch.full((3,), 2.0))
    # step 3: a DIFFERENT pair (a, c) of the same size: counting arrivals would launch after two gradients with b's slot
    # stale-zero and c's possibly missing; by identity the bucket waits (b never comes) and is reduced in finish()
    gb.zero_grad()
    ((a * (rank + 1)).sum() + (c * 4).sum()).backward()
    log.append(gb.pending[0] is not None)
    gb.finish()
    assert torch.allclose(a.grad, torch.full((3,), 1.5)) and torch.allclose(c.grad, torch.full((3,), 4.0)) and float(b.grad.abs().sum()) == 0
    # step 4: all three, c produced AFTER the expected set {a, c} ... order of arrival is autograd's; a superset never
    # launches early on a subset it was not told about: expected is now {a, c}, b is extra
    gb.zero_grad()
    try:
        ((a * (rank + 1)).sum() + (c * 4).sum() + (b * 2).sum()).backward()
        gb.finish()
        late = False
    except RuntimeError as e:
        late = "arrived after" in str(e)
    if late:                                     # the documented recovery: expectations were reset, repeat the step
        gb.zero_grad()
        ((a * (rank + 1)).sum() + (c * 4).sum() + (b * 2).sum()).backward()
        gb.finish()
    assert torch.allclose(a.grad, torch.full((3,), 1.5)) and torch.allclose(b.grad, torch.full((3,), 2.0)) and torch.allclose(c.grad, torch.full((3,), 4.0))
    gb.close()
    torch.save(log, os.path.join(out_dir, f"log{rank}.pt"))
    dist.destroy_process_group()


def test_gradient_buckets_track_parameters_by_identity(tmp_path):
    """A step that produces gradients for a different set of parameters than the previous one (a regulariser switched
    on, another head) must not launch a bucket's all-reduce after "the first k arrivals": the bucket waits for the exact
    parameters it expects, reduces in finish() when they do not all come, and never hands out a silently wrong mean."""
    mp.spawn(_identity_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        log = torch.load(os.path.join(str(tmp_path), f"log{r}.pt"))
        assert log == [False, True, False], log


def _native_train_grads(lo, hi, n, gb_factory=None):
    """One native training step's gradients (HIP path, fp32 mode) on rays [lo, hi) of a fixed n-ray batch."""
    from nonrigid_nerf_amd import render as R
    from nonrigid_nerf_amd.synthetic import build_modules
    dev = torch.device("cuda:0")
    cfg = SceneConfig(N_importance=32)
    scene = make_scene(cfg, 1)
    rays, latents = make_rays(n, 9, cfg)
    rb, coarse, fine = build_modules(scene, device=dev)
    for m in (rb, coarse, fine):
        m.requires_grad_(True)
    lat = latents.to(dev).clone().requires_grad_(True)
    target = torch.rand(n, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    R.set_precision("f32")
    gb = gb_factory(coarse, fine, rb, lat) if gb_factory else None
    issued = None
    for step in range(2 if gb else 1):           # second step: the buckets know which parameters never get a gradient
        if gb:
            gb.zero_grad()
        out = R.render_rays(rays[lo:hi].to(dev), coarse, None, cfg.N_samples, N_importance=cfg.N_importance, network_fine=fine,
                            additional_pixel_information={"ray_bending_latents": lat[lo:hi]})
        loss = ((out["rgb_map"] - target[lo:hi]) ** 2).mean() + ((out["rgb0"] - target[lo:hi]) ** 2).mean()
        loss.backward()
        if gb:
            issued = all(p is not None for p in gb.pending)
            gb.finish()
    grads = {"latents": lat.grad.detach().cpu().clone()}
    for part, mod in (("bender", rb), ("coarse", coarse), ("fine", fine)):
        grads.update({f"{part}.{k}": p.grad.detach().cpu().clone() for k, p in mod.named_parameters() if p.grad is not None})
    torch.cuda.synchronize()
    return grads, issued


def _native_dp_worker(rank, world, port, backend, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from nonrigid_nerf_amd.distributed import GradientBuckets, render_buckets
    lo, hi, _ = shard_bounds(n, world, rank)
    grads, issued = _native_train_grads(lo, hi, n, lambda c, f, rb, lat: GradientBuckets(render_buckets(c, f, rb, lat), force_collective=True))
    torch.save({"grads": grads, "issued": issued, "backend": dist.get_backend()}, os.path.join(out_dir, f"dp{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,backend", [(2, "gloo"), (1, "nccl")], ids=["two_ranks_one_gpu_gloo", "one_rank_rccl"])
def test_data_parallel_native_training_step(tmp_path, world, backend):
    """GradientBuckets around the NATIVE training step: (a) two ranks on GPU 0 over gloo (what a 1-GPU box can run of the
    N > 1 training path) render half of a 64-ray batch each -- both end up with the gradient of the full-batch mean loss;
    (b) the same code over the real backend (torch "nccl" = RCCL) with a one-rank group, collectives forced.  Every
    bucket's all-reduce must have been issued from inside backward (overlap with the rest of backward)."""
    n = 64
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_native_dp_worker, args=(r, world, port, backend, n, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    want = {}
    for r in range(world):                         # single process, same kernels: mean over the shards' gradients
        lo, hi, _ = shard_bounds(n, world, r)
        g, _ = _native_train_grads(lo, hi, n)
        for k, v in g.items():
            want[k] = want.get(k, 0) + v / world
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), f"dp{r}.pt"))
        assert res["issued"] and res["backend"] == backend
        extra = set(res["grads"]) - set(want)            # parameters render_rays never touches (views_linears): zero, not None
        assert set(want) <= set(res["grads"]) and all(not res["grads"][k].any() for k in extra), extra
        for k, g in want.items():
            assert torch.allclose(res["grads"][k], g, rtol=1e-5, atol=1e-6 * float(g.abs().max()) + 1e-12), (r, k)


def test_bench_picks_the_fitted_checkpoint_of_the_requested_architecture_family():
    """bench.py renders every compiled architecture family on ITS fitted checkpoint (so `psnr_vs_oracle_db` of a non-default
    line is measured on a realistic model, not on the synthetic stress scene); other variants fall back to synthetic weights."""
    import argparse
    import importlib.util
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module2", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ns = lambda **kw: argparse.Namespace(**{**dict(use_viewdirs=False, bend_depth=5, netwidth=256, exact_viewdirs=False), **kw})
    base = lambda p: os.path.basename(p) if p else None
    assert base(bench.fitted_checkpoint_for(ns())) == "fitted_latest.tar"
    assert base(bench.fitted_checkpoint_for(ns(use_viewdirs=True, bend_depth=7))) == "fitted_config4.tar"
    assert base(bench.fitted_checkpoint_for(ns(netwidth=128))) == "fitted_w128.tar"
    for other in (ns(netwidth=192), ns(use_viewdirs=True), ns(bend_depth=7), ns(use_viewdirs=True, bend_depth=7, exact_viewdirs=True)):
        assert bench.fitted_checkpoint_for(other) is None


# ---- frame-sharded render_path (the no-collective partitioning; reference train.py:419-553) ---------------------------
def _closed_form_frame(i, c2w, intrin, code):
    """A stand-in renderer on the CPU tier: every pixel names its frame, its own index and the frame's latent code, so a frame that lands in the
    wrong slot (or pixels of another rank's frame) shows.  The HIP path plugs in on the GPU tier (tests/test_gpu_parity.py)."""
    H, W = int(intrin["height"]), int(intrin["width"])
    px = torch.arange(H * W, dtype=torch.float32)
    rgb = torch.stack([(px % 7) / 7.0, torch.full_like(px, (i % 5) / 5.0), torch.full_like(px, float(code[0, 0]))], -1)
    return {"rgb_map": rgb, "disp_map": px + 1000.0 * i + float(c2w[0, 3]), "acc_map": torch.ones(H * W),
            "surface_pts": rgb + 1.0, "surface_rigidity": px * 0.0 + i, "median_index": (px % 3).to(torch.int32)}


def _frames_job(F):
    poses = [torch.eye(4)[:3] + 0.01 * f for f in range(F)]
    intr = [dict(height=3, width=5, focal_x=4.0, focal_y=4.0, center_x=2.5, center_y=1.5) for _ in range(F)]
    codes = torch.linspace(0.0, 1.0, max(F, 1))[:, None].expand(-1, 4).contiguous()
    return poses, intr, codes


def _worker_frames(rank, world, port, F, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from nonrigid_nerf_amd.driver import frame_shard, render_path
    poses, intr, codes = _frames_job(F)
    kw = dict(near=0.1, far=1.0, network_fn=None)
    calls = []

    def fn(i, c2w, intrin, code):
        calls.append(i)
        return _closed_form_frame(i, c2w, intrin, code)

    res = {}
    for dt in ("float32", "uint8"):
        calls.clear()
        res["all_" + dt] = render_path(poses, intr, 1024, kw, codes, rgb_dtype=dt, group=True, gather="all", _frame_fn=fn, surface_outputs=(dt == "uint8"))
        assert calls == frame_shard(F, world, rank), (calls, rank)         # this rank rendered its own frames and nothing else
    res["root"] = render_path(poses, intr, 1024, kw, codes, group=True, gather="root", _frame_fn=fn)
    res["none"] = render_path(poses, intr, 1024, kw, codes, group=True, gather=None, _frame_fn=fn)
    torch.save(res, os.path.join(out_dir, f"frames_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,F", [(2, 5), (2, 1), (8, 3), (8, 19)])
def test_frame_sharded_render_path_matches_the_single_process_sequence(tmp_path, world, F):
    """`driver.render_path(..., group=)`: rank r renders frames r, r + G, ... and ONE all-gather at the end gives every rank the sequence the
    single-process call returns -- ragged frame counts (F = 5 over 2 ranks, 19 over 8), fewer frames than ranks (F = 1 over 2, 3 over 8: five
    ranks render nothing and still take part in the collective), float and uint8 frames, gather "all" / "root" / None."""
    import numpy as np
    from nonrigid_nerf_amd.driver import frame_shard, render_path
    mp.spawn(_worker_frames, args=(world, _free_port(), F, str(tmp_path)), nprocs=world, join=True)
    poses, intr, codes = _frames_job(F)
    kw = dict(near=0.1, far=1.0, network_fn=None)
    want = {dt: render_path(poses, intr, 1024, kw, codes, rgb_dtype=dt, _frame_fn=_closed_form_frame, surface_outputs=(dt == "uint8")) for dt in ("float32", "uint8")}
    assert want["uint8"][0].dtype == np.uint8 and want["float32"][0].shape == (F, 3, 5, 3)
    for rank in range(world):
        res = torch.load(os.path.join(str(tmp_path), f"frames_{rank}.pt"), weights_only=False)
        mine = frame_shard(F, world, rank)
        for dt in ("float32", "uint8"):
            got = res["all_" + dt]
            assert np.array_equal(got[0], want[dt][0]) and np.array_equal(got[1], want[dt][1]), (rank, dt)
        det = res["all_uint8"][2]            # surface outputs stay rank-local: an entry per frame, None for the other ranks' frames
        assert len(det) == F and all((det[f] is not None) == (f in mine) for f in range(F))
        for f in mine:
            assert all(np.array_equal(det[f][k], want["uint8"][2][f][k]) for k in ("surface_pts", "surface_rigidity", "median_index"))
        own_rgb, own_disp = want["float32"][0][mine], want["float32"][1][mine]
        if rank == 0:
            assert np.array_equal(res["root"][0], want["float32"][0]) and np.array_equal(res["root"][1], want["float32"][1])
        else:
            assert np.array_equal(res["root"][0], own_rgb) and np.array_equal(res["root"][1], own_disp), rank
        assert res["none"][0].shape[0] == len(mine) and np.array_equal(res["none"][0], own_rgb) and np.array_equal(res["none"][1], own_disp)
