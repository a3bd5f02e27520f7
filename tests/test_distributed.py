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
