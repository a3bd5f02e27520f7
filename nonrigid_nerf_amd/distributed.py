"""Multi-GPU rendering: rays sharded by rank, one all-gather of the rendered pixels.

Replaces the reference's ``torch.nn.DataParallel(render_wrapper_class)`` (train.py:300-323): there,
every call re-broadcasts all parameters from GPU0, scatters rays and gathers every output to GPU0
from one Python thread per GPU.  Here it is one process per GPU (``torch.distributed``; backend
"nccl" is RCCL over xGMI on ROCm), weights packed once per rank, rays split into contiguous
``ceil(n / G)`` slices exactly like DataParallel's dim-0 scatter, and a single all-gather of the
packed ``[rgb3, disp, acc]`` pixels (20 B/ray; 3.9 MB for a 512x384 frame).  Rays are independent
(SURVEY.md section 8e), so there is no other exchange step.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous ceil(n/world) slices, last ranks possibly short/empty (DataParallel's scatter rule)."""
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    return lo, min(lo + per, n), per


def render_sharded(render_fn, rays: torch.Tensor, latents: torch.Tensor | None, group=None, force_collective: bool = False):
    """Render ``rays`` cooperatively; every rank returns the full ``[n, 5]`` = (rgb, disp, acc) image.

    ``render_fn(rays_shard, latents_shard) -> dict`` with ``rgb_map [m,3]``, ``disp_map [m]``,
    ``acc_map [m]`` (e.g. a closure over ``nonrigid_nerf_amd.render.batchify_rays``).  Every rank
    must pass the same ``rays`` / ``latents`` (as DataParallel's caller does on GPU0).  ``force_collective`` runs the
    all-gather even in a one-rank group (exercises the RCCL path on a single-GPU box).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = rays.shape[0]
    lo, hi, per = shard_bounds(n, world, rank)
    packed = torch.zeros(per, 5, dtype=torch.float32, device=rays.device)
    if hi > lo:
        out = render_fn(rays[lo:hi], latents[lo:hi] if latents is not None else None)
        packed[:hi - lo, 0:3] = out["rgb_map"]
        packed[:hi - lo, 3] = out["disp_map"]
        packed[:hi - lo, 4] = out["acc_map"]
    if world == 1 and not (force_collective and dist.is_initialized()):
        return packed[:n]
    full = torch.empty(world * per, 5, dtype=torch.float32, device=rays.device)
    dist.all_gather_into_tensor(full, packed, group=group)
    return full[:n]


def gather_pixels(packed_local: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather equally sized per-rank pixel blocks ``[m, 5]`` -> ``[world * m, 5]`` (weak-scaling bench)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return packed_local
    full = torch.empty(world * packed_local.shape[0], packed_local.shape[1], dtype=packed_local.dtype,
                       device=packed_local.device)
    dist.all_gather_into_tensor(full, packed_local.contiguous(), group=group)
    return full


class OverlappedGather:
    """Frame f's all-gather overlapped with frame f+1's render (per frame the collective moves
    20 B/ray and is latency-, not bandwidth-bound, so it is hidden behind the next frame's kernels instead of being
    waited for).

    Two pixel blocks and two result buffers are used alternately.  ``submit(i, out)`` packs frame i's ``[rgb, disp,
    acc]`` into block ``i & 1`` on the render stream and issues ``all_gather_into_tensor`` asynchronously from a side
    stream that waits for the packing; the block pair is reused two frames later, after ``wait()`` on its previous
    collective (long finished by then).  ``drain()`` joins everything (end of a sequence / before reading a result).
    With CPU tensors (gloo, the CPU test tier) there are no streams: the collectives are still issued asynchronously
    and joined the same way.
    """

    def __init__(self, rows_per_rank: int, device, group=None, force_collective: bool = False):
        self.group = group
        self.force_collective = force_collective      # run the collective even in a one-rank group (1-GPU test of the RCCL path)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.packed = [torch.empty(rows_per_rank, 5, dtype=torch.float32, device=self.device) for _ in range(2)]
        self.full = [torch.empty(self.world * rows_per_rank, 5, dtype=torch.float32, device=self.device) for _ in range(2)]
        self.pending = [None, None]
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None

    def submit(self, i: int, out: dict) -> torch.Tensor:
        """Returns the buffer that will hold all ranks' pixels of frame i once its collective has completed."""
        b = i & 1
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        buf = self.packed[b]
        buf[:, 0:3] = out["rgb_map"]
        buf[:, 3] = out["disp_map"]
        buf[:, 4] = out["acc_map"]
        if self.world == 1 and not (self.force_collective and dist.is_initialized()):
            self.full[b].copy_(buf)
            return self.full[b]
        if self.cuda:
            main = torch.cuda.current_stream(self.device)
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                self.pending[b] = dist.all_gather_into_tensor(self.full[b], buf, group=self.group, async_op=True)
        else:
            self.pending[b] = dist.all_gather_into_tensor(self.full[b], buf, group=self.group, async_op=True)
        return self.full[b]

    def drain(self):
        for b in range(2):
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None
        if self.cuda and (self.world > 1 or self.force_collective):
            torch.cuda.current_stream(self.device).wait_stream(self.side)
