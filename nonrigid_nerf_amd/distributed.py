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


class GradientBuckets:
    """Data-parallel TRAINING: every rank renders its slice of the ray batch, gradients are averaged over ranks.

    Replaces ``nn.DataParallel(training_wrapper_class)`` (train.py:300-323, 1562-1573: per step the parameters are
    re-broadcast from GPU0, the ray batch scattered, the per-GPU losses gathered and averaged on GPU0, and backward runs
    through the replicas into GPU0's parameters).  Here every process owns a replica; the only exchange of a step is one
    all-reduce per bucket of gradients (RCCL over xGMI; a ring all-reduce of B bytes moves 2 B (G-1)/G per link, so the
    4.8 MB of this model's gradients are latency-bound -- few buckets, each overlapped with the part of backward still
    running):

      * ``buckets`` is a list of parameter lists in the order their gradients become ready in backward (for
        ``render_rays``: fine network, coarse network, ray bender + latent codes).  Each bucket owns ONE flat fp32
        buffer; the parameters' ``.grad`` are views into it, so there is no flatten / unflatten copy around the
        collective and zeroing the gradients is one fill per bucket (``zero_grad``; do NOT call
        ``optimizer.zero_grad(set_to_none=True)``, it would detach the views -- ``set_to_none=False`` is fine).
      * a post-accumulate hook per parameter counts arrivals; when a bucket is complete its all-reduce is issued
        asynchronously from a side stream that waits for the producing kernels, while autograd goes on with the next
        bucket's backward on the main stream;
      * ``finish()`` (after ``loss.backward()``, before ``optimizer.step()``) joins the collectives and divides by the
        world size (sum -> mean, matching the reference's ``loss.mean()`` over equally sized per-GPU slices).
    A bucket with parameters that never receive a gradient is reduced in ``finish()`` the first time and from then on as
    soon as the parameters that do have arrived.  Parameters that take part in no bucket are left alone.  With CPU tensors (gloo; the CPU test tier) there are no
    streams and the collectives are joined the same way.  ``force_collective``: run the collectives in a one-rank group
    too (single-GPU test of the RCCL path).
    """

    def __init__(self, buckets, group=None, force_collective: bool = False):
        self.group = group
        self.force_collective = force_collective
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = dist.is_initialized() and (self.world > 1 or force_collective)
        self.flat, self.params, self.pending, self._arrived, self._expected, self._hooks, self._late = [], [], [], [], [], [], []
        for plist in buckets:
            plist = [p for p in plist if p.requires_grad]
            if not plist:
                continue
            dev = plist[0].device
            if any(p.device != dev or p.dtype != torch.float32 for p in plist):
                raise ValueError("a bucket holds fp32 parameters of one device")
            flat = torch.zeros(sum(p.numel() for p in plist), dtype=torch.float32, device=dev)
            off = 0
            for p in plist:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            bi = len(self.flat)
            self.flat.append(flat)
            self.params.append(plist)
            self.pending.append(None)
            # which parameters have arrived this step / which ones the bucket waits for before it launches: tracked by
            # IDENTITY (index in the bucket), not by count -- a step that produces gradients for other parameters than the
            # previous one (a regulariser switched on, N_importance or the head toggled) must not launch the all-reduce
            # after "the first k arrivals"
            self._arrived.append(set())
            self._expected.append(frozenset(range(len(plist))))
            self._late.append(None)
            for pi, p in enumerate(plist):
                self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, bi=bi, pi=pi: self._on_grad(bi, pi)))
        self.device = self.flat[0].device if self.flat else torch.device("cpu")
        self.cuda = self.device.type == "cuda"
        self.side = torch.cuda.Stream(device=self.device) if (self.cuda and self.active) else None

    def zero_grad(self):
        for bi, flat in enumerate(self.flat):
            flat.zero_()
            self._arrived[bi] = set()
            self._late[bi] = None
            for p in self.params[bi]:          # re-attach a view someone replaced (e.g. zero_grad(set_to_none=True))
                if p.grad is None or p.grad.untyped_storage().data_ptr() != flat.untyped_storage().data_ptr():
                    raise RuntimeError("a parameter's .grad no longer lives in its bucket (optimizer.zero_grad(set_to_none=True)?)")

    def _on_grad(self, bi: int, pi: int):
        if self.pending[bi] is not None:
            # the bucket's all-reduce is already in flight on the side stream and this gradient was just written into the
            # buffer it reads: the result is undefined.  Remember it; finish() raises (see expect_all).
            self._late[bi] = pi
            return
        self._arrived[bi].add(pi)
        if self.active and self._arrived[bi] >= self._expected[bi] and pi in self._expected[bi]:
            self._reduce(bi)

    def expect_all(self):
        """Forget which parameters the previous steps produced gradients for: the next step's buckets are reduced in
        ``finish()`` (no overlap for one step) and re-learn the set.  Call it when the loss changes which parameters
        take part (a regulariser switched on or off, a different head), or after ``finish()`` raised."""
        for bi, plist in enumerate(self.params):
            self._expected[bi] = frozenset(range(len(plist)))

    def _reduce(self, bi: int):
        flat = self.flat[bi]
        if self.cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                self.pending[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self.pending[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Join the step's collectives; afterwards every ``.grad`` holds the mean over ranks."""
        if not self.active:
            return
        for bi in range(len(self.flat)):
            if self.pending[bi] is None:
                # some of the bucket's parameters received no gradient (e.g. NeRF.views_linears without use_viewdirs,
                # rnh:196-199): reduce now, and from the next step on launch as soon as the ones that do have arrived
                if self._arrived[bi] and self._arrived[bi] != self._expected[bi]:
                    self._expected[bi] = frozenset(self._arrived[bi])
                self._reduce(bi)
        for bi in range(len(self.flat)):
            self.pending[bi].wait()
            self.pending[bi] = None
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
        late = [(bi, pi) for bi, pi in enumerate(self._late) if pi is not None]
        if late:
            self.expect_all()
            bi, pi = late[0]
            raise RuntimeError(
                f"GradientBuckets: a gradient (bucket {bi}, parameter {pi}, shape {tuple(self.params[bi][pi].shape)}) arrived after "
                "its bucket's all-reduce had been issued -- this step produced gradients for parameters the previous steps did "
                "not, or backward() ran twice before finish().  The gradients of this step are invalid; the expectations were "
                "reset (expect_all), zero_grad() and repeat the step.")
        if self.world > 1:
            for flat in self.flat:
                flat.div_(self.world)

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def render_buckets(network_fn, network_fine, ray_bender, latents=None):
    """Bucket lists for ``GradientBuckets`` in the order backward produces them for ``render_rays``: fine network first,
    then the coarse network, then the ray bender together with the latent codes (both passes contribute to those)."""
    out = []
    if network_fine is not None:
        out.append(list(network_fine.parameters()))
    out.append(list(network_fn.parameters()))
    last = list(ray_bender.parameters()) if ray_bender is not None else []
    if latents is not None:
        last += list(latents) if isinstance(latents, (list, tuple)) else [latents]
    if last:
        out.append(last)
    return out
