"""Frame driver: drop-in for the reference's ``render_path`` (train.py:419-553) on top of the HIP path.

What changes relative to the reference loop:

* rays are generated on the device from ``(c2w, intrinsics)`` by ``nrnerf_generate_rays`` (reference get_rays,
  run_nerf_helpers.py:588-605, plus render()'s packing, train.py:380-399) -- 12 floats of input per frame;
* the frame's latent code is passed once (``latent_stride = 0``) instead of being expanded per pixel
  (train.py:464-466) and per sample (train.py:82-87);
* one ``batchify_rays`` launch sequence per frame, no chunk loop, no DataParallel scatter/gather;
* frame f+1 is enqueued while frame f's pixels travel to pinned host memory on a side stream (the reference
  blocks on ``.cpu().numpy()`` for every output key of every frame, train.py:481-497).

Image writing (``savedir``) is not reproduced: it is host-side I/O that needs imageio (train.py:506-545); with
``rgb_dtype="uint8"`` the frames come back already converted the way the reference converts them for writing
(``to8b``, run_nerf_helpers.py:19), 3 bytes per pixel over PCIe instead of 12.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from . import render as R


def generate_rays(c2w, intrin: dict, near: float, far: float, use_viewdirs: bool, device) -> torch.Tensor:
    """``rays [H*W, 8|11]`` on ``device`` for one camera (row-major pixels), via the C ABI."""
    lib = _lib.load()
    H, W = int(intrin["height"]), int(intrin["width"])
    cam = _lib.Camera()
    m = np.asarray(torch.as_tensor(c2w).detach().cpu().float().numpy())[:3, :4].reshape(-1)
    for k in range(12):
        cam.c2w[k] = float(m[k])
    cam.focal_x, cam.focal_y = float(intrin["focal_x"]), float(intrin["focal_y"])
    cam.center_x, cam.center_y = float(intrin["center_x"]), float(intrin["center_y"])
    cam.height, cam.width = H, W
    stride = 11 if use_viewdirs else 8
    dev = torch.device(device)
    rays = torch.empty(H * W, stride, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.nrnerf_generate_rays(C.byref(cam), float(near), float(far), C.c_void_p(rays.data_ptr()),
                                            stride, C.c_void_p(stream)), "nrnerf_generate_rays")
    return rays


def frame_shard(n_frames: int, world: int, rank: int):
    """Frames of rank ``rank`` in a frame-sharded sequence: r, r + G, r + 2 G, ... (round robin, so consecutive frames -- which
    cost about the same -- spread evenly and a ragged tail costs at most one frame).  ``frame_slot`` is the inverse."""
    return list(range(rank, n_frames, world))


def frame_slot(frame: int, world: int, per: int) -> int:
    """Row of ``frame`` in the all-gathered stack ``[world * per, ...]`` (rank-major blocks of ``per = ceil(F / world)`` frames)."""
    return (frame % world) * per + frame // world


def _gather_frames(stack, n_frames, world, per, group, dst):
    """All ranks' frame stacks ``[per, ...]`` -> ``[n_frames, ...]`` in frame order, on every rank (``dst`` None) or on rank ``dst`` only
    (None elsewhere).  One collective for the whole sequence: 3 B/pixel with uint8 frames (1.9 GB for 300 frames of 1080p)."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    if stack.is_cuda and dist.get_backend(group) == "gloo":      # (one-GPU functional runs of the multi-process path: gloo moves host tensors)
        stack = stack.cpu()
    if dst is None:
        full = torch.empty((world * per,) + tuple(stack.shape[1:]), dtype=stack.dtype, device=stack.device)
        dist.all_gather_into_tensor(full, stack.contiguous(), group=group)
    else:
        parts = [torch.empty_like(stack) for _ in range(world)] if rank == dst else None
        dist.gather(stack.contiguous(), parts, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        if rank != dst:
            return None
        full = torch.cat(parts, 0)
    order = torch.tensor([frame_slot(f, world, per) for f in range(n_frames)], dtype=torch.long, device=full.device)
    return full.index_select(0, order)


def render_path(render_poses, intrinsics, chunk, render_kwargs, ray_bending_latents, gt_imgs=None, savedir=None,
                render_factor=0, detailed_output=False, parallelized_render_function=None, surface_outputs=False,
                rgb_dtype="float32", device=None, group=None, gather="all", _frame_fn=None):
    """Signature and return value of reference ``render_path`` (train.py:419-431, 547-553).

    ``render_kwargs`` is the dict ``create_nerf`` builds (train.py:698-719) plus ``near`` / ``far``; the networks
    are read from it.  ``parallelized_render_function`` (the DataParallel wrapper) is accepted and ignored: for
    several GPUs use one process per GPU -- ``group`` below for a sequence of frames, ``nonrigid_nerf_amd.distributed``
    for the rays of one frame.

    ``surface_outputs=True`` (extension) additionally returns, per frame, ``{"surface_pts" [H,W,3], "surface_rigidity"
    [H,W], "median_index" [H,W]}`` -- the reduction free_viewpoint_rendering.py:621-658 computes from the detailed
    outputs -- without moving the per-sample tensors to the host.

    ``rgb_dtype="uint8"`` (extension): ``rgbs`` is ``to8b`` of the render (run_nerf_helpers.py:19), converted on the
    device.  ``device`` (extension): where to render when the networks are host-resident weight holders
    (``checkpoint.load_checkpoint``); default: the networks' device if that is a GPU, else the current GPU.

    ``group`` (extension; SURVEY.md section 8e, the no-collective partitioning): a ``torch.distributed`` process group (``True`` =
    the default group) over which the SEQUENCE is sharded by whole frames -- rank r renders frames r, r + G, r + 2 G, ...
    (``frame_shard``) with its own packed weights, and nothing is exchanged while rendering: a free-viewpoint sequence
    (free_viewpoint_rendering.py:418-620 renders 120-300 poses through this function) scales with the rank count without a
    per-frame collective.  Every rank passes the same arguments.  ``gather``: "all" (default) = ONE all-gather of the finished
    frames at the end (with ``rgb_dtype="uint8"`` 3 B/pixel), every rank returns the whole sequence like the single-process call;
    "root" = gathered on rank 0 of the group only, the other ranks return their own frames; None = no collective at all, every
    rank returns its own frames (in its own order ``frame_shard(F, G, r)``).  Per-sample details / surface outputs are never
    gathered (15 KB per ray): with ``gather`` the list has one entry per frame, ``None`` for frames another rank rendered.
    The frames of a gathered sequence must share one image size.
    """
    if rgb_dtype not in ("float32", "uint8"):
        raise ValueError("rgb_dtype must be 'float32' or 'uint8'")
    if gather not in ("all", "root", None):
        raise ValueError("gather must be 'all', 'root' or None")
    if savedir is not None:
        raise NotImplementedError("image writing is host-side I/O outside the accelerated path (train.py:506-545)")
    if render_factor != 0:                                           # train.py:434-446
        scaled = []
        for intrin in intrinsics:
            s = dict(intrin)
            s["height"], s["width"] = intrin["height"] // render_factor, intrin["width"] // render_factor
            for k in ("focal_x", "focal_y", "center_x", "center_y"):
                s[k] = intrin[k] / render_factor
            scaled.append(s)
        intrinsics = scaled
    world, rank = 1, 0
    if group is not None:
        import torch.distributed as dist
        if group is True:
            group = None if not dist.is_initialized() else dist.group.WORLD
        if dist.is_initialized():
            world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_frames = min(len(render_poses), len(intrinsics))
    mine = frame_shard(n_frames, world, rank)
    kw = dict(render_kwargs)
    near, far = kw.pop("near"), kw.pop("far")
    use_viewdirs = bool(kw.pop("use_viewdirs", False))
    for k in ("ndc", "c2w_staticcam"):
        kw.pop(k, None)
    if _frame_fn is None:
        net = kw["network_fn"]
        dev = torch.device(device) if device is not None else next(net.parameters()).device
        if dev.type != "cuda":
            dev = torch.device("cuda", torch.cuda.current_device())
    else:                  # (tests: a renderer handed in, e.g. closed-form on the CPU tier -- the sharding / gather logic is the same)
        dev = torch.device(device) if device is not None else torch.device("cpu")
    cuda = dev.type == "cuda"
    copy_stream = torch.cuda.Stream(device=dev) if cuda else None
    sharded_gather = world > 1 and gather is not None
    stack_rgb = stack_disp = None          # (sharded + gather) this rank's frames, kept on the device until the one collective
    pending = []          # (pinned rgb, pinned disp, event, details)
    with torch.no_grad():
        for slot, i in enumerate(mine):
            c2w, intrin = render_poses[i], intrinsics[i]
            H, W = int(intrin["height"]), int(intrin["width"])
            code = torch.as_tensor(ray_bending_latents[i]).to(dev, torch.float32).reshape(1, -1)
            if _frame_fn is not None:
                out = _frame_fn(i, torch.as_tensor(c2w)[:3, :4], intrin, code)
            else:
                rays = generate_rays(torch.as_tensor(c2w)[:3, :4], intrin, near, far, use_viewdirs, dev)
                api = {"ray_bending_latents": code.expand(H * W, code.shape[-1])}          # stride-0 view, never materialised
                out = R.batchify_rays(rays, api, chunk=chunk, detailed_output=detailed_output, _surface=surface_outputs, **kw)
            rgb_d = out["rgb_map"]
            if rgb_dtype == "uint8":
                rgb_d = (255 * rgb_d.clamp(0, 1)).to(torch.uint8)          # to8b: clip, scale, truncate
            if sharded_gather:
                if stack_rgb is None:
                    per = (n_frames + world - 1) // world
                    stack_rgb = torch.zeros((per, H, W, 3), dtype=rgb_d.dtype, device=dev)
                    stack_disp = torch.zeros((per, H, W), dtype=torch.float32, device=dev)
                if tuple(stack_rgb.shape[1:3]) != (H, W):
                    raise ValueError("a gathered frame-sharded sequence needs frames of one size (pass gather=None otherwise)")
                stack_rgb[slot].copy_(rgb_d.view(H, W, 3))
                stack_disp[slot].copy_(out["disp_map"].view(H, W))
            keep_local = not sharded_gather
            if cuda:
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(dev))
                ctx = torch.cuda.stream(copy_stream)
            else:
                import contextlib
                ctx = contextlib.nullcontext()
            with ctx:
                if cuda:
                    copy_stream.wait_event(done)
                host = (lambda shape, dtype: torch.empty(shape, dtype=dtype, pin_memory=True)) if cuda else (lambda shape, dtype: torch.empty(shape, dtype=dtype))
                rgb_h = disp_h = None
                if keep_local:
                    rgb_h, disp_h = host((H, W, 3), rgb_d.dtype), host((H, W), torch.float32)
                    rgb_h.copy_(rgb_d.view(H, W, 3), non_blocking=True)
                    disp_h.copy_(out["disp_map"].view(H, W), non_blocking=True)
                details = None
                if detailed_output or surface_outputs:
                    details = {}
                    for k, v in out.items():
                        if k in ("rgb_map", "disp_map", "acc_map"):
                            continue
                        if not detailed_output and k not in ("surface_pts", "surface_rigidity", "median_index"):
                            continue
                        hbuf = host((H, W) + tuple(v.shape[1:]), v.dtype)
                        hbuf.copy_(v.view((H, W) + tuple(v.shape[1:])), non_blocking=True)
                        details[k] = hbuf
                ev = None
                if cuda:
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
            if cuda:
                for t in out.values():
                    t.record_stream(copy_stream)
            pending.append((rgb_h, disp_h, ev, details))
    want_details = detailed_output or surface_outputs
    rgbs, disps, my_details = [], [], []
    for rgb_h, disp_h, ev, details in pending:
        if ev is not None:
            ev.synchronize()
        if rgb_h is not None:
            rgbs.append(rgb_h.numpy())
            disps.append(disp_h.numpy())
        if want_details:
            my_details.append({k: v.numpy() for k, v in details.items()})
    if sharded_gather:
        if stack_rgb is None:           # more ranks than frames: this rank rendered nothing; the block size comes from the arguments
            per = (n_frames + world - 1) // world
            H, W = int(intrinsics[0]["height"]), int(intrinsics[0]["width"])
            stack_rgb = torch.zeros((per, H, W, 3), dtype=torch.uint8 if rgb_dtype == "uint8" else torch.float32, device=dev)
            stack_disp = torch.zeros((per, H, W), dtype=torch.float32, device=dev)
        dst = None if gather == "all" else 0
        all_rgb = _gather_frames(stack_rgb, n_frames, world, per, group, dst)
        all_disp = _gather_frames(stack_disp, n_frames, world, per, group, dst)
        if all_rgb is not None:
            rgbs_a, disps_a = all_rgb.cpu().numpy(), all_disp.cpu().numpy()
            all_details = [None] * n_frames
            for slot, i in enumerate(mine):
                if want_details:
                    all_details[i] = my_details[slot]
            return (rgbs_a, disps_a, all_details) if want_details else (rgbs_a, disps_a)
        rgbs = [stack_rgb[s].cpu().numpy() for s in range(len(mine))]       # gather="root", not the root: this rank's own frames
        disps = [stack_disp[s].cpu().numpy() for s in range(len(mine))]
    if rgbs:
        rgbs, disps = np.stack(rgbs, 0), np.stack(disps, 0)
    else:
        H, W = (int(intrinsics[0]["height"]), int(intrinsics[0]["width"])) if n_frames else (0, 0)
        rgbs = np.zeros((0, H, W, 3), dtype=np.uint8 if rgb_dtype == "uint8" else np.float32)
        disps = np.zeros((0, H, W), dtype=np.float32)
    if want_details:
        return rgbs, disps, my_details
    return rgbs, disps
