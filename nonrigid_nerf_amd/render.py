"""Drop-in replacements for the reference's ``render_rays`` / ``batchify_rays``.

Call contract mirrored:

* ``render()`` (train.py:402-408) calls ``batchify_rays(rays, additional_pixel_information,
  chunk=..., detailed_output=..., **kwargs)``, which calls ``render_rays(rays_flat[i:i+chunk],
  additional_pixel_information={"ray_bending_latents": ...}, detailed_output=..., **kwargs)``
  (train.py:125-130).  Both are looked up as module globals, so ``install(train)`` rebinds them
  for every caller (``render_path``, ``determine_nerf_volume_extent``, free_viewpoint_rendering.py).
* weights are read from the ``network_fn`` / ``network_fine`` modules (``pts_linears``,
  ``output_linear``) and from ``network_fn.ray_bender[0]`` (``network``, ``rigidity_network``);
  ``network_query_fn`` is ignored (the encoding / chunking it closes over is fused in the kernel).
* the editing knobs free_viewpoint_rendering.py:264-283 mutates on the modules are read per call.
* output: dict with exactly the reference's keys / shapes / dtypes (train.py:952-972), freshly
  allocated on the rays' device.

Everything is computed by ``libnrnerf_hip.so`` -- the compiled architectures on their specialised kernels, any other
depth / width / encoding / latent size on the run-time-parameterised one.  Calls the library has no kernel for
(training of a non-compiled architecture, exact non-rigid view directions there or under autograd, ...) are handed back to the
*reference's own* function when one was saved by ``install``; otherwise they raise.  There is
no CPU or PyTorch re-implementation in this package.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref

import numpy as np
import torch

from . import _lib

# Arithmetic of direct callers (bench.py, tests, Model(...)) when nothing is said: the headline bf16 mode.  `install()` --
# the two-line drop-in into an fp32 pipeline -- does NOT inherit this: without an explicit precision it selects "f32".
_DEFAULT_PRECISION = _lib.canonical_precision(os.environ.get("NRNERF_PRECISION", "bf16"))
_SCAN_OUTPUTS = os.environ.get("NRNERF_SCAN_OUTPUTS") == "1"
_fallbacks = {}          # {"render_rays": fn, "batchify_rays": fn} saved by install()


class FallbackWarning(UserWarning):
    """A call was handed to the reference's own function saved by install() (it runs, but not on the HIP path)."""


_fallback_seen = set()


def _note_fallback(entry: str, why: str):
    """Say ONCE per (entry point, reason) that a call went to the reference -- a training run that silently takes the eager path
    is ten times slower than it needs to be and nothing else would tell.  NRNERF_QUIET_FALLBACK=1 silences it."""
    if (entry, why) in _fallback_seen or os.environ.get("NRNERF_QUIET_FALLBACK") == "1":
        return
    _fallback_seen.add((entry, why))
    import warnings
    warnings.warn(f"nonrigid_nerf_amd: {entry} handed to the reference's own function ({why})", FallbackWarning, stacklevel=3)
_MAX_RAYS_PER_LAUNCH = 1 << 20


def set_precision(p: str):
    """Arithmetic type of the MLP contractions: "bf16" (default, headline), "f16" or "f32" (exact parity mode)."""
    global _DEFAULT_PRECISION
    _DEFAULT_PRECISION = _lib.canonical_precision(p)


def get_precision() -> str:
    return _DEFAULT_PRECISION


class Unsupported(NotImplementedError):
    """The requested configuration has no HIP kernel (and no reference function was saved to defer to)."""


# --------------------------------------------------------------------------------------------
# packed model handle
# --------------------------------------------------------------------------------------------
def _np32(t: torch.Tensor) -> np.ndarray:
    return np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())


class _Keep:
    """Keeps the numpy arrays and ctypes arrays referenced by a descriptor alive."""
    def __init__(self):
        self.objs = []

    def linear(self, mod) -> _lib.Linear:
        w = _np32(mod.weight)
        b = _np32(mod.bias) if getattr(mod, "bias", None) is not None else None
        self.objs += [w, b]
        fp = C.POINTER(C.c_float)
        return _lib.Linear(w.ctypes.data_as(fp), b.ctypes.data_as(fp) if b is not None else fp(),
                           w.shape[0], w.shape[1])

    def linear_array(self, mods):
        arr = (_lib.Linear * len(mods))(*[self.linear(m) for m in mods])
        self.objs.append(arr)
        return arr


def _mlp_desc(net, keep: _Keep) -> _lib.MlpDesc:
    d = _lib.MlpDesc()
    d.depth, d.width = int(net.D), int(net.W)
    skips = list(net.skips)
    if len(skips) > 1:
        raise Unsupported("more than one skip connection")
    d.skip = int(skips[0]) if skips else -1
    d.use_viewdirs = int(bool(net.use_viewdirs))
    d.time_conditioned = int(bool(getattr(net, "time_conditioned_baseline", False)))
    arr = keep.linear_array(list(net.pts_linears))
    d.pts_linears = C.cast(arr, C.POINTER(_lib.Linear))
    if net.use_viewdirs:
        d.output_ch = 4
        d.alpha_linear = keep.linear(net.alpha_linear)
        d.feature_linear = keep.linear(net.feature_linear)
        d.views_linear = keep.linear(net.views_linears[0])
        d.rgb_linear = keep.linear(net.rgb_linear)
    else:
        d.output_linear = keep.linear(net.output_linear)
        d.output_ch = int(net.output_linear.weight.shape[0])
    return d


def _bender_desc(rb, keep: _Keep) -> _lib.BenderDesc:
    if getattr(rb, "ray_bending_mode", "simple_neural") != "simple_neural":
        raise Unsupported(f"ray_bending_mode {rb.ray_bending_mode!r}")
    if not getattr(rb, "use_rigidity_network", True) or getattr(rb, "use_positionally_encoded_input", False):
        raise Unsupported("bender without rigidity network / with encoded input")
    if list(getattr(rb, "skips", [])) or list(getattr(rb, "rigidity_skips", [])):
        raise Unsupported("bender skip connections")
    d = _lib.BenderDesc()
    d.latent_size = int(rb.ray_bending_latent_size)
    d.depth = len(rb.network)
    d.hidden = int(rb.network[0].weight.shape[0])
    d.rigidity_depth = len(rb.rigidity_network)
    d.rigidity_hidden = int(rb.rigidity_network[0].weight.shape[0])
    na, ra = keep.linear_array(list(rb.network)), keep.linear_array(list(rb.rigidity_network))
    d.network = C.cast(na, C.POINTER(_lib.Linear))
    d.rigidity_network = C.cast(ra, C.POINTER(_lib.Linear))
    return d


def build_model_desc(network_fn, network_fine, precision: str, device_index: int, flags: int = 0):
    """ModelDesc for the C ABI from reference-style modules.  Returns (desc, keepalive)."""
    keep = _Keep()
    rb = network_fn.ray_bender[0] if getattr(network_fn, "ray_bender", None) else None
    desc = _lib.ModelDesc()
    desc.struct_size = C.sizeof(_lib.ModelDesc)
    desc.precision = _lib.PRECISIONS[precision]
    if (int(network_fn.input_ch) - 3) % 6:
        raise Unsupported("i_embed=-1 (identity embedding)")          # get_embedder, rnh:153-155
    desc.multires = (int(network_fn.input_ch) - 3) // 6
    icv = int(getattr(network_fn, "input_ch_views", 0))
    desc.multires_views = (icv - 3) // 6 if icv >= 3 else 0
    desc.device = device_index
    desc.flags = int(flags) & 0xffff            # (higher bits: Python-side markers, _lib.MODEL_PY_*)
    cm = _mlp_desc(network_fn, keep)
    keep.objs.append(cm)
    desc.coarse = C.pointer(cm)
    if network_fine is not None:
        fm = _mlp_desc(network_fine, keep)
        keep.objs.append(fm)
        desc.fine = C.pointer(fm)
    if rb is not None:
        bd = _bender_desc(rb, keep)
        keep.objs.append(bd)
        desc.bender = C.pointer(bd)
    # exact (Jacobian) view directions when the module asks for them (NeRF.approx_nonrigid_viewdirs, rnh:289-294)
    desc.exact_viewdirs = int(rb is not None and bool(getattr(network_fn, "use_viewdirs", False))
                              and not getattr(network_fn, "approx_nonrigid_viewdirs", True))
    if int(flags) & _lib.MODEL_PY_TRAINING_HANDLE:      # the caller computes the directions (training.render_rays_train)
        desc.exact_viewdirs = 0
    return desc, keep


def _linears_in_canonical_order(network_fn, network_fine):
    """nn.Linear modules in the order of the library's flat parameter vector (include/nrnerf.h,
    nrnerf_model_update_device): bender offset MLP, rigidity MLP, coarse network, fine network."""
    rb = network_fn.ray_bender[0] if getattr(network_fn, "ray_bender", None) else None
    mods = []
    if rb is not None:
        mods += list(rb.network) + list(rb.rigidity_network)
    for net in (network_fn, network_fine):
        if net is None:
            continue
        mods += list(net.pts_linears)
        if net.use_viewdirs:
            mods += [net.alpha_linear, net.feature_linear, net.views_linears[0], net.rgb_linear]
        else:
            mods.append(net.output_linear)
    return mods


def _flat_params(network_fn, network_fine, into=None):
    """All parameters as one fp32 vector in the library's canonical order, followed by the derived entries of
    nrnerf_model_update_device (networks with the view-dependent head: views_linears[0] with feature_linear folded in).
    ``into``: a (buffer, views) pair from an earlier call for the same parameter shapes -- refilled with one multi-tensor
    copy (no torch.cat: on ROCm a cat of ~90 tensors stages its argument table through host-to-device copies, every training
    step).  Returns (flat, state)."""
    parts = []
    for lin in _linears_in_canonical_order(network_fn, network_fine):
        parts.append(lin.weight.detach())
        if getattr(lin, "bias", None) is not None:
            parts.append(lin.bias.detach())
    if not parts or any(p.device != parts[0].device for p in parts) or parts[0].device.type != "cuda":
        return None, None
    for net in (network_fn, network_fine):        # derived entries: [W_v1 W_f | W_v2], W_v1 b_f + b_v  (fp32 products)
        if net is not None and net.use_viewdirs:
            wf, wv = net.feature_linear.weight.detach().float(), net.views_linears[0].weight.detach().float()
            bf, bv = net.feature_linear.bias, net.views_linears[0].bias
            k1 = int(wf.shape[0])
            folded = torch.empty(wv.shape[0], wf.shape[1] + wv.shape[1] - k1, dtype=torch.float32, device=wv.device)
            folded[:, :wf.shape[1]] = wv[:, :k1] @ wf
            folded[:, wf.shape[1]:] = wv[:, k1:]
            fb = bv.detach().float() if bv is not None else torch.zeros(wv.shape[0], dtype=torch.float32, device=wv.device)
            parts += [folded, torch.addmv(fb, wv[:, :k1], bf.detach().float()) if bf is not None else fb]
    shapes = tuple(tuple(p.shape) for p in parts)
    # parameters that already ARE one vector in this order (training.FusedAdam re-homes them so): that vector itself, nothing copied
    if all(p.dtype == torch.float32 and p.is_contiguous() for p in parts):
        end = parts[0].data_ptr()
        for p in parts:
            if p.data_ptr() != end:
                break
            end += 4 * p.numel()
        else:
            total = sum(p.numel() for p in parts)
            flat = parts[0].new_empty(0).set_(parts[0].untyped_storage(), parts[0].storage_offset(), (total,))
            return flat, None
    if into is None or into[2] != shapes or into[0].device != parts[0].device:
        flat = torch.empty(sum(p.numel() for p in parts), dtype=torch.float32, device=parts[0].device)
        views, o = [], 0
        for p in parts:
            views.append(flat[o:o + p.numel()].view(p.shape))
            o += p.numel()
        into = (flat, views, shapes)
    if all(p.dtype == torch.float32 for p in parts):
        torch._foreach_copy_(into[1], parts)
    else:
        for v, p in zip(into[1], parts):
            v.copy_(p)
    return into[0], into


_SLOTS = weakref.WeakKeyDictionary()


def _param_slots(m):
    """Every parameter slot of module ``m`` as (the owning submodule's ``_parameters`` dict, name).  ``m.parameters()`` walks
    the module tree through ``named_modules`` with de-duplication sets on every call -- 0.15 ms per network, three networks,
    three calls per training step; the walk is cached and re-validated by identity: every child still sits where it sat and
    every ``_modules`` / ``_parameters`` dict still has the same keys in the same order.  Reading the slot (not a cached tensor) sees a parameter
    object that was replaced."""
    ent = _SLOTS.get(m)
    if ent is not None and all(d.get(n) is c for d, n, c in ent[0]) and all(tuple(d) == k for d, k in ent[1]):
        return ent[2]
    children, sizes, slots, seen, stack = [], [], [], set(), [m]
    while stack:
        mod = stack.pop()
        if id(mod) in seen:
            continue
        seen.add(id(mod))
        sizes += [(mod._modules, tuple(mod._modules)), (mod._parameters, tuple(mod._parameters))]     # the key tuples, not only the lengths:
        # deleting one parameter and registering another under a different name keeps the length
        slots += [(mod._parameters, n) for n in mod._parameters]
        for n, c in mod._modules.items():
            children.append((mod._modules, n, c))
            if c is not None:
                stack.append(c)
    _SLOTS[m] = (children, sizes, slots)
    return slots


# Optimiser steps the version counters do not see.  ``torch.optim.Adam(fused=True)`` -- and every other fused optimiser --
# updates its parameters through one multi-tensor kernel that leaves ``Tensor._version`` alone (measured on torch 2.10:
# tools/experiments/debug_refresh_path.py), so (data_ptr, _version) alone would keep serving the weights packed BEFORE the step.
# A global post-step hook on all optimisers counts steps; the count is part of the fingerprint of every model that has a
# trainable parameter (frozen networks, whatever optimiser steps elsewhere, keep their handle untouched).
# (round 5, ADVICE r4: counted PER PARAMETER -- the hook bumps a counter on every parameter of the optimiser that stepped -- so a step of an
#  unrelated optimiser (one that only steps latent codes, or another model's) no longer forces a re-pack of every cached trainable model)
_OPT_HOOK = []


def _watch_optimizers():
    if not _OPT_HOOK:
        from torch.optim.optimizer import register_optimizer_step_post_hook

        def _count(optimizer, _args, _kwargs):
            for group in optimizer.param_groups:
                for p in group["params"]:
                    p._nrnerf_steps = getattr(p, "_nrnerf_steps", 0) + 1
            # an optimiser whose step re-packed the handles itself (training.FusedAdam: nrnerf_adam_step) says so: their
            # fingerprints are brought up to date instead of triggering a second re-pack at the next get_model
            after = getattr(optimizer, "_nrnerf_after_step", None)
            if after is not None:
                after()
        _OPT_HOOK.append(register_optimizer_step_post_hook(_count))


def _fingerprint(mods):
    fp = []
    for m in mods:
        if m is None:
            fp.append(None)
            continue
        ps = [p for p in (d[n] for d, n in _param_slots(m)) if p is not None]
        fp.append(tuple((p.data_ptr(), p._version, getattr(p, "_nrnerf_steps", 0)) for p in ps))
    return tuple(fp)


class Model:
    """Owns one ``nrnerf_model`` handle (packed weights resident in HBM on one device)."""

    def __init__(self, network_fn, network_fine=None, precision: str | None = None, device=None, flags: int | None = None):
        self.lib = _lib.load()
        self.flags = _lib.model_flags_from_env() if flags is None else int(flags)          # nrnerf_model_flags
        self.precision = _lib.canonical_precision(precision or _DEFAULT_PRECISION)
        dev = torch.device(device if device is not None else next(network_fn.parameters()).device)
        if dev.type != "cuda":
            raise RuntimeError("nonrigid_nerf_amd renders on a ROCm device only (got %s)" % dev)
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        desc, keep = build_model_desc(network_fn, network_fine, self.precision, self.device.index, self.flags)
        self.has_bender = bool(desc.bender)
        # nrnerf_bender_* / nrnerf_bender_divergence_* are available (the library's own rule: training_eligible in
        # csrc/nrnerf_api.cpp): a bender, not an f16 handle
        self.trains_bender = bool(desc.bender) and self.precision != "f16"
        self.needs_latents = self.has_bender or bool(desc.coarse.contents.time_conditioned)
        self.latent_size = desc.bender.contents.latent_size if self.has_bender else \
            (int(getattr(network_fn, "ray_bending_latent_size", 0)) if self.needs_latents else 0)
        self.output_ch = desc.fine.contents.output_ch if desc.fine else desc.coarse.contents.output_ch
        self.coarse_output_ch = desc.coarse.contents.output_ch
        handle = C.c_void_p()
        _lib.check(self.lib.nrnerf_model_create(C.byref(desc), C.byref(handle)), "nrnerf_model_create")
        self.handle = handle
        self.generic = self.lib.nrnerf_model_is_generic(handle) == 1      # the run-time-parameterised kernel (csrc/nrnerf_generic.h)
        self.trains_generic = self.lib.nrnerf_model_trains_generic(handle) == 1
        if self.generic:            # the bender's training kernels are compiled per BENDER shape: a generic handle has them when its bender has one
            self.trains_bender = self.lib.nrnerf_model_trains_bender(handle) == 1
        self._ws = {}            # stream -> workspace: concurrent renders on different streams never share scratch
        self._ws_lock = threading.Lock()
        self._render_events = {} # stream -> event recorded after the last render queued there (see update_from_device)
        del keep

    def update(self, network_fn, network_fine=None) -> bool:
        """Re-pack changed weights of the same architecture into this handle (``nrnerf_model_update``: no allocation,
        ordered after the work queued on the current stream).  Returns False when the modules describe a different
        model (the caller then builds a new handle)."""
        desc, keep = build_model_desc(network_fn, network_fine, self.precision, self.device.index, self.flags)
        with torch.cuda.device(self.device):
            # renders queued on OTHER streams from this cached handle may still be reading the weight buffers: wait for
            # the whole device before overwriting them (updates are rare: an optimiser step or a load_state_dict)
            torch.cuda.synchronize(self.device)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self.lib.nrnerf_model_update(self.handle, C.byref(desc), C.c_void_p(stream))
        del keep
        if rc == _lib.ERR_INVALID:
            return False
        _lib.check(rc, "nrnerf_model_update")
        return True

    def update_from_device(self, network_fn, network_fine=None) -> bool:
        """The same refresh without a host round trip (``nrnerf_model_update_device``): the parameters -- resident on this
        model's device -- are concatenated into one fp32 vector in the library's canonical order and re-packed by a
        gather kernel per image, asynchronously on the current stream.  What every training step does after
        ``optimizer.step()``.  Returns False when the modules are not on this device or describe a different model."""
        with torch.cuda.device(self.device):
            # Stream-ordered, no host synchronisation: the gather kernels run on the current stream after every render /
            # training kernel already queued there.  Work this handle has queued on OTHER streams -- renders, an earlier
            # refresh (whose gather kernels read the persistent flat buffer refilled below), training kernels -- is ordered
            # by events / a stream wait, not by draining the device: this runs once per training step.
            cur = torch.cuda.current_stream(self.device)
            with self._ws_lock:
                pending, self._render_events = self._render_events, {}
            # (while the current stream is being captured -- GraphedStep's captured refresh -- there is nothing to wait for and nothing
            #  that may be waited for: events recorded outside a capture cannot be joined from inside it; the capture is preceded by a
            #  device synchronisation, GraphedStep.__init__)
            capturing = torch.cuda.is_current_stream_capturing()
            for sid, ev in pending.items():
                if sid != cur.cuda_stream and not capturing:
                    cur.wait_event(ev)
            ts = getattr(self, "_train_stream", None)       # last stream a training kernel read these weights on (training._mstream)
            if ts is not None and ts.cuda_stream != cur.cuda_stream and not capturing:
                cur.wait_stream(ts)
            flat, self._flat_state = _flat_params(network_fn, network_fine, getattr(self, "_flat_state", None))
            if flat is None or flat.device != self.device or int(self.lib.nrnerf_model_flat_size(self.handle)) != flat.numel():
                return False
            stream = cur.cuda_stream
            rc = self.lib.nrnerf_model_update_device(self.handle, C.c_void_p(flat.data_ptr()), flat.numel(), C.c_void_p(stream))
            # the gather kernels read the persistent flat buffer: a later refresh from ANOTHER stream must not refill it
            # under them (and must see these packed images complete), so it waits for this event first -- via note_use's table
            self.note_use()
        if rc in (_lib.ERR_INVALID, _lib.ERR_UNSUPPORTED):
            return False
        _lib.check(rc, "nrnerf_model_update_device")
        return True                                       # (the flat buffer is persistent: self._flat_state)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.nrnerf_model_destroy(self.handle)
            self.handle = None

    __del__ = close

    def _workspace(self, nbytes: int, stream: int) -> torch.Tensor:
        """Scratch of the launch sequence, cached per stream (allocated while that stream is current, so the caching
        allocator's stream-ordered reuse rule covers a later, larger re-allocation)."""
        with self._ws_lock:
            ws = self._ws.get(stream)
            if ws is None or ws.numel() < nbytes:
                ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=self.device)
                self._ws[stream] = ws
            return ws

    def profile_begin(self):
        _lib.check(self.lib.nrnerf_profile_begin(self.handle), "nrnerf_profile_begin")

    def profile_end(self) -> dict:
        p = _lib.Profile()
        _lib.check(self.lib.nrnerf_profile_end(self.handle, C.byref(p)), "nrnerf_profile_end")
        return {name: dict(ms=p.ms[i], launches=p.launches[i], flops=p.flops[i], mfma_flops=p.mfma_flops[i], kernel=p.kernel_name[i].value.decode())
                for i, name in enumerate(_lib.KERNEL_NAMES)}

    def render(self, rays: torch.Tensor, latents: torch.Tensor | None, N_samples: int, N_importance: int = 0,
               retraw: bool = False, detailed_output: bool = False, rigidity_cutoff=None, test_time_scaling=None,
               removal_threshold=None, want_z_vals: bool = False, surface: bool = False, lindisp: bool = False,
               white_bkgd: bool = False, randoms: dict | None = None) -> dict:
        """One ``render_rays`` worth of work on ``rays [N, 8|11]``; returns the reference's output dict."""
        N = int(rays.shape[0])
        S, I = int(N_samples), int(N_importance)
        SF = S + I
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        rays = rays.to(**f32).contiguous()
        a = _lib.RenderArgs()
        a.struct_size = C.sizeof(_lib.RenderArgs)
        a.n_rays, a.n_samples, a.n_importance = N, S, I
        a.lindisp, a.white_bkgd = int(bool(lindisp)), int(bool(white_bkgd))
        keep_alive = []
        for key, shape in (("u_coarse", (N, S)), ("noise_coarse", (N, S)), ("u_fine", (N, I)), ("noise_fine", (N, SF))):
            t = (randoms or {}).get(key)
            if t is not None:
                t = t.to(**f32).contiguous()
                if tuple(t.shape) != shape:
                    raise ValueError(f"{key} must have shape {shape}, got {tuple(t.shape)}")
                keep_alive.append(t)
                setattr(a, key, t.data_ptr())
        a.rays, a.ray_stride = rays.data_ptr(), rays.shape[1]
        if self.needs_latents:
            if latents is None:
                raise ValueError("ray_bending_latents are required (ray bender or time-conditioned baseline)")
            # the kernel reads latent_size floats per ray: a wrong shape would be a silent out-of-bounds read where the
            # reference raises in expand / split (train.py:82-87, run_nerf_helpers.py:246)
            if latents.dim() != 2 or latents.shape[0] != N or latents.shape[1] != self.latent_size:
                raise ValueError(f"ray_bending_latents must have shape ({N}, {self.latent_size}), got {tuple(latents.shape)}")
            if latents.dim() == 2 and latents.shape[0] == N and latents.stride(0) == 0 and latents.stride(1) == 1 \
                    and latents.dtype == torch.float32 and latents.device == dev:
                a.latents, a.latent_stride = latents.data_ptr(), 0          # frame code expanded per ray (train.py:465)
            else:
                latents = latents.to(**f32).contiguous()
                a.latents, a.latent_stride = latents.data_ptr(), latents.shape[1]
        out = {}

        def new(key, *shape):
            t = torch.empty(*shape, **f32)
            out[key] = t
            return t.data_ptr()

        a.rgb_map, a.disp_map, a.acc_map = new("rgb_map", N, 3), new("disp_map", N), new("acc_map", N)
        if retraw:
            a.raw = new("raw", N, SF, self.output_ch if I > 0 else self.coarse_output_ch)
        if I > 0:
            a.rgb0, a.disp0, a.acc0, a.z_std = new("rgb0", N, 3), new("disp0", N), new("acc0", N), new("z_std", N)
        if want_z_vals:
            a.z_vals = new("_z_vals", N, SF)
        if surface:      # not reference keys: the reduction free_viewpoint_rendering.py:621-658 does on the host
            a.surface_pts, a.surface_rigidity = new("surface_pts", N, 3), new("surface_rigidity", N)
            idx = torch.empty(N, dtype=torch.int32, device=dev)
            out["median_index"] = idx
            a.median_index = idx.data_ptr()
        if detailed_output:
            def fill(so, prefix, ns):
                so.visibility_weights = new(prefix + "visibility_weights", N, ns)
                so.opacity_alpha = new(prefix + "opacity_alpha", N, ns)
                so.initial_input_pts = new(prefix + "initial_input_pts", N, ns, 3)
                so.input_pts = new(prefix + "input_pts", N, ns, 3)
                if self.has_bender:
                    so.unmasked_offsets = new(prefix + "unmasked_offsets", N, ns, 3)
                    so.masked_offsets = new(prefix + "masked_offsets", N, ns, 3)
                    so.rigidity_mask = new(prefix + "rigidity_mask", N, ns, 1)
            fill(a.coarse, "", S)
            if I > 0:
                fill(a.fine, "fine_", SF)
        a.detailed_output = int(detailed_output)
        a.flags = _lib.render_flags_from_env()
        if rigidity_cutoff is not None:
            a.has_rigidity_cutoff, a.rigidity_cutoff = 1, float(rigidity_cutoff)
        if test_time_scaling is not None:
            a.has_test_time_scaling, a.test_time_scaling = 1, float(test_time_scaling)
        if removal_threshold is not None:
            a.has_removal_threshold, a.removal_threshold = 1, float(removal_threshold)
        nbytes = self.lib.nrnerf_workspace_bytes(self.handle, N, S, I)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            ws = self._workspace(nbytes + 256, stream)
            base = (ws.data_ptr() + 255) // 256 * 256
            a.workspace, a.workspace_bytes = base, nbytes
            _lib.check(self.lib.nrnerf_render(self.handle, C.byref(a), C.c_void_p(stream)), "nrnerf_render")
            self.note_use(dev)
        return out

    def note_use(self, dev=None):
        """Kernels reading this handle's weights were just queued on the current stream: remember an event there, so that
        a later weight refresh issued from ANOTHER stream can order itself after them (update_from_device).  The
        training entry points (nonrigid_nerf_amd/training.py) do not record: a training step is single-stream by
        contract -- forward, backward, optimiser step and the refresh all run on the stream that is current."""
        if torch.cuda.is_current_stream_capturing():
            return              # (a captured record could never be waited on outside its graph: GraphedRender records after each replay)
        cur = torch.cuda.current_stream(dev if dev is not None else self.device)
        with self._ws_lock:
            ev = self._render_events.get(cur.cuda_stream)
            if ev is None:
                ev = self._render_events[cur.cuda_stream] = torch.cuda.Event()
        ev.record(cur)


# --------------------------------------------------------------------------------------------
# model cache (weights are packed once per (modules, version) -- no per-call broadcast as in DataParallel)
# --------------------------------------------------------------------------------------------
_cache = weakref.WeakKeyDictionary()       # network_fn -> {key: (fingerprint, Model | Unsupported)}
_by_bender = weakref.WeakKeyDictionary()   # ray bender -> (weakref network_fn, weakref network_fine | None, precision) of its last get_model
_cache_lock = threading.Lock()


def _bender_of(network_fn):
    rb = getattr(network_fn, "ray_bender", None)
    return rb[0] if rb else None


def _wref(obj):
    """Weak reference usable in a cache key: equal while both referents are the same live object, never equal to a new
    object that happens to reuse a dead one's address (``id()`` can)."""
    return None if obj is None else weakref.ref(obj)


def get_model(network_fn, network_fine=None, precision: str | None = None, device=None, flags: int = 0) -> Model:
    """The packed-weight handle for these modules on ``device`` (created, refreshed in place, or served from the cache).

    Staleness is detected from ``(data_ptr, _version)`` of every parameter -- ``load_state_dict``, ``copy_``, plain and
    foreach optimiser steps bump ``_version`` -- and, for networks with a trainable parameter, from a count of ALL optimiser
    steps taken in the process (fused optimisers update parameters without touching the version counters; a spurious refresh
    costs a device-side re-pack, a missed one renders with old weights).  In-place edits made through ``param.data`` are seen
    by neither: call ``invalidate(network_fn)`` (or ``mark_stale``) after such an edit, and after replaying an optimiser step
    from a HIP graph.  Architectures the library has no kernel for
    raise ``Unsupported``; that verdict is cached as well, so a fallback caller does not re-copy the weights to the
    host on every call.  ``flags``: nrnerf_model_desc.flags to OR in (a handle per flag set), e.g. ``_lib.MODEL_FORCE_GENERIC``."""
    precision = _lib.canonical_precision(precision or _DEFAULT_PRECISION)
    _watch_optimizers()
    rb = _bender_of(network_fn)
    dev = torch.device(device if device is not None else next(network_fn.parameters()).device)
    mflags = _lib.model_flags_from_env() | int(flags)
    key = (_wref(network_fine), _wref(rb), precision, str(dev), bool(getattr(network_fn, "approx_nonrigid_viewdirs", True)), mflags)
    fp = _fingerprint([network_fn, network_fine, rb])
    with _cache_lock:
        per = _cache.setdefault(network_fn, {})
        for k in [k for k in per if any(r is not None and r() is None for r in k[:2])]:
            del per[k]                                                         # entries of collected fine nets / benders
        hit = per.get(key)
        if rb is not None:
            # which handle a caller that is handed the bender alone gets (model_of_bender): the last call's -- except that a plain
            # render (flags 0) between a training call and its compute_divergence_loss does not displace the TRAINING handle of the
            # same networks (flags != 0: a generic / training-only handle; resolved with flags 0 such an architecture has no training
            # kernels and the divergence would quietly go to the reference)
            prev = _by_bender.get(rb)
            if int(flags) or prev is None or prev[3] == 0 or prev[0]() is not network_fn:
                _by_bender[rb] = (weakref.ref(network_fn), _wref(network_fine), precision, int(flags))
        if hit is not None and hit[0] == fp:
            if isinstance(hit[1], Exception):
                raise hit[1]
            return hit[1]
        if hit is not None and isinstance(hit[1], Model) and (hit[1].update_from_device(network_fn, network_fine)
                                                               or hit[1].update(network_fn, network_fine)):
            per[key] = (fp, hit[1])                                            # weights changed: refreshed in place
            return hit[1]
        try:
            model = Model(network_fn, network_fine, precision, dev, flags=mflags)
        except Unsupported as e:
            per[key] = (fp, e)
            raise
        except _lib.NrnerfError as e:
            if e.status != _lib.ERR_UNSUPPORTED:
                raise
            err = Unsupported(str(e))
            per[key] = (fp, err)
            raise err from e
        per[key] = (fp, model)
        return model


def model_of_bender(ray_bender, device):
    """The up-to-date packed model (with training kernels) whose ray bender is this module, on ``device`` -- for callers
    that are handed the bender alone (``compute_divergence_loss``, run_nerf_helpers.py:22).  The modules that were
    rendered with this bender last are looked up and their handle refreshed like any ``get_model`` call; None if the
    bender has not been rendered through the HIP path (or its model has no training kernels)."""
    with _cache_lock:
        ent = _by_bender.get(ray_bender)
    if ent is None:
        return None
    nf, nfine, precision = ent[0](), (ent[1]() if ent[1] is not None else None), ent[2]
    if precision == "f16":
        precision = "bf16"            # as training.render_rays_train: an f16 handle has no training kernels
    if nf is None or (ent[1] is not None and nfine is None):
        return None
    try:
        model = get_model(nf, nfine, precision=precision, device=device, flags=ent[3])      # (the handle the last call used: same flags)
    except Unsupported:
        return None
    return model if model.trains_bender else None


def mark_stale(network_fn, replay_stream=None):
    """The packed weights of ``network_fn``'s handles no longer match the parameters although the version counters say
    they do -- an optimiser step replayed from a HIP graph updates the parameters without touching the counters.  The
    next ``get_model`` re-packs them IN PLACE (device-side refresh), unlike ``invalidate``, which drops the handles.
    ``replay_stream``: the stream the graph was just replayed on -- the training kernels inside it read the packed weights there,
    so a refresh issued from another stream has to order itself after it (``Model.update_from_device`` waits on ``_train_stream``;
    the stream the model remembers from the capture is the capture's, long gone)."""
    with _cache_lock:
        per = _cache.get(network_fn)
        if per:
            for k, (fp, m) in list(per.items()):
                if isinstance(m, Model):
                    per[k] = (None, m)
                    if replay_stream is not None:
                        m._train_stream = replay_stream


def note_repacked(network_fn, model):
    """``model`` (a handle of ``network_fn`` from this cache) was just re-packed from the CURRENT parameters by its caller (the fused
    optimiser step): record their fingerprint, so that the next ``get_model`` serves the handle as it is."""
    with _cache_lock:
        per = _cache.get(network_fn)
        if not per:
            return
        for k, (fp, m) in list(per.items()):
            if m is model:
                nfine, rb = (k[0]() if k[0] is not None else None), (k[1]() if k[1] is not None else None)
                per[k] = (_fingerprint([network_fn, nfine, rb]), m)


def forget_streams(network_fn):
    """After a device synchronisation: drop the recorded render events and the remembered training stream of ``network_fn``'s handles
    (all that work is complete) -- GraphedStep calls it right before a capture."""
    with _cache_lock:
        per = _cache.get(network_fn)
        if per:
            for _, m in per.values():
                if isinstance(m, Model):
                    with m._ws_lock:
                        m._render_events = {}
                    m._train_stream = None


def invalidate(network_fn=None):
    """Forget the packed weights of ``network_fn`` (all models when None): the next call re-packs from the modules.
    Needed only after edits the version counters cannot see (``param.data.copy_(...)``, ``param.data *= ...``)."""
    with _cache_lock:
        if network_fn is None:
            _cache.clear()
        else:
            _cache.pop(network_fn, None)


# --------------------------------------------------------------------------------------------
# eligibility + the two drop-in entry points
# --------------------------------------------------------------------------------------------
def _trains(network_fn, network_fine, ray_batch, latents):
    """Is this a call autograd has to see?  (The bender is deliberately not a submodule of the NeRF modules --
    run_nerf_helpers.py:213-215 -- so its parameters are checked explicitly, like the fine network's.)"""
    if not torch.is_grad_enabled():
        return False
    if ray_batch.requires_grad or (latents is not None and latents.requires_grad):
        return True
    for m in (network_fn, network_fine, _bender_of(network_fn)):
        if m is not None and any(p.requires_grad for p in m.parameters()):
            return True
    return False


def _why_unsupported(ray_batch, network_fn, network_fine, N_samples, N_importance, lindisp, perturb, white_bkgd,
                     raw_noise_std, pytest, latents):
    if _trains(network_fn, network_fine if N_importance > 0 else None, ray_batch, latents):
        return "autograd is enabled (training path)"
    if pytest:
        return "pytest flag (numpy-seeded random numbers, train.py:863-867)"
    if ray_batch.device.type != "cuda":
        return "rays are not on a ROCm device"
    if getattr(network_fn, "use_viewdirs", False):
        has_bender = _bender_of(network_fn) is not None
        exact = has_bender and not getattr(network_fn, "approx_nonrigid_viewdirs", True)
        if (not has_bender or exact) and ray_batch.shape[-1] < 11:
            return "use_viewdirs without view directions in the ray batch"
        if network_fine is not None and getattr(network_fine, "approx_nonrigid_viewdirs", True) == exact:
            return "coarse and fine networks disagree about approx_nonrigid_viewdirs"
    if N_samples < 2 or N_samples + N_importance > _lib.MAX_SAMPLES:
        return f"more than {_lib.MAX_SAMPLES} samples per ray"
    a = getattr(network_fn, "test_time_nonrigid_object_removal_threshold", None)
    b = getattr(network_fine, "test_time_nonrigid_object_removal_threshold", a) if network_fine is not None else a
    if a != b:
        return "different removal thresholds on coarse and fine networks"
    return None


def _eligible(ray_batch, latents, network_fn, network_fine=None, N_samples=64, N_importance=0, lindisp=False,
              perturb=0.0, white_bkgd=False, raw_noise_std=0.0, pytest=False, **_):
    """(model, None) when the HIP path takes this call, (None, reason) otherwise.  Decided once per ``batchify_rays``."""
    why = _why_unsupported(ray_batch, network_fn, network_fine, N_samples, N_importance, lindisp, perturb,
                           white_bkgd, raw_noise_std, pytest, latents)
    if why is not None:
        return None, why
    try:
        return get_model(network_fn, network_fine if N_importance > 0 else None, device=ray_batch.device), None
    except Unsupported as e:
        return None, str(e)


def render_rays(ray_batch, network_fn, network_query_fn=None, N_samples=64, retraw=False, lindisp=False, perturb=0.0,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0.0,
                additional_pixel_information=None, detailed_output=False, verbose=False, pytest=False,
                **dummy_kwargs):
    """Signature and semantics of reference ``render_rays`` (train.py:792-809)."""
    latents = None
    if additional_pixel_information is not None:
        latents = additional_pixel_information.get("ray_bending_latents")
    model = dummy_kwargs.pop("_model", None)           # batchify_rays already decided
    why = None
    if model is None and _trains(network_fn, network_fine if N_importance > 0 else None, ray_batch, latents):
        # training call (train.py:152-287): the native autograd path (nonrigid_nerf_amd/training.py) when it has kernels
        from . import training
        why = training.why_not_trainable(network_fn, network_fine, N_samples, N_importance, lindisp, pytest, ray_batch)
        if why is None:
            try:
                return training.render_rays_train(
                    ray_batch, network_fn, N_samples, retraw=retraw, perturb=perturb, N_importance=N_importance,
                    network_fine=network_fine, white_bkgd=white_bkgd, raw_noise_std=raw_noise_std,
                    additional_pixel_information=additional_pixel_information, detailed_output=detailed_output,
                    want_z_vals=bool(dummy_kwargs.get("_want_z_vals", False)), lindisp=bool(lindisp),
                    only_details=dummy_kwargs.get("_only_details"), divergence_share=dummy_kwargs.get("_divergence_share"),
                    randoms=dummy_kwargs.get("_randoms"))
            except (Unsupported, _lib.NrnerfError) as e:
                if isinstance(e, _lib.NrnerfError) and e.status != _lib.ERR_UNSUPPORTED:
                    raise
                why = str(e)
    elif model is None:
        model, why = _eligible(ray_batch, latents, network_fn, network_fine, N_samples, N_importance, lindisp, perturb,
                               white_bkgd, raw_noise_std, pytest)
    if why is None and N_importance == 0 and detailed_output:
        # the reference raises UnboundLocalError here (train.py:900-908 vs 967-970); keep that contract
        raise UnboundLocalError("local variable 'visibility_weights_0' referenced before assignment "
                                "(reference render_rays cannot do detailed_output with N_importance == 0)")
    if why is not None:
        ref = _fallbacks.get("render_rays")
        if ref is None:
            raise Unsupported(f"no HIP kernel for this call ({why}) and no reference function installed to defer to")
        _note_fallback("render_rays", why)
        return ref(ray_batch, network_fn, network_query_fn, N_samples, retraw=retraw, lindisp=lindisp,
                   perturb=perturb, N_importance=N_importance, network_fine=network_fine, white_bkgd=white_bkgd,
                   raw_noise_std=raw_noise_std, additional_pixel_information=additional_pixel_information,
                   detailed_output=detailed_output, verbose=verbose, pytest=pytest, **dummy_kwargs)
    rb = _bender_of(network_fn)
    randoms = _draw_randoms(ray_batch, N_samples, N_importance, perturb, raw_noise_std)
    ret = model.render(
        ray_batch, latents, N_samples, N_importance, retraw=retraw, detailed_output=detailed_output,
        rigidity_cutoff=getattr(rb, "rigidity_test_time_cutoff", None) if rb is not None else None,
        test_time_scaling=getattr(rb, "test_time_scaling", None) if rb is not None else None,
        removal_threshold=getattr(network_fn, "test_time_nonrigid_object_removal_threshold", None),
        want_z_vals=bool(dummy_kwargs.get("_want_z_vals", False)), surface=bool(dummy_kwargs.get("_surface", False)),
        lindisp=lindisp, white_bkgd=white_bkgd, randoms=randoms)
    if _SCAN_OUTPUTS:
        # the reference's per-key scan (train.py:974-978, on whenever its DEBUG global is True): one reduction and one
        # host sync per output key per call, hence opt-in here (NRNERF_SCAN_OUTPUTS=1)
        for k, v in ret.items():
            if torch.isnan(v).any() or torch.isinf(v).any():
                print(f"! [Numerical Error] {k} contains nan or inf.")
    return ret


def _draw_randoms(ray_batch, N_samples, N_importance, perturb, raw_noise_std):
    """The random numbers of render_rays' stochastic branches, drawn with the reference's own calls in the reference's
    order on the rays' device -- ``t_rand = torch.rand(z_vals.shape)`` (train.py:860), coarse ``torch.randn(raw[..., 3].shape)
    * raw_noise_std`` (:753), ``u = torch.rand(list(cdf.shape[:-1]) + [N_importance])`` (run_nerf_helpers.py:665), fine
    noise (:753) -- so a seeded call consumes torch's generator exactly like the reference and renders the same image."""
    stochastic_z = bool(perturb) and perturb > 0.0
    noisy = bool(raw_noise_std) and raw_noise_std > 0.0
    if not (stochastic_z or noisy):
        return None
    n, dev = ray_batch.shape[0], ray_batch.device
    out = {}
    if stochastic_z:
        out["u_coarse"] = torch.rand([n, N_samples], device=dev)
    if noisy:
        # (normal_(0, std) IS randn * std -- ATen's kernel returns rand * std + mean for the same draws -- in one launch instead of two)
        out["noise_coarse"] = torch.empty([n, N_samples], device=dev).normal_(0.0, float(raw_noise_std))
    if N_importance > 0:
        if stochastic_z:
            out["u_fine"] = torch.rand([n, N_importance], device=dev)
        if noisy:
            out["noise_fine"] = torch.empty([n, N_samples + N_importance], device=dev).normal_(0.0, float(raw_noise_std))
    return out


class GraphedRender:
    """``render_rays`` for a FIXED number of rays and fixed keyword arguments, captured once in a HIP graph and replayed.

    A small call -- the 1024-ray batches of a training forward, the ``chunk=128`` probes of
    ``determine_nerf_volume_extent`` (run_nerf_helpers.py:918-1051) -- is four kernels of 10-150 us: the host side of an eager
    call (argument struct, output tensors, four launches) costs as much as the kernels when the caller waits for every
    result, and leaves gaps between them when it does not.  A replay is one launch.

        g = GraphedRender(example_rays, network_fn, N_samples=64, N_importance=128, network_fine=fine, latents=example_codes)
        out = g(rays, codes)          # the reference's output dict; the SAME tensors every call (copy what must survive)

    Weights: the graph reads the packed handle's buffers, which ``get_model`` refreshes in place when a parameter changed
    -- checked on every call (``check_weights=False`` skips that for frozen networks); a handle that had to be REBUILT
    (other architecture) re-captures.  Stochastic calls (``perturb`` / ``raw_noise_std``) draw fresh numbers on every
    replay through torch's graph-safe generator state.  Only calls the HIP path takes can be captured (``Unsupported``
    otherwise); inference only (no autograd: ``training.GraphedStep`` is the training twin)."""

    def __init__(self, example_rays, network_fn, latents=None, warmup=2, check_weights=True, **render_kwargs):
        if "additional_pixel_information" in render_kwargs:
            raise ValueError("pass the latent codes as `latents`")
        if _SCAN_OUTPUTS:
            raise Unsupported("NRNERF_SCAN_OUTPUTS synchronises with the host after every call: nothing to capture")
        self.network_fn, self.kwargs, self.check_weights = network_fn, dict(render_kwargs), bool(check_weights)
        dev = example_rays.device
        self.rays = example_rays.detach().to(torch.float32).clone().contiguous()
        self.latents = latents.detach().to(torch.float32).clone().contiguous() if latents is not None else None
        self._fine = render_kwargs.get("network_fine") if int(render_kwargs.get("N_importance", 0)) > 0 else None
        with torch.no_grad():
            self.model, why = _eligible(self.rays, self.latents, network_fn, **render_kwargs)
            if self.model is None:
                raise Unsupported(f"this call does not run on the HIP path ({why}): nothing to capture")
            self.stream = torch.cuda.Stream(dev)
            self._capture(int(warmup))

    def _api(self):
        return {"ray_bending_latents": self.latents} if self.latents is not None else None

    def _capture(self, warmup):
        cur = torch.cuda.current_stream(self.rays.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for _ in range(max(1, warmup)):             # (also sizes the handle's workspace for this stream)
                render_rays(self.rays, self.network_fn, additional_pixel_information=self._api(), _model=self.model, **self.kwargs)
        cur.wait_stream(self.stream)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.out = render_rays(self.rays, self.network_fn, additional_pixel_information=self._api(), _model=self.model, **self.kwargs)

    @torch.no_grad()
    def __call__(self, ray_batch, latents=None):
        if tuple(ray_batch.shape) != tuple(self.rays.shape):
            raise ValueError(f"captured for rays of shape {tuple(self.rays.shape)}, got {tuple(ray_batch.shape)}")
        if (latents is None) != (self.latents is None) or (latents is not None and tuple(latents.shape) != tuple(self.latents.shape)):
            raise ValueError("latent codes differ in presence / shape from the captured call")
        if self.check_weights:
            model = get_model(self.network_fn, self._fine, device=self.rays.device)      # refreshes changed weights in place
            if model is not self.model:
                self.model = model
                self._capture(1)
        self.rays.copy_(ray_batch, non_blocking=True)
        if latents is not None:
            self.latents.copy_(latents, non_blocking=True)
        self.graph.replay()
        self.model.note_use(self.rays.device)
        return self.out


def batchify_rays(rays_flat, additional_pixel_information, chunk=1024 * 32, detailed_output=False, **kwargs):
    """Signature and semantics of reference ``batchify_rays`` (train.py:108-137).

    ``chunk`` exists in the reference to bound memory and "does not affect final results"
    (train.py:344-345).  The fused kernels need ~5 KB of scratch per ray instead of the reference's
    ~70 KB, so on the HIP path rays are processed in launches of up to 2^20 rays (or ``chunk`` if larger).
    Whether the HIP path takes the call is decided ONCE, here: a call it cannot take goes to the reference's own
    ``batchify_rays`` with the caller's ``chunk`` (its memory bound must hold for the reference's ~70 KB per ray).
    """
    lat = additional_pixel_information["ray_bending_latents"] if additional_pixel_information else None
    model, why = _eligible(rays_flat, lat, **kwargs)
    if model is None and _trains(kwargs.get("network_fn"), kwargs.get("network_fine"), rays_flat, lat):
        from . import training
        if training.why_not_trainable(kwargs.get("network_fn"), kwargs.get("network_fine"), kwargs.get("N_samples", 64),
                                      kwargs.get("N_importance", 0), kwargs.get("lindisp", False), kwargs.get("pytest", False),
                                      rays_flat) is None:
            # native training path: the chunk loop of the reference (train.py:115-137), chunk = the caller's (the random
            # numbers are drawn per chunk; activations are kept for the backward pass, so memory scales with chunk)
            pieces = {}
            for i in range(0, rays_flat.shape[0], int(chunk)):
                api = {"ray_bending_latents": lat[i:i + chunk, :]} if lat is not None else None
                ret = render_rays(rays_flat[i:i + chunk], additional_pixel_information=api, detailed_output=detailed_output, **kwargs)
                for k, v in ret.items():
                    pieces.setdefault(k, []).append(v)
            return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in pieces.items()}
    if model is None:
        ref = _fallbacks.get("batchify_rays")
        if ref is None:
            raise Unsupported(f"no HIP kernel for this call ({why}) and no reference function installed to defer to")
        _note_fallback("batchify_rays", why)
        return ref(rays_flat, additional_pixel_information, chunk=chunk, detailed_output=detailed_output, **kwargs)
    n = rays_flat.shape[0]
    step = max(int(chunk), _MAX_RAYS_PER_LAUNCH)
    if (kwargs.get("perturb") or 0) > 0 or (kwargs.get("raw_noise_std") or 0) > 0:
        step = int(chunk)        # the chunk shapes decide which random numbers each ray gets: keep the reference's
    pieces = {}
    for i in range(0, n, step):
        api = {"ray_bending_latents": lat[i:i + step, :]} if lat is not None else None    # train.py:119-123
        ret = render_rays(rays_flat[i:i + step], additional_pixel_information=api,
                          detailed_output=detailed_output, _model=model, **kwargs)
        for k, v in ret.items():
            pieces.setdefault(k, []).append(v)
    return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in pieces.items()}


def install(train_module, precision: str | None = None):
    """Rebind ``train_module.render_rays`` / ``.batchify_rays`` to the HIP path.

    ``precision``: "f32" (default: exact fp32 MFMA, the parity mode), "bf16" (the benchmarked throughput mode, 65 dB
    against the fp32 render on a fitted model) or "f16".  The originals are kept and used only for calls the library
    has no kernel for.  Returns a callable that undoes the patch.
    """
    _lib.load()      # fail now, loudly, if the library is missing
    # A drop-in must not silently change an fp32 pipeline's arithmetic: without an explicit request (argument, or the
    # NRNERF_PRECISION environment variable) the exact fp32 kernels are selected; the 16-bit modes are one keyword away.
    previous_precision = _DEFAULT_PRECISION          # restored by uninstall(): direct callers (Model, get_model, bench.py) keep theirs
    set_precision(precision if precision is not None else os.environ.get("NRNERF_PRECISION", "f32"))
    orig = (train_module.render_rays, train_module.batchify_rays)
    _fallbacks["render_rays"], _fallbacks["batchify_rays"] = orig
    train_module.render_rays = render_rays
    train_module.batchify_rays = batchify_rays
    # the divergence regulariser of the training iteration (train.py:266 calls the name it star-imported from
    # run_nerf_helpers, i.e. a global of the train module): second order through the ray bender, native as well
    orig_div = getattr(train_module, "compute_divergence_loss", None)
    if orig_div is not None:
        from . import training
        _fallbacks["compute_divergence_loss"] = orig_div
        train_module.compute_divergence_loss = training.compute_divergence_loss

    def uninstall():
        train_module.render_rays, train_module.batchify_rays = orig
        if orig_div is not None:
            train_module.compute_divergence_loss = orig_div
        _fallbacks.clear()
        set_precision(previous_precision)
    return uninstall
