"""Reference checkpoint (``logs/latest.tar``) -> render kwargs for the HIP path, without the reference's model code.

The reference writes its checkpoints with ``torch.save`` of a dict (train.py:1680-1698):
``global_step``, ``network_fn_state_dict``, ``network_fine_state_dict`` (or None), ``ray_bender_state_dict``
(or None), ``optimizer_state_dict``, ``ray_bending_latent_codes`` ``[frames, latent]``, ``intrinsics``,
``scripts_dict``, ``dataset_extras``.  Reading one back normally needs ``create_nerf`` (train.py:556-721: the
``configargparse`` arguments of the experiment, ``.cuda()`` modules, a ``logs/`` directory) followed by
``load_state_dict`` (train.py:666-682, free_viewpoint_rendering.py:42-63).

Every architectural number ``create_nerf`` takes from the arguments is also visible in the tensor shapes, so this
loader infers it from the state dicts, fills the weight holders of ``nonrigid_nerf_amd.modules`` (host memory; the
boundary packs from host arrays) and returns the same ``render_kwargs_test`` dictionary ``create_nerf`` builds
(train.py:698-719).  The only numbers not stored in a checkpoint are the sample counts (``N_samples``,
``N_importance`` live in the experiment's config file): pass them, the defaults are the shipped configs' 64 / 64.
"""
from __future__ import annotations

import dataclasses
import re
from typing import Any

import torch

from .modules import NeRFWeights, RayBenderWeights


def _count(sd, prefix):
    idx = [int(m.group(1)) for k in sd for m in [re.match(re.escape(prefix) + r"\.(\d+)\.weight$", k)] if m]
    if sorted(idx) != list(range(len(idx))):
        raise ValueError(f"state dict has no contiguous '{prefix}.<i>.weight' entries")
    return len(idx)


def _is_posenc_width(c: int) -> bool:
    return c >= 3 and (c - 3) % 6 == 0          # 3 + 3 * 2 * multires  (run_nerf_helpers.py:153-168)


def infer_bender(sd) -> dict:
    """``ray_bending.__init__`` arguments visible in its state dict (run_nerf_helpers.py:388-505)."""
    depth = _count(sd, "network")
    hidden, first_in = sd["network.0.weight"].shape
    if depth < 2 or sd[f"network.{depth - 1}.weight"].shape[0] != 3 or f"network.{depth - 1}.bias" in sd:
        raise ValueError("not a ray_bending offset network (last layer must be hidden -> 3 without bias)")
    rdepth = _count(sd, "rigidity_network")
    return dict(latent_size=int(first_in) - 3, hidden=int(hidden), depth=depth,
                rigidity_hidden=int(sd["rigidity_network.0.weight"].shape[0]), rigidity_depth=rdepth)


def infer_nerf(sd, latent_size: int | None, has_bender: bool, time_conditioned_baseline: bool | None = None) -> dict:
    """``NeRF.__init__`` arguments visible in its state dict (run_nerf_helpers.py:172-238)."""
    D = _count(sd, "pts_linears")
    W, net_in = (int(x) for x in sd["pts_linears.0.weight"].shape)
    skips = [i - 1 for i in range(1, D) if sd[f"pts_linears.{i}.weight"].shape[1] == W + net_in]
    for i in range(1, D):
        if sd[f"pts_linears.{i}.weight"].shape[1] not in (W, W + net_in):
            raise ValueError(f"pts_linears.{i} has an input width that is neither W nor W + input")
    tcb = time_conditioned_baseline
    if tcb is None:
        # the time-conditioned baseline feeds [encoding, latent] to the trunk and has no bender (train.py:571-575)
        tcb = (not has_bender) and (not _is_posenc_width(net_in)) and latent_size is not None \
            and _is_posenc_width(net_in - latent_size)
    input_ch = net_in - (latent_size if tcb else 0)
    if not _is_posenc_width(input_ch):
        raise ValueError(f"trunk input width {net_in} is not a positional encoding of xyz"
                         + (" plus the latent code" if tcb else ""))
    use_viewdirs = "alpha_linear.weight" in sd
    input_ch_views = int(sd["views_linears.0.weight"].shape[1]) - W
    # the view-dependent head has no output_linear; create_nerf's value (5 with a fine network, train.py:593) is kept by the caller
    output_ch = None if use_viewdirs else int(sd["output_linear.weight"].shape[0])
    return dict(D=D, W=W, input_ch=input_ch, input_ch_views=input_ch_views, output_ch=output_ch, skips=tuple(skips),
                use_viewdirs=use_viewdirs, time_conditioned_baseline=bool(tcb))


@dataclasses.dataclass
class Checkpoint:
    global_step: int
    ray_bender: RayBenderWeights | None
    network_fn: NeRFWeights
    network_fine: NeRFWeights | None
    latents: torch.Tensor | None              # [frames, latent]  (train.py:1693)
    intrinsics: Any
    arch: dict
    render_kwargs_test: dict                  # what create_nerf returns as render_kwargs_test (train.py:698-719)
    raw: dict                                 # the remaining checkpoint entries, untouched


def load_checkpoint(path_or_dict, N_samples: int = 64, N_importance: int | None = None, device=None,
                    approx_nonrigid_viewdirs: bool = True, time_conditioned_baseline: bool | None = None) -> Checkpoint:
    """Read a reference checkpoint and return weight holders + ``render_kwargs_test`` for ``render.batchify_rays`` /
    ``driver.render_path``.  ``N_importance=None``: 64 when the checkpoint has a fine network, else 0.

    ``device``: where the latent codes go (the weight holders stay on the host: ``render.get_model`` packs them for
    the device of the rays it is called with)."""
    ck = path_or_dict if isinstance(path_or_dict, dict) else \
        torch.load(path_or_dict, map_location="cpu", weights_only=False)     # holds numpy / python objects
    sd_c, sd_f, sd_b = ck["network_fn_state_dict"], ck.get("network_fine_state_dict"), ck.get("ray_bender_state_dict")
    latents = ck.get("ray_bending_latent_codes")
    if latents is not None:
        latents = torch.as_tensor(latents).detach().to(torch.float32)
        if latents.numel() == 0:
            latents = None
    if N_importance is None:
        N_importance = 64 if sd_f is not None else 0
    if (N_importance > 0) != (sd_f is not None):
        raise ValueError("N_importance > 0 needs a fine network in the checkpoint (and vice versa)")

    rb, barch = None, None
    if sd_b is not None:
        barch = infer_bender(sd_b)
        rb = RayBenderWeights(**barch)
        rb.load_state_dict({k: torch.as_tensor(v).detach().float() for k, v in sd_b.items()}, strict=True)
    latent_size = barch["latent_size"] if barch else (int(latents.shape[1]) if latents is not None else None)

    def make(sd, num_ray_samples):
        arch = infer_nerf(sd, latent_size, rb is not None, time_conditioned_baseline)
        if arch["output_ch"] is None:
            arch["output_ch"] = 5 if N_importance > 0 else 4
        net = NeRFWeights(ray_bender=None, ray_bending_latent_size=latent_size if latent_size is not None else 32,
                          num_ray_samples=num_ray_samples, approx_nonrigid_viewdirs=approx_nonrigid_viewdirs, **arch)
        net.load_state_dict({k: torch.as_tensor(v).detach().float() for k, v in sd.items()}, strict=True)
        net.ray_bender = (rb,)                   # 1-tuple, like the reference (run_nerf_helpers.py:213-215)
        return net, arch

    coarse, arch = make(sd_c, N_samples)
    fine = None
    if sd_f is not None:
        fine, arch_f = make(sd_f, N_samples + N_importance)
        if (arch_f["use_viewdirs"], arch_f["time_conditioned_baseline"]) != (arch["use_viewdirs"], arch["time_conditioned_baseline"]):
            raise ValueError("coarse and fine networks disagree about the head / conditioning")
    for m in (rb, coarse, fine):
        if m is not None:
            m.requires_grad_(False)
    if latents is not None and device is not None:
        latents = latents.to(device)

    kwargs = {"network_query_fn": None, "perturb": False, "N_importance": N_importance, "network_fine": fine,
              "N_samples": N_samples, "network_fn": coarse, "ray_bender": rb, "use_viewdirs": arch["use_viewdirs"],
              "white_bkgd": False, "raw_noise_std": 0.0, "ndc": False, "lindisp": False}
    rest = {k: v for k, v in ck.items() if k not in ("network_fn_state_dict", "network_fine_state_dict",
                                                     "ray_bender_state_dict", "ray_bending_latent_codes")}
    return Checkpoint(global_step=int(ck.get("global_step", 0)), ray_bender=rb, network_fn=coarse, network_fine=fine,
                      latents=latents, intrinsics=ck.get("intrinsics"), arch=dict(arch, bender=barch),
                      render_kwargs_test=kwargs, raw=rest)
