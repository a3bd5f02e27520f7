"""Generate tests/golden/*.npz by running the UNMODIFIED reference on CPU.

Run in the build container only (needs /root/reference; the GPU box does not
have it):   python oracle/make_golden.py

Recipe: pre-seed ``sys.modules`` with empty ``imageio``
/ ``configargparse`` (train.py:11, 985 import them, the hot path never touches
them), make ``Tensor.get_device`` return the device object (the reference is
CUDA-only as written: get_device() == -1 on CPU breaks every ``device=``
argument), build ``ray_bending`` / ``NeRF`` with the arguments of
train.py:564-630 (``create_nerf`` itself calls .cuda() unconditionally), load
the seeded synthetic weights of ``nonrigid_nerf_amd.synthetic`` into them, and
call the reference ``train.render`` -> ``batchify_rays`` -> ``render_rays``.

Each fixture stores the seeds/config needed to regenerate the inputs and the
reference outputs (fp32).  Weights are NOT stored (4 MB per scene): they are a
pure function of the seed on torch's CPU generator.
"""
from __future__ import annotations

import os
import sys
import types
from math import gcd

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = os.environ.get("NRNERF_REFERENCE", "/root/reference")

from nonrigid_nerf_amd.synthetic import SceneConfig, make_rays, make_scene  # noqa: E402


def import_reference():
    sys.path.insert(0, REF)
    for m in ("imageio", "configargparse"):
        sys.modules.setdefault(m, types.ModuleType(m))
    torch.Tensor.get_device = lambda self: self.device
    import run_nerf_helpers as H
    import train as T
    T.DEBUG = False
    T.device = torch.device("cpu")
    return H, T


def reference_kwargs(H, T, scene):
    cfg = scene.cfg
    S, I = cfg.N_samples, cfg.N_importance
    embed_fn, input_ch = H.get_embedder(cfg.multires, 0)
    rb = None
    if scene.bender is not None:
        rb = H.ray_bending(input_ch, cfg.latent_size, "simple_neural", embed_fn)
        if cfg.bend_depth != rb.network_depth:
            # the reference hard-codes depth 5 (run_nerf_helpers.py:406-407); its forward() only loops over
            # self.network, so a deeper offset MLP is the same module with a longer ModuleList (BASELINE config 4)
            hd = rb.hidden_dimensions
            rb.network_depth = cfg.bend_depth
            rb.network = torch.nn.ModuleList(
                [torch.nn.Linear(3 + cfg.latent_size, hd)] + [torch.nn.Linear(hd, hd) for _ in range(cfg.bend_depth - 2)]
                + [torch.nn.Linear(hd, 3, bias=False)])
        rb.load_state_dict({k: v.clone() for k, v in scene.bender.items()}, strict=True)
    embeddirs_fn, input_ch_views = (None, 0)
    netchunk = 1024 * 64
    if cfg.use_viewdirs:
        embeddirs_fn, input_ch_views = H.get_embedder(cfg.multires_views, 0)
        lcm = S * (S + I) // gcd(S, S + I)
        netchunk = (netchunk // lcm) * lcm                      # train.py:584-592

    def mk(arrays, ns, cfg):
        m = H.NeRF(D=cfg.netdepth, W=cfg.netwidth, input_ch=input_ch, output_ch=cfg.output_ch,
                   skips=list(cfg.skips), input_ch_views=input_ch_views, use_viewdirs=cfg.use_viewdirs,
                   ray_bender=rb, ray_bending_latent_size=cfg.latent_size, embeddirs_fn=embeddirs_fn,
                   num_ray_samples=ns, approx_nonrigid_viewdirs=cfg.approx_nonrigid_viewdirs,
                   time_conditioned_baseline=cfg.time_conditioned_baseline)
        m.load_state_dict({k: v.clone() for k, v in arrays.items()}, strict=True)
        return m

    coarse = mk(scene.coarse, S, cfg)
    fine = mk(scene.fine, S + I, cfg.for_fine()) if scene.fine is not None else None      # create_nerf, train.py:612-630

    def network_query_fn(inputs, viewdirs, api, network_fn, detailed_output=False):   # train.py:633-649
        return T.run_network(inputs, viewdirs, api, network_fn, embed_fn=embed_fn,
                             embeddirs_fn=embeddirs_fn, netchunk=netchunk,
                             detailed_output=detailed_output)

    kw = dict(network_query_fn=network_query_fn, perturb=0.0, N_importance=I, network_fine=fine,
              N_samples=S, network_fn=coarse, ray_bender=rb, use_viewdirs=cfg.use_viewdirs,
              white_bkgd=False, raw_noise_std=0.0, ndc=False, lindisp=False,
              near=cfg.near, far=cfg.far)                        # train.py:698-719
    return kw, rb, coarse, fine


CASES = {
    # name: (SceneConfig kwargs, n_rays, chunk, detailed_output, retraw, knobs)
    "coarse_only_1k":   (dict(N_importance=0), 1024, 32768, False, False, {}),
    "headline_64_128":  (dict(), 192, 32768, False, True, {}),
    "detailed_64_128":  (dict(), 24, 32768, True, True, {}),
    "ragged_chunks":    (dict(N_importance=64), 37, 16, False, False, {}),
    "knobs_64_64":      (dict(N_importance=64), 24, 32768, True, False,
                         dict(rigidity_test_time_cutoff=0.45, test_time_scaling=0.5, removal_threshold=0.6)),
    "viewdirs_64_64":   (dict(N_importance=64, use_viewdirs=True), 48, 32768, False, True, {}),
    "no_bender_64_64":  (dict(N_importance=64, ray_bending=False), 48, 32768, False, False, {}),
    "time_conditioned_64_64": (dict(N_importance=64, ray_bending=False, time_conditioned_baseline=True), 48, 32768, False, True, {}),
    "config4_deep_bender_viewdirs": (dict(N_importance=64, use_viewdirs=True, bend_depth=7), 40, 32768, True, True, {}),
    # the two render_rays flags create_nerf always passes as False (train.py:707,715); knobs starting with "render_" are
    # keyword overrides of the reference render() call, not module attributes
    "lindisp_white_bkgd_64_64": (dict(N_importance=64), 48, 32768, False, True, dict(render_lindisp=True, render_white_bkgd=True)),
    # the stochastic branches, seeded: torch.manual_seed(render_seed) right before the reference's render() call;
    # chunk 16 of 40 rays, so the random draws interleave across chunks exactly as in batchify_rays
    "exact_viewdirs_64_64": (dict(N_importance=64, use_viewdirs=True, approx_nonrigid_viewdirs=False), 40, 32768, False, True, {}),
    "exact_viewdirs_knobs": (dict(N_importance=64, use_viewdirs=True, approx_nonrigid_viewdirs=False), 24, 32768, True, False,
                             dict(rigidity_test_time_cutoff=0.45, test_time_scaling=0.5)),
    "config4_exact_viewdirs": (dict(N_importance=64, use_viewdirs=True, bend_depth=7, approx_nonrigid_viewdirs=False), 32, 32768, False, True, {}),
    # --netwidth 128 --netwidth_fine 128 (train.py:1004-1010): compiled architecture 5
    "narrow_128_64_64": (dict(N_importance=64, netwidth=128), 48, 32768, True, True, {}),
    "narrow_128_no_bender": (dict(N_importance=64, netwidth=128, ray_bending=False), 37, 16, False, True, {}),
    "stochastic_64_64": (dict(N_importance=64), 40, 16, False, True, dict(render_perturb=1.0, render_raw_noise_std=0.7, render_seed=1234)),
    # architectures outside the compiled set (the run-time-parameterised kernel, csrc/nrnerf_generic.h): --netdepth 6 --netwidth 192
    # --netdepth_fine 10 --netwidth_fine 320 --multires 8 --ray_bending_latent_size 16 (train.py:1004-1010, 1060, 1133-1139), with
    # detailed outputs and the editing knobs; and the view-dependent head at --netwidth 96 / 160, --multires 6, --multires_views 2
    "generic_192_320_detailed": (dict(N_importance=64, netdepth=6, netwidth=192, netdepth_fine=10, netwidth_fine=320, multires=8, latent_size=16),
                                 40, 32768, True, True, dict(rigidity_test_time_cutoff=0.45, test_time_scaling=0.5, removal_threshold=0.6)),
    "generic_viewdirs_96_160": (dict(N_importance=48, N_samples=40, netdepth=7, netwidth=96, netwidth_fine=160, multires=6, multires_views=2,
                                     use_viewdirs=True), 37, 16, False, True, {}),
    "generic_shallow_no_bender": (dict(N_importance=64, netdepth=4, netwidth=64, ray_bending=False), 48, 32768, False, True, {}),
    # exact Jacobian view directions (approx_nonrigid_viewdirs=False, rnh:358-385) on a NON-COMPILED trunk (--netwidth 192 --netwidth_fine 160,
    # --netdepth 6), the reference's own bender: the library computes J d with the bender's divergence kernel (round 6)
    "generic_exact_viewdirs_192": (dict(N_importance=64, netdepth=6, netwidth=192, netwidth_fine=160, skips=(2,), use_viewdirs=True,
                                        approx_nonrigid_viewdirs=False), 37, 16, False, True, dict(rigidity_test_time_cutoff=0.3, test_time_scaling=0.8)),
    "generic_time_conditioned_448": (dict(N_importance=32, netdepth=8, netwidth=448, multires=12, latent_size=24, ray_bending=False,
                                          time_conditioned_baseline=True, use_viewdirs=True, multires_views=6), 24, 32768, False, True, {}),
}


def run_case(H, T, name, seed=0):
    cfg_kw, n, chunk, detailed, retraw, knobs = CASES[name]
    cfg = SceneConfig(**cfg_kw)
    scene = make_scene(cfg, seed)
    rays, latents = make_rays(n, seed, cfg)
    kw, rb, coarse, fine = reference_kwargs(H, T, scene)
    kw.update({k[len("render_"):]: v for k, v in knobs.items() if k.startswith("render_") and k != "render_seed"})
    if "render_seed" in knobs:
        torch.manual_seed(knobs["render_seed"])
    if rb is not None:
        rb.rigidity_test_time_cutoff = knobs.get("rigidity_test_time_cutoff")
        rb.test_time_scaling = knobs.get("test_time_scaling")
    for m in (coarse, fine):
        if m is not None:
            m.test_time_nonrigid_object_removal_threshold = knobs.get("removal_threshold")
    rays_d = rays[:, 3:6]
    with torch.no_grad():
        rgb, disp, acc, extras = T.render(rays[:, 0:3], rays_d, chunk=chunk,
                                          additional_pixel_information={"ray_bending_latents": latents},
                                          detailed_output=detailed, retraw=retraw, **kw)
    out = {"rgb_map": rgb, "disp_map": disp, "acc_map": acc, **extras}
    arrays = {"out__" + k: v.detach().cpu().numpy().astype(np.float32) for k, v in out.items()}
    meta = dict(seed=seed, n_rays=n, chunk=chunk, detailed=int(detailed), retraw=int(retraw))
    arrays["meta_json"] = np.frombuffer(
        __import__("json").dumps(dict(cfg=cfg_kw, knobs=knobs, **meta)).encode(), dtype=np.uint8)
    # inputs are stored too so a generator drift is detected rather than silently re-based
    arrays["in__rays"] = rays.numpy()
    arrays["in__latents"] = latents.numpy()
    return arrays, out


def synthetic_camera(k, H=24, W=32):
    """A small, deterministic camera: pose k of a little orbit, intrinsics scaled from example_sequence (hwf 384,512,256.6)."""
    import math
    a = 0.35 * k - 0.4
    R = torch.tensor([[math.cos(a), 0.0, math.sin(a)], [0.05 * k, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
    R = torch.linalg.qr(R)[0]
    t = torch.tensor([[0.1 * math.sin(a)], [0.02 * k], [0.15 - 0.03 * k]])
    c2w = torch.cat([R, t], 1).float()
    intrin = dict(height=H, width=W, focal_x=256.6 * W / 512, focal_y=256.6 * H / 384 * 1.01,
                  center_x=W / 2 - 0.3, center_y=H / 2 + 0.2, ray_bending_latent_size=32)
    return c2w, intrin


def run_raygen(H_ref):
    """Reference get_rays (run_nerf_helpers.py:588-605) on three synthetic cameras."""
    arrays = {}
    for k in range(3):
        c2w, intrin = synthetic_camera(k)
        ro, rd = H_ref.get_rays(c2w, intrin)
        arrays[f"out__rays_o_{k}"] = ro.numpy().astype(np.float32)
        arrays[f"out__rays_d_{k}"] = rd.numpy().astype(np.float32)
        arrays[f"in__c2w_{k}"] = c2w.numpy()
    return arrays


CHECKPOINT_CONFIGS = {           # = tests/test_checkpoint.py::CONFIGS
    "default":          dict(),
    "coarse_only":      dict(N_importance=0),
    "viewdirs":         dict(N_importance=64, use_viewdirs=True),
    "no_bender":        dict(N_importance=64, ray_bending=False),
    "time_conditioned": dict(N_importance=64, ray_bending=False, time_conditioned_baseline=True),
    "deep_bender":      dict(N_importance=64, use_viewdirs=True, bend_depth=7),
}


def run_checkpoint_layout(H, T):
    """Keys and shapes of what train.py:1680-1698 stores, read off the reference's own modules' state_dict()."""
    layout = {}
    for name, cfg_kw in CHECKPOINT_CONFIGS.items():
        scene = make_scene(SceneConfig(**cfg_kw), 0)
        kw, rb, coarse, fine = reference_kwargs(H, T, scene)
        shapes = lambda m: None if m is None else {k: list(v.shape) for k, v in m.state_dict().items()}
        layout[name] = {"network_fn": shapes(coarse), "network_fine": shapes(fine), "ray_bender": shapes(rb)}
    return layout


GRAD_PARAMS = {"bender": ["network.4.weight", "network.0.bias", "rigidity_network.2.weight"],
               "coarse": ["pts_linears.0.bias", "output_linear.weight"],
               "fine": ["pts_linears.7.bias", "output_linear.weight"]}


GRAD_PARAMS_VIEWS = {"bender": ["network.4.weight", "network.0.bias", "rigidity_network.2.weight"],
                     "coarse": ["pts_linears.0.bias", "alpha_linear.weight", "feature_linear.bias", "views_linears.0.weight", "rgb_linear.weight"],
                     "fine": ["pts_linears.7.bias", "alpha_linear.bias", "feature_linear.weight", "views_linears.0.bias", "rgb_linear.bias"]}


# a NON-COMPILED architecture (--netdepth 6 --netwidth 192 --netwidth_fine 320, 8 frequencies; the run-time-parameterised training kernels):
# pts_linears.3 is the skip layer of a 6-layer trunk with skips = [2] (its weight has the [encoding | activation] columns)
GRAD_CFG_GENERIC = dict(netdepth=6, netwidth=192, netwidth_fine=320, multires=8, skips=(2,))
GRAD_PARAMS_GENERIC = {"bender": ["network.4.weight", "network.0.bias", "rigidity_network.2.weight"],
                       "coarse": ["pts_linears.0.weight", "pts_linears.3.weight", "pts_linears.5.bias", "output_linear.weight", "output_linear.bias"],
                       "fine": ["pts_linears.0.bias", "pts_linears.3.weight", "pts_linears.5.weight", "output_linear.weight"]}


GRAD_PARAMS_TCB = {"coarse": ["pts_linears.0.weight", "pts_linears.5.bias", "output_linear.weight"],
                   "fine": ["pts_linears.5.weight", "pts_linears.0.bias", "output_linear.bias"]}


def run_gradients(H, T, seed=0, cfg_kw=None, grad_params=None):
    """Reference autograd through render() (the training data term, train.py:1560-1580 restricted to rgb): gradients of
    sum(rgb_map) + sum(rgb0) wrt a few small parameters and the latent codes -- the yardstick for a future backward pass.
    ``cfg_kw``: scene settings beyond 64 + 64 samples (e.g. use_viewdirs=True: the view-dependent head with
    finite-difference directions, whose gradient also reaches the bent points of neighbouring samples)."""
    cfg = SceneConfig(N_importance=64, **(cfg_kw or {}))
    grad_params = grad_params or GRAD_PARAMS
    scene = make_scene(cfg, seed)
    rays, latents = make_rays(16, seed, cfg)
    kw, rb, coarse, fine = reference_kwargs(H, T, scene)
    latents = latents.clone().requires_grad_(True)
    rgb, disp, acc, extras = T.render(rays[:, 0:3], rays[:, 3:6], chunk=32768,
                                      additional_pixel_information={"ray_bending_latents": latents}, **kw)     # (viewdirs=None: render() derives them, train.py:377-381)
    loss = rgb.sum() + extras["rgb0"].sum()
    loss.backward()
    out = {"loss": loss.detach().numpy().astype(np.float64), "grad__latents": latents.grad.numpy()}
    for part, mod in (("bender", rb), ("coarse", coarse), ("fine", fine)):
        if mod is None:
            continue
        params = dict(mod.named_parameters())
        for name in grad_params[part]:
            out[f"grad__{part}__{name}"] = params[name].grad.numpy()
    return out


TRAIN_STEP = dict(n_rays=16, n_frames=4, seed=0, render_seed=4321, global_step=100000,
                  # configs/example_sequence.txt:14-16, 22, 26-28, 35
                  offsets_loss_weight=60.0, divergence_loss_weight=3.0, rigidity_loss_weight=0.0005, N_iters=200000,
                  N_samples=64, N_importance=64, chunk=32768, raw_noise_std=1.0, perturb=1.0)
# every parameter's gradient NORM is stored; these tensors (all of the bender, the codes, a few of each trunk) in full
TRAIN_STEP_FULL = {"coarse": ["pts_linears.0.weight", "pts_linears.5.bias", "output_linear.weight"],
                   "fine": ["pts_linears.7.weight", "pts_linears.0.bias", "output_linear.bias"]}


def run_train_step(H, T):
    """The REFERENCE's training iteration on CPU, unmodified: ``training_wrapper_class.forward`` (train.py:152-287) with
    the shipped regulariser weights (configs/example_sequence.txt) -- render with detailed outputs, data term on both
    images, offsets + rigidity regulariser, divergence regulariser (compute_divergence_loss, double backward through the
    ray bender) -- then ``loss.mean().backward()`` as train.py:1594-1597 does.  Seeded right before the call."""
    import argparse
    ts = TRAIN_STEP
    cfg = SceneConfig(N_importance=ts["N_importance"])
    scene = make_scene(cfg, ts["seed"])
    rays, _ = make_rays(ts["n_rays"], ts["seed"], cfg)
    kw, rb, coarse, fine = reference_kwargs(H, T, scene)
    kw.update(perturb=ts["perturb"], raw_noise_std=ts["raw_noise_std"])                      # train.py:698-719 (training kwargs)
    g = torch.Generator().manual_seed(11)
    codes = [(torch.randn(cfg.latent_size, generator=g) * 0.1).requires_grad_(True) for _ in range(ts["n_frames"])]
    image_ids = torch.randint(0, ts["n_frames"], (ts["n_rays"],), generator=g)
    target = torch.rand(ts["n_rays"], 3, generator=g)
    args = argparse.Namespace(offsets_loss_weight=ts["offsets_loss_weight"], divergence_loss_weight=ts["divergence_loss_weight"],
                              rigidity_loss_weight=ts["rigidity_loss_weight"], chunk=ts["chunk"], N_iters=ts["N_iters"],
                              N_samples=ts["N_samples"], ray_bending_latent_size=cfg.latent_size)
    wrapper = T.training_wrapper_class(coarse, codes, fine_model=fine, ray_bender=rb)
    batch_pixel_indices = torch.stack([image_ids, torch.zeros_like(image_ids), torch.zeros_like(image_ids)], 1)
    torch.manual_seed(ts["render_seed"])
    loss = wrapper(args, rays[:, 0:3], rays[:, 3:6], 100, dict(kw), target, ts["global_step"], 0,
                   {"imageid_to_timestepid": list(range(ts["n_frames"]))}, batch_pixel_indices)
    assert tuple(loss.shape) == (ts["n_rays"],)
    loss.mean().backward()                                                                    # train.py:1594-1597
    # the same seeded call without the divergence term (its probes are the last random numbers drawn, so everything else
    # sees the same draws): tells a reader how much of the loss the second-order term is
    args0 = argparse.Namespace(**{**vars(args), "divergence_loss_weight": 0.0})
    torch.manual_seed(ts["render_seed"])
    with torch.no_grad():
        loss0 = wrapper(args0, rays[:, 0:3], rays[:, 3:6], 100, dict(kw), target, ts["global_step"], 0,
                        {"imageid_to_timestepid": list(range(ts["n_frames"]))}, batch_pixel_indices)
    out = {"out__loss_per_ray": loss.detach().numpy().astype(np.float64), "out__loss_per_ray_without_divergence": loss0.numpy().astype(np.float64),
           "in__rays": rays.numpy(),
           "in__codes": torch.stack([c.detach() for c in codes]).numpy(), "in__image_ids": image_ids.numpy(), "in__target": target.numpy(),
           "grad__codes": torch.stack([c.grad if c.grad is not None else torch.zeros_like(c) for c in codes]).numpy()}
    for part, mod in (("bender", rb), ("coarse", coarse), ("fine", fine)):
        for name, prm in mod.named_parameters():
            if prm.grad is None:
                continue
            out[f"gradnorm__{part}__{name}"] = np.float64(prm.grad.double().norm())
            if part == "bender" or name in TRAIN_STEP_FULL[part]:
                out[f"grad__{part}__{name}"] = prm.grad.numpy()
    out["meta_json"] = np.frombuffer(__import__("json").dumps(ts).encode(), dtype=np.uint8)
    return out


def run_render_path(H, T, seed=0):
    """Reference ``train.render_path`` (train.py:419-553) on two tiny frames with detailed outputs, and the per-pixel
    surface reduction free_viewpoint_rendering.py:621-658 performs on those outputs (the same torch ops, lifted: the
    script itself needs a checkpoint directory and a GPU).  Pins ``O.render_path`` and ``O.surface_from_details``."""
    cfg = SceneConfig(N_importance=64)
    scene = make_scene(cfg, seed)
    kw, rb, coarse, fine = reference_kwargs(H, T, scene)
    cams = [synthetic_camera(k, H=8, W=12) for k in range(2)]
    poses, intrins = [c for c, _ in cams], [i for _, i in cams]
    codes = torch.randn(2, cfg.latent_size, generator=torch.Generator().manual_seed(3)) * 0.1
    with torch.no_grad():
        rgbs, disps, details = T.render_path(poses, intrins, 32768, kw, codes, detailed_output=True)
    out = {"out__rgbs": rgbs.astype(np.float32), "out__disps": disps.astype(np.float32), "in__codes": codes.numpy(),
           "in__poses": np.stack([p.numpy() for p in poses], 0)}
    for i, image_details in enumerate(details):
        accumulated_visibility = torch.cumsum(torch.Tensor(image_details["fine_visibility_weights"]), dim=-1)      # fvr:623-625
        median_indices = torch.min(torch.abs(accumulated_visibility - 0.5), dim=-1)[1]                               # fvr:626-628
        height, width = median_indices.shape
        surface_pixels = image_details["fine_input_pts"].reshape(height * width, -1, 3)[
            np.arange(height * width), median_indices.cpu().reshape(-1), :].reshape(height, width, 3)               # fvr:631-637
        rigidity = image_details["fine_rigidity_mask"].reshape(height * width, -1)[
            np.arange(height * width), median_indices.cpu().reshape(-1)].reshape(height, width)                     # fvr:648-654
        out[f"out__median_indices_{i}"] = median_indices.numpy().astype(np.int32)
        out[f"out__surface_pixels_{i}"] = surface_pixels.astype(np.float32)
        out[f"out__rigidity_{i}"] = rigidity.astype(np.float32)
        if i == 0:       # the reference's own detail tensors of one frame, so the reduction can be checked on identical inputs
            for k in ("fine_visibility_weights", "fine_input_pts", "fine_rigidity_mask"):
                out["out__" + k + "_0"] = image_details[k].astype(np.float32)
    return out


def main():
    H, T = import_reference()
    os.makedirs(os.path.join(REPO, "tests", "golden"), exist_ok=True)
    with open(os.path.join(REPO, "tests", "golden", "checkpoint_layout.json"), "w") as f:
        __import__("json").dump(run_checkpoint_layout(H, T), f, indent=0, sort_keys=True)
    if "--only-layout" in sys.argv:
        return
    if "--only-render-path" in sys.argv or not [a for a in sys.argv if a.startswith("--case=")]:
        np.savez_compressed(os.path.join(REPO, "tests", "golden", "render_path_2frames.npz"), **run_render_path(H, T))
        if "--only-render-path" in sys.argv:
            return
    if "--only-grads" in sys.argv or not [a for a in sys.argv if a.startswith("--case=")]:
        np.savez_compressed(os.path.join(REPO, "tests", "golden", "gradients_64_64.npz"), **run_gradients(H, T))
        np.savez_compressed(os.path.join(REPO, "tests", "golden", "gradients_viewdirs_64_64.npz"),
                            **run_gradients(H, T, cfg_kw=dict(use_viewdirs=True), grad_params=GRAD_PARAMS_VIEWS))
        np.savez_compressed(os.path.join(REPO, "tests", "golden", "gradients_exact_viewdirs_64_64.npz"),
                            **run_gradients(H, T, cfg_kw=dict(use_viewdirs=True, approx_nonrigid_viewdirs=False), grad_params=GRAD_PARAMS_VIEWS))
        np.savez_compressed(os.path.join(REPO, "tests", "golden", "gradients_time_conditioned_64_64.npz"),
                            **run_gradients(H, T, cfg_kw=dict(ray_bending=False, time_conditioned_baseline=True), grad_params=GRAD_PARAMS_TCB))
        np.savez_compressed(os.path.join(REPO, "tests", "golden", "gradients_generic_192_320_64_64.npz"),
                            **run_gradients(H, T, cfg_kw=GRAD_CFG_GENERIC, grad_params=GRAD_PARAMS_GENERIC))
        if "--only-grads" in sys.argv:
            return
    if "--only-train-step" in sys.argv or not [a for a in sys.argv if a.startswith("--case=")]:
        np.savez_compressed(os.path.join(REPO, "tests", "golden", "train_step_64_64.npz"), **run_train_step(H, T))
        if "--only-train-step" in sys.argv:
            return
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--case=")]
    if not only:
        np.savez_compressed(os.path.join(REPO, "tests", "golden", "raygen.npz"), **run_raygen(H))
    for name in (only or CASES):
        arrays, out = run_case(H, T, name)
        path = os.path.join(REPO, "tests", "golden", name + ".npz")
        np.savez_compressed(path, **arrays)
        acc = out["acc_map"]
        print(f"{name:18s} keys={len(out):2d} acc[min/mean/max]={acc.min():.3f}/{acc.mean():.3f}/{acc.max():.3f} "
              f"nan_disp={int(torch.isnan(out['disp_map']).sum())} -> {os.path.getsize(path)/1024:.0f} KiB")


if __name__ == "__main__":
    main()
