"""Seeded synthetic weights / rays / latents.

No trained checkpoint ships with the reference and random-init networks give
sigma <= 0 almost everywhere, which would make parity vacuous (SURVEY.md
section 7, "Hard parts").  The generators here make the scene non-degenerate:
the density row is biased and scaled, and the (zero-initialised in the
reference) last layers of the bender and the rigidity network are made
non-zero so the deformation actually moves points.

Everything is a pure function of an integer seed and runs on CPU, so the
golden-vector script (which loads these arrays into the *reference* modules),
the oracle, the GPU tests and ``bench.py`` all see identical numbers.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch

from .modules import NeRFWeights, RayBenderWeights, load_named_arrays

NEAR, FAR = 0.0022, 1.0024  # example_sequence bounds, SURVEY.md section 8c


@dataclass
class SceneConfig:
    """The hot-path-relevant flags of train.py:983-1219, with reference defaults."""
    N_samples: int = 64
    N_importance: int = 128
    netdepth: int = 8
    netwidth: int = 256
    netdepth_fine: int | None = None          # --netdepth_fine / --netwidth_fine (train.py:1004-1010); None: as the coarse network
    netwidth_fine: int | None = None
    multires: int = 10
    multires_views: int = 4
    use_viewdirs: bool = False
    latent_size: int = 32
    bend_hidden: int = 64
    bend_depth: int = 5
    rigidity_hidden: int = 32
    rigidity_depth: int = 3
    skips: tuple = (4,)
    ray_bending: bool = True
    time_conditioned_baseline: bool = False
    approx_nonrigid_viewdirs: bool = True     # False: directions = normalised J(bent wrt xyz) . d   (rnh:358-385)
    near: float = NEAR
    far: float = FAR

    def for_fine(self) -> "SceneConfig":
        """The same settings with the fine network's depth / width in ``netdepth`` / ``netwidth`` (create_nerf, train.py:612-630)."""
        import dataclasses
        return dataclasses.replace(self, netdepth=self.netdepth_fine or self.netdepth, netwidth=self.netwidth_fine or self.netwidth,
                                   netdepth_fine=None, netwidth_fine=None)

    @property
    def input_ch(self) -> int:
        return 3 + 6 * self.multires

    @property
    def input_ch_views(self) -> int:
        return (3 + 6 * self.multires_views) if self.use_viewdirs else 0

    @property
    def output_ch(self) -> int:
        return 5 if self.N_importance > 0 else 4  # train.py:593


def _linear(gen, out_f, in_f, bias=True, gain=1.0):
    """U(-b, b) weights with b = gain / sqrt(in_f); gain sqrt(6) is He-uniform.

    torch's default Linear init (gain 1) makes an 8-layer ReLU trunk nearly
    constant over space (measured raw std 0.008 around a 0.05 offset), which
    would make compositing parity vacuous; trained NeRF weights are in the
    variance-preserving regime, so the trunk uses He-uniform.
    """
    bound = gain / math.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound
    b = (torch.rand(out_f, generator=gen) * 2 - 1) * (1.0 / math.sqrt(in_f)) if bias else None
    return w, b


def _probe_head_stats(arrays: dict, cfg: "SceneConfig", head: str, seed: int):
    """Mean/std of one linear head's pre-activation over random points of the volume.

    Generator utility only (used to centre the synthetic density / colour rows so
    opacity spans (0,1)); it is not part of any render path.
    """
    gen = torch.Generator().manual_seed(4242 + seed)
    p = torch.cat([torch.randn(4096, 2, generator=gen) * 0.35, -torch.rand(4096, 1, generator=gen)], -1)
    cols = [p]
    for k in range(cfg.multires):
        cols += [torch.sin(p * 2.0 ** k), torch.cos(p * 2.0 ** k)]
    x = torch.cat(cols, -1)
    if cfg.time_conditioned_baseline:
        x = torch.cat([x, torch.randn(4096, cfg.latent_size, generator=gen) * 0.1], -1)
    h = x
    for i in range(cfg.netdepth):
        h = torch.relu(h @ arrays[f"pts_linears.{i}.weight"].T + arrays[f"pts_linears.{i}.bias"])
        if i in cfg.skips:
            h = torch.cat([x, h], -1)
    y = h @ arrays[head + ".weight"].T + arrays[head + ".bias"]
    return y.mean(0), y.std(0)


def bender_arrays(cfg: SceneConfig, seed: int) -> dict:
    gen = torch.Generator().manual_seed(seed)
    out = {}
    dims = [3 + cfg.latent_size] + [cfg.bend_hidden] * (cfg.bend_depth - 1) + [3]
    for i in range(cfg.bend_depth):
        last = i == cfg.bend_depth - 1
        w, b = _linear(gen, dims[i + 1], dims[i], bias=not last)
        if last:  # reference zero-initialises this layer; make it bend for real
            w = torch.randn(dims[i + 1], dims[i], generator=gen) * 0.15
        out[f"network.{i}.weight"] = w
        if b is not None:
            out[f"network.{i}.bias"] = b
    rdims = [3] + [cfg.rigidity_hidden] * (cfg.rigidity_depth - 1) + [1]
    for i in range(cfg.rigidity_depth):
        w, b = _linear(gen, rdims[i + 1], rdims[i])
        if i == cfg.rigidity_depth - 1:
            w = torch.randn(rdims[i + 1], rdims[i], generator=gen) * 2.5
        out[f"rigidity_network.{i}.weight"] = w
        out[f"rigidity_network.{i}.bias"] = b
    return out


def nerf_arrays(cfg: SceneConfig, seed: int) -> dict:
    gen = torch.Generator().manual_seed(seed)
    W, D = cfg.netwidth, cfg.netdepth
    net_in = cfg.input_ch + (cfg.latent_size if cfg.time_conditioned_baseline else 0)
    out = {}
    for i in range(D):
        in_f = net_in if i == 0 else (W + net_in if (i - 1) in cfg.skips else W)
        w, b = _linear(gen, W, in_f, gain=math.sqrt(6.0))
        out[f"pts_linears.{i}.weight"], out[f"pts_linears.{i}.bias"] = w, b
    w, b = _linear(gen, W // 2, cfg.input_ch_views + W)
    out["views_linears.0.weight"], out["views_linears.0.bias"] = w, b
    if cfg.use_viewdirs:
        w, b = _linear(gen, W, W)
        out["feature_linear.weight"], out["feature_linear.bias"] = w, b
        w, b = _linear(gen, 1, W)
        out["alpha_linear.weight"], out["alpha_linear.bias"] = w, b
        _calibrate(out, cfg, "alpha_linear", seed, rows=[0], std=SIGMA_STD, shift=SIGMA_SHIFT)
        w, b = _linear(gen, 3, W // 2, gain=math.sqrt(6.0))
        out["rgb_linear.weight"], out["rgb_linear.bias"] = w, b
    else:
        w, b = _linear(gen, cfg.output_ch, W)
        out["output_linear.weight"], out["output_linear.bias"] = w, b
        _calibrate(out, cfg, "output_linear", seed, rows=[0, 1, 2], std=1.5, shift=0.0)
        _calibrate(out, cfg, "output_linear", seed, rows=[3], std=SIGMA_STD, shift=SIGMA_SHIFT)
    return out


SIGMA_STD, SIGMA_SHIFT = 6.0, -2.0   # ~37 % of samples have sigma > 0, mean relu(sigma) ~ 1.4


def _calibrate(arrays, cfg, head, seed, rows, std, shift):
    """Rescale/centre ``rows`` of a head so its pre-activation is ~N(shift, std^2) over the volume."""
    mean, sd = _probe_head_stats(arrays, cfg, head, seed)
    for r in rows:
        k = std / float(sd[r])
        arrays[head + ".weight"][r] *= k
        arrays[head + ".bias"][r] = (arrays[head + ".bias"][r] - mean[r]) * k + shift


@dataclass
class Scene:
    cfg: SceneConfig
    bender: dict
    coarse: dict
    fine: dict | None
    modules: dict = field(default_factory=dict)


def make_scene(cfg: SceneConfig | None = None, seed: int = 0) -> Scene:
    cfg = cfg or SceneConfig()
    bender = bender_arrays(cfg, seed * 7919 + 1) if cfg.ray_bending else None
    coarse = nerf_arrays(cfg, seed * 7919 + 2)
    fine = nerf_arrays(cfg.for_fine(), seed * 7919 + 3) if cfg.N_importance > 0 else None
    return Scene(cfg, bender, coarse, fine)


def build_modules(scene: Scene, device="cpu", dtype=torch.float32):
    """Instantiate the parameter holders for ``scene`` (ray_bender, coarse, fine)."""
    cfg = scene.cfg
    rb = None
    if scene.bender is not None:
        rb = RayBenderWeights(cfg.latent_size, cfg.bend_hidden, cfg.bend_depth,
                              cfg.rigidity_hidden, cfg.rigidity_depth)
        load_named_arrays(rb, scene.bender)
        rb = rb.to(device=device, dtype=dtype)

    def mk(arrays, ns, cfg):
        m = NeRFWeights(D=cfg.netdepth, W=cfg.netwidth, input_ch=cfg.input_ch,
                        input_ch_views=cfg.input_ch_views, output_ch=cfg.output_ch,
                        skips=cfg.skips, use_viewdirs=cfg.use_viewdirs, ray_bender=None,
                        ray_bending_latent_size=cfg.latent_size, num_ray_samples=ns,
                        approx_nonrigid_viewdirs=cfg.approx_nonrigid_viewdirs,
                        time_conditioned_baseline=cfg.time_conditioned_baseline)
        load_named_arrays(m, arrays)
        m = m.to(device=device, dtype=dtype)
        m.ray_bender = (rb,)
        return m

    coarse = mk(scene.coarse, cfg.N_samples, cfg)
    fine = mk(scene.fine, cfg.N_samples + cfg.N_importance, cfg.for_fine()) if scene.fine is not None else None
    return rb, coarse, fine


def make_rays(n: int, seed: int = 0, cfg: SceneConfig | None = None, with_viewdirs=None):
    """Synthetic ray batch in the layout render() builds (train.py:389-399).

    Returns ``rays [n, 8 | 11]`` = ``[o3, d3, near, far (, unit viewdir3)]`` and
    ``latents [n, L]``.
    """
    cfg = cfg or SceneConfig()
    gen = torch.Generator().manual_seed(1000 + seed)
    o = torch.randn(n, 3, generator=gen) * 0.1
    d = torch.cat([torch.randn(n, 2, generator=gen) * 0.35, -torch.ones(n, 1)], -1)
    near = torch.full((n, 1), cfg.near)
    far = torch.full((n, 1), cfg.far)
    rays = torch.cat([o, d, near, far], -1)
    if cfg.use_viewdirs if with_viewdirs is None else with_viewdirs:
        rays = torch.cat([rays, d / torch.norm(d, dim=-1, keepdim=True)], -1)
    latents = torch.randn(n, cfg.latent_size, generator=gen) * 0.1
    return rays.float().contiguous(), latents.float().contiguous()
