"""CPU oracle for the NR-NeRF per-ray hot path.  TEST INFRASTRUCTURE ONLY.

This file is a checker, not a product path: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  ``nonrigid_nerf_amd`` never imports anything under ``oracle/``;
the product fails loudly if the HIP library is missing.

What it is: a functional restatement, in plain PyTorch CPU ops, of the
algorithm of ``render_rays`` (reference train.py:792-980) and everything it
calls.  The reference's arithmetic *is* PyTorch eager ops (third-party
``pytorch``, pinned =1.6.0 in environment.yml:8, torch 2.10 here), so the
restatement uses the same primitives (linear, relu, sin/cos, cumprod, cumsum,
searchsorted, sort) and is dtype-generic: fp32 gives the reference's own
numbers, fp64 gives a higher-precision yardstick.

Parity pinning: the reference ships no golden vectors or tests for this path
(SURVEY.md section 4).  The oracle is therefore pinned against outputs of the
reference itself, generated in the build container by ``oracle/make_golden.py``
(which imports /root/reference unmodified) and committed under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks the oracle against
them bit-for-bit-tolerance on every run.

Inputs are plain tensors and ``{state_dict key: tensor}`` dicts (see
``nonrigid_nerf_amd.synthetic``), not modules.  The code is device-agnostic: two GPU tests also run it on the ROCm
device (``scene_on``) -- as the eager-PyTorch context number and to give the stochastic branches the device's own
random stream -- still only as the checker.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class Knobs:
    """Test-time editing knobs (free_viewpoint_rendering.py:264-283)."""
    rigidity_test_time_cutoff: float | None = None     # run_nerf_helpers.py:563-564
    test_time_scaling: float | None = None             # run_nerf_helpers.py:568-569
    removal_threshold: float | None = None             # run_nerf_helpers.py:308-311


def posenc(x: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """[x, sin(2^k x), cos(2^k x)]_k  -- Embedder.embed, run_nerf_helpers.py:120-150.

    Frequencies are ``2 ** linspace(0, L-1, L)`` = exact powers of two
    (run_nerf_helpers.py:137, 157-164); layout: identity, then per frequency
    sin(xyz) followed by cos(xyz).
    """
    cols = [x]
    for k in range(n_freqs):
        xs = x * float(2 ** k)
        cols.append(torch.sin(xs))
        cols.append(torch.cos(xs))
    return torch.cat(cols, -1)


def _lin(arrs, name, x, dtype):
    w = arrs[name + ".weight"].to(device=x.device, dtype=dtype)
    b = arrs.get(name + ".bias")
    return F.linear(x, w, None if b is None else b.to(device=x.device, dtype=dtype))


def bend_points(pts, latents, bender, knobs: Knobs | None = None):
    """ray_bending.forward, run_nerf_helpers.py:507-584.

    pts [M,3], latents [M,L] -> bent points [M,3] plus the three detail
    tensors the reference exposes (unmasked_offsets, rigidity_mask,
    masked_offsets).
    """
    dt = pts.dtype
    knobs = knobs or Knobs()
    n_off = 1 + max(int(k.split(".")[1]) for k in bender if k.startswith("network."))
    h = torch.cat([pts, latents.to(dt)], -1)                       # :525
    for i in range(n_off):
        h = _lin(bender, f"network.{i}", h, dt)                    # :527
        if i != n_off - 1:
            h = F.relu(h)                                          # :533-536
    unmasked = h
    n_rig = 1 + max(int(k.split(".")[1]) for k in bender if k.startswith("rigidity_network."))
    h = pts                                                        # :546
    for i in range(n_rig):
        h = _lin(bender, f"rigidity_network.{i}", h, dt)
        if i != n_rig - 1:
            h = F.relu(h)
    mask = (torch.tanh(h) + 1) / 2                                 # :559-561
    if knobs.rigidity_test_time_cutoff is not None:
        mask = torch.where(mask <= knobs.rigidity_test_time_cutoff, torch.zeros_like(mask), mask)  # :563-564
    masked = mask * unmasked                                       # :567
    if knobs.test_time_scaling is not None:
        masked = masked * knobs.test_time_scaling                  # :568-569
    return pts + masked, dict(unmasked_offsets=unmasked, rigidity_mask=mask, masked_offsets=masked)


def finite_difference_dirs(bent, samples_per_ray: int):
    """NeRF.viewdirs_via_finite_differences (backward), run_nerf_helpers.py:316-356."""
    p = bent.view(-1, samples_per_ray, 3)
    diff = p[:, 1:] - p[:, :-1]
    diff = diff / (torch.norm(diff, dim=-1, keepdim=True) + 0.000001)
    return torch.cat([diff[:, :1], diff], 1).reshape(-1, 3)


def exact_dirs(flat, lat, bender, knobs, unbent_dirs):
    """NeRF.exact_nonrigid_viewdirs, run_nerf_helpers.py:358-385: the bent point's Jacobian wrt the straight point,
    applied to the ray's unit direction.  Inference (nothing requires a gradient): only J . d is needed, so one forward-mode
    product replaces the reference's three reverse passes (_get_minibatch_jacobian, :81-104); same value up to fp32 rounding.
    Under autograd (training): the Jacobian row by row with create_graph=True exactly as :81-104 -- the reference's loss
    differentiates THROUGH the Jacobian (second order in the bender's parameters and the latent codes)."""
    trains = torch.is_grad_enabled() and (lat.requires_grad or any(v.requires_grad for v in bender.values()))
    if not trains:
        with torch.enable_grad():
            _, jd = torch.autograd.functional.jvp(lambda x: bend_points(x, lat, bender, knobs)[0], (flat,), (unbent_dirs,))
        jd = jd.detach()
        return jd / torch.norm(jd, dim=-1, keepdim=True) + 0.000001        # :374-378 (eps lands outside the division)
    x = flat.detach().requires_grad_(True)                                 # (the reference's points are a leaf here too: train.py:871-873)
    y = bend_points(x, lat, bender, knobs)[0]
    rows = []
    for j in range(3):                                                     # :93-103
        rows.append(torch.autograd.grad(y[:, j], x, torch.ones_like(y[:, j]), retain_graph=True, create_graph=True)[0].unsqueeze(1))
    jac = torch.cat(rows, 1)                                               # [M, 3 (outputs), 3 (inputs)]
    jd = torch.matmul(jac, unbent_dirs.reshape(-1, 3, 1)).view(-1, 3)      # :367-368
    return jd / torch.norm(jd, dim=-1, keepdim=True) + 0.000001            # :371-376


def canonical_mlp(enc, net, cfg, enc_dirs=None, latents=None):
    """NeRF.forward after bending, run_nerf_helpers.py:272-306."""
    dt = enc.dtype
    D = sum(1 for k in net if k.startswith("pts_linears.") and k.endswith(".weight"))   # (the fine network may be deeper / wider: train.py:1004-1010)
    x_in = enc if not cfg.time_conditioned_baseline else torch.cat([enc, latents.to(dt)], -1)  # :273-274
    h = x_in
    for i in range(D):
        h = F.relu(_lin(net, f"pts_linears.{i}", h, dt))           # :276-277
        if i in cfg.skips:
            h = torch.cat([x_in, h], -1)                           # :278-282
    if cfg.use_viewdirs:
        alpha = _lin(net, "alpha_linear", h, dt)                   # :285
        feat = _lin(net, "feature_linear", h, dt)                  # :286
        h = F.relu(_lin(net, "views_linears.0", torch.cat([feat, enc_dirs], -1), dt))  # :296-301
        return torch.cat([_lin(net, "rgb_linear", h, dt), alpha], -1)   # :303-304
    return _lin(net, "output_linear", h, dt)                       # :306


def query_network(pts, viewdirs, latents, net, bender, cfg, knobs=None, detailed=False):
    """run_network + NeRF.forward (train.py:57-105, run_nerf_helpers.py:240-314).

    pts [N,S,3]; viewdirs [N,3] or None; latents [N,L].  Returns raw [N,S,C]
    and (if ``detailed``) the per-sample detail dict, reshaped [N,S,-1].
    """
    N, S, _ = pts.shape
    dt = pts.dtype
    flat = pts.reshape(-1, 3)
    lat = latents[:, None, :].expand(N, S, latents.shape[-1]).reshape(N * S, -1)   # train.py:79-87
    details = {}
    if detailed:
        details["initial_input_pts"] = flat.clone()                # rnh:250-252
    if bender is not None:
        bent, bd = bend_points(flat, lat, bender, knobs)           # rnh:268
        if detailed:
            details.update(bd)
    else:
        bent = flat
    if detailed:
        details["input_pts"] = bent.clone()                        # rnh:270
    enc = posenc(bent, cfg.multires)                               # rnh:582-584
    enc_dirs = None
    if cfg.use_viewdirs:
        if bender is not None and not getattr(cfg, "approx_nonrigid_viewdirs", True):
            d = viewdirs[:, None, :].expand(N, S, 3).reshape(-1, 3).to(dt)
            dirs = exact_dirs(flat, lat, bender, knobs, d)         # rnh:291-294
        elif bender is not None:
            dirs = finite_difference_dirs(bent, S)                 # rnh:288-290
        else:
            dirs = viewdirs[:, None, :].expand(N, S, 3).reshape(-1, 3).to(dt)   # train.py:73-76
        enc_dirs = posenc(dirs, cfg.multires_views)
    raw = canonical_mlp(enc, net, cfg, enc_dirs, lat)
    if detailed and knobs is not None and knobs.removal_threshold is not None and bender is not None:
        kill = details["rigidity_mask"].flatten() >= knobs.removal_threshold       # rnh:308-311
        raw = raw.clone()
        raw[kill, 3] *= 0.0
    raw = raw.reshape(N, S, -1)
    if detailed:
        details = {k: v.reshape(N, S, -1) for k, v in details.items()}             # train.py:95-98
        return raw, details
    return raw


def composite(raw, z_vals, rays_d, white_bkgd=False, raw_noise_std=0.0):
    """raw2outputs, train.py:724-789."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]                                     # :743
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)          # :744-746
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)                       # :748
    rgb = torch.sigmoid(raw[..., :3])                                              # :750
    noise = 0.0
    if raw_noise_std > 0.0:
        noise = torch.randn(raw[..., 3].shape, device=raw.device).to(raw.dtype) * raw_noise_std   # :753
    alpha = 1.0 - torch.exp(-F.relu(raw[..., 3] + noise) * dists)                  # :740-741, 761
    trans = torch.cumprod(
        torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]   # :763-775
    weights = alpha * trans
    rgb_map = torch.sum(weights[..., None] * rgb, -2)                              # :776
    depth_map = torch.sum(weights * z_vals, -1)                                    # :778
    acc_map = torch.sum(weights, -1)                                               # :779
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)   # :781-784
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - acc_map[..., None])                             # :786-787
    return rgb_map, disp_map, acc_map, alpha, weights, depth_map


def sample_pdf_det(bins, weights, n_samples: int, det: bool = True):
    """sample_pdf, run_nerf_helpers.py:651-698 (det=True at test time: perturb == 0)."""
    weights = weights + 1e-5                                                       # :654
    pdf = weights / torch.sum(weights, -1, keepdim=True)                           # :655
    cdf = torch.cumsum(pdf, -1)                                                    # :656
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)                     # :657-659
    if det:
        u = torch.linspace(0.0, 1.0, steps=n_samples, dtype=torch.float32).to(cdf)   # :663
        u = u.expand(list(cdf.shape[:-1]) + [n_samples]).contiguous()              # :664, 680
    else:
        u = torch.rand(list(cdf.shape[:-1]) + [n_samples], device=cdf.device).to(cdf.dtype)   # :665
    inds = torch.searchsorted(cdf, u, right=False)                                 # :681
    below = torch.clamp(inds - 1, min=0)                                           # :683
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)                               # :684
    cdf_lo, cdf_hi = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)    # :690
    bin_lo, bin_hi = torch.gather(bins, -1, below), torch.gather(bins, -1, above)  # :691
    denom = cdf_hi - cdf_lo                                                        # :693
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)               # :694
    t = (u - cdf_lo) / denom                                                       # :695
    return bin_lo + t * (bin_hi - bin_lo)                                          # :696


def render_rays(ray_batch, latents, scene, *, retraw=False, detailed_output=False,
                knobs: Knobs | None = None, dtype=torch.float32, lindisp=False, white_bkgd=False,
                perturb=0.0, raw_noise_std=0.0, z_fine_override=None):
    """render_rays (train.py:792-980); the stochastic branches draw from torch's generator in the reference's order.

    ``scene`` is a ``nonrigid_nerf_amd.synthetic.Scene`` (or anything with
    ``cfg``, ``bender``, ``coarse``, ``fine``).  Output dict: same keys/shapes
    as the reference.
    """
    cfg = scene.cfg
    S, I = cfg.N_samples, cfg.N_importance
    if I == 0 and detailed_output:
        # train.py:900-908 vs 967-970: the reference itself raises here.
        raise UnboundLocalError("reference render_rays cannot do detailed_output with N_importance == 0")
    rb = ray_batch.to(dtype)
    rays_o, rays_d = rb[:, 0:3], rb[:, 3:6]                                        # :842
    viewdirs = rb[:, -3:] if rb.shape[-1] > 8 else None                            # :843
    near, far = rb[:, 6:7], rb[:, 7:8]                                             # :844-845
    t_vals = torch.linspace(0.0, 1.0, steps=S, dtype=torch.float32).to(device=rb.device, dtype=dtype)      # :847
    if not lindisp:
        z_vals = near * (1.0 - t_vals) + far * t_vals                              # :849
    else:
        z_vals = 1.0 / (1.0 / near * (1.0 - t_vals) + 1.0 / far * t_vals)          # :851
    z_vals = z_vals.expand(rb.shape[0], S)                                         # :853
    if perturb > 0.0:
        mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])                          # :857
        upper = torch.cat([mids, z_vals[..., -1:]], -1)                            # :858
        lower = torch.cat([z_vals[..., :1], mids], -1)                             # :859
        t_rand = torch.rand(z_vals.shape, device=z_vals.device).to(dtype)          # :860
        z_vals = lower + (upper - lower) * t_rand                                  # :868
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[:, :, None]             # :871-873
    lat = latents.to(dtype)
    out = query_network(pts, viewdirs, lat, scene.coarse, scene.bender, cfg, knobs, detailed_output)
    raw, details = out if detailed_output else (out, None)
    rgb_map, disp_map, acc_map, alpha, weights, _ = composite(raw, z_vals, rays_d, white_bkgd, raw_noise_std)  # :898
    ret = {}
    if I > 0:
        rgb0, disp0, acc0, alpha0, weights0 = rgb_map, disp_map, acc_map, alpha, weights   # :902-908
        z_mid = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])                         # :910
        z_samples = sample_pdf_det(z_mid, weights[..., 1:-1], I, det=(perturb == 0.0)).detach()   # :911-918 (no gradient through the sample positions)
        z_vals, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)             # :920
        if z_fine_override is not None:
            # test hook (not in the reference): evaluate the fine pass at given merged depths, so that a checked
            # implementation whose sample_pdf branch (rnh:694) fell the other way on a few rays can be compared tightly
            z_vals = z_fine_override.to(z_vals)
        pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[:, :, None]         # :921-923
        net = scene.fine if scene.fine is not None else scene.coarse               # :925
        out = query_network(pts, viewdirs, lat, net, scene.bender, cfg, knobs, detailed_output)
        raw, fine_details = out if detailed_output else (out, None)
        rgb_map, disp_map, acc_map, alpha, weights, _ = composite(raw, z_vals, rays_d, white_bkgd, raw_noise_std)   # :943-950
    ret.update(rgb_map=rgb_map, disp_map=disp_map, acc_map=acc_map)                # :952
    if retraw:
        ret["raw"] = raw                                                           # :953-954
    if I > 0:
        ret.update(rgb0=rgb0, disp0=disp0, acc0=acc0,
                   z_std=torch.std(z_samples, dim=-1, unbiased=False))             # :955-959
        if detailed_output:
            ret["fine_visibility_weights"] = weights                               # :962
            ret["fine_opacity_alpha"] = alpha                                      # :964
            for k, v in fine_details.items():
                ret["fine_" + k] = v                                               # :965-966
    if detailed_output:
        ret["visibility_weights"] = weights0                                       # :969
        ret["opacity_alpha"] = alpha0                                              # :970
        ret.update(details)                                                        # :971-972
    # internal extra (not a reference key): the merged sample depths, handy for stage-wise parity
    ret["_z_vals"] = z_vals
    return ret


def compute_divergence_loss(input_points, point_latents, bender, exact, chunk, n_rays, weights=None,
                            backprop_into_weights=True, knobs: Knobs | None = None):
    """compute_divergence_loss + divergence_approx / divergence_exact (run_nerf_helpers.py:22-116), through autograd with
    create_graph=True exactly as the reference: per chunk of points, offsets = masked offsets of the bender evaluated on a
    leaf copy of the points; approx: e = randn_like(offsets), e . (J^T e); exact: trace of the Jacobian built row by row."""
    input_points = input_points.detach().requires_grad_(True)                      # :39
    parts = []
    for i in range(0, input_points.shape[0], chunk):                               # :52-59
        sub = input_points[i:i + chunk, :]
        _, details = bend_points(sub, point_latents[i:i + chunk, :], bender, knobs)    # :42 (special_loss_return)
        offsets = details["masked_offsets"]                                        # :43-47
        if exact:
            rows = []
            for j in range(offsets.shape[1]):                                      # _get_minibatch_jacobian, :81-104
                rows.append(torch.autograd.grad(offsets[:, j], sub, torch.ones_like(offsets[:, j]), retain_graph=True,
                                                create_graph=True)[0].view(sub.shape[0], -1).unsqueeze(1))
            jac = torch.cat(rows, 1)
            parts.append(torch.sum(jac.view(jac.shape[0], -1)[:, ::(jac.shape[1] + 1)], 1))    # :72-77
        else:
            e = torch.randn_like(offsets)                                          # :106
            e_dydx = torch.autograd.grad(offsets, sub, e, create_graph=True)[0]    # :107-109
            parts.append((e_dydx * e).view(offsets.shape[0], -1).sum(dim=1))       # :110-112
    divergence_loss = torch.cat(parts, 0)
    divergence_loss = torch.abs(divergence_loss) ** 2                              # :61-62
    if weights is not None:
        if not backprop_into_weights:
            weights = weights.detach()                                             # :65-66
        divergence_loss = weights * divergence_loss                                # :67
    return torch.mean(divergence_loss.view(n_rays, -1), dim=-1)                    # :69


def training_loss(ray_batch, latents, scene, target_s, *, offsets_loss_weight=0.0, divergence_loss_weight=0.0,
                  rigidity_loss_weight=0.0, global_step=0, n_iters=200000, chunk=1024 * 32, perturb=1.0, raw_noise_std=1.0,
                  z_fine_override=None):
    """training_wrapper_class.forward (train.py:152-287) for one ray batch: render with retraw (and detailed outputs when a
    regulariser is on), data term on the fine and the coarse image, offsets + rigidity regulariser, divergence regulariser,
    both with the reference's increasing schedule.  Returns the per-ray loss [N_rays] (the caller takes the mean,
    train.py:1594) and the render outputs.  Random numbers are drawn in the reference's order: render's, then the probes."""
    n_rays = ray_batch.shape[0]
    img2mse = lambda x, y: torch.mean((x - y) ** 2, dim=-1)                        # rnh: img2mse(x, y, N_rays) -> [N_rays]
    detailed = offsets_loss_weight > 0.0 or divergence_loss_weight > 0.0           # :193-196
    out = render_rays(ray_batch, latents, scene, retraw=True, detailed_output=detailed, perturb=perturb,
                      raw_noise_std=raw_noise_std, z_fine_override=z_fine_override)
    loss = img2mse(out["rgb_map"], target_s)                                       # :210-212
    if "rgb0" in out:
        loss = loss + img2mse(out["rgb0"], target_s)                               # :215-218
    schedule = (1.0 / 100.0) ** (1 - (global_step / n_iters))                      # :240, 285
    if scene.bender is not None and offsets_loss_weight > 0.0:                     # :221-242
        weights = out["visibility_weights"].detach().reshape(-1)
        offsets_loss = torch.mean((weights * torch.pow(torch.norm(out["unmasked_offsets"].reshape(-1, 3), dim=-1),
                                                       2.0 - out["rigidity_mask"].reshape(-1))).view(n_rays, -1), dim=-1)
        offsets_loss = offsets_loss + rigidity_loss_weight * torch.mean((weights * out["rigidity_mask"].reshape(-1)).view(n_rays, -1), dim=-1)
        loss = loss + offsets_loss_weight * schedule * offsets_loss
    if scene.bender is not None and divergence_loss_weight > 0.0:                  # :245-287
        pts = out["initial_input_pts"].reshape(-1, 3)
        weights = 1.0 - torch.exp(-F.relu(out["opacity_alpha"].reshape(-1)))       # :264 (sic: of the alphas)
        n_samples = out["initial_input_pts"].shape[1]
        lat = latents.view(n_rays, 1, -1).expand(n_rays, n_samples, latents.shape[-1]).reshape(-1, latents.shape[-1])   # :256-262
        div = compute_divergence_loss(pts, lat, scene.bender, False, chunk, n_rays, weights=weights, backprop_into_weights=False)
        loss = loss + divergence_loss_weight * schedule * div
    return loss, out


def scene_on(scene, device):
    """A shallow copy of ``scene`` with the weight arrays on ``device`` (the oracle is device-agnostic torch code: on a
    ROCm device it is "the reference's eager ops on the GPU", a second baseline; never the product path)."""
    import copy
    out = copy.copy(scene)
    for name in ("bender", "coarse", "fine"):
        d = getattr(scene, name)
        setattr(out, name, None if d is None else {k: v.to(device) for k, v in d.items()})
    return out


def batchify_rays(rays_flat, latents, scene, chunk=1024 * 32, **kw):
    """batchify_rays, train.py:108-137: chunk loop + per-key concatenation."""
    pieces = {}
    for i in range(0, rays_flat.shape[0], chunk):
        r = render_rays(rays_flat[i:i + chunk], latents[i:i + chunk], scene, **kw)
        for k, v in r.items():
            pieces.setdefault(k, []).append(v)
    return {k: torch.cat(v, 0) for k, v in pieces.items()}


# ------------------------------------------------------------------------------------------------
# frame driver: ray generation + per-frame render loop
# ------------------------------------------------------------------------------------------------
def get_rays(c2w, intrin, dtype=torch.float32):
    """get_rays, run_nerf_helpers.py:588-605.  c2w [3,4]; intrin dict with height, width, focal_x/y, center_x/y.
    Returns rays_o, rays_d of shape [H, W, 3]."""
    H, W = intrin["height"], intrin["width"]
    c2w = torch.as_tensor(c2w, dtype=dtype)
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W, dtype=dtype), torch.linspace(0, H - 1, H, dtype=dtype),
                          indexing="ij")                                                        # :592
    i, j = i.t(), j.t()                                                                          # :593-594
    dirs = torch.stack([(i - intrin["center_x"]) / intrin["focal_x"],
                        -(j - intrin["center_y"]) / intrin["focal_y"], -torch.ones_like(i)], -1)  # :599
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)                                     # :602
    rays_o = c2w[:3, -1].expand(rays_d.shape)                                                    # :604
    return rays_o, rays_d


def pack_rays(rays_o, rays_d, near, far, use_viewdirs):
    """The `rays` tensor render() builds for batchify_rays, train.py:380-399."""
    rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    cols = [rays_o, rays_d, near * torch.ones_like(rays_d[:, :1]), far * torch.ones_like(rays_d[:, :1])]
    if use_viewdirs:
        cols.append(rays_d / torch.norm(rays_d, dim=-1, keepdim=True))                          # :380
    return torch.cat(cols, -1)


def render_path(render_poses, intrinsics, scene, ray_bending_latents, chunk=1024 * 32, detailed_output=False, **kw):
    """render_path, train.py:419-553 (no image writing): per frame get_rays, one latent code expanded to every
    pixel (:464-466), render, reshape to [H, W, ...].  Returns rgbs [F,H,W,3], disps [F,H,W] as float32 tensors, plus
    -- with ``detailed_output`` -- the list of per-frame dicts of the remaining keys reshaped [H, W, ...] (:487-497)."""
    cfg = scene.cfg
    rgbs, disps, details = [], [], []
    for c2w, intrin, code in zip(render_poses, intrinsics, ray_bending_latents):
        ro, rd = get_rays(torch.as_tensor(c2w)[:3, :4], intrin)
        H, W = ro.shape[:2]
        rays = pack_rays(ro, rd, cfg.near, cfg.far, cfg.use_viewdirs)
        lat = torch.as_tensor(code).reshape(1, -1).expand(H * W, -1)
        out = batchify_rays(rays, lat, scene, chunk=chunk, detailed_output=detailed_output, **kw)
        rgbs.append(out["rgb_map"].reshape(H, W, 3))
        disps.append(out["disp_map"].reshape(H, W))
        if detailed_output:
            details.append({k: v.reshape((H, W) + tuple(v.shape[1:])) for k, v in out.items()
                            if k not in ("rgb_map", "disp_map", "acc_map") and not k.startswith("_")})
    if detailed_output:
        return torch.stack(rgbs, 0), torch.stack(disps, 0), details
    return torch.stack(rgbs, 0), torch.stack(disps, 0)


def surface_from_details(visibility_weights, input_pts, rigidity_mask=None):
    """The per-pixel reduction free_viewpoint_rendering.py:621-648 performs on the detailed outputs: the sample whose
    accumulated visibility is closest to 0.5, the bent point there and the rigidity there."""
    acc = torch.cumsum(visibility_weights, dim=-1)                                  # fvr:623-625
    idx = torch.min(torch.abs(acc - 0.5), dim=-1)[1]                                # fvr:626-628
    n = visibility_weights.shape[0]
    pts = input_pts[torch.arange(n), idx, :]                                        # fvr:631-637
    rig = rigidity_mask.reshape(n, -1)[torch.arange(n), idx] if rigidity_mask is not None else None   # fvr:648-654
    return idx, pts, rig
