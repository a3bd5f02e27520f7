"""Host-side pieces of the training path (nonrigid_nerf_amd/training.py) that need no GPU: eligibility, the row-block
arithmetic of the torch-op bender's batched layers, and the autograd-side bender against the oracle's."""
import os

import pytest
import torch

from nonrigid_nerf_amd import training as T
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_rays, make_scene
from oracle import nrnerf_oracle as O


def test_row_blocks_divide_evenly_and_stay_large():
    for m in (1, 100, 4095, 8191, 8192, 8384, 65536, 196608, 131 * 85, 1024 * 192, 7 * 4096 + 13):
        b = T._chunks(m)
        assert b >= 1 and m % b == 0
        assert b == 1 or m // b >= 4096


@pytest.mark.parametrize("batched", [False, True])
def test_autograd_bender_matches_the_oracle(batched):
    """training.bend (F.linear / batched GEMMs over row blocks on the modules' parameters) against the oracle's
    bend_points (reference ray_bending.forward, run_nerf_helpers.py:507-577): values, detail tensors, knobs, and the
    gradients wrt points, latents and every parameter -- including a second derivative (the divergence regulariser
    differentiates through the bender twice, run_nerf_helpers.py:22-116)."""
    cfg = SceneConfig()
    scene = make_scene(cfg, 0)
    rb, _, _ = build_modules(scene)
    rb.requires_grad_(True)
    rb.rigidity_test_time_cutoff, rb.test_time_scaling = 0.3, 0.7
    g = torch.Generator().manual_seed(1)
    n = 8192 if batched else 257
    pts = (torch.randn(n, 3, generator=g) * 0.3).requires_grad_(True)
    lat = (torch.randn(n, 32, generator=g) * 0.1).requires_grad_(True)
    old = T.BATCHED_BENDER
    T.BATCHED_BENDER = batched
    try:
        bent, d = T.bend(rb, pts, lat)
    finally:
        T.BATCHED_BENDER = old
    arrays = {k: v.detach() for k, v in rb.state_dict().items()}
    ref, dref = O.bend_points(pts.detach(), lat.detach(), arrays, O.Knobs(rigidity_test_time_cutoff=0.3, test_time_scaling=0.7))
    assert torch.allclose(bent, ref, atol=2e-6) and all(torch.allclose(d[k], dref[k], atol=2e-6) for k in dref)
    # first and second derivatives exist and agree with plain F.linear autograd
    (gp,) = torch.autograd.grad(bent.pow(2).sum(), pts, create_graph=True)
    gp.pow(2).sum().backward()
    assert pts.grad is not None and all(p.grad is not None and torch.isfinite(p.grad).all() for p in rb.parameters())


def test_eligibility_of_training_calls():
    cfg = SceneConfig(N_importance=64)
    rb, coarse, fine = build_modules(make_scene(cfg, 0))
    rays, _ = make_rays(4, 0, cfg)
    assert T.why_not_trainable(coarse, fine, 64, 64, False, False, rays) == "rays are not on a ROCm device"
    cuda_like = rays.to("meta") if False else rays          # device checks come first; the rest is architecture
    cfgv = SceneConfig(N_importance=64, use_viewdirs=True)
    _, cv, fv = build_modules(make_scene(cfgv, 0))

    class OnGpu:
        """a stand-in whose .device says cuda: the remaining checks only read module attributes"""
        device = torch.device("cuda", 0)
        requires_grad = False

    class OnGpuWithGrad(OnGpu):
        requires_grad = True

    # rays that require a gradient (camera refinement): the kernels treat rays as data, so such a call must reach the
    # reference's autograd graph instead of coming back with a silently missing gradient
    assert "requires a gradient" in T.why_not_trainable(coarse, fine, 64, 64, False, False, OnGpuWithGrad)
    class OnGpu8(OnGpu):
        shape = (4, 8)

    class OnGpu11(OnGpu):
        shape = (4, 11)

    # view-dependent head: trains natively (both branches in the training kernels) with finite-difference directions -- which
    # need nothing from the ray batch --, with the exact Jacobian directions and without a bender when the batch carries the
    # rays' unit directions (columns 8..10, train.py:73-76), not without them
    assert T.why_not_trainable(cv, fv, 64, 64, False, False, OnGpu8) is None
    cv.approx_nonrigid_viewdirs = fv.approx_nonrigid_viewdirs = False
    assert T.why_not_trainable(cv, fv, 64, 64, False, False, OnGpu11) is None
    assert "without view directions" in T.why_not_trainable(cv, fv, 64, 64, False, False, OnGpu8)
    cv.approx_nonrigid_viewdirs = fv.approx_nonrigid_viewdirs = True
    _, cn, fn = build_modules(make_scene(SceneConfig(N_importance=64, use_viewdirs=True, ray_bending=False), 0))
    assert T.why_not_trainable(cn, fn, 64, 64, False, False, OnGpu11) is None
    assert "without view directions" in T.why_not_trainable(cn, fn, 64, 64, False, False, OnGpu8)
    assert T.why_not_trainable(coarse, fine, 64, 64, True, False, OnGpu) is None                  # lindisp trains natively
    assert T.why_not_trainable(coarse, fine, 64, 64, False, True, OnGpu).startswith("pytest flag")
    assert T.why_not_trainable(coarse, fine, 200, 100, False, False, OnGpu) is None               # up to 1024 samples per pass
    assert T.why_not_trainable(coarse, fine, 600, 500, False, False, OnGpu) == "more than 1024 samples per ray"
    assert T.why_not_trainable(coarse, fine, 64, 64, False, False, OnGpu) is None
    cfgw = SceneConfig(N_importance=64, netwidth=128)
    _, cw, fw = build_modules(make_scene(cfgw, 0))
    assert T.why_not_trainable(cw, fw, 64, 64, False, False, OnGpu) is None      # the second compiled trunk width
    cfgw = SceneConfig(N_importance=64, netwidth=192)
    _, cw, fw = build_modules(make_scene(cfgw, 0))
    # round 5: a plain trunk outside the compiled set trains on the run-time-parameterised kernel (training._GenericTrunk) ...
    assert T.why_not_trainable(cw, fw, 64, 64, False, False, OnGpu) is None
    for kw in (dict(netwidth=320, netdepth=5, skips=(2,), multires=6), dict(netwidth=64, netdepth=3, skips=()), dict(netwidth=512, netdepth=2, skips=(0,)),
               dict(netwidth=192, use_viewdirs=True), dict(netwidth=480, netdepth=3, use_viewdirs=True, multires_views=2),      # (and with the view-dependent head)
               dict(netwidth=128, use_viewdirs=True),      # (renders on compiled kernels, trains on a generic handle of its own: render_rays_train forces one)
               dict(netwidth=192, ray_bending=False, time_conditioned_baseline=True), dict(ray_bending=False, time_conditioned_baseline=True, latent_size=16),
               dict(netwidth=192, use_viewdirs=True, approx_nonrigid_viewdirs=False), dict(netwidth=128, use_viewdirs=True, approx_nonrigid_viewdirs=False, bend_depth=7)):
        _, cw, fw = build_modules(make_scene(SceneConfig(N_importance=64, **kw), 0))
        assert T.why_not_trainable(cw, fw, 64, 64, False, False, OnGpu11) is None, kw
    # ... but not with a width the kernel's 4-element rows do not divide, two skip connections, widths beyond the kernel's,
    # or exact Jacobian directions with a bender of another shape (their tangent runs through the bender's compiled training kernels)
    for kw, why in ((dict(netwidth=190), "non-default trunk"), (dict(netwidth=192, skips=(2, 5)), "non-default trunk"), (dict(netwidth=640), "non-default trunk"),
                    (dict(netwidth=512, use_viewdirs=True), "non-default trunk"),
                    (dict(netwidth=192, use_viewdirs=True, approx_nonrigid_viewdirs=False, bend_hidden=96), "exact Jacobian"),
                    (dict(netwidth=128, use_viewdirs=True, approx_nonrigid_viewdirs=False, latent_size=16), "exact Jacobian")):
        _, cw, fw = build_modules(make_scene(SceneConfig(N_importance=64, **kw), 0))
        assert why in T.why_not_trainable(cw, fw, 64, 64, False, False, OnGpu11), kw


@pytest.mark.parametrize("D,W,skips,views", [(6, 24, (2,), False), (3, 16, (), False), (4, 20, (0,), True), (2, 12, (), True)])
def test_generic_weight_gradient_plan_against_torch_autograd(D, W, skips, views):
    """training._generic_grad_plan -- which products of the saved arrays make up a non-compiled trunk's weight / bias gradients, and
    where each lands in the flat result (nrnerf_tn_products executes the list on the GPU) -- emulated on the CPU with plain matrix
    products over activations / pre-activation gradients taken from a torch MLP of the reference's shape (rnh:253-306), against that
    MLP's autograd: every parameter's gradient, in _generic_trunk_params order, positions no product covers zero."""
    g = torch.Generator().manual_seed(D * 100 + W)
    M, n_in, n_dir, half, C_out = 50, 9, 6, W // 2, 5
    lin = lambda o, i: torch.nn.Linear(i, o).double()
    pts = [lin(W, n_in)] + [lin(W, W + (n_in if (i - 1) in skips else 0)) for i in range(1, D)]
    x, xv = torch.randn(M, n_in, generator=g).double(), torch.randn(M, n_dir, generator=g).double()
    acts, pres, h = [], [], x
    for i, l in enumerate(pts):
        inp = x if i == 0 else (torch.cat([x, h], -1) if (i - 1) in skips else h)
        z = l(inp)
        z.retain_grad()
        pres.append(z)
        h = torch.relu(z)
        acts.append(h)
    if not views:
        head = lin(C_out, W)
        raw = head(h)
        params = [p for l in pts for p in (l.weight, l.bias)] + [head.weight, head.bias]
    else:
        alpha, feat, vl, rgb = lin(1, W), lin(W, W), lin(half, W + n_dir), lin(3, half)
        f = feat(h)
        f.retain_grad()
        zv = vl(torch.cat([f, xv], -1))
        zv.retain_grad()
        hv = torch.relu(zv)
        raw = torch.cat([rgb(hv), alpha(h)], -1)
        params = [p for l in pts for p in (l.weight, l.bias)] + [alpha.weight, alpha.bias, feat.weight, feat.bias, vl.weight, vl.bias, rgb.weight, rgb.bias]
    gr = torch.randn(M, 4, generator=g).double()
    (raw[:, :4] * gr).sum().backward()
    ops = {"g": gr, "enc": x, "encv": xv}
    for i in range(D):
        ops[("acts", i)], ops[("d_pre", i)] = acts[i].detach(), pres[i].grad
    if views:
        pad = lambda t: torch.cat([t, torch.zeros(M, W - t.shape[1], dtype=t.dtype)], 1)          # the saved arrays are [M, W] rows
        ops[("acts", D)], ops[("d_pre", D)] = f.detach(), f.grad
        ops[("acts", D + 1)], ops[("d_pre", D + 1)] = pad(hv.detach()), pad(zv.grad)
    plan, shapes, total = T._generic_grad_plan(D, W, n_in, skips, C_out, views, half, n_dir)
    flat = torch.zeros(total, dtype=torch.float64)
    for a, ac, b, bc, wo, wi, ldo, ow, ob in plan:
        prod = ops[a][:, ac:ac + wo].T @ ops[b][:, bc:bc + wi]
        for o in range(wo):
            flat[ow + o * ldo: ow + o * ldo + wi] += prod[o]
        if ob is not None:
            flat[ob:ob + wo] += ops[a][:, ac:ac + wo].sum(0)
    assert [tuple(p.shape) for p in params] == shapes and total == sum(p.numel() for p in params)
    o = 0
    for p, shp in zip(params, shapes):
        want = p.grad if p.grad is not None else torch.zeros_like(p)
        assert torch.allclose(flat[o:o + p.numel()].view(shp), want, atol=1e-10), shp
        o += p.numel()


def test_which_handle_a_training_call_works_on():
    """training._training_handle_flags: shapes that render on compiled kernels but have no compiled TRAINING kernels get a generic handle
    of their own (MODEL_FORCE_GENERIC); exact Jacobian directions off the compiled set a handle described without them
    (MODEL_PY_TRAINING_HANDLE: the training path computes the directions itself); everything else the handle rendering uses."""
    from nonrigid_nerf_amd import _lib
    FG, TH = _lib.MODEL_FORCE_GENERIC, _lib.MODEL_PY_TRAINING_HANDLE
    cases = [(dict(), 0), (dict(netwidth=128), 0), (dict(use_viewdirs=True), 0), (dict(use_viewdirs=True, approx_nonrigid_viewdirs=False), 0),
             (dict(netwidth=192), 0), (dict(netwidth=192, use_viewdirs=True), 0), (dict(bend_depth=7, use_viewdirs=True), 0),
             (dict(ray_bending=False, time_conditioned_baseline=True), 0),
             (dict(netwidth=128, use_viewdirs=True), FG), (dict(netwidth_fine=128, use_viewdirs=True), FG),
             (dict(ray_bending=False, time_conditioned_baseline=True, latent_size=16), FG), (dict(ray_bending=False, time_conditioned_baseline=True, netwidth=128), FG),
             (dict(netwidth=192, use_viewdirs=True, approx_nonrigid_viewdirs=False), TH), (dict(netwidth=128, use_viewdirs=True, approx_nonrigid_viewdirs=False), FG | TH),
             (dict(netwidth=192, ray_bending=False, use_viewdirs=True, approx_nonrigid_viewdirs=False), 0)]       # (no bender: the rays' own directions)
    for kw, want in cases:
        rb, c, f = build_modules(make_scene(SceneConfig(N_importance=64, **kw), 0))
        assert T._training_handle_flags([c, f], rb) == want, (kw, T._training_handle_flags([c, f], rb), want)
    assert not (TH & 0xffff), "the Python-side marker must not reach nrnerf_model_desc.flags"
    # ... and what the description handed to the library looks like for such a handle
    from nonrigid_nerf_amd.render import build_model_desc
    rb, c, f = build_modules(make_scene(SceneConfig(N_importance=64, netwidth=192, use_viewdirs=True, approx_nonrigid_viewdirs=False), 0))
    d_render, _k1 = build_model_desc(c, f, "f32", 0, 0)
    d_train, _k2 = build_model_desc(c, f, "f32", 0, FG | TH)
    assert d_render.exact_viewdirs == 1 and d_render.flags == 0
    assert d_train.exact_viewdirs == 0 and d_train.flags == FG


def _apply_index(record, index, n_partials_full, n_partials_short):
    """What nrnerf_reduce_partials computes, on the host, for `n` identical records: value x number of records added."""
    from nonrigid_nerf_amd import _lib
    idx = index.long()
    pos = torch.where(idx >= 0, idx & (_lib.REDUCE_SHORT - 1), torch.zeros_like(idx))
    count = torch.where((idx & _lib.REDUCE_SHORT) != 0, n_partials_short, n_partials_full).to(record.dtype)
    return torch.where(idx >= 0, record[pos] * count, torch.zeros((), dtype=record.dtype))


@pytest.mark.parametrize("cfg_kw", [dict(), dict(netwidth=128), dict(use_viewdirs=True), dict(ray_bending=False, time_conditioned_baseline=True)],
                         ids=["default", "w128", "viewdirs", "time_conditioned"])
def test_trunk_gradient_index_table_matches_the_record_layout(cfg_kw):
    """training._trunk_grad_index (the map nrnerf_reduce_partials adds the partial sums through) against the record layout of
    include/nrnerf.h assembled by hand: dw_hidden [D-1][W][W], dw_enc [2][W][64], dw_head^T [W][64], db [D+1][W]; which
    positions stop at the short record count; what stays zero (latent columns, 5th channel, the head's bias)."""
    from nonrigid_nerf_amd import _lib
    cfg = SceneConfig(N_importance=64, **cfg_kw)
    _, coarse, _ = build_modules(make_scene(cfg, 0))
    D, W = int(coarse.D), int(coarse.W)
    views = bool(coarse.use_viewdirs)
    C_out = 4 if views else int(coarse.output_linear.weight.shape[0])
    params = T._trunk_params(coarse)
    n_enc = 63
    n_lat = int(coarse.pts_linears[0].weight.shape[1]) - n_enc
    index, shapes, hb = T._trunk_grad_index(coarse, D, W, C_out, views, n_lat, "cpu")
    assert [tuple(p.shape) for p in params] == [tuple(s) for s in shapes]
    n_colour = sum(int(p.numel()) for p in params[2 * D + 2:]) if views else 0            # the colour branch follows alpha_linear
    assert int(index.shape[0]) == sum(int(p.numel()) for p in params) and hb == int(index.shape[0]) - n_colour - (1 if views else C_out)
    rec = torch.arange(_lib.wgrad_stride_views(D, W) if views else _lib.wgrad_stride(D, W), dtype=torch.float64) + 1.0    # one record, every slot distinct and non-zero
    flat = _apply_index(rec, index, torch.tensor(7), torch.tensor(3))
    got = T._split_flat(flat, shapes)
    o = 0
    dwh = rec[o:o + (D - 1) * W * W].view(D - 1, W, W); o += (D - 1) * W * W
    dwe = rec[o:o + 2 * W * 64].view(2, W, 64); o += 2 * W * 64
    dwo = rec[o:o + W * 64].view(W, 64); o += W * 64
    db = rec[o:o + (D + 1) * W].view(D + 1, W)
    skips = set(int(s) for s in coarse.skips)
    zl = torch.zeros(W, n_lat, dtype=torch.float64)
    for i in range(D):
        if i == 0:
            want = torch.cat([dwe[0][:, :n_enc] * 3, zl], 1)
        elif (i - 1) in skips:
            want = torch.cat([dwe[1][:, :n_enc] * 3, zl, dwh[i - 1] * 7], 1)
        else:
            want = dwh[i - 1] * 7
        assert torch.equal(got[2 * i], want), i
        assert torch.equal(got[2 * i + 1], db[i] * (3 if i == 0 else 7)), i
    if views:
        assert torch.equal(got[2 * D], (dwo[:, 3:4] * 3).t()) and torch.equal(got[2 * D + 1], torch.zeros(1, dtype=torch.float64))
    else:
        want = torch.zeros(C_out, W, dtype=torch.float64)
        want[:4] = (dwo[:, :4] * 3).t()
        assert torch.equal(got[2 * D], want) and torch.equal(got[2 * D + 1], torch.zeros(C_out, dtype=torch.float64))


@pytest.mark.parametrize("cfg_kw", [dict(), dict(use_viewdirs=True)], ids=["default", "viewdirs"])
def test_trunk_gradient_index_marks_the_head_bias_for_the_aux_reduction(cfg_kw):
    """training._trunk_grad_index(head_sums=True): exactly the head's bias positions -- output_linear.bias's first four channels, or
    alpha_linear.bias and rgb_linear.bias -- carry -2 ("filled by nrnerf_reduce_partials_aux from nrnerf_wgrad_args.head_sums"), every
    other entry equals the plain table's; the positions handed to the aux reduction (training._Trunk._weight_grads) are those."""
    cfg = SceneConfig(N_importance=64, **cfg_kw)
    _, coarse, _ = build_modules(make_scene(cfg, 0))
    D, W = int(coarse.D), int(coarse.W)
    views = bool(coarse.use_viewdirs)
    C_out = 4 if views else int(coarse.output_linear.weight.shape[0])
    plain, shapes, hb = T._trunk_grad_index(coarse, D, W, C_out, views, 0, "cpu")
    aux, shapes2, hb2 = T._trunk_grad_index(coarse, D, W, C_out, views, 0, "cpu", head_sums=True)
    assert shapes == shapes2 and hb == hb2
    n = int(plain.shape[0])
    pos = (n - 3, n - 2, n - 1, hb) if views else tuple(hb + c for c in range(4))
    marked = (aux == -2).nonzero().flatten().tolist()
    assert sorted(marked) == sorted(pos)
    keep = torch.ones(n, dtype=torch.bool)
    keep[list(pos)] = False
    assert torch.equal(plain[keep], aux[keep]) and bool((plain[~keep] == -1).all())


def test_pooled_draws_partition_two_generator_calls():
    """training._pooled_draws (POOLED_DRAWS): the dict render._draw_randoms builds and the divergence term's probe vectors as views of ONE
    uniform and ONE normal draw -- shapes, disjointness, which keys exist for which flags, the noise scaled by raw_noise_std."""
    rays = torch.zeros(10, 11)
    torch.manual_seed(0)
    rnd, e = T._pooled_draws(rays, dict(N_samples=6, N_importance=4, perturb=1.0, raw_noise_std=2.0), True)
    assert {k: tuple(v.shape) for k, v in rnd.items()} == {"u_coarse": (10, 6), "u_fine": (10, 4), "noise_coarse": (10, 6), "noise_fine": (10, 10)}
    assert tuple(e.shape) == (60, 3)
    torch.manual_seed(0)
    u, nrm = torch.rand(100), torch.randn(160 + 180)
    assert torch.equal(rnd["u_coarse"].flatten(), u[:60]) and torch.equal(rnd["u_fine"].flatten(), u[60:])
    assert torch.equal(rnd["noise_coarse"].flatten(), nrm[:60] * 2.0) and torch.equal(rnd["noise_fine"].flatten(), nrm[60:160] * 2.0)
    assert torch.equal(e.flatten(), nrm[160:])
    rnd, e = T._pooled_draws(rays, dict(N_samples=6, N_importance=0, perturb=0.0, raw_noise_std=1.0), False)
    assert set(rnd) == {"noise_coarse"} and e is None
    rnd, e = T._pooled_draws(rays, dict(N_samples=6, N_importance=4, perturb=1.0, raw_noise_std=0.0), False)
    assert set(rnd) == {"u_coarse", "u_fine"} and e is None


@pytest.mark.parametrize("divergence", [False, True])
@pytest.mark.parametrize("depth", [5, 7])
def test_bender_gradient_index_table_matches_the_slots(divergence, depth):
    """training._bender_grad_index against the slot layout of nrnerf_bender_wgrad / nrnerf_bender_divergence_backward: one
    [64][64] + [64] slot per layer (the divergence call: network[0] in two slots, point columns and latent columns)."""
    from nonrigid_nerf_amd import _lib
    rb, _, _ = build_modules(make_scene(SceneConfig(bend_depth=depth), 0))
    layers = list(rb.network) + list(rb.rigidity_network)
    index, shapes = T._bender_grad_index(rb, "cpu", divergence)
    params = T._bender_params(rb)
    assert [tuple(p.shape) for p in params] == [tuple(s) for s in shapes]
    slot = _lib.BENDER_WGRAD_SLOT
    nj = len(layers) + (1 if divergence else 0)
    rec = torch.arange(nj * slot, dtype=torch.float64) + 1.0
    got = T._split_flat(_apply_index(rec, index, torch.tensor(1), torch.tensor(1)), shapes)
    tot = rec.view(nj, slot)
    dW, dB = tot[:, :4096].view(nj, 64, 64), tot[:, 4096:]
    k_out = 0
    for k, lin in enumerate(layers):
        o, i_ = int(lin.weight.shape[0]), int(lin.weight.shape[1])
        if divergence and k == 0:
            want, job = torch.cat([dW[0, :o, :3], dW[1, :o, :i_ - 3]], 1), 0
        else:
            job = k + 1 if divergence else k
            want = dW[job, :o, :i_]
        assert torch.equal(got[k_out], want), k
        k_out += 1
        if lin.bias is not None:
            assert torch.equal(got[k_out], dB[job, :o]), k
            k_out += 1


def test_folded_colour_parameters_equal_the_two_layers():
    """training._colour_params: the colour branch of the view-dependent head as the kernels evaluate it -- feature_linear
    folded into views_linears[0] (no nonlinearity between them, rnh:286-301) -- against the reference's layers one by one:
    rgb logits, and the gradients wrt the last hidden activation and every parameter carried back through the fold by
    autograd (the kernels return the gradient wrt the FOLDED weights; the chain rule through the two small products is
    torch's, in parameter space)."""
    import torch.nn.functional as F
    cfg = SceneConfig(use_viewdirs=True, N_importance=64)
    _, net, _ = build_modules(make_scene(cfg, 0))
    net.requires_grad_(True)
    g = torch.Generator().manual_seed(0)
    h = torch.randn(320, 256, generator=g).requires_grad_(True)
    d = F.normalize(torch.randn(320, 3, generator=g), dim=-1)
    enc = T.posenc(d, 4)
    w = torch.randn(320, 3, generator=g)
    names = ["feature_linear.weight", "feature_linear.bias", "views_linears.0.weight", "views_linears.0.bias", "rgb_linear.weight", "rgb_linear.bias"]
    params = dict(net.named_parameters())

    def run(folded):
        for p in net.parameters():
            p.grad = None
        h.grad = None
        if folded:
            wfold, bfold, wdir, wr, br = T._colour_params(net)
            assert tuple(wfold.shape) == (128, 256) and tuple(wdir.shape) == (128, 27)
            hv = F.relu(F.linear(h, wfold) + F.linear(enc, wdir) + bfold)
            y = F.linear(hv, wr, br)
        else:
            feat = net.feature_linear(h)                                                     # rnh:286
            y = net.rgb_linear(F.relu(net.views_linears[0](torch.cat([feat, enc], -1))))     # rnh:296-303
        (y * w).sum().backward()
        return [y.detach().clone(), h.grad.clone()] + [params[n].grad.clone() for n in names]

    for a, b in zip(run(False), run(True)):
        assert float((a - b).abs().max()) <= 5e-6 * float(a.abs().max()) + 1e-9


def test_trunk_gradient_index_places_the_colour_branch():
    """training._trunk_grad_index for a view-dependent head: every parameter of _trunk_params order finds its slot in one
    record of nrnerf_trunk_wgrad (NRNERF_WGRAD_STRIDE_VIEWS): the folded matrix and its bias after the plain record, the
    direction columns and rgb_linear (transposed there) in the 64-column products, the biases the caller sums as -1."""
    from nonrigid_nerf_amd import _lib
    cfg = SceneConfig(use_viewdirs=True, N_importance=64)
    _, net, _ = build_modules(make_scene(cfg, 0))
    D, W = 8, 256
    index, shapes, hb = T._trunk_grad_index(net, D, W, 4, True, 0, torch.device("cpu"))
    assert [tuple(p.shape) for p in T._trunk_params(net)] == [tuple(x) for x in shapes]
    stride, base = _lib.wgrad_stride_views(D, W), _lib.wgrad_stride(D, W)
    idx = index.numpy().astype("int64")
    pos = idx & (_lib.REDUCE_SHORT - 1)
    assert int(pos[idx >= 0].max()) < stride and idx[hb] == -1 and (idx[-3:] == -1).all()
    o = int(sum(int(torch.tensor(x).prod()) for x in shapes[:-5]))
    fold = idx[o:o + 128 * 256].reshape(128, 256)
    assert fold[0, 0] == base and fold[1, 0] == base + 256 and fold[127, 255] == base + 128 * 256 - 1      # n_partials records
    bias = idx[o + 128 * 256:o + 128 * 256 + 128]
    assert bias[0] == base + 128 * 256 + 2 * 128 * 64 and bias[127] == stride - 1
    dirs = idx[o + 128 * 257:o + 128 * 257 + 128 * 27].reshape(128, 27)
    assert dirs[2, 5] == ((base + 128 * 256 + 2 * 64 + 5) | _lib.REDUCE_SHORT)
    rgb = idx[o + 128 * 257 + 128 * 27:o + 128 * 257 + 128 * 27 + 3 * 128].reshape(3, 128)
    assert rgb[1, 7] == ((base + 128 * 256 + 128 * 64 + 7 * 64 + 1) | _lib.REDUCE_SHORT)


def test_param_token_is_rebuilt_when_parameters_are_frozen_and_unfrozen():
    """freezing the bender (fitting test-time latent codes) and unfreezing it bumps no version counter; a token
    cached while frozen would carry no gradient to the parameters for ever after."""
    lin = torch.nn.Linear(4, 3)
    params = list(lin.parameters())
    with torch.enable_grad():
        t0 = T._param_token(lin, params)
        assert t0.requires_grad and T._param_token(lin, params) is t0            # cached within an iteration
        lin.requires_grad_(False)
        t1 = T._param_token(lin, params)
        assert not t1.requires_grad
        lin.requires_grad_(True)
        t2 = T._param_token(lin, params)
        assert t2 is not t1 and t2.requires_grad
        (t2 * torch.arange(t2.numel(), dtype=torch.float32)).sum().backward()
    assert lin.weight.grad is not None and lin.bias.grad is not None
    assert torch.equal(lin.weight.grad.reshape(-1), torch.arange(12, dtype=torch.float32))


def test_param_slot_walk_notices_a_renamed_parameter():
    """same number of parameters, different names -- the cached walk must be rebuilt, not raise KeyError."""
    from nonrigid_nerf_amd import render as R
    m = torch.nn.Linear(4, 3)
    assert [n for _, n in R._param_slots(m)] == ["weight", "bias"]
    del m._parameters["bias"]
    m.register_parameter("scale", torch.nn.Parameter(torch.ones(3)))
    assert [n for _, n in R._param_slots(m)] == ["weight", "scale"]
    R._fingerprint([m])


def test_fingerprint_counts_the_steps_of_the_optimisers_that_own_a_networks_parameters():
    """Fused optimisers (torch.optim.Adam(fused=True)) update their parameters without bumping Tensor._version (measured on
    the GPU: tools/experiments/debug_refresh_path.py), so the staleness check of render.get_model also counts optimiser steps
    -- PER PARAMETER (round 5): a step of an optimiser that owns none of a network's parameters (one that steps only
    latent codes, another model's) leaves that network's fingerprint, hence its packed handle, alone."""
    from nonrigid_nerf_amd import render as R
    R._watch_optimizers()
    mine, frozen, other = torch.nn.Linear(4, 3), torch.nn.Linear(4, 3).requires_grad_(False), torch.nn.Linear(2, 2)
    opt_other = torch.optim.SGD(other.parameters(), lr=0.1)
    opt_mine = torch.optim.SGD(mine.parameters(), lr=0.0)          # (lr 0: the values -- and a fused optimiser's version counters -- do not move)
    fp_m, fp_f, fp_o = R._fingerprint([mine, None]), R._fingerprint([frozen]), R._fingerprint([other])
    assert R._fingerprint([mine, None]) == fp_m
    other.weight.grad, other.bias.grad = torch.ones(2, 2), torch.ones(2)
    opt_other.step()                                         # somebody else's optimiser
    assert R._fingerprint([mine, None]) == fp_m and R._fingerprint([frozen]) == fp_f and R._fingerprint([other]) != fp_o
    mine.weight.grad, mine.bias.grad = torch.zeros(3, 4), torch.zeros(3)
    v = (mine.weight._version, mine.bias._version)
    opt_mine.step()
    if (mine.weight._version, mine.bias._version) == v:      # (an optimiser that leaves the version counters alone, like the fused ones)
        assert R._fingerprint([mine, None]) != fp_m
    assert getattr(mine.weight, "_nrnerf_steps", 0) == 1 and getattr(frozen.weight, "_nrnerf_steps", 0) == 0
    assert R._fingerprint([mine, None]) != fp_m
    # the version counters keep working on their own
    fp_f = R._fingerprint([frozen])
    with torch.no_grad():
        frozen.weight.mul_(2.0)
    assert R._fingerprint([frozen]) != fp_f


def test_a_fallback_to_the_reference_is_announced_once_per_reason():
    """A call the HIP path cannot take runs on the reference's own function saved by install() -- and says so ONCE per (entry
    point, reason) with a FallbackWarning: a training run that silently takes the eager path is ten times slower than it
    needs to be.  NRNERF_QUIET_FALLBACK=1 silences it."""
    import types
    import warnings
    from nonrigid_nerf_amd import render as R
    cfg = SceneConfig(N_importance=0)
    _, coarse, _ = build_modules(make_scene(cfg, 0))
    rays, lat = make_rays(4, 0, cfg)
    seen = []
    mod = types.SimpleNamespace(render_rays=lambda *a, **k: seen.append("rr") or {"rgb_map": torch.zeros(4, 3)},
                                batchify_rays=lambda *a, **k: seen.append("br") or {"rgb_map": torch.zeros(4, 3)})
    undo = R.install(mod)
    R._fallback_seen.clear()
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            for _ in range(3):                                   # CPU rays: "rays are not on a ROCm device"
                with torch.no_grad():
                    mod.batchify_rays(rays, {"ray_bending_latents": lat}, network_fn=coarse, network_query_fn=None, N_samples=64)
            notes = [x for x in w if issubclass(x.category, R.FallbackWarning)]
        assert seen == ["br"] * 3 and len(notes) == 1 and "not on a ROCm device" in str(notes[0].message)
        os.environ["NRNERF_QUIET_FALLBACK"] = "1"
        R._fallback_seen.clear()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            with torch.no_grad():
                mod.batchify_rays(rays, {"ray_bending_latents": lat}, network_fn=coarse, network_query_fn=None, N_samples=64)
            assert not [x for x in w if issubclass(x.category, R.FallbackWarning)]
    finally:
        os.environ.pop("NRNERF_QUIET_FALLBACK", None)
        undo()


def test_install_restores_the_previous_precision_on_uninstall():
    """install() selects "f32" for the drop-in; uninstall() must give direct callers their precision back."""
    import types
    from nonrigid_nerf_amd import render as R
    before = R.get_precision()
    try:
        R.set_precision("bf16")
        mod = types.SimpleNamespace(render_rays=lambda *a, **k: None, batchify_rays=lambda *a, **k: None)
        orig = (mod.render_rays, mod.batchify_rays)
        undo = R.install(mod)
        assert R.get_precision() == "f32" and mod.render_rays is R.render_rays
        undo()
        assert R.get_precision() == "bf16" and (mod.render_rays, mod.batchify_rays) == orig
    finally:
        R.set_precision(before)


def test_pinned_numbers_are_held_to_twice_the_committed_value(tmp_path, monkeypatch):
    """tests/helpers.py::check_pinned (the fp32 end-to-end fractions of the GPU tier): <= max(2 x pinned, pinned + 2 elements)
    passes, more fails, an unpinned case fails unless a recording file is given."""
    import json
    from tests import helpers
    pin = tmp_path / "pins.json"
    pin.write_text(json.dumps({"case_a": {"moved": 0.010, "rgb": 0.0}}))
    monkeypatch.setattr(helpers, "PIN_FILE", str(pin))
    monkeypatch.delenv("NRNERF_PIN_RECORD", raising=False)
    helpers.check_pinned("case_a", {"moved": 0.019, "rgb": 2.0 / 1000}, {"moved": 10000, "rgb": 1000})      # 2 x, and two elements of 1000
    with pytest.raises(AssertionError):
        helpers.check_pinned("case_a", {"moved": 0.021, "rgb": 0.0}, {"moved": 10000, "rgb": 1000})
    with pytest.raises(AssertionError):
        helpers.check_pinned("case_a", {"moved": 0.0, "rgb": 3.0 / 1000}, {"moved": 10000, "rgb": 1000})
    with pytest.raises(AssertionError):
        helpers.check_pinned("case_unpinned", {"moved": 0.0}, {"moved": 10})
    rec = tmp_path / "rec.jsonl"
    monkeypatch.setenv("NRNERF_PIN_RECORD", str(rec))
    helpers.check_pinned("case_unpinned", {"moved": 0.5}, {"moved": 10})                                     # recording: no pin needed
    assert json.loads(rec.read_text().splitlines()[-1]) == {"case": "case_unpinned", "measured": {"moved": 0.5}}
