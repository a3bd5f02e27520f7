"""Training path of ``render_rays``: the reference trains through autograd (training_wrapper_class.forward,
train.py:152-287; backward + optimiser step, train.py:1594-1610); here the same call runs on the HIP library under
``torch.autograd``.

What is native (C ABI, include/nrnerf.h; kernels in csrc/nrnerf_train.h, csrc/nrnerf_train_bend.h, csrc/nrnerf_composite.hip):
  * the canonical network -- positional encoding, 8x256 (or 8x128) trunk, head (output_linear, or the view-dependent head:
    alpha_linear + feature_linear / views_linears[0] / rgb_linear on the samples' view directions) -- forward with saved activations, the fused
    backward-data pass on MFMA and every weight / bias gradient in one launch (``nrnerf_trunk_forward / _backward / _wgrad``),
    fp32 (v_mfma_f32_32x32x2_f32 throughout) or bf16;
  * the ray-bending and rigidity MLPs (35->64->64->64->64->3 and 3->32->32->1; ``nrnerf_bender_forward / _backward / _wgrad``):
    forward with saved activations and backward-data in exact fp32, down to the latent codes; the fine pass bends only its
    N_importance new samples and re-uses the coarse pass' bent points (SPLIT_FINE_BENDER; ``nrnerf_merge_rows`` puts the rows
    in merged-depth order and undoes that for the gradient);
  * the partial sums of every weight-gradient call added up straight into the parameters' layouts (``nrnerf_reduce_partials``);
  * the divergence regulariser (compute_divergence_loss, run_nerf_helpers.py:22-116; second order in autograd's terms): one
    forward-mode tangent through the bender and a two-chain backward (``nrnerf_bender_divergence_forward / _backward``);
  * compositing forward (raw2outputs, train.py:724-789), hierarchical sampling + merge (run_nerf_helpers.py:651-698,
    train.py:910-920, no gradient: the reference detaches the sample positions), the compositing backward, the coarse depths
    (``nrnerf_composite_forward / _backward``, ``nrnerf_sample_depths``).
What is left to libraries, as plumbing:
  * products in PARAMETER space: feature_linear folded into views_linears[0] for the view-dependent head (``_colour_params``:
    the kernels evaluate both branches of that head and return the gradient wrt the folded weights; autograd carries it
    back through the two small products);
  * the two small GEMMs that turn a time-conditioned baseline's latent columns into per-ray biases.
``NATIVE_BENDER = False`` runs the bender's MLPs as torch ops on the modules' parameters instead (the gradient-parity tests use
it because it reproduces the reference's bent points bit for bit).
Eligible: the compiled architectures (render.py / README.md) in precision fp32 or bf16 (``render.set_precision``; "f16" trains
in bf16: unscaled f16 gradients underflow), exact Jacobian view directions included (the tangent J d and its gradient come
from the divergence regulariser's kernels).  Anything else is handed to the reference
by ``render.render_rays`` as before.  ``training_loss`` is the reference's whole iteration on these entry points,
``GraphedStep`` the same captured in one HIP graph.
"""
from __future__ import annotations

import ctypes as C
import weakref

import torch
import torch.nn.functional as F

from . import _lib
from . import render as R


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _mstream(model, dev):
    """The current stream for a kernel that READS the packed weights of `model`; the model remembers it, so that a weight
    refresh issued from another stream orders itself after this work (render.Model.update_from_device)."""
    cur = torch.cuda.current_stream(dev)
    model._train_stream = cur
    return C.c_void_p(cur.cuda_stream)


def posenc(x: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """Embedder.embed (run_nerf_helpers.py:120-150) -- only used to materialise the first operand of the weight-gradient
    GEMMs of layers 0 and skip+1 (the kernels compute the encoding in registers and never write it)."""
    cols = [x]
    for k in range(n_freqs):
        xs = x * float(2 ** k)
        cols += [torch.sin(xs), torch.cos(xs)]
    return torch.cat(cols, -1)


def _chunks(m: int, target: int = 4096) -> int:
    """Number of equal row blocks to cut an [m, .] operand into: the largest divisor of m that keeps blocks >= target rows
    (and at most ~256 blocks: enough to fill the chip, and the fp32 partial products that are added afterwards stay small)."""
    target = max(target, m // 256)
    if m < 2 * target:
        return 1
    b = m // target
    while b > 1 and m % b:
        b -= 1
    return b


def _rows4(x: torch.Tensor, M: int) -> torch.Tensor:
    """[.., 3] float32 -> the [M,4] rows (xyz + one float the kernels ignore) the C ABI takes.  A tensor that already is the
    xyz part of such rows (what _Bender returns and _Trunk.backward hands back) is re-viewed, anything else copied."""
    n = x.shape[-2] if x.dim() == 3 else 1
    if x.dtype == torch.float32 and x.dim() == 3 and x.stride() == (n * 4, 4, 1) and x.untyped_storage().nbytes() >= (x.storage_offset() + M * 4) * 4 \
            and x.storage_offset() % 4 == 0:
        return x.as_strided((M, 4), (4, 1), x.storage_offset())
    rows = torch.zeros(M, 4, dtype=torch.float32, device=x.device)
    rows[:, :3] = x.reshape(M, 3)
    return rows


def _is_f32(model) -> bool:
    """Layout of the saved arrays (row-major fp32 vs bf16 block tiles) as the LIBRARY sees this handle -- asked of the handle,
    not inferred from a precision string, so an alias can never pair bf16-sized buffers with the fp32 kernels."""
    return int(model.lib.nrnerf_model_precision(model.handle)) == _lib.PRECISIONS["f32"]


_GRAD_INDEX = {}


def _reduce_partials(parts: torch.Tensor, n_short: int, index: torch.Tensor, aux=None, aux_pos=None) -> torch.Tensor:
    """Flat fp32 gradient buffer [len(index)] = the records of partial sums `parts` [P, stride] added up and laid out as the
    parameters are (nrnerf_reduce_partials: one launch instead of a sum, zero-fills of the unwritten records and a copy per
    parameter whose gradient is not a contiguous block of the record).  ``aux`` [R,4] + ``aux_pos`` (four positions, -1 = none): its
    column sums go to those positions in the same launch (nrnerf_reduce_partials_aux; `index` must be -2 there)."""
    out = torch.empty(int(index.shape[0]), dtype=torch.float32, device=parts.device)
    with torch.cuda.device(parts.device):
        if aux is None:
            _lib.check(_lib.load().nrnerf_reduce_partials(parts.data_ptr(), int(parts.stride(0)), int(parts.shape[0]), int(n_short), index.data_ptr(),
                                                          int(index.shape[0]), out.data_ptr(), _stream(parts.device)), "nrnerf_reduce_partials")
        else:
            pos = (C.c_int64 * 4)(*[int(x) for x in aux_pos])
            _lib.check(_lib.load().nrnerf_reduce_partials_aux(parts.data_ptr(), int(parts.stride(0)), int(parts.shape[0]), int(n_short), index.data_ptr(),
                                                              int(index.shape[0]), out.data_ptr(), aux.data_ptr(), int(aux.shape[0]), pos,
                                                              _stream(parts.device)), "nrnerf_reduce_partials_aux")
    return out


def _split_flat(flat: torch.Tensor, shapes):
    out, o = [], 0
    for shp in shapes:
        n = 1
        for d in shp:
            n *= int(d)
        out.append(flat[o:o + n].view(shp))
        o += n
    return out


def _trunk_grad_index(net, D, W, C_out, views, n_lat, dev, head_sums=False):
    """(index [n] int32 on dev, parameter shapes, offset of the head's bias) for nrnerf_reduce_partials: position in one record
    of nrnerf_trunk_wgrad (include/nrnerf.h) of every element of the trunk's weights and biases in _trunk_params order; -1
    (zero) for what no kernel produces: the latent columns of the time-conditioned baseline's two input layers (their
    gradient comes through ray_bias), the 5th output channel, and the head's bias (summed from d_raw4 by the caller) -- or, with
    ``head_sums``, -2 there: nrnerf_reduce_partials_aux fills the head's bias from nrnerf_wgrad_args.head_sums."""
    import numpy as np
    L = (int(net.input_ch) - 3) // 6
    n_enc = 3 + 6 * L
    skips = tuple(sorted(int(k) for k in net.skips))
    key = ("trunk", D, W, C_out, bool(views), int(getattr(net, "input_ch_views", 0)) if views else 0, n_enc, n_lat, skips, str(dev), bool(head_sums))
    HB = -2 if head_sums else -1
    if key in _GRAD_INDEX:
        return _GRAD_INDEX[key]
    SH = _lib.REDUCE_SHORT
    o_h, o_e = 0, (D - 1) * W * W
    o_o = o_e + 2 * W * 64
    o_db = o_o + W * 64
    rows = np.arange(W, dtype=np.int64)[:, None]
    lat = np.full((W, n_lat), -1, dtype=np.int64)
    segs, shapes = [], []
    for i in range(D):
        if i == 0:
            w = np.concatenate([(o_e + rows * 64 + np.arange(n_enc)[None]) | SH, lat], 1)
        elif (i - 1) in skips:                                                               # x = [encoding, (latent,) h] (rnh:278-282)
            w = np.concatenate([(o_e + W * 64 + rows * 64 + np.arange(n_enc)[None]) | SH, lat, o_h + (i - 1) * W * W + rows * W + np.arange(W)[None]], 1)
        else:
            w = o_h + (i - 1) * W * W + rows * W + np.arange(W)[None]
        b = o_db + i * W + np.arange(W)
        segs += [w.reshape(-1), (b | SH) if i == 0 else b]
        shapes += [tuple(w.shape), (W,)]
    if views:                                                  # alpha_linear (1 x W): the sigma channel's column of dw_head^T
        segs += [((o_o + np.arange(W) * 64 + 3) | SH), np.full(1, HB)]
        shapes += [(1, W), (1,)]
        head_bias_at = sum(int(x.shape[0]) for x in segs) - 1
        # the colour branch (_colour_params order): folded views layer's hidden columns [W/2, W] and bias [W/2], its direction
        # columns [W/2, 3 + 6 LV], rgb_linear [3, W/2] (from dw_rgb^T) and its bias (summed from d_raw4 by the caller)
        V = W // 2
        n_dir = int(net.input_ch_views)
        o_f = o_db + (D + 1) * W
        o_d = o_f + V * W
        o_r = o_d + V * 64
        o_bv = o_r + V * 64
        vrows = np.arange(V, dtype=np.int64)[:, None]
        segs += [(o_f + vrows * W + np.arange(W)[None]).reshape(-1), o_bv + np.arange(V),
                 ((o_d + vrows * 64 + np.arange(n_dir)[None]) | SH).reshape(-1),
                 ((o_r + np.arange(V)[None] * 64 + np.arange(3)[:, None]) | SH).reshape(-1), np.full(3, HB)]
        shapes += [(V, W), (V,), (V, n_dir), (3, V), (3,)]
    else:
        w = np.full((C_out, W), -1, dtype=np.int64)
        for ch in range(min(4, C_out)):
            w[ch] = (o_o + np.arange(W) * 64 + ch) | SH
        segs += [w.reshape(-1), np.where(np.arange(C_out) < 4, HB, -1)]
        shapes += [(C_out, W), (C_out,)]
    flat = np.concatenate(segs).astype(np.int32)
    if not views:
        head_bias_at = int(flat.shape[0]) - int(shapes[-1][0])
    _GRAD_INDEX[key] = (torch.from_numpy(flat).to(dev), shapes, head_bias_at)
    return _GRAD_INDEX[key]


def _bender_grad_index(rb, dev, divergence):
    """As _trunk_grad_index for the ray bender's two MLPs (_bender_params order) over the records of nrnerf_bender_wgrad
    (one slot per layer) or -- ``divergence`` -- of nrnerf_bender_divergence_backward (network[0] in two slots: the point's
    columns, the latent code's columns)."""
    import numpy as np
    layers = list(rb.network) + list(rb.rigidity_network)
    key = ("bender", bool(divergence), tuple((tuple(l.weight.shape), l.bias is not None) for l in layers), str(dev))
    if key in _GRAD_INDEX:
        return _GRAD_INDEX[key]
    slot = _lib.BENDER_WGRAD_SLOT
    segs, shapes = [], []
    for k, lin in enumerate(layers):
        o, i_ = int(lin.weight.shape[0]), int(lin.weight.shape[1])
        rows = np.arange(o, dtype=np.int64)[:, None]
        if divergence and k == 0:                   # jobs 0 / 1: the point's and the latent code's columns of network[0]
            w = np.concatenate([rows * 64 + np.arange(3)[None], slot + rows * 64 + np.arange(i_ - 3)[None]], 1)
            job = 0
        else:
            job = k + 1 if divergence else k
            w = job * slot + rows * 64 + np.arange(i_)[None]
        segs.append(w.reshape(-1))
        shapes.append((o, i_))
        if lin.bias is not None:
            segs.append(job * slot + 4096 + np.arange(o))
            shapes.append((o,))
    _GRAD_INDEX[key] = (torch.from_numpy(np.concatenate(segs).astype(np.int32)).to(dev), shapes)
    return _GRAD_INDEX[key]


class _ParamToken(torch.autograd.Function):
    """A [total] fp32 tensor that STANDS FOR a list of parameters in the autograd graph (its values are never read: the
    kernels take the weights from the packed model): a function that would return one gradient per parameter returns ONE
    flat gradient for the token instead, and this node hands each parameter its slice.  Several uses of the same parameters
    in an iteration (the bender: coarse samples, new samples, divergence term) share one token, so autograd adds their flat
    gradients with one launch each instead of one per parameter and use."""

    @staticmethod
    def forward(ctx, *params):
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.set_materialize_grads(False)
        return torch.empty(sum(int(p.numel()) for p in params), dtype=torch.float32, device=params[0].device)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * len(ctx.shapes)
        return tuple(_split_flat(g, ctx.shapes))


_TOKENS = weakref.WeakKeyDictionary()


def _param_token(owner, params):
    """The token of `params` for this iteration: one per (owner module, parameter objects and versions, grad mode).  An
    optimiser step bumps the versions, so the next iteration builds a fresh node; a token that is re-used after a backward
    pass is harmless (the node saves nothing)."""
    # requires_grad is part of the key: freezing the bender for a while (fitting test-time latent codes) and unfreezing it
    # bumps no version counter, and a token built while frozen carries no gradient to the parameters
    key = (tuple((id(p), p._version, bool(p.requires_grad)) for p in params), torch.is_grad_enabled())
    hit = _TOKENS.get(owner)
    if hit is not None and hit[0] == key:
        return hit[1]
    tok = _ParamToken.apply(*params)
    _TOKENS[owner] = (key, tok)
    return tok


class _Trunk(torch.autograd.Function):
    """raw4 [N,S,4] (differentiable), raw [N,S,C] (the reference's "raw" key; no gradient) = NeRF trunk(points)."""

    @staticmethod
    def forward(ctx, pts, model, net, which, ray_bias, dirs, *params):
        N, S = int(pts.shape[0]), int(pts.shape[1])
        M, dev = N * S, pts.device
        D, W = int(net.D), int(net.W)
        f32 = _is_f32(model)
        pts4 = _rows4(pts.detach(), M)               # the bender's own [M,4] rows when the points come from _Bender
        # saved activations.  fp32 mode: [layer][sample][width] rows; bf16 mode: [layer][block][width][32
        # samples] for nrnerf_trunk_wgrad (blocks of 32 consecutive samples of a ray) + 16 relu bits per lane and tile (one record per lane and layer)
        nblk = N * ((S + 31) // 32)
        acts = torch.empty(D, M, W, dtype=torch.float32, device=dev) if f32 else torch.empty(D, nblk, W, 32, dtype=torch.bfloat16, device=dev)
        mask = None if f32 else torch.empty(D, nblk, 64, W // 32, dtype=torch.int16, device=dev)
        raw4 = torch.empty(M, 4, dtype=torch.float32, device=dev)
        views = bool(net.use_viewdirs)
        # view-dependent head (rnh:284-304): both branches run behind the trunk in the same kernel, on the directions handed in;
        # raw4 = [rgb logits, density logit]
        C_out = 4 if views else int(net.output_linear.weight.shape[0])
        raw = torch.empty(M, C_out, dtype=torch.float32, device=dev)
        a = _lib.TrunkArgs()
        a.struct_size = C.sizeof(_lib.TrunkArgs)
        a.which, a.n_rays, a.n_samples = int(which), N, S
        a.pts4, a.acts, a.raw4, a.raw, a.raw_ch = pts4.data_ptr(), acts.data_ptr(), raw4.data_ptr(), raw.data_ptr(), C_out
        a.relu_mask = None if f32 else mask.data_ptr()
        saved = [pts4, acts] if f32 else [pts4, acts, mask]
        if views:
            d3 = dirs.detach().to(torch.float32).reshape(M, 3).contiguous()
            hv = torch.empty(M, W // 2, dtype=torch.float32, device=dev) if f32 else torch.empty(nblk, W // 2, 32, dtype=torch.bfloat16, device=dev)
            hvm = None if f32 else torch.empty(nblk, 64, W // 64, dtype=torch.int16, device=dev)
            a.dirs, a.hv, a.hv_mask = d3.data_ptr(), hv.data_ptr(), (None if f32 else hvm.data_ptr())
            saved += [d3, hv] if f32 else [d3, hv, hvm]
        if ray_bias is not None:        # time-conditioned baseline: W[:, latent columns] . latent of the two input layers, per ray
            rbias = ray_bias.detach().to(torch.float32).contiguous()
            assert tuple(rbias.shape) == (N, 2, W)
            a.ray_bias = rbias.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(model.lib.nrnerf_trunk_forward(model.handle, C.byref(a), _mstream(model, dev)), "nrnerf_trunk_forward")
        ctx.model, ctx.net, ctx.which, ctx.dims, ctx.views, ctx.tcb = model, net, int(which), (N, S, D, W, C_out), views, ray_bias is not None
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(raw)
        ctx.set_materialize_grads(False)             # an output nobody differentiates arrives as None, not as a zero-filled tensor
        return raw4.view(N, S, 4), raw.view(N, S, C_out)

    @staticmethod
    def backward(ctx, g_raw4, _g_raw):
        model, net = ctx.model, ctx.net
        f32 = _is_f32(model)
        pts4, acts = ctx.saved_tensors[:2]
        N, S, D, W, C_out = ctx.dims
        M, dev = N * S, pts4.device
        n_params = 2 * D + 2 + (5 if ctx.views else 0)
        if g_raw4 is None:
            return (None,) * (6 + n_params)
        g = g_raw4.contiguous().reshape(M, 4).float()
        d_pre = torch.empty_like(acts)
        d_pts4 = torch.empty(M, 4, dtype=torch.float32, device=dev)
        a = _lib.TrunkArgs()
        a.struct_size = C.sizeof(_lib.TrunkArgs)
        a.which, a.n_rays, a.n_samples = ctx.which, N, S
        a.pts4, a.acts, a.d_raw4, a.d_pre, a.d_pts4 = pts4.data_ptr(), acts.data_ptr(), g.data_ptr(), d_pre.data_ptr(), d_pts4.data_ptr()
        a.relu_mask = None if f32 else ctx.saved_tensors[2].data_ptr()
        colour, d_dirs = None, None
        if ctx.views:                   # the colour branch's saved arrays; gradient wrt the directions when somebody wants it
            d3, hv = ctx.saved_tensors[2 if f32 else 3], ctx.saved_tensors[3 if f32 else 4]
            d_pre_v = torch.empty_like(hv)
            a.dirs, a.hv, a.d_pre_v = d3.data_ptr(), hv.data_ptr(), d_pre_v.data_ptr()
            a.hv_mask = None if f32 else ctx.saved_tensors[5].data_ptr()
            if ctx.needs_input_grad[5]:
                d_dirs = torch.empty(M, 3, dtype=torch.float32, device=dev)
                a.d_dirs = d_dirs.data_ptr()
            colour = (d3, hv, d_pre_v)
        with torch.cuda.device(dev):
            _lib.check(model.lib.nrnerf_trunk_backward(model.handle, C.byref(a), _mstream(model, dev)), "nrnerf_trunk_backward")
        skip1 = int(list(net.skips)[0]) + 1
        g_bias = None
        if ctx.tcb:             # gradient wrt the per-ray biases: d_pre of the two input layers summed over the ray's samples
            if f32:
                g_bias = torch.stack([d_pre[0].view(N, S, W).sum(1), d_pre[skip1].view(N, S, W).sum(1)], 1)
            else:               # [block][feature][32 samples] tiles, zero in the padded columns
                bpr = (S + 31) // 32
                # (row sums over the 32 samples of every tile row by the library's own kernel, then the ray's blocks: torch's
                #  reduction over both axes of the array converted to fp32 cost 7 ms per 16 384-ray step, a GEMM against ones 13)
                rows = torch.empty(2, N * bpr * W, dtype=torch.float32, device=dev)
                with torch.cuda.device(dev):
                    for j, k in enumerate((0, skip1)):
                        _lib.check(_lib.load().nrnerf_tile_row_sums(d_pre[k].data_ptr(), N * bpr * W, rows[j].data_ptr(), _stream(dev)), "nrnerf_tile_row_sums")
                g_bias = rows.view(2, N, bpr, W).sum(2).permute(1, 0, 2)
        n_lat = int(net.pts_linears[0].weight.shape[1]) - (3 + 6 * ((int(net.input_ch) - 3) // 6)) if ctx.tcb else 0

        # every weight and bias gradient from ONE launch over the two saved arrays (nrnerf_trunk_wgrad: bf16 block tiles, or the
        # fp32 mode's rows); the time-conditioned baseline's latent columns are already in the flat buffer's layout, zero
        return (d_pts4.view(N, S, 4)[..., :3], None, None, None, g_bias, (d_dirs.view(N, S, 3) if d_dirs is not None else None),
                *_Trunk._weight_grads(model, net, ctx.dims, pts4, acts, d_pre, g, ctx.views, n_lat, colour))

    @staticmethod
    def _weight_grads(model, net, dims, pts4, acts, d_pre, g, views=False, n_lat=0, colour=None):
        """Every weight and bias gradient of the trunk from one call of nrnerf_trunk_wgrad over the two saved arrays (reads
        each once; a library route read them twice and reduced d_pre a third time): bf16 [block][feature][32 samples] tiles on
        the bf16 matrix pipe, or (fp32 mode) [sample][feature] rows on v_mfma_f32_32x32x2_f32.  One record of partial sums
        per workgroup, added here with one reduction."""
        N, S, D, W, C_out = dims
        dev = acts.device
        f32 = acts.dtype == torch.float32
        nblk = N * ((S + 31) // 32)
        # encoding / head-gradient operands the call fills: bf16 tiles, or fp32 rows [M][64]
        ns = 3 if views else 2          # (+ the direction encoding with the view-dependent head)
        scratch = torch.empty(ns, N * S, 64, dtype=torch.float32, device=dev) if f32 else torch.empty(ns, nblk, 64, 32, dtype=torch.bfloat16, device=dev)
        # records of partial sums; the launch has (D - 1) * kch + 3 * (10/16 or 12/16) kch workgroups: one per CU at most
        # (view-dependent head: one more job of kch and two more of 10/16 kch workgroups)
        short = 10 if W == 256 else 12
        kch = max(1, min(nblk, (_num_cus(dev) * 16) // ((D - (0 if views else 1)) * 16 + (5 if views else 3) * short)))
        # only the records the kernel does not write need zeroing: the three 64-column products (and their bias rows) are cut
        # into NRNERF_WGRAD_SHORT_PARTIALS <= kch partial sums (include/nrnerf.h); zero-filling the whole array was 74 MB per call
        parts = torch.empty(kch, _lib.wgrad_stride_views(D, W) if views else _lib.wgrad_stride(D, W), dtype=torch.float32, device=dev)
        # the three 64-column products (and their bias rows) are cut into NRNERF_WGRAD_SHORT_PARTIALS <= kch partial sums
        # (include/nrnerf.h); the other records' slots are never read (nrnerf_reduce_partials), so nothing is zero-filled
        kl = _lib.wgrad_short_partials(kch, W)
        a = _lib.WgradArgs()
        a.struct_size = C.sizeof(_lib.WgradArgs)
        a.n_rays, a.n_samples, a.n_partials = N, S, kch
        a.acts, a.d_pre, a.pts4, a.d_raw4 = acts.data_ptr(), d_pre.data_ptr(), pts4.data_ptr(), g.data_ptr()
        a.enc, a.g_head, a.partials = scratch[0].data_ptr(), scratch[1].data_ptr(), parts.data_ptr()
        # the head's bias gradient = column sums of d raw: per-block sums from the call's operand kernel, added up by the reduction below
        # (bf16 arrays, up to 16384 blocks; otherwise torch.sum -- a memset and a reduction)
        hs = None if (f32 or nblk > 16384) else torch.empty(nblk, 4, dtype=torch.float32, device=dev)
        if hs is not None:
            a.head_sums = hs.data_ptr()
        if views:
            d3, hv, d_pre_v = colour
            a.dirs, a.hv, a.d_pre_v, a.encv = d3.data_ptr(), hv.data_ptr(), d_pre_v.data_ptr(), scratch[2].data_ptr()
        with torch.cuda.device(dev):
            _lib.check(model.lib.nrnerf_trunk_wgrad(model.handle, C.byref(a), _stream(dev)), "nrnerf_trunk_wgrad")
        # every weight and bias, each in its own shape, back to back in one buffer: one launch
        index, shapes, hb = _trunk_grad_index(net, D, W, C_out, views, n_lat, dev, head_sums=hs is not None)
        if hs is not None:
            n_flat = int(index.shape[0])
            pos = (n_flat - 3, n_flat - 2, n_flat - 1, hb) if views else tuple(hb + c if c < C_out else -1 for c in range(4))
            return _split_flat(_reduce_partials(parts, kl, index, hs, pos), shapes)
        flat = _reduce_partials(parts, kl, index)
        if views:                                              # biases of alpha_linear and rgb_linear: column sums of d raw
            sums = g.sum(0)
            flat[hb:hb + 1].copy_(sums[3:4])
            flat[flat.shape[0] - 3:].copy_(sums[0:3])
        else:                                                  # (the 5th channel never reaches the loss: zero)
            torch.sum(g, 0, out=flat[hb:hb + 4])
        return _split_flat(flat, shapes)


class _GenericTrunk(torch.autograd.Function):
    """_Trunk for an architecture outside the compiled set (any depth, width % 4 == 0, at most one skip connection, plain
    output_linear head or the view-dependent one; fp32 or bf16): the run-time-parameterised kernel runs the forward with every
    activation saved and the backward-data pass from transposed weights (nrnerf_generic_trunk_forward / _backward,
    include/nrnerf.h); the weight gradients dW_i = d_pre_i^T x_i are library GEMMs over the two saved arrays and the gradients of
    the points / directions follow from their encodings'.  ``params``: _generic_trunk_params(net)."""

    @staticmethod
    def forward(ctx, pts, model, net, which, ray_bias, dirs, *params):
        N, S = int(pts.shape[0]), int(pts.shape[1])
        M, dev = N * S, pts.device
        D, W = int(net.D), int(net.W)
        f32 = _is_f32(model)
        views = bool(net.use_viewdirs)
        pts4 = _rows4(pts.detach(), M)
        # (view-dependent head: + feature_linear's outputs and the colour branch's activations, include/nrnerf.h)
        acts = torch.empty(D + (2 if views else 0), M, W, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
        C_out = 4 if views else int(net.output_linear.weight.shape[0])
        raw4 = torch.empty(M, 4, dtype=torch.float32, device=dev)
        raw = torch.empty(M, C_out, dtype=torch.float32, device=dev)
        a = _lib.GenericTrunkArgs()
        a.struct_size = C.sizeof(_lib.GenericTrunkArgs)
        a.which, a.n_rays, a.n_samples = int(which), N, S
        a.pts4, a.acts, a.raw4, a.raw, a.raw_ch = pts4.data_ptr(), acts.data_ptr(), raw4.data_ptr(), raw.data_ptr(), C_out
        saved = [pts4, acts]
        # (bf16, plain head: the forward also leaves one relu bit per activation, and the backward call that is handed them runs on the
        #  16x16x32 kernels' dataflow, csrc/nrnerf_gx16_bwd.h)
        nbits = int(model.lib.nrnerf_generic_trunk_bits_bytes(model.handle, int(which), N, S))
        bits = torch.empty(nbits, dtype=torch.uint8, device=dev) if nbits > 0 else None
        if bits is not None:
            a.relu_bits = bits.data_ptr()
        ctx.bits = bits
        if views:
            d3 = dirs.detach().to(torch.float32).reshape(M, 3).contiguous()
            a.dirs = d3.data_ptr()
            saved.append(d3)
        tcb = bool(getattr(net, "time_conditioned_baseline", False))
        if tcb:         # (``ray_bias`` carries the rays' latent codes [N, latent size] here: they are columns of the two input layers)
            codes = ray_bias.detach().to(torch.float32).contiguous()
            assert tuple(codes.shape) == (N, int(net.pts_linears[0].weight.shape[1]) - int(net.input_ch))
            a.latents = codes.data_ptr()
            saved.append(codes)
        ctx.tcb = tcb
        with torch.cuda.device(dev):
            _lib.check(model.lib.nrnerf_generic_trunk_forward(model.handle, C.byref(a), _mstream(model, dev)), "nrnerf_generic_trunk_forward")
        ctx.model, ctx.net, ctx.which, ctx.dims, ctx.views = model, net, int(which), (N, S, D, W, C_out), views
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(raw)
        ctx.set_materialize_grads(False)
        return raw4.view(N, S, 4), raw.view(N, S, C_out)

    @staticmethod
    def backward(ctx, g_raw4, _g_raw):
        model, net, views = ctx.model, ctx.net, ctx.views
        pts4, acts = ctx.saved_tensors[:2]
        N, S, D, W, C_out = ctx.dims
        M, dev = N * S, pts4.device
        if g_raw4 is None:
            return (None,) * (6 + 2 * D + (8 if views else 2))
        n_freqs = (int(net.input_ch) - 3) // 6
        n_enc = 3 + 6 * n_freqs
        n_lat = int(net.pts_linears[0].weight.shape[1]) - n_enc if ctx.tcb else 0
        skips = [int(k) for k in net.skips if 0 <= int(k) <= D - 2]
        g = g_raw4.contiguous().reshape(M, 4).float()
        d_pre = torch.empty_like(acts)
        d_enc = torch.empty(2 if skips else 1, M, n_enc + n_lat, dtype=torch.float32, device=dev)
        a = _lib.GenericTrunkArgs()
        a.struct_size = C.sizeof(_lib.GenericTrunkArgs)
        a.which, a.n_rays, a.n_samples = ctx.which, N, S
        a.acts, a.d_raw4, a.d_pre, a.d_enc0 = acts.data_ptr(), g.data_ptr(), d_pre.data_ptr(), d_enc[0].data_ptr()
        a.d_enc1 = d_enc[1].data_ptr() if skips else None
        if ctx.bits is not None:
            a.relu_bits = ctx.bits.data_ptr()
        if views:
            nv = (int(net.input_ch_views) - 3) // 6
            d_encv = torch.empty(M, 3 + 6 * nv, dtype=torch.float32, device=dev)
            a.d_encv = d_encv.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(model.lib.nrnerf_generic_trunk_backward(model.handle, C.byref(a), _mstream(model, dev)), "nrnerf_generic_trunk_backward")
        # the gradient wrt the points through the encoding's transposed Jacobian (both layers that read the encoding: added in the kernel),
        # and the encoding's rows once more as the first operand of those layers' weight gradients (nrnerf_encoding_*: one launch each)
        cdt = acts.dtype
        d_pts, d_codes = None, None
        if ctx.needs_input_grad[0]:
            d_pts = _encoding_backward(pts4, 4, n_freqs, d_enc[0], d_enc[1] if skips else None, n_enc + n_lat, M).view(N, S, 4)[..., :3]
        codes = ctx.saved_tensors[-1] if ctx.tcb else None
        n_in = n_enc + n_lat
        enc_c = _encoding_rows(pts4, 4, n_freqs, M, _pad8(n_in), cdt, codes, S)       # [M, n_in padded to whole 16-byte pieces]
        if ctx.tcb and ctx.needs_input_grad[4]:     # the codes' gradient: their columns of both layers, summed over the ray's samples
            d_codes = d_enc[:, :, n_enc:].sum(0).reshape(N, S, n_lat).sum(1)

        # every weight and bias gradient, dW = d_pre^T x and db = column sums of d_pre, as ONE call over the two saved arrays
        # (nrnerf_tn_products) straight into one buffer in _generic_trunk_params order
        half = int(net.views_linears[0].weight.shape[0]) if views else 0
        plan, shapes, off = _generic_grad_plan(D, W, n_in, skips, C_out, views, half, 3 + 6 * nv if views else 0)
        operands = {"g": g.to(cdt), "enc": enc_c}
        for i in range(D + (2 if views else 0)):
            operands[("d_pre", i)], operands[("acts", i)] = d_pre[i], acts[i]
        d_dirs = None
        if views:
            d3 = ctx.saved_tensors[2]
            n_dir = 3 + 6 * nv
            if ctx.needs_input_grad[5]:
                d_dirs = _encoding_backward(d3, 3, nv, d_encv, None, n_dir, M).view(N, S, 4)[..., :3]
            operands["encv"] = _encoding_rows(d3, 3, nv, M, _pad8(n_dir), cdt, None, 1)
        jobs = [(operands[a], ac, operands[b], bc, wo, wi, ldo, ow, ob) for a, ac, b, bc, wo, wi, ldo, ow, ob in plan]
        flat = _tn_products(jobs, M, off, dev)
        return (d_pts, None, None, None, d_codes, d_dirs, *_split_flat(flat, shapes))


def _generic_grad_plan(D, W, n_in, skips, C_out, views, half=0, n_dir=0):
    """The products that make up a non-compiled trunk's weight / bias gradients, as (a, first column of a, b, first column of b, wo, wi,
    ldo, offset of the weight, offset of the bias | None) with a / b naming the saved arrays -- ("d_pre", i), ("acts", i), "enc" (the
    network's input rows), "encv" (the directions' encoding), "g" (d raw [M, 4]) -- + the parameter shapes in _generic_trunk_params
    order + the length of the flat result.  out[offset + o * ldo + k] = sum_m a[m, col_a + o] b[m, col_b + k]; bias = column sums of a.
    Host-only logic (tests/test_training_host.py emulates it against torch autograd)."""
    plan, shapes, off = [], [], 0

    def param(shape):
        nonlocal off
        shapes.append(tuple(int(d) for d in shape))
        o, n = off, 1
        for d in shape:
            n *= int(d)
        off += n
        return o

    for i in range(D):              # pts_linears[i]: weight, bias (rnh:253-258: layer skip + 1 reads [encoding, activation])
        if i == 0:
            ow = param((W, n_in))
            plan.append([("d_pre", 0), 0, "enc", 0, W, n_in, n_in, ow, None])
        elif (i - 1) in skips:
            ow = param((W, n_in + W))
            plan.append([("d_pre", i), 0, "enc", 0, W, n_in, n_in + W, ow, None])
            plan.append([("d_pre", i), 0, ("acts", i - 1), 0, W, W, n_in + W, ow + n_in, None])
        else:
            ow = param((W, W))
            plan.append([("d_pre", i), 0, ("acts", i - 1), 0, W, W, W, ow, None])
        plan[-1][-1] = param((W,))
    if not views:
        ow = param((C_out, W))      # output_linear (a 5th channel never reaches a loss: its row and bias stay zero)
        plan.append(["g", 0, ("acts", D - 1), 0, 4, W, W, ow, param((C_out,))])
    else:
        # rnh:284-304: sigma = alpha_linear(h); feature = feature_linear(h); hv = relu(views_linears[0]([feature, enc(dir)])); rgb = rgb_linear(hv)
        # saved: acts[D] = feature, acts[D + 1][:, :half] = hv; d_pre[D] = d feature, d_pre[D + 1][:, :half] = d (views layer's pre-activation)
        ow = param((1, W))
        plan.append(["g", 3, ("acts", D - 1), 0, 1, W, W, ow, param((1,))])                                # alpha_linear
        ow = param((W, W))
        plan.append([("d_pre", D), 0, ("acts", D - 1), 0, W, W, W, ow, param((W,))])                       # feature_linear
        ow = param((half, W + n_dir))                                                                      # views_linears[0] on [feature, enc(dir)]
        plan.append([("d_pre", D + 1), 0, ("acts", D), 0, half, W, W + n_dir, ow, None])
        plan.append([("d_pre", D + 1), 0, "encv", 0, half, n_dir, W + n_dir, ow + W, param((half,))])
        ow = param((3, half))
        plan.append(["g", 0, ("acts", D + 1), 0, 3, half, half, ow, param((3,))])                          # rgb_linear
    return [tuple(p) for p in plan], shapes, off


def _pad8(n: int) -> int:
    return (int(n) + 7) // 8 * 8


def _encoding_rows(src, src_stride, n_freqs, M, cols, dtype, codes, rows_per_code):
    """Embedder.embed (rnh:120-150) of the rows' first three columns as [M, cols] rows of ``dtype`` (+ the rays' codes behind the encoding
    for the time-conditioned baseline, zero padding behind those): nrnerf_encoding_forward, one launch."""
    enc = torch.empty(M, cols, dtype=dtype, device=src.device)
    a = _lib.EncodingArgs()
    a.struct_size = C.sizeof(_lib.EncodingArgs)
    a.n_freqs, a.n_rows = int(n_freqs), int(M)
    a.src, a.src_stride = src.data_ptr(), int(src_stride)
    a.enc, a.enc_cols, a.enc_is_bf16 = enc.data_ptr(), int(cols), int(dtype == torch.bfloat16)
    if codes is not None:
        a.codes, a.n_lat, a.rows_per_code = codes.data_ptr(), int(codes.shape[1]), int(rows_per_code)
    with torch.cuda.device(src.device):
        _lib.check(_lib.load().nrnerf_encoding_forward(C.byref(a), _stream(src.device)), "nrnerf_encoding_forward")
    return enc


def _encoding_backward(src, src_stride, n_freqs, d_enc0, d_enc1, d_stride, M):
    """J^T (d_enc0 [+ d_enc1]) of that encoding: the gradient wrt the rows' first three columns as [M, 4] (column 3 zero), one launch."""
    out = torch.empty(M, 4, dtype=torch.float32, device=src.device)
    a = _lib.EncodingArgs()
    a.struct_size = C.sizeof(_lib.EncodingArgs)
    a.n_freqs, a.n_rows = int(n_freqs), int(M)
    a.src, a.src_stride = src.data_ptr(), int(src_stride)
    a.d_enc0, a.d_enc1, a.d_enc_stride = d_enc0.data_ptr(), (d_enc1.data_ptr() if d_enc1 is not None else None), int(d_stride)
    a.d_src, a.d_src_stride = out.data_ptr(), 4
    with torch.cuda.device(src.device):
        _lib.check(_lib.load().nrnerf_encoding_backward(C.byref(a), _stream(src.device)), "nrnerf_encoding_backward")
    return out


def _tn_products(jobs, n_rows, total, dev):
    """nrnerf_tn_products over ``jobs`` = (a tensor, first column of a, b tensor, first column of b, wo, wi, ldo, out offset, bias offset | None):
    out[offset + o * ldo + k] = sum_m a[m, col_a + o] b[m, col_b + k] (+ the column sums of a at the bias offset) -> the flat fp32 result."""
    lib = _lib.load()
    dt = jobs[0][0].dtype
    es = 2 if dt == torch.bfloat16 else 4
    arr = (_lib.TnJob * len(jobs))()
    per, copies = 16 // es, {}

    def operand(t, col, w):
        """(pointer, leading dimension) of columns [col, col + w) of `t` as the kernel wants them: 16-byte aligned rows padded to whole
        16-byte pieces -- the arrays the training kernels save qualify as they are whenever the width is a multiple of 8 (bf16) / 4
        (fp32); anything else (a 100-wide trunk, the [M, 4] head gradient, a column of it) is copied once into a padded buffer"""
        ptr, ld = t.data_ptr() + es * int(col), int(t.stride(0))
        if t.stride(1) == 1 and ptr % 16 == 0 and (ld * es) % 16 == 0 and (w + per - 1) // per * per <= ld - 0 and col + (w + per - 1) // per * per <= t.shape[1] + (ld - t.shape[1]):
            return ptr, ld
        key = (t.data_ptr(), int(col), int(w))
        if key not in copies:
            pad = torch.zeros(t.shape[0], (w + per - 1) // per * per, dtype=dt, device=t.device)
            pad[:, :w] = t[:, col:col + w]
            copies[key] = pad
        return copies[key].data_ptr(), int(copies[key].stride(0))

    for j, (a_t, a_col, b_t, b_col, wo, wi, ldo, ow, ob) in enumerate(jobs):
        assert a_t.dtype == dt and b_t.dtype == dt
        (pa, lda), (pb, ldb) = operand(a_t, a_col, wo), operand(b_t, b_col, wi)
        arr[j] = _lib.TnJob(pa, pb, lda, ldb, int(wo), int(wi), int(ldo), 0, int(ow), -1 if ob is None else int(ob))
    out = torch.empty(int(total), dtype=torch.float32, device=dev)
    a = _lib.TnArgs()
    a.struct_size = C.sizeof(_lib.TnArgs)
    a.n_jobs, a.is_bf16, a.n_rows, a.out_floats = len(jobs), int(dt == torch.bfloat16), int(n_rows), int(total)
    a.jobs, a.out = arr, out.data_ptr()
    with torch.cuda.device(dev):
        nbytes = int(lib.nrnerf_tn_workspace_bytes(C.byref(a)))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
        _lib.check(lib.nrnerf_tn_products(C.byref(a), _stream(dev)), "nrnerf_tn_products")
    return out


def _generic_trunk_params(net):
    """The parameters _GenericTrunk returns gradients for, in its order (the view-dependent head unfolded, unlike _colour_params)."""
    ps = []
    for lin in net.pts_linears:
        ps += [lin.weight, lin.bias]
    if not net.use_viewdirs:
        return ps + [net.output_linear.weight, net.output_linear.bias]
    return ps + [net.alpha_linear.weight, net.alpha_linear.bias, net.feature_linear.weight, net.feature_linear.bias,
                 net.views_linears[0].weight, net.views_linears[0].bias, net.rgb_linear.weight, net.rgb_linear.bias]


_NUM_CUS = {}


def _num_cus(dev) -> int:
    k = torch.device(dev).index or 0
    if k not in _NUM_CUS:
        _NUM_CUS[k] = int(torch.cuda.get_device_properties(k).multi_processor_count)
    return _NUM_CUS[k]


def _trunk_params(net):
    ps = []
    for lin in net.pts_linears:
        ps += [lin.weight, lin.bias]
    if not net.use_viewdirs:
        return ps + [net.output_linear.weight, net.output_linear.bias]
    return ps + [net.alpha_linear.weight, net.alpha_linear.bias] + _colour_params(net)


def _colour_params(net):
    """The colour branch of the view-dependent head as the kernels see it (rnh:286-303): no nonlinearity sits between
    feature_linear and views_linears[0], so they are ONE layer on [h, direction encoding],
        relu(Wv [Wf h + bf, enc] + bv) = relu((Wv1 Wf) h + Wv2 enc + (Wv1 bf + bv)),   Wv = [Wv1 | Wv2],
    and the library's weight images hold the folded matrix (the packer, or render._flat_params for the device-side re-pack,
    forms it).  The same two small products in PARAMETER space here, under autograd: the kernels return the gradient wrt
    the folded weights, autograd carries it back to Wf, bf, Wv and bv.  -> [Wv1 Wf, Wv1 bf + bv, Wv2, W_rgb, b_rgb]."""
    wf, bf = net.feature_linear.weight, net.feature_linear.bias
    wv, bv = net.views_linears[0].weight, net.views_linears[0].bias
    k1 = int(wf.shape[0])
    return [wv[:, :k1] @ wf, wv[:, :k1] @ bf + bv, wv[:, k1:], net.rgb_linear.weight, net.rgb_linear.bias]


def finite_difference_dirs(bent: torch.Tensor) -> torch.Tensor:
    """NeRF.viewdirs_via_finite_differences (run_nerf_helpers.py:316-356, "backward" differences) on bent points [N,S,3],
    under autograd: normalised p_j - p_{j-1} (eps added to the norm), sample 0 takes sample 1's direction."""
    diff = bent[:, 1:, :] - bent[:, :-1, :]
    diff = diff / (torch.norm(diff, dim=-1, keepdim=True) + 0.000001)
    out = torch.empty_like(bent)
    out[:, 1:, :] = diff
    out[:, 0, :] = diff[:, 0, :]
    return out


# True: the finite-difference view directions of the bent points and their encoding come from nrnerf_direction_encoding (one
# launch each way); False: from torch ops under autograd (finite_difference_dirs, posenc: ~40 launches per pass).
NATIVE_DIRECTION_ENCODING = True


class _DirectionEncoding(torch.autograd.Function):
    """[N*S, 3 + 6 L] encoding of the finite-difference view directions of bent points [N,S,3] (finite_difference_dirs +
    posenc: run_nerf_helpers.py:316-356, 120-150) in `dtype`, on the HIP library; gradient: the bent points."""

    @staticmethod
    def forward(ctx, bent, n_freqs, dtype):
        N, S = int(bent.shape[0]), int(bent.shape[1])
        dev = bent.device
        b4 = _rows4(bent.detach(), N * S)
        enc = torch.empty(N * S, 3 + 6 * int(n_freqs), dtype=dtype, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().nrnerf_direction_encoding(b4.data_ptr(), N, S, int(n_freqs), enc.data_ptr(), int(dtype == torch.bfloat16), None,
                                                            _stream(dev)), "nrnerf_direction_encoding")
        ctx.save_for_backward(b4)
        ctx.dims = (N, S, int(n_freqs))
        return enc

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (b4,) = ctx.saved_tensors
        N, S, L = ctx.dims
        g = g.contiguous()
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        out = torch.empty(N, S, 4, dtype=torch.float32, device=b4.device)
        with torch.cuda.device(b4.device):
            _lib.check(_lib.load().nrnerf_direction_encoding(b4.data_ptr(), N, S, L, g.data_ptr(), int(g.dtype == torch.bfloat16), out.data_ptr(),
                                                            _stream(b4.device)), "nrnerf_direction_encoding")
        return out[..., :3], None, None


class _Composite(torch.autograd.Function):
    """raw2outputs (train.py:724-789) of one pass, plus -- ``n_importance > 0`` -- the hierarchical sampling and merge
    (run_nerf_helpers.py:651-698, train.py:910-920) in the same launch.  Differentiable outputs: rgb_map, disp_map,
    acc_map, weights; alpha, the merged depths and z_std carry no gradient (the reference detaches the samples)."""

    @staticmethod
    def forward(ctx, raw4, rays, z, noise, white_bkgd, n_importance, u):
        N, S, I = int(raw4.shape[0]), int(raw4.shape[1]), int(n_importance)
        dev = raw4.device
        f32 = dict(dtype=torch.float32, device=dev)
        raw4 = raw4.detach().contiguous()
        rgb, disp, acc = torch.empty(N, 3, **f32), torch.empty(N, **f32), torch.empty(N, **f32)
        weights, alpha = torch.empty(N, S, **f32), torch.empty(N, S, **f32)
        z_std, z_merged = torch.empty(N if I > 0 else 0, **f32), torch.empty(N if I > 0 else 0, S + I, **f32)
        z_new = torch.empty(N if I > 0 else 0, I, **f32)
        rank_new = torch.empty(N if I > 0 else 0, I, dtype=torch.uint8, device=dev)
        a = _lib.CompositeArgs()
        a.struct_size = C.sizeof(_lib.CompositeArgs)
        a.n_rays, a.n_samples, a.n_importance = N, S, I
        a.rays, a.ray_stride, a.raw4, a.z = rays.data_ptr(), int(rays.shape[1]), raw4.data_ptr(), z.data_ptr()
        a.white_bkgd = int(bool(white_bkgd))
        if noise is not None:
            a.noise = noise.data_ptr()
        if u is not None:
            a.u = u.data_ptr()
        a.rgb, a.disp, a.acc, a.weights, a.alpha = rgb.data_ptr(), disp.data_ptr(), acc.data_ptr(), weights.data_ptr(), alpha.data_ptr()
        if I > 0:
            a.z_std, a.z_merged = z_std.data_ptr(), z_merged.data_ptr()
            if S + I <= SPLIT_MAX_SAMPLES:          # (8-bit ranks: beyond that the fine pass bends all merged samples again)
                a.z_new, a.rank_new = z_new.data_ptr(), rank_new.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(_lib.load().nrnerf_composite_forward(C.byref(a), _stream(dev)), "nrnerf_composite_forward")
        ctx.white_bkgd = bool(white_bkgd)
        ctx.noise = noise
        ctx.save_for_backward(raw4, rays, z)
        ctx.mark_non_differentiable(alpha, z_std, z_merged, z_new, rank_new)
        ctx.set_materialize_grads(False)
        return rgb, disp, acc, weights, alpha, z_merged, z_std, z_new, rank_new

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_w, _g_alpha, _g_zm, _g_zs, _g_zn=None, _g_rn=None):
        raw4, rays, z = ctx.saved_tensors
        N, S = int(raw4.shape[0]), int(raw4.shape[1])
        dev = raw4.device
        if g_rgb is None and g_disp is None and g_acc is None and g_w is None:
            return (None,) * 7
        d_raw4 = torch.empty(N, S, 4, dtype=torch.float32, device=dev)
        keep = [t.contiguous().float() if t is not None else None for t in (g_rgb, g_disp, g_acc, g_w)]
        if keep[0] is None:
            keep[0] = torch.zeros(N, 3, dtype=torch.float32, device=dev)
        a = _lib.CompositeArgs()
        a.struct_size = C.sizeof(_lib.CompositeArgs)
        a.n_rays, a.n_samples, a.n_importance = N, S, 0
        a.rays, a.ray_stride, a.raw4, a.z = rays.data_ptr(), int(rays.shape[1]), raw4.data_ptr(), z.data_ptr()
        a.white_bkgd = int(ctx.white_bkgd)
        if ctx.noise is not None:
            a.noise = ctx.noise.data_ptr()
        a.g_rgb = keep[0].data_ptr()
        for name, t in (("g_disp", keep[1]), ("g_acc", keep[2]), ("g_weights", keep[3])):
            if t is not None:
                setattr(a, name, t.data_ptr())
        a.d_raw4 = d_raw4.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(_lib.load().nrnerf_composite_backward(C.byref(a), _stream(dev)), "nrnerf_composite_backward")
        return d_raw4, None, None, None, None, None, None


class _Bender(torch.autograd.Function):
    """bent points [N,S,3], unmasked offsets [N,S,3], rigidity mask [N,S,1] = ray_bending(o + d z, latents)
    (run_nerf_helpers.py:507-577) on the HIP library, fp32.  Gradients: latents and the bender's parameters."""

    @staticmethod
    def forward(ctx, latents, model, rb, rays, z, token, share=None):
        """``share`` (a dict, or None): this evaluation will ALSO carry the divergence regulariser (training_loss: the term is taken at the
        coarse samples' points, train.py:248-262).  A fifth output -- a one-float handle -- then ties ``_DivergenceOnBender`` to this node:
        autograd calls this node's backward after the divergence's, and ONE nrnerf_bender_divergence_backward serves both uses (it takes
        the render pass' cotangents as well): no nrnerf_bender_backward / _wgrad launches of their own for the coarse samples.
        When the dict already holds the probe vectors ``e`` and the samples' points ``pts`` (training_loss with POOLED_DRAWS: the probes
        are drawn up front), the divergence FORWARD makes this evaluation too (nrnerf_divergence_args.bent4): no nrnerf_bender_forward
        launch for the coarse samples either."""
        N, S = int(z.shape[0]), int(z.shape[1])
        M, dev = N * S, z.device
        BD, BW = len(rb.network), int(rb.network[0].weight.shape[0])
        RD, RW = len(rb.rigidity_network), int(rb.rigidity_network[0].weight.shape[0])
        lat = latents.detach().to(torch.float32).contiguous()
        z = z.detach().contiguous()
        bent4 = torch.empty(M, 4, dtype=torch.float32, device=dev)
        off4 = torch.empty(M, 4, dtype=torch.float32, device=dev)
        ctx.fused = share is not None and share.get("e") is not None and share.get("pts") is not None and int(share["pts"].numel()) == 3 * M
        if ctx.fused:
            share.update(model=model, rb=rb, lat=lat, dims=(N, S))
            _divergence_forward(share, share["pts"].reshape(M, 3), share["e"], bent4=bent4, off4=off4)
            ctx.model, ctx.rb, ctx.dims, ctx.share = model, rb, (N, S, BD, BW, RD, RW), share
            ctx.save_for_backward(rays, lat, z, bent4, off4)
            ctx.set_materialize_grads(False)
            b = bent4.view(N, S, 4)
            return (b[..., :3], off4.view(N, S, 4)[..., :3], b[..., 3:4], bent4.view(N, S, 4)[..., :3], torch.empty(1, dtype=torch.float32, device=dev))
        # saved arrays: fp32 for an fp32 model, bf16 otherwise (include/nrnerf.h, nrnerf_bender_args): only the weight-gradient
        # kernel reads their values, and in that mode it rounds them to bf16 for the matrix pipe anyway
        sdt = torch.float32 if _is_f32(model) else torch.bfloat16
        acts_b = torch.empty(BD - 1, M, BW, dtype=sdt, device=dev)
        acts_r = torch.empty(RD - 1, M, RW, dtype=sdt, device=dev)
        a = _bender_args(rb, rays, lat, z, N, S, bent4, off4, acts_b, acts_r)
        with torch.cuda.device(dev):
            _lib.check(model.lib.nrnerf_bender_forward(model.handle, C.byref(a), _mstream(model, dev)), "nrnerf_bender_forward")
        ctx.model, ctx.rb, ctx.dims = model, rb, (N, S, BD, BW, RD, RW)
        ctx.save_for_backward(rays, lat, z, bent4, off4, acts_b, acts_r)
        ctx.set_materialize_grads(False)
        b = bent4.view(N, S, 4)
        # the bent points twice: a graph that reads them in two places (coarse trunk; rows of the merged samples, _merge_rows) takes one
        # output for each, so that the two gradients arrive separately and the kernel adds them (autograd's own add of two [.,4]-row
        # views makes a packed [N,S,3] tensor the kernel cannot take: an add, a zero-fill and a copy per step)
        ctx.share = share
        outs = (b[..., :3], off4.view(N, S, 4)[..., :3], b[..., 3:4], bent4.view(N, S, 4)[..., :3])
        if share is None:
            return outs
        share.update(model=model, rb=rb, lat=lat, dims=(N, S))
        return outs + (torch.empty(1, dtype=torch.float32, device=dev),)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_bent, g_unmasked, g_mask, g_bent_b=None, _g_handle=None):
        if ctx.fused:
            rays, lat, z, bent4, off4 = ctx.saved_tensors
            acts_b = acts_r = None
        else:
            rays, lat, z, bent4, off4, acts_b, acts_r = ctx.saved_tensors
        model, rb = ctx.model, ctx.rb
        N, S, BD, BW, RD, RW = ctx.dims
        M, dev, LAT = N * S, z.device, int(lat.shape[1])
        if g_bent is None:
            g_bent, g_bent_b = g_bent_b, None
        g4 = _rows4(g_bent, M) if g_bent is not None else torch.zeros(M, 4, dtype=torch.float32, device=dev)
        g4b = _rows4(g_bent_b, M) if g_bent_b is not None else None
        gu = g_unmasked.reshape(M, 3).float().contiguous() if g_unmasked is not None else None
        gm = g_mask.reshape(M).float().contiguous() if g_mask is not None else None
        share = getattr(ctx, "share", None)
        if share is not None and (share.get("g_div") is not None or ctx.fused):
            # the divergence term was taken at this evaluation's points: its backward (value + tangent chain) takes this node's cotangents too
            # (the saved arrays stay in `share` until the graph is freed: a second backward with retain_graph, train.py:1594-1604, finds them)
            g_div = share.pop("g_div", None)
            if g_div is None:            # (the divergence forward made this evaluation, but nobody differentiated the term itself)
                g_div = torch.zeros(M, dtype=torch.float32, device=dev)
            d_lat, flat = _divergence_backward(share, g_div, render=(g4 if g_bent is not None else None, g4b, gu, gm))
            return (d_lat.view(N, S, LAT).sum(1), None, None, None, None, flat if ctx.needs_input_grad[5] else None, None)
        dz_b, dz_r = torch.empty_like(acts_b), torch.empty_like(acts_r)
        dz_out4 = torch.empty(M, 4, dtype=torch.float32, device=dev)
        d_lat = torch.empty(M, LAT, dtype=torch.float32, device=dev)
        a = _bender_args(rb, rays, lat, z, N, S, bent4, off4, acts_b, acts_r)
        a.g_bent4 = g4.data_ptr()
        a.g_bent4_b = g4b.data_ptr() if g4b is not None else None
        a.g_unmasked_offsets = gu.data_ptr() if gu is not None else None
        a.g_rigidity_mask = gm.data_ptr() if gm is not None else None
        a.dz_offsets, a.dz_rigidity, a.dz_out4, a.d_latents = dz_b.data_ptr(), dz_r.data_ptr(), dz_out4.data_ptr(), d_lat.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(model.lib.nrnerf_bender_backward(model.handle, C.byref(a), _mstream(model, dev)), "nrnerf_bender_backward")
        if not ctx.needs_input_grad[5]:              # frozen bender (e.g. fitting test-time latent codes): no weight gradients
            return (d_lat.view(N, S, LAT).sum(1), None, None, None, None, None, None)
        # every weight / bias gradient of both MLPs in one launch (nrnerf_bender_wgrad): partial sums per wave, added here
        nparts = 4 * max(1, min(_num_cus(dev), (M + 1023) // 1024))
        nj = BD + RD
        parts = torch.empty(nparts, nj, _lib.BENDER_WGRAD_SLOT, dtype=torch.float32, device=dev)
        w = _lib.BenderWgradArgs()
        w.struct_size = C.sizeof(_lib.BenderWgradArgs)
        w.n_rays, w.n_samples, w.n_partials = N, S, nparts
        w.rays, w.ray_stride, w.latents, w.latent_stride, w.z = rays.data_ptr(), int(rays.stride(0)), lat.data_ptr(), int(lat.stride(0)), z.data_ptr()
        w.acts_offsets, w.acts_rigidity = acts_b.data_ptr(), acts_r.data_ptr()
        w.dz_offsets, w.dz_rigidity, w.dz_out4, w.partials = dz_b.data_ptr(), dz_r.data_ptr(), dz_out4.data_ptr(), parts.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(model.lib.nrnerf_bender_wgrad(model.handle, C.byref(w), _stream(dev)), "nrnerf_bender_wgrad")
        # added up into ONE flat gradient (the parameters' shapes back to back) for the bender's token
        index, _ = _bender_grad_index(rb, dev, divergence=False)
        return (d_lat.view(N, S, LAT).sum(1), None, None, None, None, _reduce_partials(parts.view(nparts, -1), nparts, index), None)


def _bender_args(rb, rays, lat, z, N, S, bent4, off4, acts_b, acts_r):
    a = _lib.BenderArgs()
    a.struct_size = C.sizeof(_lib.BenderArgs)
    a.n_rays, a.n_samples = N, S
    a.rays, a.ray_stride = rays.data_ptr(), int(rays.stride(0))
    a.latents, a.latent_stride = lat.data_ptr(), int(lat.stride(0))
    a.z = z.data_ptr()
    cutoff, scaling = getattr(rb, "rigidity_test_time_cutoff", None), getattr(rb, "test_time_scaling", None)
    if cutoff is not None:
        a.has_rigidity_cutoff, a.rigidity_cutoff = 1, float(cutoff)
    if scaling is not None:
        a.has_test_time_scaling, a.test_time_scaling = 1, float(scaling)
    a.bent4, a.off4, a.acts_offsets, a.acts_rigidity = bent4.data_ptr(), off4.data_ptr(), acts_b.data_ptr(), acts_r.data_ptr()
    return a


def _bender_params(rb):
    ps = []
    for lin in list(rb.network) + list(rb.rigidity_network):
        ps.append(lin.weight)
        if lin.bias is not None:
            ps.append(lin.bias)
    return ps


def bend_native(model, rb, rays, z, latents, details=True, masked=True, second=None, share=None):
    """As `bend`, on the HIP library: rays [N,>=6], z [N,S], latents [N,L] -> bent [N,S,3], dict of [N,S,.] tensors (``masked`` False:
    without masked_offsets, one launch less).  ``second``: a list that receives the second handle on the bent points (_Bender.forward);
    ``share``: _Bender.forward's dict (the divergence regulariser on this evaluation)."""
    outs = _Bender.apply(latents, model, rb, rays, z, _param_token(rb, _bender_params(rb)), share)
    bent, unmasked, mask, bent_b = outs[:4]
    if share is not None:
        share["handle"] = outs[4]
    if second is not None:
        second.append(bent_b)
    if not details:
        return bent, {}
    bd = dict(unmasked_offsets=unmasked, rigidity_mask=mask)
    if masked:
        m = mask * unmasked                                            # rnh:567
        scaling = getattr(rb, "test_time_scaling", None)
        if scaling is not None:
            m = m * scaling                                            # rnh:568-569
        bd["masked_offsets"] = m
    return bent, bd


# True: the fine pass bends only its N_importance new samples and re-uses the coarse pass' bent points (render_rays_train);
# False: it bends all S + I merged samples again, as the reference's graph does (same values, a third more bender work).
SPLIT_FINE_BENDER = True
SPLIT_MAX_SAMPLES = 256         # merged samples per ray up to which it applies (the new samples' ranks are 8-bit: include/nrnerf.h)


def _pack4(xyz: torch.Tensor, w, M: int) -> torch.Tensor:
    """[M,4] fp32 rows (xyz, w) for nrnerf_merge_rows.  What _Bender hands out -- xyz and w as the two parts of the same
    [M,4] rows -- is re-viewed; anything else is copied (w None: the 4th float is unspecified)."""
    n = int(xyz.shape[-2])
    rows_like = xyz.dtype == torch.float32 and xyz.dim() == 3 and xyz.stride() == (n * 4, 4, 1) and xyz.storage_offset() % 4 == 0 \
        and xyz.untyped_storage().nbytes() >= (xyz.storage_offset() + M * 4) * 4
    if rows_like and (w is None or (w.dtype == torch.float32 and w.stride() == (n * 4, 4, 1) and w.storage_offset() == xyz.storage_offset() + 3
                                    and w.untyped_storage().data_ptr() == xyz.untyped_storage().data_ptr())):
        return xyz.as_strided((M, 4), (4, 1), xyz.storage_offset())
    if w is None:
        return _rows4(xyz, M)
    return torch.cat([xyz.reshape(M, 3).float(), w.reshape(M, 1).float()], 1)


class _MergeRows(torch.autograd.Function):
    """(bent points, unmasked offsets, rigidity mask) of the S coarse samples and of the I importance samples -> the same in
    merged-depth order [N, S + I, .]: importance sample i goes to row rank_new[:, i] (nrnerf_composite_args.rank_new), the
    coarse samples fill the other rows in order.  One launch each way (nrnerf_merge_rows; a permutation, so the gradient is
    the inverse permutation), where gathers / where under autograd were ~45 launches per iteration."""

    @staticmethod
    def forward(ctx, bent_c, unm_c, mask_c, bent_n, unm_n, mask_n, rank_new):
        N, S, I = int(bent_c.shape[0]), int(bent_c.shape[1]), int(bent_n.shape[1])
        dev = bent_c.device
        det = unm_c is not None                      # with the bender's details, or (None there) the bent points only
        ca = _pack4(bent_c.detach(), mask_c.detach() if det else None, N * S)
        na = _pack4(bent_n.detach(), mask_n.detach() if det else None, N * I)
        cb, nb = (_pack4(unm_c.detach(), None, N * S), _pack4(unm_n.detach(), None, N * I)) if det else (None, None)
        ma = torch.empty(N, S + I, 4, dtype=torch.float32, device=dev)
        mb = torch.empty(N, S + I, 4, dtype=torch.float32, device=dev) if det else None
        rk = rank_new.contiguous()
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(dev):
            _lib.check(_lib.load().nrnerf_merge_rows(rk.data_ptr(), N, S, I, ca.data_ptr(), ptr(cb), na.data_ptr(), ptr(nb), ma.data_ptr(),
                                                     ptr(mb), 0, _stream(dev)), "nrnerf_merge_rows")
        ctx.dims = (N, S, I)
        ctx.save_for_backward(rk)
        ctx.set_materialize_grads(False)
        if not det:
            return ma[..., :3], None, None
        return ma[..., :3], mb[..., :3], ma[..., 3:4]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_bent, g_unm, g_mask):
        (rk,) = ctx.saved_tensors
        N, S, I = ctx.dims
        T, dev = S + I, rk.device
        if g_bent is None and g_unm is None and g_mask is None:
            return (None,) * 7
        f32 = dict(dtype=torch.float32, device=dev)
        if g_mask is None:           # the common case (the trunk's gradient wrt the bent points): its [.,4] rows as they are
            ga = _rows4(g_bent, N * T) if g_bent is not None else None
        else:
            ga = torch.cat([g_bent.reshape(N * T, 3).float() if g_bent is not None else torch.zeros(N * T, 3, **f32), g_mask.reshape(N * T, 1).float()], 1)
        gb = _rows4(g_unm, N * T) if g_unm is not None else None
        first, second = (ga, gb) if ga is not None else (gb, None)
        oc = [torch.empty(N, S, 4, **f32), torch.empty(N, S, 4, **f32) if second is not None else None]
        on = [torch.empty(N, I, 4, **f32), torch.empty(N, I, 4, **f32) if second is not None else None]
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(dev):
            _lib.check(_lib.load().nrnerf_merge_rows(rk.data_ptr(), N, S, I, ptr(oc[0]), ptr(oc[1]), ptr(on[0]), ptr(on[1]), ptr(first), ptr(second),
                                                     1, _stream(dev)), "nrnerf_merge_rows")
        (ca, cb), (na, nb) = ((oc[0], oc[1]), (on[0], on[1])) if ga is not None else ((None, oc[0]), (None, on[0]))
        part = lambda t, lo, hi: t[..., lo:hi] if t is not None else None
        return (part(ca, 0, 3) if g_bent is not None else None, part(cb, 0, 3), part(ca, 3, 4) if g_mask is not None else None,
                part(na, 0, 3) if g_bent is not None else None, part(nb, 0, 3), part(na, 3, 4) if g_mask is not None else None, None)


def _merge_rows(coarse_parts, new_parts, rank_new, rays, z_merged, scaling, detailed_output, want=lambda key: True):
    """(points, bent points, bender details) of the merged samples from those of the coarse and of the new samples; of the derived
    details only those ``want`` asks for."""
    (_, bent_c, bd_c), (_, bent_n, bd_n) = coarse_parts, new_parts
    if not bd_c:                                     # no details asked for: the bent points only
        bent, _, _ = _MergeRows.apply(bent_c, None, None, bent_n, None, None, rank_new)
        return None, bent, {}
    bent, unmasked, mask = _MergeRows.apply(bent_c, bd_c["unmasked_offsets"], bd_c["rigidity_mask"], bent_n, bd_n["unmasked_offsets"],
                                            bd_n["rigidity_mask"], rank_new)
    # train.py:921-923 (no gradient: rays, depths)
    pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * z_merged[:, :, None]) if want("fine_initial_input_pts") else None
    bd = dict(unmasked_offsets=unmasked, rigidity_mask=mask)
    if want("fine_masked_offsets"):
        masked = mask * unmasked                                                                 # rnh:567
        if scaling is not None:
            masked = masked * scaling                                                            # rnh:568-569
        bd["masked_offsets"] = masked
    return pts, bent, bd


# True: the ray bender runs on the HIP library (_Bender).  False: as torch ops on the module's parameters (`bend`).
NATIVE_BENDER = True

# False: the bender's layers run as plain F.linear (bit-identical bent points to the reference's ops; used by the gradient
# parity tests, because gradients through the 2^9 encoding frequency move by 1e-2 of their scale under a 1-ulp change of
# the bent points).  True: batched GEMMs over row blocks (see _linear_rows), 4 ms less per 1024-ray step.
BATCHED_BENDER = True


def _linear_rows(x, lin, B):
    """F.linear(x, W, b) for x [M, in] with M huge and in / out <= 64.  Written as a batched GEMM over B row blocks
    against the (stride-0) expanded weight: autograd then forms the weight gradient per block with one batched GEMM and
    adds the blocks (expand's backward), instead of one [out x M] x [M x in] GEMM that has a single output tile to work
    on.  Pure torch ops (double-differentiable)."""
    if B == 1:
        return F.linear(x, lin.weight, lin.bias)
    M = x.shape[0]
    y = torch.bmm(x.view(B, M // B, -1), lin.weight.t().unsqueeze(0).expand(B, -1, -1)).reshape(M, -1)
    return y if lin.bias is None else y + lin.bias


def bend(rb, pts, latents):
    """ray_bending.forward (run_nerf_helpers.py:507-577) on the module's own parameters, under autograd.
    pts [M,3], latents [M,L] -> bent points [M,3], dict(unmasked_offsets, rigidity_mask, masked_offsets)."""
    B = _chunks(pts.shape[0], 2048) if BATCHED_BENDER else 1
    h = torch.cat([pts, latents], -1)                                  # :525
    n = len(rb.network)
    for i, lin in enumerate(rb.network):
        h = _linear_rows(h, lin, B)                                    # :527
        if i != n - 1:
            h = F.relu(h)                                              # :533-536
    unmasked = h
    r = pts                                                            # :546
    n = len(rb.rigidity_network)
    for i, lin in enumerate(rb.rigidity_network):
        r = _linear_rows(r, lin, B)
        if i != n - 1:
            r = F.relu(r)
    mask = (torch.tanh(r) + 1) / 2                                     # :559-561
    cutoff = getattr(rb, "rigidity_test_time_cutoff", None)
    if cutoff is not None:
        mask = torch.where(mask <= cutoff, torch.zeros_like(mask), mask)   # :563-564
    masked = mask * unmasked                                           # :567
    scaling = getattr(rb, "test_time_scaling", None)
    if scaling is not None:
        masked = masked * scaling                                      # :568-569
    return pts + masked, dict(unmasked_offsets=unmasked, rigidity_mask=mask, masked_offsets=masked)


class _Divergence(torch.autograd.Function):
    """d [M] = e^T J e,  J = d(masked offsets)/d(point) of the ray bender, on the HIP library (nrnerf_bender_divergence_*):
    one forward-mode tangent through both MLPs instead of the reference's vector-Jacobian product with create_graph=True,
    and a backward pass over the value and the tangent chain instead of autograd's double backward.  Gradients: the
    per-point latent rows and the bender's parameters (the points are a leaf nobody reads in the reference)."""

    @staticmethod
    def forward(ctx, point_latents, model, rb, pts, e, token, tangent=False):
        """tangent: return J e itself, [M,3] -- the directional derivative of the masked offsets along e -- instead of e^T J e
        (exact view directions, rnh:358-385: the bent point's Jacobian applied to the ray direction is e + this)."""
        M, dev = int(pts.shape[0]), pts.device
        BD, BW = len(rb.network), int(rb.network[0].weight.shape[0])
        RD, RW = len(rb.rigidity_network), int(rb.rigidity_network[0].weight.shape[0])
        lat = point_latents.detach().to(torch.float32)
        if lat.stride(-1) != 1 or (lat.stride(0) != 0 and lat.stride(0) < lat.shape[1]):
            lat = lat.contiguous()
        pts = pts.detach().to(torch.float32).contiguous()
        e = e.detach().to(torch.float32).contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        div = torch.empty(M, **f32)
        off4, toff4 = torch.empty(M, 4, **f32), torch.empty(M, 4, **f32)
        sd = dict(dtype=torch.float32 if _is_f32(model) else torch.bfloat16, device=dev)      # saved arrays, as in _Bender
        acts_b, tacts_b = torch.empty(BD - 1, M, BW, **sd), torch.empty(BD - 1, M, BW, **sd)
        acts_r, tacts_r = torch.empty(RD - 1, M, RW, **sd), torch.empty(RD - 1, M, RW, **sd)
        a = _divergence_args(rb, pts, lat, e, div, off4, toff4, acts_b, tacts_b, acts_r, tacts_r)
        tvec = torch.empty(M, 3, **f32) if tangent else None
        if tangent:
            a.tangent = tvec.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(model.lib.nrnerf_bender_divergence_forward(model.handle, C.byref(a), _mstream(model, dev)), "nrnerf_bender_divergence_forward")
        ctx.model, ctx.rb, ctx.dims, ctx.tangent = model, rb, (M, BD, BW, RD, RW), bool(tangent)
        ctx.save_for_backward(pts, lat, e, div, off4, toff4, acts_b, tacts_b, acts_r, tacts_r)
        ctx.set_materialize_grads(False)
        return tvec if tangent else div

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_div):
        if g_div is None:
            return (None,) * 7
        pts, lat, e, div, off4, toff4, acts_b, tacts_b, acts_r, tacts_r = ctx.saved_tensors
        model, rb = ctx.model, ctx.rb
        M, BD, BW, RD, RW = ctx.dims
        dev, LAT = pts.device, int(lat.shape[1])
        f32 = dict(dtype=torch.float32, device=dev)
        g = g_div.contiguous().float()
        dz_b, dtz_b, dz_r, dtz_r = torch.empty_like(acts_b), torch.empty_like(acts_b), torch.empty_like(acts_r), torch.empty_like(acts_r)
        dz_out4, dtz_out4 = torch.empty(M, 4, **f32), torch.empty(M, 4, **f32)
        d_lat = torch.empty(M, LAT, **f32)
        nparts = 4 * max(1, min(_num_cus(dev), (M + 1023) // 1024))
        nj = BD + RD + 1
        parts = torch.empty(nparts, nj, _lib.BENDER_WGRAD_SLOT, **f32)
        a = _divergence_args(rb, pts, lat, e, div, off4, toff4, acts_b, tacts_b, acts_r, tacts_r)
        if ctx.tangent:
            a.g_tangent = g.data_ptr()          # [M,3]: gradient wrt the tangent vector
        else:
            a.g_divergence = g.data_ptr()
        a.dz_offsets, a.dtz_offsets, a.dz_rigidity, a.dtz_rigidity = dz_b.data_ptr(), dtz_b.data_ptr(), dz_r.data_ptr(), dtz_r.data_ptr()
        a.dz_out4, a.dtz_out4, a.d_latents = dz_out4.data_ptr(), dtz_out4.data_ptr(), d_lat.data_ptr()
        a.n_partials, a.partials = nparts, parts.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(model.lib.nrnerf_bender_divergence_backward(model.handle, C.byref(a), _mstream(model, dev)), "nrnerf_bender_divergence_backward")
        index, _ = _bender_grad_index(rb, dev, divergence=True)
        return (d_lat, None, None, None, None, _reduce_partials(parts.view(nparts, -1), nparts, index), None)


def _divergence_args(rb, pts, lat, e, div, off4, toff4, acts_b, tacts_b, acts_r, tacts_r):
    a = _lib.DivergenceArgs()
    a.struct_size = C.sizeof(_lib.DivergenceArgs)
    a.n_points = int(pts.shape[0])
    a.points, a.latents, a.latent_stride, a.probe = pts.data_ptr(), lat.data_ptr(), int(lat.stride(0)), e.data_ptr()
    cutoff, scaling = getattr(rb, "rigidity_test_time_cutoff", None), getattr(rb, "test_time_scaling", None)
    if cutoff is not None:
        a.has_rigidity_cutoff, a.rigidity_cutoff = 1, float(cutoff)
    if scaling is not None:
        a.has_test_time_scaling, a.test_time_scaling = 1, float(scaling)
    a.divergence, a.off4, a.toff4 = div.data_ptr(), off4.data_ptr(), toff4.data_ptr()
    a.acts_offsets, a.tacts_offsets, a.acts_rigidity, a.tacts_rigidity = acts_b.data_ptr(), tacts_b.data_ptr(), acts_r.data_ptr(), tacts_r.data_ptr()
    return a


def _divergence_forward(share, pts, e, bent4=None, off4=None):
    """nrnerf_bender_divergence_forward for the evaluation ``share`` describes (model, rb, lat [N, L] per ray, dims): fills ``share`` with the
    arrays its backward needs and the per-point values ``div``; with ``bent4`` it writes the render pass' rows as well."""
    model, rb, lat = share["model"], share["rb"], share["lat"]
    N, S = share["dims"]
    M, dev = int(pts.shape[0]), pts.device
    if M != N * S:
        raise ValueError("the divergence points are not the bender evaluation's points")
    BD, BW = len(rb.network), int(rb.network[0].weight.shape[0])
    RD, RW = len(rb.rigidity_network), int(rb.rigidity_network[0].weight.shape[0])
    lat_pts = lat[:, None, :].expand(N, S, lat.shape[1]).reshape(M, -1)                               # train.py:256-262
    pts = pts.detach().to(torch.float32).contiguous()
    e = e.detach().to(torch.float32).contiguous()
    f32 = dict(dtype=torch.float32, device=dev)
    div = torch.empty(M, **f32)
    off4 = torch.empty(M, 4, **f32) if off4 is None else off4
    toff4 = torch.empty(M, 4, **f32)
    sd = dict(dtype=torch.float32 if _is_f32(model) else torch.bfloat16, device=dev)
    acts_b, tacts_b = torch.empty(BD - 1, M, BW, **sd), torch.empty(BD - 1, M, BW, **sd)
    acts_r, tacts_r = torch.empty(RD - 1, M, RW, **sd), torch.empty(RD - 1, M, RW, **sd)
    a = _divergence_args(rb, pts, lat_pts, e, div, off4, toff4, acts_b, tacts_b, acts_r, tacts_r)
    if bent4 is not None:
        a.bent4 = bent4.data_ptr()
    with torch.cuda.device(dev):
        _lib.check(model.lib.nrnerf_bender_divergence_forward(model.handle, C.byref(a), _mstream(model, dev)), "nrnerf_bender_divergence_forward")
    share.update(pts=pts, lat_pts=lat_pts, e=e, div=div, off4=off4, toff4=toff4, acts_b=acts_b, tacts_b=tacts_b, acts_r=acts_r, tacts_r=tacts_r,
                 computed=True)
    return div


def _divergence_backward(saved, g, render=None):
    """nrnerf_bender_divergence_backward on the arrays a divergence forward saved (``saved``: model, rb, pts, lat_pts, e, div, off4, toff4,
    acts_b, tacts_b, acts_r, tacts_r) -> (d_latents [M, LAT] per point, the bender's flat parameter gradient).  ``render``: the cotangents
    (g_bent4 rows, second g_bent4 rows, g_unmasked [M,3], g_mask [M]; each may be None) of a render pass over the same evaluation."""
    model, rb = saved["model"], saved["rb"]
    pts, lat, e = saved["pts"], saved["lat_pts"], saved["e"]
    acts_b, tacts_b, acts_r, tacts_r = saved["acts_b"], saved["tacts_b"], saved["acts_r"], saved["tacts_r"]
    M, dev, LAT = int(pts.shape[0]), pts.device, int(lat.shape[1])
    BD, RD = len(rb.network), len(rb.rigidity_network)
    f32 = dict(dtype=torch.float32, device=dev)
    g = g.contiguous().float()
    dz_b, dtz_b, dz_r, dtz_r = torch.empty_like(acts_b), torch.empty_like(acts_b), torch.empty_like(acts_r), torch.empty_like(acts_r)
    dz_out4, dtz_out4 = torch.empty(M, 4, **f32), torch.empty(M, 4, **f32)
    d_lat = torch.empty(M, LAT, **f32)
    nparts = 4 * max(1, min(_num_cus(dev), (M + 1023) // 1024))
    parts = torch.empty(nparts, BD + RD + 1, _lib.BENDER_WGRAD_SLOT, **f32)
    a = _divergence_args(rb, pts, lat, e, saved["div"], saved["off4"], saved["toff4"], acts_b, tacts_b, acts_r, tacts_r)
    a.g_divergence = g.data_ptr()
    if render is not None:
        ptr = lambda t: None if t is None else t.data_ptr()
        a.render_g_bent4, a.render_g_bent4_b, a.render_g_unmasked_offsets, a.render_g_rigidity_mask = [ptr(t) for t in render]
    a.dz_offsets, a.dtz_offsets, a.dz_rigidity, a.dtz_rigidity = dz_b.data_ptr(), dtz_b.data_ptr(), dz_r.data_ptr(), dtz_r.data_ptr()
    a.dz_out4, a.dtz_out4, a.d_latents = dz_out4.data_ptr(), dtz_out4.data_ptr(), d_lat.data_ptr()
    a.n_partials, a.partials = nparts, parts.data_ptr()
    with torch.cuda.device(dev):
        _lib.check(model.lib.nrnerf_bender_divergence_backward(model.handle, C.byref(a), _mstream(model, dev)), "nrnerf_bender_divergence_backward")
    index, _ = _bender_grad_index(rb, dev, divergence=True)
    return d_lat, _reduce_partials(parts.view(nparts, -1), nparts, index)


class _DivergenceOnBender(torch.autograd.Function):
    """The divergence regulariser's per-point values e^T J e [M] for an evaluation of the bender that a ``_Bender`` node of the same graph
    made (``share`` = that node's dict, ``handle`` its fifth output): forward as ``_Divergence``; backward only PARKS the gradient -- the
    ``_Bender`` node's backward, which autograd runs afterwards (it waits for the handle's gradient), runs the one pass for both uses."""

    @staticmethod
    def forward(ctx, handle, share, pts, e):
        # (already made by the _Bender node when it had the probe vectors up front: its divergence forward WAS the bender evaluation)
        div = share["div"] if share.get("computed") else _divergence_forward(share, pts, e)
        ctx.share = share
        ctx.set_materialize_grads(False)
        return div

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_div):
        if g_div is None:
            return None, None, None, None
        ctx.share["g_div"] = g_div
        return torch.empty(1, dtype=torch.float32, device=g_div.device), None, None, None       # (the handle's "gradient": never read)


def why_no_native_divergence(ray_bender, input_points, point_latents):
    """None when compute_divergence_loss runs on the HIP library."""
    if ray_bender is None:
        return "no ray bender"
    if input_points.device.type != "cuda":
        return "points are not on a ROCm device"
    if input_points.dim() != 2 or input_points.shape[1] != 3 or point_latents.dim() != 2 or point_latents.shape[0] != input_points.shape[0]:
        return "unexpected shapes"
    if input_points.grad_fn is not None:
        return "the points are part of an autograd graph"          # the reference would raise when it sets requires_grad
    if R.model_of_bender(ray_bender, input_points.device) is None:
        return "no packed model with training kernels for this ray bender (render through the HIP path first)"
    return None


def _divergence_values(input_points, point_latents, ray_bender, exact, chunk, e=None):
    """Per point the (Hutchinson estimate ``e^T J e`` of the, or with ``exact`` the exact) divergence of the masked offsets field, [M], on the
    native kernels (the caller has checked why_no_native_divergence).  The probe vectors are drawn with the reference's own call per
    ``chunk`` of points (``torch.randn_like`` on a [chunk, 3] tensor, rnh:106)."""
    model = R.model_of_bender(ray_bender, input_points.device)
    M = int(input_points.shape[0])
    pts = input_points.detach()
    token = _param_token(ray_bender, _bender_params(ray_bender))
    if exact:                                            # divergence_exact (rnh:72-77): trace of J = sum_k unit_k^T J unit_k
        div = None
        for k in range(3):
            e = torch.zeros(M, 3, dtype=torch.float32, device=pts.device)
            e[:, k] = 1.0
            d = _Divergence.apply(point_latents, model, ray_bender, pts, e, token)
            div = d if div is None else div + d
        return div
    if e is None:
        e = torch.empty_like(pts)                    # divergence_approx (rnh:103-113), one draw per chunk as in rnh:52-59:
        for i in range(0, M, int(chunk)):            # randn_like(offsets) per chunk = empty_like().normal_(): same draws, no cat
            e[i:i + chunk, :].normal_()
    return _Divergence.apply(point_latents, model, ray_bender, pts, e, token)


class _FusedLoss(torch.autograd.Function):
    """training_wrapper_class.forward's loss terms (train.py:207-287) over the render outputs as nrnerf_loss_forward / _backward: one
    launch each instead of ~30 + ~50 eager torch launches of 2-5 us (a third of a graphed 1024-ray step's launches).  Differentiable
    inputs: rgb_map, rgb0, unmasked offsets, rigidity mask, divergence; the visibility weights and the opacity are constants, as the
    reference detaches them (train.py:223, rnh:65-66)."""

    @staticmethod
    def forward(ctx, rgb_map, rgb0, target, weights, offsets, rigidity, alpha, div, offsets_weight, rigidity_weight, divergence_weight, schedule=None,
                want_mean=False):
        """-> (per-ray loss [N], mean over the rays or None).  The mean (train.py:1594) is torch's reduction of the per-ray losses, but its
        gradient goes back into the kernel as the scalar it is (``loss.mean()`` under autograd: a division launch making a [N] tensor of g / N).
        (The mean inside the loss kernel -- last workgroup adds up -- was built and cost 23 us: its device-scope release writes back the
        dirty L2 lines of the whole die, as in nrnerf_optim.hip.)"""
        lib = _lib.load()
        dev = rgb_map.device
        f32 = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        rgb_map_, rgb0_, target_, weights_, alpha_, div_ = map(f32, (rgb_map, rgb0, target, weights, alpha, div))
        N = int(rgb_map_.shape[0])
        S = int(weights_.numel() // N) if weights_ is not None else (int(div_.numel() // N) if div_ is not None else 0)

        def rows(t, width):
            """(tensor, floats from one sample to the next): a [N,S,width] part of wider rows -- what render_rays hands out -- as it lies"""
            if t is None:
                return None, 0
            t = t.detach()
            if t.dtype == torch.float32 and t.dim() == 3 and int(t.shape[2]) == width and t.stride(2) == 1 and t.stride(1) >= width \
                    and t.stride(0) == t.stride(1) * int(t.shape[1]):
                return t, int(t.stride(1))
            return t.to(torch.float32).contiguous(), width
        offsets_, o_stride = rows(offsets, 3)
        rigidity_, r_stride = rows(rigidity if rigidity is None or rigidity.dim() == 3 else rigidity.reshape(N, S, 1), 1)
        loss = torch.empty(N, dtype=torch.float32, device=dev)

        a = _lib.LossArgs()
        a.struct_size = C.sizeof(_lib.LossArgs)
        a.n_rays, a.n_samples = N, S
        ptr = lambda t: None if t is None else t.data_ptr()
        a.rgb_map, a.rgb0, a.target = ptr(rgb_map_), ptr(rgb0_), ptr(target_)
        a.weights, a.offsets, a.rigidity, a.alpha, a.divergence = ptr(weights_), ptr(offsets_), ptr(rigidity_), ptr(alpha_), ptr(div_)
        a.offsets_stride, a.rigidity_stride = o_stride, r_stride
        a.offsets_weight, a.rigidity_weight, a.divergence_weight = float(offsets_weight), float(rigidity_weight), float(divergence_weight)
        # the schedule factor as a device scalar when the caller has it as a tensor (GraphedStep: a graph input)
        sched = None if schedule is None else schedule.detach().to(device=dev, dtype=torch.float32).reshape(1).contiguous()
        a.schedule = ptr(sched)
        a.loss = loss.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(lib.nrnerf_loss_forward(C.byref(a), _stream(dev)), "nrnerf_loss_forward")
        ctx.sched = sched
        ctx.save_for_backward(*[t for t in (rgb_map_, rgb0_, target_, weights_, offsets_, rigidity_, alpha_, div_) if t is not None])
        ctx.have = [t is not None for t in (rgb_map_, rgb0_, target_, weights_, offsets_, rigidity_, alpha_, div_)]
        ctx.scal = (N, S, float(offsets_weight), float(rigidity_weight), float(divergence_weight), o_stride, r_stride)
        ctx.shapes = (None if offsets is None else offsets.shape, None if rigidity is None else rigidity.shape, None if div is None else div.shape)
        ctx.set_materialize_grads(False)
        return loss, (loss.mean() if want_mean else None)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loss, g_mean=None):
        lib = _lib.load()
        it = iter(ctx.saved_tensors)
        rgb_map, rgb0, target, weights, offsets, rigidity, alpha, div = [next(it) if h else None for h in ctx.have]
        N, S, ow, rw, dw, o_stride, r_stride = ctx.scal
        dev = rgb_map.device
        if g_loss is None and g_mean is None:
            return (None,) * 13
        g_loss = None if g_loss is None else g_loss.to(torch.float32).contiguous()
        g_mean = None if g_mean is None else g_mean.to(torch.float32).reshape(1).contiguous()
        f32d = dict(dtype=torch.float32, device=dev)
        new = lambda t: None if t is None else torch.empty_like(t)
        g_map, g_0, g_div = new(rgb_map), new(rgb0), new(div)
        g_off = None if offsets is None else torch.empty(N, S, 3, **f32d)        # (packed, whatever the inputs' strides were)
        g_rig = None if rigidity is None else torch.empty(N, S, 1, **f32d)
        a = _lib.LossArgs()
        a.struct_size = C.sizeof(_lib.LossArgs)
        a.n_rays, a.n_samples = N, S
        ptr = lambda t: None if t is None else t.data_ptr()
        a.rgb_map, a.rgb0, a.target = ptr(rgb_map), ptr(rgb0), ptr(target)
        a.weights, a.offsets, a.rigidity, a.alpha, a.divergence = ptr(weights), ptr(offsets), ptr(rigidity), ptr(alpha), ptr(div)
        a.offsets_stride, a.rigidity_stride = o_stride, r_stride
        a.offsets_weight, a.rigidity_weight, a.divergence_weight = ow, rw, dw
        a.schedule = ptr(ctx.sched)
        a.g_loss, a.g_mean = ptr(g_loss), ptr(g_mean)
        a.g_rgb_map, a.g_rgb0, a.g_offsets, a.g_rigidity, a.g_divergence = ptr(g_map), ptr(g_0), ptr(g_off), ptr(g_rig), ptr(g_div)
        with torch.cuda.device(dev):
            _lib.check(lib.nrnerf_loss_backward(C.byref(a), _stream(dev)), "nrnerf_loss_backward")
        so, sr, sd = ctx.shapes
        return (g_map, g_0, None, None, None if g_off is None else g_off.view(so), None if g_rig is None else g_rig.view(sr), None,
                None if g_div is None else g_div.view(sd), None, None, None, None, None)


# training_loss: every uniform number of an iteration from ONE torch.rand call and every normal one from ONE torch.randn call (stratified
# jitter + sample_pdf's u; sigma noise of both passes + the divergence term's probe vectors) instead of the reference's six calls in the
# reference's order (train.py:860, 753, rnh:665, 753, 106 per chunk).  Same distributions, another position in the generator's stream: a
# seeded iteration no longer draws the numbers the reference draws, hence off by default (bench.py's train_step legs turn it on and say so).
POOLED_DRAWS = False


def _pooled_draws(rays_flat, kw, with_probe):
    """(the dict render._draw_randoms builds, the divergence term's probe vectors [N * N_samples, 3] or None) from two generator calls."""
    N, dev = int(rays_flat.shape[0]), rays_flat.device
    S, I = int(kw["N_samples"]), int(kw.get("N_importance", 0))
    perturb, std = kw.get("perturb", 0.0), kw.get("raw_noise_std", 0.0)
    stochastic_z, noisy = bool(perturb) and perturb > 0.0, bool(std) and std > 0.0
    out = {}
    if stochastic_z:
        u = torch.rand(N * (S + I), device=dev)
        out["u_coarse"] = u[:N * S].view(N, S)
        if I > 0:
            out["u_fine"] = u[N * S:].view(N, I)
    n_noise = N * (S + (S + I if I > 0 else 0)) if noisy else 0
    n_probe = N * S * 3 if with_probe else 0
    e = None
    if n_noise + n_probe:
        nrm = torch.randn(n_noise + n_probe, device=dev)
        if noisy:
            noise = nrm[:n_noise] if float(std) == 1.0 else nrm[:n_noise] * float(std)
            out["noise_coarse"] = noise[:N * S].view(N, S)
            if I > 0:
                out["noise_fine"] = noise[N * S:].view(N, S + I)
        if with_probe:
            e = nrm[n_noise:].view(N * S, 3)
    return out, e


# training_loss: the divergence regulariser and the coarse samples' bender evaluation share ONE backward pass (_DivergenceOnBender); False:
# two separate autograd nodes, as the reference's graph has them (same gradients up to the order of additions)
SHARED_DIVERGENCE = True

FUSED_LOSS = True        # training_loss: the loss terms as nrnerf_loss_forward / _backward (False: eager torch ops, the gradient-parity reference)


def compute_divergence_loss(offsets_of_inputs, input_points, point_latents, ray_bender, exact, chunk, N_rays, weights=None,
                            backprop_into_weights=True):
    """Signature and semantics of reference ``compute_divergence_loss`` (run_nerf_helpers.py:22-69), what
    training_wrapper_class.forward calls for the divergence regulariser (train.py:244-287): per point the (Hutchinson
    estimate ``e^T J e`` of the, or with ``exact`` the exact) divergence of the masked offsets field, absolute value,
    squared, weighted, mean per ray -> [N_rays].  ``offsets_of_inputs`` is unused, as in the reference (it re-evaluates the
    bender on ``input_points``).  The probe vectors are drawn with the reference's own call per ``chunk`` of points
    (``torch.randn_like`` on a [chunk, 3] tensor, rnh:106), so a seeded step consumes the generator like the reference."""
    why = why_no_native_divergence(ray_bender, input_points, point_latents)
    if why is not None:
        ref = R._fallbacks.get("compute_divergence_loss")
        if ref is None:
            raise R.Unsupported(f"no HIP kernel for this divergence call ({why}) and no reference function installed to defer to")
        R._note_fallback("compute_divergence_loss", why)
        return ref(offsets_of_inputs, input_points, point_latents, ray_bender, exact, chunk, N_rays, weights=weights,
                   backprop_into_weights=backprop_into_weights)
    input_points.requires_grad = True                                                        # rnh:39 (kept: callers may look at it)
    div = _divergence_values(input_points, point_latents, ray_bender, exact, chunk)
    divergence_loss = torch.abs(div)                                                         # rnh:61
    divergence_loss = divergence_loss ** 2                                                   # rnh:62
    if weights is not None:
        if not backprop_into_weights:
            weights = weights.detach()                                                       # rnh:65-66
        divergence_loss = weights * divergence_loss                                          # rnh:67
    return torch.mean(divergence_loss.view(N_rays, -1), dim=-1)                              # rnh:69


def _trains_on_compiled_kernels(net) -> bool:
    """Whether the compiled training kernels (csrc/nrnerf_train*.h) cover this network: 8 layers, skip behind layer 4, 10 frequencies,
    256 or 128 wide -- with the view-dependent head only 256 wide on 4 direction frequencies.  (The 128-wide trunk with that head RENDERS on
    compiled kernels but trains on the run-time-parameterised one: render_rays_train asks for a generic handle then.)"""
    views = bool(getattr(net, "use_viewdirs", False))
    return _is_default_shape(net) and (not views or (int(net.W) == 256 and int(getattr(net, "input_ch_views", 0)) == 27))


def _bender_has_compiled_shape(rb) -> bool:
    """The reference's hard-coded bender (rnh:406-407: 5 -- or 7 -- layers of 64, rigidity 3 x 32, latent 32): the shapes the bender's
    training kernels are compiled for, whatever the trunk (nrnerf_model_trains_bender)."""
    lins = [l for l in rb.network if hasattr(l, "weight")]
    rig = [l for l in rb.rigidity_network if hasattr(l, "weight")]
    return (len(lins) in (5, 7) and int(lins[0].weight.shape[1]) == 3 + 32 and all(int(l.weight.shape[0]) == 64 for l in lins[:-1])
            and len(rig) == 3 and int(rig[0].weight.shape[1]) == 3 and all(int(l.weight.shape[0]) == 32 for l in rig[:-1]))


def _is_default_shape(net) -> bool:
    return int(net.D) == 8 and int(net.W) in (256, 128) and list(net.skips) == [4] and int(net.input_ch) == 63


def _training_handle_flags(nets, rb) -> int:
    """nrnerf_model_desc.flags (+ Python-side markers) of the handle render_rays_train works on, beyond the environment's."""
    def renders_compiled_trains_generic(n):
        # a shape that RENDERS on compiled kernels without having compiled TRAINING kernels (the 128-wide trunk with the view-dependent head;
        # the time-conditioned baseline with another code size than 32 or on the 128-wide trunk): a generic handle of its own
        if not _is_default_shape(n):
            return False
        if getattr(n, "time_conditioned_baseline", False) and not (int(n.W) == 256 and int(n.pts_linears[0].weight.shape[1]) == int(n.input_ch) + 32):
            return True
        return not _trains_on_compiled_kernels(n)
    flags = _lib.MODEL_FORCE_GENERIC if any(renders_compiled_trains_generic(n) for n in nets) else 0
    if rb is not None and any(getattr(n, "use_viewdirs", False) and not getattr(n, "approx_nonrigid_viewdirs", True) and not _trains_on_compiled_kernels(n)
                              for n in nets):
        # exact Jacobian directions off the compiled set: computed by render_rays_train (the tangent of the divergence kernels); the handle is one
        # for the training entry points only (the generic RENDER kernel has no Jacobian directions, so the library would refuse the description)
        flags |= _lib.MODEL_PY_TRAINING_HANDLE
    return flags


def why_not_trainable(network_fn, network_fine, N_samples, N_importance, lindisp, pytest, ray_batch):
    """None when the native training path takes this call."""
    if pytest:
        return "pytest flag (numpy-seeded random numbers)"
    if ray_batch.device.type != "cuda":
        return "rays are not on a ROCm device"
    if ray_batch.requires_grad:
        # the kernels treat rays as data (sample positions carry no gradient): a caller who differentiates with respect to
        # the rays (camera refinement) must get the reference's autograd graph, not a silently missing gradient
        return "the ray batch requires a gradient"
    for net in (network_fn, network_fine if N_importance > 0 else None):
        if net is None:
            continue
        tcb = bool(getattr(net, "time_conditioned_baseline", False))
        if tcb and R._bender_of(network_fn) is not None:
            return "time-conditioned baseline with a bender under autograd"              # (train.py:574-576 rules it out as well)
        views = bool(getattr(net, "use_viewdirs", False))
        # (the time-conditioned baseline's compiled training kernels: the 256-wide trunk with a 32-float code)
        compiled = _trains_on_compiled_kernels(net) and (not tcb or (int(net.W) == 256 and int(net.pts_linears[0].weight.shape[1]) == int(net.input_ch) + 32))
        if views:
            has_bender = R._bender_of(network_fn) is not None
            exact = has_bender and not getattr(net, "approx_nonrigid_viewdirs", True)
            if exact and not compiled and not _bender_has_compiled_shape(R._bender_of(network_fn)):
                # (the Jacobian's tangent runs through the bender's compiled training kernels, which a generic handle carries for those shapes only)
                return "exact Jacobian view directions with a non-default ray bender under autograd"
            if (not has_bender or exact) and ray_batch.shape[-1] < 11:
                return "use_viewdirs without view directions in the ray batch"
        if not compiled:
            # outside the compiled set: the run-time-parameterised kernel trains the trunk, plain or view-dependent head
            # (_GenericTrunk; the library decides -- nrnerf_model_trains_generic -- and render_rays_train raises Unsupported when it says no)
            D, W = int(net.D), int(net.W)
            skips = [int(k) for k in net.skips if 0 <= int(k) <= D - 2]
            n_in = int(net.pts_linears[0].weight.shape[1])
            if (W % 4 or W > (480 if views else 512) or D < 1 or D > 16 or len(skips) > 1 or (int(net.input_ch) - 3) % 6
                    or (n_in != int(net.input_ch) if not tcb else not (int(net.input_ch) < n_in <= int(net.input_ch) + 64))):
                return "non-default trunk under autograd"
            if views and ((int(net.input_ch_views) - 3) % 6 or int(net.views_linears[0].weight.shape[1]) != W + int(net.input_ch_views)
                          or int(net.views_linears[0].weight.shape[0]) > W or int(net.feature_linear.weight.shape[0]) != W):
                return "view-dependent head of an unusual shape under autograd"
            if not views and int(net.output_linear.weight.shape[0]) not in (4, 5):
                return "non-default trunk under autograd"
    if N_samples < 2 or N_samples + N_importance > _lib.MAX_SAMPLES:
        return f"more than {_lib.MAX_SAMPLES} samples per ray"
    if R.get_precision() == "f16":
        return None            # trains in bf16 (see module docstring)
    return None


def render_rays_train(ray_batch, network_fn, N_samples, retraw=False, perturb=0.0, N_importance=0, network_fine=None,
                      white_bkgd=False, raw_noise_std=0.0, additional_pixel_information=None, detailed_output=False,
                      want_z_vals=False, lindisp=False, only_details=None, divergence_share=None, randoms=None):
    """reference render_rays (train.py:792-980) with autograd: same output dict, attached to the graph of the networks',
    the bender's and the latent codes' parameters.  ``only_details`` (not a reference argument; training_loss passes it): the
    ``detailed_output`` keys the caller is going to read -- the others that cost launches of their own (the sample points, the masked
    offsets) are then left out of the dict.  ``divergence_share`` (training_loss): a dict the COARSE samples' bender evaluation fills when
    it runs on the native training kernels (``_Bender.forward``), so that the divergence regulariser can ride on it.  ``randoms``
    (training_loss with POOLED_DRAWS): the dict ``render._draw_randoms`` would draw, drawn by the caller."""
    want = (lambda key: True) if only_details is None else (lambda key: key in only_details)
    dev = ray_batch.device
    precision = "bf16" if R.get_precision() == "f16" else R.get_precision()
    rb = R._bender_of(network_fn)
    latents = additional_pixel_information.get("ray_bending_latents") if additional_pixel_information else None
    nets = [network_fn] + ([network_fine] if (N_importance > 0 and network_fine is not None) else [])
    model = R.get_model(network_fn, network_fine if N_importance > 0 else None, precision=precision, device=dev, flags=_training_handle_flags(nets, rb))
    if model.generic and not model.trains_generic:
        raise R.Unsupported("this architecture has no training kernels (view-dependent head / time-conditioned baseline off the compiled set)")
    trunk = _GenericTrunk if model.generic else _Trunk
    rays = ray_batch.detach().to(torch.float32).contiguous()
    N, S, I = int(rays.shape[0]), int(N_samples), int(N_importance)
    if model.needs_latents:
        # the kernels read latent_size floats per ray and write a [N * S, latent_size] gradient: a wrong shape would be an
        # out-of-bounds access where the reference raises in expand / split (train.py:82-87, run_nerf_helpers.py:246)
        if latents is None:
            raise ValueError("ray_bending_latents are required with a ray bender")
        if latents.dim() != 2 or tuple(latents.shape) != (N, model.latent_size):
            raise ValueError(f"ray_bending_latents must have shape ({N}, {model.latent_size}), got {tuple(latents.shape)}")
    rays_o, rays_d = rays[:, 0:3], rays[:, 3:6]
    if I == 0 and detailed_output:
        raise UnboundLocalError("local variable 'visibility_weights_0' referenced before assignment "
                                "(reference render_rays cannot do detailed_output with N_importance == 0)")
    # random numbers in the reference's order (train.py:860, 753; run_nerf_helpers.py:665; 753)
    rnd = randoms if randoms is not None else (R._draw_randoms(rays, S, I, perturb, raw_noise_std) or {})
    # coarse depths (:847-868): linspace between near and far (or in inverse depth) and the stratified jitter, one launch -- which also
    # writes the coarse samples' points (:871-873) when the caller reads them (detailed_output's initial_input_pts)
    z_vals = torch.empty(N, S, dtype=torch.float32, device=dev)
    u_c = rnd.get("u_coarse")
    coarse_pts = torch.empty(N, S, 3, dtype=torch.float32, device=dev) if (detailed_output and want("initial_input_pts")) else None
    with torch.cuda.device(dev):
        if coarse_pts is not None:
            _lib.check(model.lib.nrnerf_sample_depths_points(rays.data_ptr(), int(rays.shape[1]), u_c.data_ptr() if u_c is not None else None, N, S,
                                                             int(bool(lindisp)), z_vals.data_ptr(), coarse_pts.data_ptr(), _stream(dev)),
                       "nrnerf_sample_depths_points")
        else:
            _lib.check(model.lib.nrnerf_sample_depths(rays.data_ptr(), int(rays.shape[1]), u_c.data_ptr() if u_c is not None else None, N, S,
                                                      int(bool(lindisp)), z_vals.data_ptr(), _stream(dev)), "nrnerf_sample_depths")

    def bend_samples(z, for_merge=False):
        """(points, bent points, bender details) of the samples at depths z [N, ns]; points only when something needs them
        (``for_merge``: the new samples of the split fine bender -- what _merge_rows takes, nothing derived)."""
        native = rb is not None and NATIVE_BENDER and model.trains_bender     # (a generic handle: the bender as library GEMMs, ``bend``)
        ns = int(z.shape[1])
        pts, bd = None, {}
        pre = "" if ns == S else "fine_"
        if for_merge and native:
            if latents is None:
                raise ValueError("ray_bending_latents are required with a ray bender")
            bent, unmasked, mask, _ = _Bender.apply(latents, model, rb, rays, z, _param_token(rb, _bender_params(rb)), None)
            return None, bent, (dict(unmasked_offsets=unmasked, rigidity_mask=mask) if detailed_output else {})
        if z is z_vals and coarse_pts is not None:
            pts = coarse_pts                                                                 # :871-873, from the depths' own launch
        elif (detailed_output and want(pre + "initial_input_pts")) or not native:
            pts = rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]                    # :871-873 / 921-923
        if rb is None:
            return pts, pts, bd
        if latents is None:
            raise ValueError("ray_bending_latents are required with a ray bender")
        if native:
            bent, bd = bend_native(model, rb, rays, z, latents, details=detailed_output, masked=want(pre + "masked_offsets"), second=second_handle,
                                   share=divergence_share if ns == S and not second_handle else None)
        else:
            lat = latents[:, None, :].expand(N, ns, latents.shape[-1]).reshape(N * ns, -1)   # train.py:79-87
            bent, bd = bend(rb, pts.reshape(-1, 3), lat.to(torch.float32))
            bent = bent.reshape(N, ns, 3)
        bd = {k: v.reshape(N, ns, -1) for k, v in bd.items()} if detailed_output else {}
        return pts, bent, bd

    def query(z, net, which, bent_parts=None):
        ns = int(z.shape[1])
        details = {}
        if bent_parts is None:
            pts, bent, bd = bend_samples(z)
        else:
            pts, bent, bd = bent_parts
        if detailed_output:
            if pts is not None:
                details["initial_input_pts"] = pts                                           # rnh:250-252
            details.update(bd)
            details["input_pts"] = bent                                                      # rnh:270
        ray_bias = None
        if getattr(net, "time_conditioned_baseline", False):
            # naive baseline (rnh:207-209, 273-282): the latent code is concatenated to the inputs of pts_linears[0] and of the
            # skip layer; constant along a ray, so its columns act as per-ray biases W[:, latent columns] . latent -- formed here
            # (library GEMMs under autograd: their backward yields the codes' and those columns' gradients), added in the kernel
            n_enc, lat = int(net.input_ch), latents.to(torch.float32)
            if model.generic:               # (_GenericTrunk: the codes themselves, as input columns)
                ray_bias = lat
            else:
                sk = int(list(net.skips)[0]) + 1
                ray_bias = torch.stack([F.linear(lat, net.pts_linears[0].weight[:, n_enc:n_enc + lat.shape[1]]),
                                        F.linear(lat, net.pts_linears[sk].weight[:, n_enc:n_enc + lat.shape[1]])], 1)
        dirs = None
        if net.use_viewdirs:
            # view-dependent head (rnh:284-304): the direction of every sample -- the finite differences of the bent points
            # (rnh:288-290, approx_nonrigid_viewdirs; one launch each way) or the ray's own (train.py:73-76); the head itself runs
            # behind the trunk in the same kernels
            if rb is not None and not getattr(net, "approx_nonrigid_viewdirs", True):
                # exact directions (rnh:291-294, 358-385): the bent point's Jacobian applied to the ray's unit direction, J d = d +
                # d(masked offsets)/dp . d -- one forward-mode tangent through the bender (the reference: three reverse passes with
                # create_graph=True), differentiated by the two-chain backward of the divergence kernels
                d_unit = rays[:, None, 8:11].expand(N, ns, 3).reshape(N * ns, 3)
                straight = (rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]).reshape(N * ns, 3)
                lat_pts = latents.view(N, 1, -1).expand(N, ns, latents.shape[-1]).reshape(N * ns, -1)
                jd = d_unit + _Divergence.apply(lat_pts, model, rb, straight, d_unit, _param_token(rb, _bender_params(rb)), True)
                dirs = (jd / torch.norm(jd, dim=-1, keepdim=True) + 0.000001).view(N, ns, 3)        # rnh:371-376: eps outside the division
            elif rb is not None and NATIVE_DIRECTION_ENCODING and bent.is_cuda and ns >= 2:
                dirs = _DirectionEncoding.apply(bent, 0, torch.float32).view(N, ns, 3)
            elif rb is not None:
                dirs = finite_difference_dirs(bent)
            else:
                dirs = rays[:, None, 8:11].expand(N, ns, 3)
        raw4, raw = trunk.apply(bent, model, net, which, ray_bias, dirs, *(_generic_trunk_params(net) if model.generic else _trunk_params(net)))
        return raw4, raw, details

    second_handle = []
    if divergence_share is not None and coarse_pts is not None:
        divergence_share["pts"] = coarse_pts
    coarse_parts = bend_samples(z_vals)
    raw4, raw, details = query(z_vals, network_fn, 0, coarse_parts)
    noise_c = rnd.get("noise_coarse")
    # :898, and -- same launch -- sample_pdf + merge (:910-920; no gradient: the reference detaches the samples)
    rgb_map, disp_map, acc_map, weights, alpha, z_merged, z_std, z_new, rank_new = _Composite.apply(raw4, rays, z_vals, noise_c, white_bkgd, I,
                                                                                                    rnd.get("u_fine"))
    ret = {}
    if I > 0:
        rgb0, disp0, acc0, weights0, alpha0 = rgb_map, disp_map, acc_map, weights, alpha     # :902-908
        net_f = network_fine if network_fine is not None else network_fn                     # :925
        fine_parts = None
        if rb is not None and SPLIT_FINE_BENDER and model.trains_bender and S + I <= SPLIT_MAX_SAMPLES:
            # The bender is shared by both networks (rnh:213-215) and the coarse depths are a subset of the merged depths
            # (:920): bend only the I new samples and put every sample's point / bent point / details at its row among the
            # merged depths (as nrnerf_render's split-bender path).  Same values as bending all S + I points again; the
            # coarse samples' bender evaluation now receives the gradient of both passes in ONE backward call.
            new_parts = bend_samples(z_new, for_merge=True)
            if second_handle:       # the coarse bent points' second output: its gradient reaches the bender's backward on its own (_Bender)
                coarse_parts = (coarse_parts[0], second_handle[0], coarse_parts[2])
            fine_parts = _merge_rows(coarse_parts, new_parts, rank_new, rays, z_merged, getattr(rb, "test_time_scaling", None), detailed_output, want)
        raw4, raw, fine_details = query(z_merged, net_f, 1 if network_fine is not None else 0, fine_parts)
        rgb_map, disp_map, acc_map, weights, alpha, _, _, _, _ = _Composite.apply(raw4, rays, z_merged, rnd.get("noise_fine"),
                                                                                  white_bkgd, 0, None)          # :943-950
    ret.update(rgb_map=rgb_map, disp_map=disp_map, acc_map=acc_map)                          # :952
    if retraw:
        ret["raw"] = raw                                                                     # :953-954
    if I > 0:
        ret.update(rgb0=rgb0, disp0=disp0, acc0=acc0, z_std=z_std)                           # :955-959
        if detailed_output:
            ret["fine_visibility_weights"] = weights                                         # :962
            ret["fine_opacity_alpha"] = alpha                                                # :964
            for k, v in fine_details.items():
                ret["fine_" + k] = v                                                         # :965-966
    if detailed_output:
        ret["visibility_weights"] = weights0                                                 # :969
        ret["opacity_alpha"] = alpha0                                                        # :970
        ret.update(details)                                                                  # :971-972
    if want_z_vals:
        ret["_z_vals"] = z_merged if I > 0 else z_vals          # not a reference key: the depths of the final pass
    return ret


# detailed_output keys training_loss reads (render_rays_train's ``only_details``)
_LOSS_DETAILS = frozenset(("initial_input_pts", "unmasked_offsets", "rigidity_mask", "visibility_weights", "opacity_alpha"))


def training_loss(rays_flat, ray_bending_latents, target_s, render_kwargs, *, offsets_loss_weight=0.0, divergence_loss_weight=0.0,
                  rigidity_loss_weight=0.0, global_step=0, N_iters=200000, chunk=1024 * 32, mean=False):
    """The loss of one training iteration as ``training_wrapper_class.forward`` builds it (train.py:190-287), on the
    drop-in entry points: ``batchify_rays`` (``retraw=True``; ``detailed_output`` when a regulariser is on, train.py:193-196),
    data term on the fine and the coarse image (:207-218), offsets + rigidity regulariser (:221-242) and divergence
    regulariser (:245-287, ``compute_divergence_loss``) with the increasing schedule.  Returns the per-ray loss [N_rays]
    (the training loop takes ``loss.mean()``, train.py:1594) and the render outputs.  ``render_kwargs`` = the reference's
    ``render_kwargs_train`` (network_fn, network_fine, N_samples, N_importance, perturb, raw_noise_std, ...).  A caller of
    the reference does not need this function -- its own ``training_wrapper_class`` lands on the same entry points through
    ``install()``; bench.py and the tests use it where the reference is not importable.  ``mean=True``: returns the mean over the rays
    (a scalar: what the training loop differentiates, train.py:1594) instead of the per-ray loss -- from the loss kernel's own launch."""
    N_rays = int(rays_flat.shape[0])
    ray_bender = R._bender_of(render_kwargs["network_fn"])
    detailed_output = offsets_loss_weight > 0.0 or divergence_loss_weight > 0.0                  # :193-196
    kw = {k: v for k, v in render_kwargs.items() if k not in ("retraw", "ray_bender", "near", "far", "ndc", "use_viewdirs")}
    if detailed_output:      # the detail keys the terms below read (the others that cost launches of their own are not produced)
        kw["_only_details"] = _LOSS_DETAILS
    pooled_e = None
    if POOLED_DRAWS and rays_flat.is_cuda and N_rays <= int(chunk):
        kw["_randoms"], pooled_e = _pooled_draws(rays_flat, kw, ray_bender is not None and divergence_loss_weight > 0.0 and FUSED_LOSS)
    share = None
    if SHARED_DIVERGENCE and FUSED_LOSS and ray_bender is not None and divergence_loss_weight > 0.0 and rays_flat.is_cuda and N_rays <= int(chunk):
        share = kw["_divergence_share"] = {}         # (one render_rays call: the coarse bender evaluation is the one the term is taken at)
        if pooled_e is not None:
            share["e"] = pooled_e                    # (probes known up front: the divergence forward can BE that evaluation, _Bender.forward)
    extras = R.batchify_rays(rays_flat, {"ray_bending_latents": ray_bending_latents}, chunk=chunk, detailed_output=detailed_output,
                             retraw=True, **kw)
    schedule = (1.0 / 100.0) ** (1 - (global_step / N_iters))                                    # :240, 285
    use_off = ray_bender is not None and offsets_loss_weight > 0.0
    use_div = ray_bender is not None and divergence_loss_weight > 0.0
    div_pts = div_lat = None
    riding = use_div and share is not None and "handle" in share      # (then the points and codes are the bender evaluation's own)
    if use_div:
        n_samples = int(extras["initial_input_pts"].shape[1])
        lat = ray_bending_latents
        if not riding:
            div_lat = lat.view(N_rays, 1, -1).expand((N_rays, n_samples, lat.shape[-1])).reshape(-1, lat.shape[-1])              # :256-262
        div_pts = extras["initial_input_pts"].view(-1, 3)
    if FUSED_LOSS and extras["rgb_map"].is_cuda and (not use_div or riding or why_no_native_divergence(ray_bender, div_pts, div_lat) is None):
        # the same terms as below, one launch forward and one backward (nrnerf_loss_forward / _backward)
        div = None
        if riding:
            # the term rides on the coarse samples' bender evaluation: one backward pass + one weight-gradient launch for both (_Bender)
            e = pooled_e
            if e is None:
                e = torch.empty_like(div_pts)
                for i in range(0, int(div_pts.shape[0]), int(chunk)):                            # randn_like per chunk, rnh:52-59, 106
                    e[i:i + chunk, :].normal_()
            div = _DivergenceOnBender.apply(share["handle"], share, div_pts, e)
        elif use_div:
            div_pts.requires_grad = True                                                         # rnh:39
            div = _divergence_values(div_pts, div_lat, ray_bender, False, chunk, e=pooled_e)
        loss, loss_mean = _FusedLoss.apply(extras["rgb_map"], extras.get("rgb0"), target_s,
                                extras["visibility_weights"] if use_off else None, extras["unmasked_offsets"] if use_off else None,
                                extras["rigidity_mask"] if use_off else None, extras["opacity_alpha"] if use_div else None, div,
                                *((offsets_loss_weight if use_off else 0.0, rigidity_loss_weight, divergence_loss_weight if use_div else 0.0, schedule)
                                  if torch.is_tensor(schedule) else
                                  (offsets_loss_weight * schedule if use_off else 0.0, rigidity_loss_weight,
                                   divergence_loss_weight * schedule if use_div else 0.0, None)), bool(mean))
        return (loss_mean if mean else loss), extras
    img2mse = lambda x, y: torch.mean(((x - y) ** 2).view(N_rays, -1), dim=1)                    # rnh:10-13
    loss = img2mse(extras["rgb_map"], target_s)                                                  # :207-212
    if "rgb0" in extras:
        loss = loss + img2mse(extras["rgb0"], target_s)                                          # :214-218
    if ray_bender is not None and offsets_loss_weight > 0.0:                                     # :221-242
        weights = extras["visibility_weights"].detach().view(-1)
        offsets_loss = torch.mean((weights * torch.pow(torch.norm(extras["unmasked_offsets"].view(-1, 3), dim=-1),
                                                       2.0 - extras["rigidity_mask"].view(-1))).view(N_rays, -1), dim=-1)
        offsets_loss = offsets_loss + rigidity_loss_weight * torch.mean((weights * extras["rigidity_mask"].view(-1)).view(N_rays, -1), dim=-1)
        loss = loss + offsets_loss_weight * schedule * offsets_loss
    if ray_bender is not None and divergence_loss_weight > 0.0:                                  # :245-287
        initial_input_pts = extras["initial_input_pts"].view(-1, 3)
        offsets = (extras["masked_offsets"] if "masked_offsets" in extras else extras["unmasked_offsets"]).view(-1, 3)
        weights = extras["opacity_alpha"].view(-1)
        n_samples = int(extras["initial_input_pts"].shape[1])
        lat = ray_bending_latents
        divergence_latents = lat.view(N_rays, 1, -1).expand((N_rays, n_samples, lat.shape[-1])).reshape(-1, lat.shape[-1])   # :256-262
        weights = 1.0 - torch.exp(-F.relu(weights))                                              # :264
        divergence_loss = compute_divergence_loss(offsets, initial_input_pts, divergence_latents, ray_bender, exact=False, chunk=chunk,
                                                  N_rays=N_rays, weights=weights, backprop_into_weights=False)
        loss = loss + divergence_loss_weight * schedule * divergence_loss
    return (loss.mean() if mean else loss), extras


class _SelectCodes(torch.autograd.Function):
    """codes[index]: the per-ray latent codes of a batch (training_wrapper_class.forward, train.py:173-188, indexes the stacked
    per-frame codes by each ray's time step).  Backward as ONE small GEMM, onehot(index)^T g -- deterministic, and 0.9 ms less
    per 16384-ray step than the sort-based accumulation autograd's indexing backward runs for 16384 rows landing on 8."""

    @staticmethod
    def forward(ctx, codes, index):
        ctx.save_for_backward(index)
        ctx.n = int(codes.shape[0])
        return codes.index_select(0, index)

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        if g.is_cuda and g.dtype == torch.float32 and index.dtype == torch.int64 and int(g.shape[1]) <= 256:
            # one launch, rays added in order (nrnerf_code_gradients); the one-hot GEMM below is a fill, a scatter and a GEMM
            g = g.contiguous()
            out = torch.empty(ctx.n, int(g.shape[1]), dtype=torch.float32, device=g.device)
            with torch.cuda.device(g.device):
                _lib.check(_lib.load().nrnerf_code_gradients(index.contiguous().data_ptr(), g.data_ptr(), int(g.shape[0]), int(g.shape[1]), ctx.n,
                                                            out.data_ptr(), _stream(g.device)), "nrnerf_code_gradients")
            return out, None
        onehot = torch.zeros(ctx.n, int(index.shape[0]), dtype=g.dtype, device=g.device)
        onehot.scatter_(0, index[None, :], 1.0)
        return onehot @ g, None


def select_codes(codes: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """codes[index] with the cheap backward above while the one-hot matrix stays small; plain indexing otherwise."""
    if codes.dim() != 2 or index.dim() != 1 or int(codes.shape[0]) * int(index.shape[0]) > (1 << 26):
        return codes[index]
    return _SelectCodes.apply(codes, index)


class FusedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam(params, lr, betas, eps)`` -- what train.py:655-658 builds over ``grad_vars`` -- whose ``step()`` AND the device-side
    re-pack of the networks' packed weights the next forward needs are ONE call = two launches (``nrnerf_adam_step``, csrc/nrnerf_optim.hip)
    instead of torch's three multi-tensor launches + the copy of the parameters into the library's flat vector + the re-pack launch (144 us
    of a 1.8 ms step at N_rand = 1024).

        opt = FusedAdam(grad_vars, lr=5e-4, betas=(0.9, 0.999), networks=(network_fn, network_fine))

    ``networks``: the (coarse, fine) modules whose handle the step refreshes.  Their parameters (and their ray bender's) are RE-HOMED at
    construction: ``p.data`` becomes a view of ONE fp32 vector in the library's canonical order (``nrnerf_model_update_device``), so the
    kernel updates the parameters in place and packs straight from them -- modules, ``state_dict()`` and checkpoints see ordinary
    tensors.  Without ``networks`` (or for networks with the view-dependent head, whose packed images need two derived matrices) the step
    is the fused Adam alone and the next forward re-packs as before.  No weight decay / amsgrad (the reference uses neither).  The state
    has torch's keys (``step``, ``exp_avg``, ``exp_avg_sq``), so ``state_dict()`` round-trips with ``torch.optim.Adam`` checkpoints
    (train.py:1680-1698 saves the optimiser's).  ``param_group["lr"]`` is read every step (train.py:1625-1630 decays it on the host); a
    0-dim device tensor there is passed as a device scalar (a schedule inside ``GraphedStep``).  Capturable: the step count lives on the
    device."""

    repacks_weights = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, networks=None):
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        self._lib = _lib.load()
        allp = [p for g in self.param_groups for p in g["params"]]
        if not allp or any(p.device.type != "cuda" or p.dtype != torch.float32 or p.device != allp[0].device for p in allp):
            raise ValueError("FusedAdam: fp32 parameters on one ROCm device")
        self._dev = allp[0].device
        self._networks = None
        self._flat = None
        homed = {}
        if networks is not None:
            nf, nfine = (networks if isinstance(networks, (tuple, list)) else (networks, None))
            mods = R._linears_in_canonical_order(nf, nfine)
            slots = []
            for lin in mods:
                slots.append(lin.weight)
                if getattr(lin, "bias", None) is not None:
                    slots.append(lin.bias)
            views = any(getattr(n, "use_viewdirs", False) for n in (nf, nfine) if n is not None)
            if not views and all(t.device == self._dev and t.dtype == torch.float32 for t in slots):
                with torch.no_grad():
                    flat = torch.empty(sum(t.numel() for t in slots), dtype=torch.float32, device=self._dev)
                    o = 0
                    for t in slots:
                        v = flat[o:o + t.numel()].view(t.shape)
                        v.copy_(t)
                        t.data = v                    # the module's parameter now lives in the flat vector
                        homed[t] = o
                        o += t.numel()
                self._flat, self._networks = flat, (nf, nfine)
                self.repacks_weights = True
        # state: torch's keys; exp_avg / exp_avg_sq of the re-homed parameters are views of two vectors parallel to the flat one
        self._step = torch.zeros((), dtype=torch.float32, device=self._dev)
        self._barrier = torch.zeros(2, dtype=torch.int32, device=self._dev)
        m_flat = torch.zeros_like(self._flat) if self._flat is not None else None
        v_flat = torch.zeros_like(self._flat) if self._flat is not None else None
        self._homed = homed
        for p in allp:
            if p in homed:
                o = homed[p]
                st = dict(step=self._step, exp_avg=m_flat[o:o + p.numel()].view(p.shape), exp_avg_sq=v_flat[o:o + p.numel()].view(p.shape))
            else:
                st = dict(step=self._step, exp_avg=torch.zeros_like(p, memory_format=torch.contiguous_format),
                          exp_avg_sq=torch.zeros_like(p, memory_format=torch.contiguous_format))
            self.state[p] = st
        # parameter order of a step: the re-homed ones by their offset (runs merge), then the others
        self._order = sorted([p for p in allp if p in homed], key=lambda p: homed[p]) + [p for p in allp if p not in homed]

    def load_state_dict(self, state_dict):
        """torch's checkpoints load INTO the flat state (the loaded tensors are copied, the views stay)."""
        keep = {p: dict(st) for p, st in self.state.items()}
        super().load_state_dict(state_dict)
        with torch.no_grad():
            step = None
            for p, st in list(self.state.items()):
                old = keep.get(p)
                if old is None:
                    continue
                for k in ("exp_avg", "exp_avg_sq"):
                    if k in st and st[k] is not old[k]:
                        old[k].copy_(st[k])
                if "step" in st and st["step"] is not old["step"]:
                    step = float(st["step"])
                self.state[p] = old
            if step is not None:
                self._step.fill_(step)

    def _model(self):
        nf, nfine = self._networks
        prec = "bf16" if R.get_precision() == "f16" else R.get_precision()
        nets = [nf] + ([nfine] if nfine is not None else [])
        return R.get_model(nf, nfine, precision=prec, device=self._dev, flags=_training_handle_flags(nets, R._bender_of(nf)))

    def _nrnerf_after_step(self):
        if self.repacks_weights and getattr(self, "_stepped_model", None) is not None:
            R.note_repacked(self._networks[0], self._stepped_model)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = self._lib
        model = None
        if self.repacks_weights:
            try:
                model = self._model()
            except R.Unsupported:
                model = None
        self._stepped_model = model
        groups = [g for g in self.param_groups if any(p.grad is not None for p in g["params"])]
        with torch.cuda.device(self._dev):
            stream = _stream(self._dev)
            for gi, group in enumerate(groups):
                members = set(group["params"])
                segs = []
                keep_alive = []
                for p in self._order:
                    if p not in members or p.grad is None:
                        continue
                    g = p.grad
                    if g.is_sparse:
                        raise RuntimeError("FusedAdam does not support sparse gradients")
                    if g.dtype != torch.float32 or not g.is_contiguous():
                        g = g.to(torch.float32).contiguous()
                        keep_alive.append(g)
                    st = self.state[p]
                    cur = [p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()]
                    if segs and all(segs[-1][k] + 4 * segs[-1][4] == cur[k] for k in range(4)):
                        segs[-1][4] += cur[4]              # the run goes on: one segment
                    else:
                        segs.append(cur)
                last_group = gi == len(groups) - 1
                self.last_segments = len(segs)
                for c0 in range(0, max(len(segs), 1), _lib.ADAM_MAX_SEGMENTS):
                    chunk = segs[c0:c0 + _lib.ADAM_MAX_SEGMENTS]
                    last = last_group and c0 + _lib.ADAM_MAX_SEGMENTS >= len(segs)
                    a = _lib.AdamArgs()
                    a.struct_size = C.sizeof(_lib.AdamArgs)
                    a.n_segments = len(chunk)
                    for i, (pp, gp, mp, vp, n) in enumerate(chunk):
                        a.segments[i] = _lib.AdamSegment(pp, gp, mp, vp, n)
                    lr = group["lr"]
                    if torch.is_tensor(lr):
                        lr_t = lr.detach().to(device=self._dev, dtype=torch.float32).reshape(())
                        keep_alive.append(lr_t)
                        a.lr_device, a.lr = lr_t.data_ptr(), 0.0
                    else:
                        a.lr = float(lr)
                    a.beta1, a.beta2, a.eps = float(group["betas"][0]), float(group["betas"][1]), float(group["eps"])
                    # (several launches of one step -- more than ADAM_MAX_SEGMENTS runs, or several groups -- share the step count: only
                    #  the last one advances it, the others work on a copy)
                    step_t = self._step if last else self._step.clone()
                    keep_alive.append(step_t)
                    a.step = step_t.data_ptr()
                    a.barrier = self._barrier.data_ptr()
                    handle = None
                    if last and model is not None:
                        a.flat_params, a.n_floats = self._flat.data_ptr(), self._flat.numel()
                        handle = model.handle
                    _lib.check(lib.nrnerf_adam_step(handle, C.byref(a), stream), "nrnerf_adam_step")
            if model is not None:
                model.note_use(self._dev)
        return loss


class GraphedStep:
    """One whole training iteration -- device-side weight re-pack, forward, loss, backward, optimiser step -- captured in
    a HIP graph and replayed: the ~150 launches of a 1024-ray step (N_rand of the shipped config) cost no host time and no
    launch gaps.  The iteration the reference runs per step (train.py:1543-1610) is launch-bound at that batch size; its
    Python loop can adopt this with the changes INTEGRATION.md lists (static input tensors, ``capturable=True`` optimiser).

        graphed = GraphedStep(step_fn, inputs, optimizer, [network_fn])      # step_fn(**inputs) -> scalar loss; it must NOT
        loss = graphed(rays=..., target=..., ...)                            # call backward / optimizer.step itself

    ``inputs``: dict of example tensors; the same keys are accepted per call and copied into the static buffers
    (shapes fixed).  Anything that changes per step must be one of them -- e.g. the regularisers' schedule as a 0-dim tensor
    (``training_loss(global_step=tensor)``).  Random numbers are drawn inside the graph from torch's generator (graph-safe
    Philox offsets), so a replayed step draws fresh numbers.  ``networks``: the ``network_fn`` modules whose packed weights
    the step reads: their handles are re-packed from the parameters at the START of every replay (the captured refresh);
    after each replay they are marked stale (the replayed optimiser step does not touch the parameters' version counters),
    so a render outside the graph re-packs first."""

    def __init__(self, step_fn, inputs, optimizer, networks, warmup=3):
        self.networks = list(networks)
        self.static = {k: v.clone() for k, v in inputs.items()}
        self.optimizer = optimizer
        dev = next(iter(self.static.values())).device

        # (an optimiser whose step re-packs the handles itself -- FusedAdam(networks=...) -- leaves them fresh at the end of every
        #  step, captured or replayed: no refresh at the start of the step, nothing to mark stale after a replay)
        self.repacks = bool(getattr(optimizer, "repacks_weights", False))
        self._one = None

        def one():
            if not self.repacks:
                for nf in self.networks:
                    R.mark_stale(nf)                               # the refresh kernels are part of the captured step
            loss = step_fn(**self.static)
            if self._one is None or self._one.shape != loss.shape:
                self._one = torch.ones_like(loss)
            loss.backward(self._one)                               # (a fixed d loss / d loss: backward() fills a fresh one per step)
            optimizer.step()
            return loss

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.enable_grad():         # warm-up on a side stream, as torch's graph recipe asks
            for _ in range(warmup):
                optimizer.zero_grad(set_to_none=True)
                one()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        for nf in networks:                                        # the warm-up's render events / training stream are complete: nothing
            R.forget_streams(nf)                                   # recorded outside the capture may be waited for inside it
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)                      # .grad tensors are allocated from the graph's pool
        with torch.enable_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss = one()

    def __call__(self, **inputs):
        """Replays the step on `inputs` (keys and shapes of the example inputs; a key that is left out KEEPS the data of the
        previous replay -- deliberate: a fixed ray batch is passed once).  Returns the graph's STATIC loss tensor: the next
        replay overwrites it -- ``.clone()`` it to keep a value across steps."""
        unknown = inputs.keys() - self.static.keys()
        if unknown:
            raise KeyError(f"GraphedStep: unknown inputs {sorted(unknown)}; captured with {sorted(self.static)}")
        for k, v in inputs.items():
            if tuple(v.shape) != tuple(self.static[k].shape):
                raise ValueError(f"GraphedStep: input {k!r} has shape {tuple(v.shape)}, captured with {tuple(self.static[k].shape)}")
            self.static[k].copy_(v)
        self.graph.replay()
        self.sync()          # host-only: the replayed optimiser step bumped no version counter, so a render / evaluation
        return self.loss     # outside the graph would otherwise read the weights packed at the START of this replay

    def sync(self):
        """Mark the networks' packed weights stale (done after every replay; kept as a public no-cost call)."""
        if self.repacks:
            return
        for nf in self.networks:
            dev = next(nf.parameters()).device
            R.mark_stale(nf, torch.cuda.current_stream(dev) if dev.type == "cuda" else None)


# configs/example_sequence.txt:14-16, 26-28, 35 -- the recipe the reference ships
SHIPPED_RECIPE = dict(N_samples=64, N_importance=64, N_rand=1024, perturb=1.0, raw_noise_std=1.0, offsets_loss_weight=60.0,
                      divergence_loss_weight=3.0, rigidity_loss_weight=0.0005, N_iters=200000, chunk=32768)


def _fresh_training_modules(cfg, dev, n_importance):
    """Modules at the reference's initialisation (create_nerf train.py:595-630; ray_bending.__init__ rnh:436-455, 487-505:
    kaiming hidden layers, zero biases, zero last layers) -- the synthetic stress weights of the inference benchmark would
    die (sigma < 0 everywhere, zero gradients) after one Adam step."""
    from .modules import NeRFWeights, RayBenderWeights
    rb = RayBenderWeights(cfg.latent_size, cfg.bend_hidden, cfg.bend_depth, cfg.rigidity_hidden, cfg.rigidity_depth) if cfg.ray_bending else None
    if rb is not None:
        with torch.no_grad():
            for net in (rb.network, rb.rigidity_network):
                for layer in list(net)[:-1]:
                    torch.nn.init.kaiming_uniform_(layer.weight, a=0, mode="fan_in", nonlinearity="relu")
                    torch.nn.init.zeros_(layer.bias)
                net[-1].weight.zero_()
                if net[-1].bias is not None:
                    net[-1].bias.zero_()
    out_ch = 5 if n_importance > 0 else 4                                                   # train.py:593
    mk = lambda ns: NeRFWeights(D=cfg.netdepth, W=cfg.netwidth, input_ch=cfg.input_ch, input_ch_views=cfg.input_ch_views, output_ch=out_ch,
                                skips=cfg.skips, use_viewdirs=cfg.use_viewdirs, ray_bending_latent_size=cfg.latent_size, num_ray_samples=ns,
                                approx_nonrigid_viewdirs=cfg.approx_nonrigid_viewdirs, time_conditioned_baseline=cfg.time_conditioned_baseline)
    coarse, fine = mk(cfg.N_samples), (mk(cfg.N_samples + n_importance) if n_importance > 0 else None)
    for m in (rb, coarse, fine):
        if m is not None:
            m.to(dev)
    coarse.ray_bender = (rb,)
    if fine is not None:
        fine.ray_bender = (rb,)
    return rb, coarse, fine


def _time_training(cfg, dev, precision, n_rays, n_importance, steps, warmup, regularised, graph=False, repeats=1, torch_adam=False):
    import time

    from .synthetic import make_rays
    rec = SHIPPED_RECIPE
    rb, coarse, fine = _fresh_training_modules(cfg, dev, n_importance)
    params = []
    for m in (rb, coarse, fine):
        if m is not None:
            m.requires_grad_(True)
            params += list(m.parameters())
    codes = torch.zeros(8, cfg.latent_size, device=dev, requires_grad=True)
    # train.py:655-658: Adam over grad_vars -- as FusedAdam (the step and the weight re-pack in one launch) unless `torch_adam`
    if torch_adam:
        opt = torch.optim.Adam(params + [codes], lr=5e-4, betas=(0.9, 0.999), fused=True, capturable=bool(graph))
    else:
        opt = FusedAdam(params + [codes], lr=5e-4, betas=(0.9, 0.999), networks=(coarse, fine))
    rays, _ = make_rays(n_rays, 5, cfg)
    rays = rays.to(dev)
    frame = torch.randint(0, 8, (n_rays,), device=dev)
    target = 0.5 + 0.4 * torch.sin(3.0 * rays[:, 3:6])                     # a smooth colour field of the ray direction
    kw = dict(network_fn=coarse, network_fine=fine, network_query_fn=None, N_samples=cfg.N_samples, N_importance=n_importance,
              perturb=rec["perturb"], raw_noise_std=rec["raw_noise_std"])
    weights = dict(offsets_loss_weight=rec["offsets_loss_weight"], divergence_loss_weight=rec["divergence_loss_weight"],
                   rigidity_loss_weight=rec["rigidity_loss_weight"]) if (regularised and rb is not None) else {}
    state = {"i": 0}

    def loss_of(rays, target, frame, global_step):
        loss, _ = training_loss(rays, select_codes(codes, frame), target, kw, global_step=global_step, N_iters=rec["N_iters"], chunk=rec["chunk"],
                                mean=True, **weights)                        # train.py:1594: loss.mean()
        return loss

    def step():
        opt.zero_grad(set_to_none=True)
        loss = loss_of(rays, target, frame, state["i"])
        loss.backward()
        opt.step()
        state["i"] += 1
        return loss

    prev = R.get_precision()
    R.set_precision(precision)
    global POOLED_DRAWS
    prev_pool, POOLED_DRAWS = POOLED_DRAWS, True       # (two generator calls per iteration instead of six; see POOLED_DRAWS)
    try:
        if graph:
            gstep = torch.zeros((), device=dev)
            graphed = GraphedStep(loss_of, dict(rays=rays, target=target, frame=frame, global_step=gstep), opt, [coarse], warmup=max(warmup, 2))

            def step():                                                     # noqa: F811  (the replayed step)
                state["i"] += 1
                gstep.fill_(float(state["i"]))
                return graphed(global_step=gstep)
            warmup = 2
        with torch.enable_grad():
            for _ in range(warmup):
                step()
            dt = float("inf")
            for _ in range(max(1, repeats)):                # eager steps are host-bound at 1024 rays: best of `repeats` timed loops
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(steps):
                    loss = step()
                torch.cuda.synchronize(dev)
                dt = min(dt, (time.perf_counter() - t0) / steps)
    finally:
        R.set_precision(prev)
        POOLED_DRAWS = prev_pool
    return dt, float(loss.detach())


def bench_train_step(scene, cfg, dev, precision="bf16", n_rays=None, steps=30, warmup=5):
    """bench.py's ``train_step`` leg: the reference's training iteration (train.py:1543-1642) with the recipe it ships
    (configs/example_sequence.txt): N_rand = 1024 rays, 64 + 64 samples, perturb, raw_noise_std = 1, detailed outputs, loss =
    data term (fine + coarse) + 60 x offsets / rigidity regulariser + 3 x divergence regulariser (increasing schedule), then
    backward, Adam step and the device-side weight refresh the next forward needs -- everything through the drop-in entry
    points (training_loss: batchify_rays under autograd, compute_divergence_loss), no eager reference-module call.
    ``data_term_only`` is last round's lighter step (64 + 128 samples, data term only) for continuity."""
    rec = SHIPPED_RECIPE
    n_rays = n_rays or rec["N_rand"]
    # (each leg: the best of three timed loops of `steps` iterations -- an eager 1024-ray step is bound by the host's launch
    #  rate, which another process' threads on the same host move by a factor of two)
    dt, final = _time_training(cfg, dev, precision, n_rays, rec["N_importance"], steps, warmup, regularised=True, repeats=3)
    dt0, final0 = _time_training(cfg, dev, precision, n_rays, 128, steps, warmup, regularised=False, repeats=3)
    peak = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}[precision]
    S, I = cfg.N_samples, rec["N_importance"]

    def roof(dt_, n_imp, with_div):
        samples = n_rays * (2 * S + n_imp)                                  # coarse pass + fine pass network evaluations
        trunk_macs = 63 * 256 + 4 * 256 * 256 + 319 * 256 + 2 * 256 * 256 + 256 * 5
        bend_macs = 15872                                                   # offset + rigidity MLPs, SURVEY.md section 8d
        flops = 3.0 * 2.0 * (trunk_macs + (bend_macs if cfg.ray_bending else 0)) * samples   # forward + backward-data + backward-weights
        if with_div and cfg.ray_bending:
            flops += 2.0 * 3.0 * 2.0 * bend_macs * n_rays * S               # value + tangent chain of the divergence term
        # HBM bytes the designed dataflow must move per sample: every hidden activation and every pre-activation gradient
        # written once and read once by the weight-gradient kernels (bf16 mode 2 B, fp32 mode 4 B per value), the bender's
        # arrays in fp32
        eb = 4 if precision == "f32" else 2
        per_sample = 4 * 8 * 256 * eb + (4 * (4 * 64 + 2 * 32) * 4 if cfg.ray_bending else 0)
        hbm = per_sample * samples
        return {"mfma": {"achieved": round(flops / dt_ / 1e12, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(flops / dt_ / 1e12 / peak, 4)},
                "hbm": {"achieved": round(hbm / dt_ / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(hbm / dt_ / 8e12, 4),
                        "bytes_per_sample_by_design": per_sample}}
    try:        # the same iteration replayed from a HIP graph (GraphedStep): no host time, no launch gaps
        dtg, finalg = _time_training(cfg, dev, precision, n_rays, rec["N_importance"], steps, warmup, regularised=True, graph=True, repeats=3)
        graph = {"rays_per_s": round(n_rays / dtg, 1), "ms_per_step": round(dtg * 1e3, 3), "final_loss": round(finalg, 5),
                 "what": "the same iteration (forward, loss, backward, Adam + weight re-pack as one launch) captured once in a HIP graph and replayed (training.GraphedStep)",
                 "roofline": roof(dtg, I, True)}
        try:        # round 5's optimiser for continuity: torch.optim.Adam(fused, capturable) + re-pack at the start of the step
            dtt, _ = _time_training(cfg, dev, precision, n_rays, rec["N_importance"], steps, warmup, regularised=True, graph=True, repeats=3, torch_adam=True)
            graph["with_torch_adam_ms_per_step"] = round(dtt * 1e3, 3)
        except Exception as e:
            graph["with_torch_adam_error"] = f"{type(e).__name__}: {str(e)[:200]}"
    except Exception as e:                                                  # capture is best effort: report, do not fail the bench
        graph = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
    launches = None
    try:        # device launches (kernels + copies) of ONE eager iteration: torch.profiler's device events of 3 steps minus those of 1, halved
        from torch.profiler import ProfilerActivity, profile
        counts = []
        for k in (1, 3):
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                _time_training(cfg, dev, precision, n_rays, rec["N_importance"], k, 2, regularised=True)
            counts.append(sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA))
        launches = (counts[1] - counts[0]) // 2
    except Exception:                                                       # (best effort: the tracer is not part of the measurement)
        launches = None
    r = roof(dt, I, True)
    eager = {"rays_per_s": round(n_rays / dt, 1), "ms_per_step": round(dt * 1e3, 3), "final_loss": round(final, 5),
             "what": "the same iteration as an eager Python loop: host-bound (about forty launches and the autograd graph's Python per 1.5 ms of "
                     "kernels), so its time follows the host's single-thread speed, not the kernels'",
             "roofline": {"bound": "hbm", **r["hbm"], "mfma": r["mfma"]}}
    graphed = "ms_per_step" in graph
    # the step as this library runs it = replayed from ONE HIP graph (training.GraphedStep); the eager loop beside it.  (Through round 5 the
    # top-level figures of this record were the eager loop's and the graph's sat under "hip_graph" -- that key stays, with the same numbers.)
    top_dt = graph["ms_per_step"] / 1e3 if graphed else dt
    rt = roof(top_dt, I, True)
    return {"rays_per_s": round(n_rays / top_dt, 1), "ms_per_step": round(top_dt * 1e3, 3), "rays_per_step": n_rays,
            "timed": ("the iteration replayed from one HIP graph (training.GraphedStep): forward, loss, backward, Adam + weight re-pack; the eager "
                      "Python loop is under 'eager_loop'") if graphed else "the eager Python loop (the HIP-graph capture failed: see 'hip_graph')",
            "device_launches_per_step": launches,
            "samples_per_ray": f"{S}+{I}", "dtype": precision, "final_loss": round(graph.get("final_loss", final), 5),
            "what": "the reference's training iteration with its shipped recipe (configs/example_sequence.txt): render under autograd with "
                    "detailed outputs (perturb, raw_noise_std 1), loss = mse(rgb_map) + mse(rgb0) + 60 x (offsets + 5e-4 rigidity) "
                    "regulariser + 3 x divergence regulariser (native second-order path) with the increasing schedule, backward, training.FusedAdam ("
                    "Adam step + device-side weight re-pack as two launches); all through render.batchify_rays / training.training_loss. "
                    "The iteration's random numbers (stratified jitter, sample_pdf's u, sigma noise of both passes, the divergence term's probe "
                    "vectors) come from ONE torch.rand and ONE torch.randn call (training.POOLED_DRAWS: same distributions, not the reference's six "
                    "calls in the reference's order). Every leg: best of three timed loops",
            "loss_terms": ["mse(rgb_map)", "mse(rgb0)", "offsets", "rigidity", "divergence"],
            "hip_graph": graph,
            "eager_loop": eager,
            "roofline": {"bound": "hbm", **rt["hbm"], "mfma": rt["mfma"],
                         "note": "algorithmic work (3 x forward flops of trunk + bender, + the divergence chains; saved arrays written "
                                 "once and read once) over the whole step's wall time, incl. optimiser, small loss ops and launch overheads"},
            "data_term_only": {"rays_per_s": round(n_rays / dt0, 1), "ms_per_step": round(dt0 * 1e3, 3), "samples_per_ray": f"{S}+128",
                               "final_loss": round(final0, 5), "what": "round 2's step: mse(rgb_map) + mse(rgb0) only, no detailed outputs",
                               "roofline": roof(dt0, 128, False)}}
