"""CPU checks of the host logic in libnrnerf_hip.so: symbol export, descriptor validation and the
MFMA-fragment weight packer (``nrnerf_pack_host``).

The packer is checked by *emulating the kernel's register dataflow* in numpy from the documented
CDNA4 MFMA layouts (nrnerf_plan.h): A fragment lane l, element e = A[i = l&31][k = KH*(l>>5)+e];
B slab lane (j, h), element e = B[k = KH*h+e][j]; D register r of lane (j, h) = D[tile_row(r,h)][j];
the next layer's B slab t*SP+u takes element e from D register u*KH+e.  If the packed stream,
bias table and unit table are right, chaining those rules over every layer must reproduce the
plain ``F.linear`` network.  No compute entry point of the library is called (no GPU needed).
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from nonrigid_nerf_amd import _lib
from nonrigid_nerf_amd.render import build_model_desc
from nonrigid_nerf_amd.synthetic import SceneConfig, build_modules, make_scene


def tile_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    assert lib.nrnerf_abi_version() == _lib.ABI_VERSION
    for name in _lib.EXPORTS:
        assert hasattr(lib, name), name
    # every prototype in the public header is bound
    import os, re
    hdr = open(os.path.join(os.path.dirname(_lib._HERE), "include", "nrnerf.h")).read()
    declared = set(re.findall(r"\b(nrnerf_[a-z_]+)\s*\(", hdr)) - {"nrnerf_workspace_bytes"} | {"nrnerf_workspace_bytes"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert _lib.strerror(0) == "ok" and "unsupported" in _lib.strerror(_lib.ERR_UNSUPPORTED)


def _pack(scene_cfg, precision, which):
    scene = make_scene(scene_cfg, 3)
    rb, coarse, fine = build_modules(scene)
    desc, keep = build_model_desc(coarse, fine, precision, 0)
    lib = _lib.load()
    info = _lib.PackedInfo()
    null_u32 = C.POINTER(C.c_uint32)()
    null_f = C.POINTER(C.c_float)()
    _lib.check(lib.nrnerf_pack_host(C.byref(desc), which, C.byref(info), None, 0, null_u32, null_f), "size query")
    stream = np.zeros(info.stream_bytes, dtype=np.uint8)
    units = np.zeros(info.n_units + 1, dtype=np.uint32)
    bias = np.zeros(info.n_bias_tiles * 32, dtype=np.float32)
    _lib.check(lib.nrnerf_pack_host(C.byref(desc), which, C.byref(info), stream.ctypes.data_as(C.c_void_p),
                                    stream.nbytes, units.ctypes.data_as(C.POINTER(C.c_uint32)),
                                    bias.ctypes.data_as(C.POINTER(C.c_float))), "pack")
    return scene, (rb, coarse, fine), info, stream, units, bias


class FragReader:
    """Walks the packed stream fragment by fragment, decoding A[32][2*KH] matrices.

    In the 16-bit modes a fragment is f16 when its B operand is bounded by construction or feeds the
    encoding (bender, rigidity, encoding slabs; nrnerf_plan.h frag_is_f16) and bf16/f16 otherwise.
    """
    def __init__(self, stream, precision, frag_bytes):
        self.KH = 1 if precision == "f32" else 8
        self.precision = precision
        if precision == "f32":
            self.f32 = stream.view(np.float32).astype(np.float64)
        else:
            u = stream.view(np.uint16).astype(np.uint32) << 16
            self.bf16 = u.view(np.float32).astype(np.float64)
            self.f16 = stream.view(np.float16).astype(np.float64)
        self.per = 64 * self.KH
        assert frag_bytes == self.per * (4 if precision == "f32" else 2)
        self.pos = 0

    def next(self, as_f16=False):
        vals = self.f32 if self.precision == "f32" else (self.f16 if (as_f16 or self.precision == "f16") else self.bf16)
        f = vals[self.pos * self.per:(self.pos + 1) * self.per].reshape(64, self.KH)
        self.pos += 1
        A = np.zeros((32, 2 * self.KH))
        for lane in range(64):
            A[lane & 31, self.KH * (lane >> 5):self.KH * (lane >> 5) + self.KH] = f[lane]
        return A


def dense_emul(fr, bias_tab, tile0, ns, nt, slabs, split=False, f16_slabs=0):
    """slabs: [ns][2*KH][nsamp] -> list of nt D tiles [32][nsamp].

    Stream order (nrnerf_plan.h): tiles in pairs with interleaved slabs -- (2p,0) (2p+1,0) (2p,1) (2p+1,1) ... --
    an odd last tile alone.  split: every (tile, slab) has a (hi, lo * 2^11) fragment pair (3-term split product).
    f16_slabs: the first f16_slabs slabs are f16 fragments even in bf16 mode (-1: all)."""
    def bias_of(t):
        b = np.zeros(32)
        for h in range(2):
            for r in range(16):
                b[tile_row(r, h)] = bias_tab[(tile0 + t) * 32 + h * 16 + r]
        return np.repeat(b[:, None], slabs[0].shape[1], 1)

    def next_A(s):
        f16 = f16_slabs < 0 or s < f16_slabs
        A = fr.next(f16)
        if split:
            lo = fr.next(f16) / 2048.0       # lo parts are stored pre-scaled by 2^11 (no f16 subnormals)
            assert np.abs(lo).max() <= np.abs(A).max() * 2.0 ** -10 + 1e-30
            A = A + lo
        return A

    out = [None] * nt
    for p in range(0, nt - 1, 2):
        D0, D1 = bias_of(p), bias_of(p + 1)
        for s in range(ns):
            D0 = D0 + next_A(s) @ slabs[s]
            D1 = D1 + next_A(s) @ slabs[s]
        out[p], out[p + 1] = D0, D1
    if nt & 1:
        D = bias_of(nt - 1)
        for s in range(ns):
            D = D + next_A(s) @ slabs[s]
        out[nt - 1] = D
    return out


def repack(tiles, KH, relu=True, rnd=None):
    """D tiles -> B slabs of the next layer (the in-register hand-off)."""
    SP = 16 // KH
    slabs = []
    for D in tiles:
        X = np.maximum(D, 0) if relu else D
        if rnd is not None:
            X = rnd(X)
        for u in range(SP):
            sl = np.zeros((2 * KH, D.shape[1]))
            for h in range(2):
                for e in range(KH):
                    sl[KH * h + e] = X[tile_row(u * KH + e, h)]
            slabs.append(sl)
    return slabs


def enc_slabs(p, L, KH, rnd):
    """Kernel-side positional encoding order: per half h slots [id0,id1,(sin,cos) per (fl,c)], f = h*F0+fl."""
    F0 = (L + 1) // 2
    nslot = -(-(2 + 6 * F0) // KH) * KH
    ev = np.zeros((2, nslot, p.shape[0]))
    for h in range(2):
        ev[h, 0] = p[:, 2] if h else p[:, 0]
        ev[h, 1] = 0 if h else p[:, 1]
        for fl in range(F0):
            for c in range(3):
                arg = p[:, c] * float(2 ** (h * F0 + fl))
                ev[h, 2 + 2 * (3 * fl + c)] = np.sin(arg)
                ev[h, 2 + 2 * (3 * fl + c) + 1] = np.cos(arg)
    ns = nslot // KH
    slabs = []
    for s in range(ns):
        sl = np.zeros((2 * KH, p.shape[0]))
        for h in range(2):
            sl[KH * h:KH * h + KH] = ev[h, s * KH:(s + 1) * KH]
        slabs.append(rnd(sl))
    return slabs


def vec_slabs(v, KH, rnd):
    """Logical input vector v [len, nsamp] -> slabs with element (s,h,e) = v[(2s+h)*KH+e]."""
    n = -(-v.shape[0] // (2 * KH))
    vp = np.zeros((n * 2 * KH, v.shape[1]))
    vp[:v.shape[0]] = v
    return [rnd(vp[s * 2 * KH:(s + 1) * 2 * KH]) for s in range(n)]


def rounder(precision):
    if precision == "f32":
        return lambda x: x
    dt = torch.bfloat16 if precision == "bf16" else torch.float16
    return lambda x: torch.from_numpy(np.asarray(x, dtype=np.float32)).to(dt).to(torch.float64).numpy()


@pytest.mark.parametrize("precision", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("bend,views,tcb,width", [(True, False, False, 256), (False, False, False, 256), (True, True, False, 256),
                                                  (False, True, False, 256), (False, False, True, 256),
                                                  (True, False, False, 128), (False, False, False, 128)])
def test_packed_stream_reproduces_the_network(precision, bend, views, tcb, width):
    cfg = SceneConfig(N_importance=128, ray_bending=bend, use_viewdirs=views, time_conditioned_baseline=tcb, netwidth=width)
    NT = width // 32
    scene, (rb, coarse, fine), info, stream, units, bias = _pack(cfg, precision, which=1)
    KH = 1 if precision == "f32" else 8
    SP = 16 // KH
    rnd = rounder(precision)
    split = precision == "f16"                 # "f16" mode: 3-term split product in the bender; "bf16": single f16 product
    rnd_b = (lambda x: x) if precision != "bf16" else rounder("f16")   # hi + lo carries ~22 bits: emulate as exact
    rnd_e = rounder("f16") if split else rnd   # encoding slabs are f16 in both 16-bit modes
    fr = FragReader(stream, precision, info.frag_bytes)
    gen = torch.Generator().manual_seed(5)
    ns_ = 32
    p = (torch.randn(ns_, 3, generator=gen) * 0.4).double().numpy()
    lat = (torch.randn(ns_, 32, generator=gen) * 0.1).double().numpy()
    tile0 = 0
    pt = torch.from_numpy(p)
    mfma = 0
    if bend:
        m3 = 3 if split else 1
        kwb = dict(split=split, f16_slabs=-1)
        v = np.concatenate([p.T, np.zeros((5, ns_)), lat.T], 0)                    # bin vector (nrnerf_plan.h)
        slabs = vec_slabs(v, KH, rnd_b)
        nt_b = 2
        tiles = dense_emul(fr, bias, tile0, len(slabs), nt_b, slabs, **kwb); mfma += m3 * len(slabs) * nt_b; tile0 += nt_b
        for _ in range(3):
            slabs = repack(tiles, KH, True, rnd_b)
            tiles = dense_emul(fr, bias, tile0, len(slabs), nt_b, slabs, **kwb); mfma += m3 * len(slabs) * nt_b; tile0 += nt_b
        slabs = repack(tiles, KH, True, rnd_b)
        D = dense_emul(fr, bias, tile0, len(slabs), 1, slabs, **kwb)[0]; mfma += m3 * len(slabs); tile0 += 1
        off = D[0:3]                         # lanes of half 0, acc[0..2]
        assert np.allclose(D[4:7], off)      # duplicated rows feed half 1
        v = np.concatenate([p.T, np.zeros((5, ns_))], 0)
        slabs = vec_slabs(v, KH, rnd_b)
        tiles = dense_emul(fr, bias, tile0, len(slabs), 1, slabs, **kwb); mfma += m3 * len(slabs); tile0 += 1
        slabs = repack(tiles, KH, True, rnd_b)
        tiles = dense_emul(fr, bias, tile0, len(slabs), 1, slabs, **kwb); mfma += m3 * len(slabs); tile0 += 1
        slabs = repack(tiles, KH, True, rnd_b)
        D = dense_emul(fr, bias, tile0, len(slabs), 1, slabs, **kwb)[0]; mfma += m3 * len(slabs); tile0 += 1
        logit = D[0]
        assert np.allclose(D[4], logit)
        # reference (fp64 torch) bender
        with torch.no_grad():
            h = torch.cat([pt, torch.from_numpy(lat)], -1)
            for i, l in enumerate(rb.network):
                h = F.linear(h, l.weight.double(), None if l.bias is None else l.bias.double())
                if i != len(rb.network) - 1:
                    h = F.relu(h)
            r = pt
            for i, l in enumerate(rb.rigidity_network):
                r = F.linear(r, l.weight.double(), l.bias.double())
                if i != len(rb.rigidity_network) - 1:
                    r = F.relu(r)
        tol = {"f32": 1e-9, "f16": 2e-6, "bf16": 4e-3}[precision]     # split product: hi + lo (~22 bits); single f16 product
        scale = float(h.abs().max())
        assert np.abs(off.T - h.numpy()).max() <= tol * max(scale, 1e-3) + 1e-12, "bender offsets"
        assert np.abs(logit - r.numpy()[:, 0]).max() <= tol * max(float(r.abs().max()), 1.0), "rigidity logit"
    # ---- trunk
    slabs_enc = enc_slabs(p, 10, KH, rnd_e)
    if tcb:        # time-conditioned baseline: latent slabs follow the encoding (element (s,h,e) = latent[(2s+h)*KH+e])
        slabs_enc = slabs_enc + vec_slabs(lat.T, KH, rnd_e)
    tiles = dense_emul(fr, bias, tile0, len(slabs_enc), NT, slabs_enc, f16_slabs=-1); mfma += len(slabs_enc) * NT; tile0 += NT
    for i in range(1, 8):
        slabs = repack(tiles, KH, True, rnd)
        n16 = 0
        if i - 1 == 4:
            slabs = slabs_enc + slabs
            n16 = len(slabs_enc)
        tiles = dense_emul(fr, bias, tile0, len(slabs), NT, slabs, f16_slabs=n16); mfma += len(slabs) * NT; tile0 += NT
    slabs = repack(tiles, KH, True, rnd)
    dirs = (torch.randn(ns_, 3, generator=gen)).double()
    dirs = (dirs / dirs.norm(dim=-1, keepdim=True)).numpy()
    if views:
        D = dense_emul(fr, bias, tile0, len(slabs), 1, slabs)[0]; mfma += len(slabs); tile0 += 1        # alpha_linear
        alpha = D[0]
        assert np.allclose(D[4], alpha)
        # (feature_linear has no layer of its own: folded into the views layer's hidden columns by the packer, nrnerf_plan.h)
        dslabs = enc_slabs(dirs, 4, KH, rnd_e)
        vs = dslabs + slabs
        vtiles = dense_emul(fr, bias, tile0, len(vs), 4, vs, f16_slabs=len(dslabs)); mfma += len(vs) * 4; tile0 += 4
        vslabs = repack(vtiles, KH, True, rnd)
        D = dense_emul(fr, bias, tile0, len(vslabs), 1, vslabs)[0]; mfma += len(vslabs); tile0 += 1     # rgb_linear
        assert np.allclose(D[4:7], D[0:3])
        raw = np.stack([D[0], D[1], D[2], alpha], -1)
    else:
        D = dense_emul(fr, bias, tile0, len(slabs), 1, slabs)[0]; mfma += len(slabs); tile0 += 1
        raw = np.stack([D[0], D[1], D[2], D[3], D[8]], -1)       # acc[0..4] of half-0 lanes
    used = fr.pos * info.frag_bytes          # the stream is zero-padded to whole 16 KiB units, a multiple of the ring depth
    assert used <= info.stream_bytes < used + 8 * info.slot_bytes and not stream[used:].any(), "stream fully consumed"
    assert info.stream_bytes == info.n_units * info.slot_bytes and info.slot_bytes % 16384 == 0
    assert tile0 == info.n_bias_tiles and mfma == info.mfma_per_block
    with torch.no_grad():
        cols = [pt]
        for k in range(10):
            cols += [torch.sin(pt * 2.0 ** k), torch.cos(pt * 2.0 ** k)]
        x = torch.cat(cols + ([torch.from_numpy(lat)] if tcb else []), -1)
        h = x
        for i, l in enumerate(fine.pts_linears):
            h = F.relu(F.linear(h, l.weight.double(), l.bias.double()))
            if i == 4:
                h = torch.cat([x, h], -1)
        if views:
            lin = lambda m, x: F.linear(x, m.weight.double(), m.bias.double())
            dt = torch.from_numpy(dirs)
            dcols = [dt]
            for k in range(4):
                dcols += [torch.sin(dt * 2.0 ** k), torch.cos(dt * 2.0 ** k)]
            al = lin(fine.alpha_linear, h)
            hv = F.relu(lin(fine.views_linears[0], torch.cat([lin(fine.feature_linear, h), torch.cat(dcols, -1)], -1)))
            ref = torch.cat([lin(fine.rgb_linear, hv), al], -1).numpy()
        else:
            ref = F.linear(h, fine.output_linear.weight.double(), fine.output_linear.bias.double()).numpy()
    # (f32 with the view-dependent head: the folded weights W_v1 W_f are rounded to fp32 once, 1e-7 of their scale)
    tol = (1e-6 if views else 1e-9) if precision == "f32" else (8e-2 if precision == "bf16" else 1e-2)
    err = np.abs(raw - ref).max()
    assert err <= tol * np.abs(ref).max(), f"trunk+head mismatch {err} vs scale {np.abs(ref).max()}"
    # unit table: uniform 16 KiB units (offsets in 16-byte words)
    assert units[0] == 0 and int(units[-1]) * 16 == info.stream_bytes
    assert (np.diff(units.astype(np.int64)) * 16 == info.slot_bytes).all()


def _x16_enc_col(L, s, g, e):
    """nrnerf_plan.h::x16_enc_col, restated: slot (k-step s, lane group g, element e) of the encoding -> reference column."""
    q = 8 * s + e
    m = 4 * (q // 2) + g
    if m < 3 * L:
        return 3 + 6 * (m // 3) + 3 * (q & 1) + (m % 3)
    spare = sum(1 for qq in range(q) if 4 * (qq // 2) + g >= 3 * L)
    before = sum(1 for gg in range(g) for qq in range(16) if 4 * (qq // 2) + gg >= 3 * L)
    return before + spare if before + spare < 3 else -1


def _x16_dir_col(LV, g, e):
    """nrnerf_plan.h::x16_dir_col, restated: the direction encoding fills ONE k-step (8 slots per lane group)."""
    m = 4 * (e // 2) + g
    if m < 3 * LV:
        return 3 + 6 * (m // 3) + 3 * (e & 1) + (m % 3)
    spare = sum(1 for qq in range(e) if 4 * (qq // 2) + g >= 3 * LV)
    before = sum(1 for gg in range(g) for qq in range(8) if 4 * (qq // 2) + gg >= 3 * LV)
    return before + spare if before + spare < 3 else -1


@pytest.mark.parametrize("cfg_kw", [dict(), dict(netwidth=128), dict(use_viewdirs=True), dict(use_viewdirs=True, bend_depth=7)],
                         ids=["w256", "w128", "w256_viewdirs", "config4"])
@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_x16_stream_reproduces_the_trunk(precision, cfg_kw):
    """The trunk-only image of the 16x16x32 kernel (csrc/nrnerf_net_x16.h, PlanX16; nrnerf_pack_host which = 10), emulated in
    numpy as the kernel consumes it: a fragment is W[16 rows][32 k] with lane (r, g) holding k positions 8 g .. 8 g + 7; the B
    operand of k-step s holds, at position 8 g + e, input column x16_in_col(s, g, e) -- for the encoding the slot layout of
    x16_enc_col (every one of the 63 columns exactly once), for hidden layers features 32 s + 4 g + e (e < 4, D tile 2 s) and
    32 s + 16 + 4 g + (e - 4) (D tile 2 s + 1): what a lane holds after two consecutive 16-row tiles; tiles in pairs with their
    k-steps interleaved, the head alone.  Must reproduce the fp64 network up to the operand rounding."""
    cfg = SceneConfig(N_importance=128, **cfg_kw)
    scene, (rb, coarse, fine), info, stream, units, bias = _pack(cfg, precision, which=10)
    assert info.frag_bytes == 1024
    views, W = cfg.use_viewdirs, cfg.netwidth
    rnd, rnd_e = rounder(precision), rounder("f16")
    u16 = stream.view(np.uint16)
    as_bf16 = (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    as_f16 = stream.view(np.float16).astype(np.float64)
    pos = [0]

    def next_A(f16):
        vals = as_f16 if (f16 or precision == "f16") else as_bf16
        f = vals[pos[0] * 512:(pos[0] + 1) * 512].reshape(64, 8)
        pos[0] += 1
        A = np.zeros((16, 32))
        for lane in range(64):
            A[lane & 15, 8 * (lane >> 4):8 * (lane >> 4) + 8] = f[lane]
        return A

    def dense(tile0, ns, nt, B, n_f16):
        """B: [ns][32 positions][samples] -> nt D tiles [16][samples]"""
        def bias_of(t):
            return np.repeat(bias[(tile0 + t) * 16:(tile0 + t) * 16 + 16].astype(np.float64)[:, None], B[0].shape[1], 1)
        out = [None] * nt
        for p_ in range(0, nt - 1, 2):
            D0, D1 = bias_of(p_), bias_of(p_ + 1)
            for s_ in range(ns):
                D0 = D0 + next_A(s_ < n_f16) @ B[s_]
                D1 = D1 + next_A(s_ < n_f16) @ B[s_]
            out[p_], out[p_ + 1] = D0, D1
        if nt & 1:
            D = bias_of(nt - 1)
            for s_ in range(ns):
                D = D + next_A(s_ < n_f16) @ B[s_]
            out[nt - 1] = D
        return out

    def hand_off(tiles):
        """16 D tiles -> 8 B operands: position 8 g + e = relu(feature 32 s + 4 g + e | 32 s + 16 + 4 g + e - 4), rounded"""
        B = []
        for s_ in range(len(tiles) // 2):
            b = np.zeros((32, tiles[0].shape[1]))
            for g in range(4):
                for e in range(8):
                    b[8 * g + e] = tiles[2 * s_][4 * g + e] if e < 4 else tiles[2 * s_ + 1][4 * g + e - 4]
            B.append(rnd(np.maximum(b, 0.0)))
        return B

    gen = torch.Generator().manual_seed(6)
    ns_ = 24
    p = (torch.randn(ns_, 3, generator=gen) * 0.4).double()
    cols = [p]
    for k in range(10):
        cols += [torch.sin(p * 2.0 ** k), torch.cos(p * 2.0 ** k)]
    x = torch.cat(cols, -1)                                                   # [ns, 63], reference column order
    xn = x.numpy()
    seen = sorted(c for s_ in range(2) for g in range(4) for e in range(8) if (c := _x16_enc_col(10, s_, g, e)) >= 0)
    assert seen == list(range(63)), "every encoding column sits in exactly one slot"
    Benc = []
    for s_ in range(2):
        b = np.zeros((32, ns_))
        for g in range(4):
            for e in range(8):
                c = _x16_enc_col(10, s_, g, e)
                if c >= 0:
                    b[8 * g + e] = xn[:, c]
        Benc.append(rnd_e(b))
    NT = W // 16
    tile0, mfma = 0, 0
    tiles = dense(tile0, 2, NT, Benc, 2); tile0 += NT; mfma += 2 * NT
    for i in range(1, 8):
        B = hand_off(tiles)
        n16 = 0
        if i - 1 == 4:
            B, n16 = Benc + B, 2
        tiles = dense(tile0, len(B), NT, B, n16); tile0 += NT; mfma += len(B) * NT
    B = hand_off(tiles)
    if views:
        # view-dependent head (PlanX16<.., VIEWS>): k-steps [direction encoding (one, x16_dir_col), trunk output] -> W / 32 feature tiles
        # of relu(views_linears[0] o feature_linear) + the alpha tile (row 0, no relu, zero weights in the direction k-step); then rgb_linear
        d = torch.randn(ns_, 3, generator=gen).double()
        d = d / d.norm(dim=-1, keepdim=True)
        dcols = [d]
        for k in range(4):
            dcols += [torch.sin(d * 2.0 ** k), torch.cos(d * 2.0 ** k)]
        xd = torch.cat(dcols, -1)                                             # [ns, 27]
        seen_d = sorted(c for g in range(4) for e in range(8) if (c := _x16_dir_col(4, g, e)) >= 0)
        assert seen_d == list(range(27)), "every direction-encoding column sits in exactly one slot"
        bd = np.zeros((32, ns_))
        for g in range(4):
            for e in range(8):
                c = _x16_dir_col(4, g, e)
                if c >= 0:
                    bd[8 * g + e] = xd.numpy()[:, c]
        Bv = [rnd_e(bd)] + B
        NTV = W // 32
        vt = dense(tile0, len(Bv), NTV + 1, Bv, 1); tile0 += NTV + 1; mfma += len(Bv) * (NTV + 1)
        sigma = vt[NTV][0]
        Bh = hand_off(vt[:NTV])
        D = dense(tile0, len(Bh), 1, Bh, 0)[0]; tile0 += 1; mfma += len(Bh)
        raw = np.concatenate([D[0:3].T, sigma[:, None]], 1)                  # [rgb, sigma]
    else:
        D = dense(tile0, len(B), 1, B, 0)[0]; tile0 += 1; mfma += len(B)
        raw = D[0:5].T                                                       # channels 0..3: group 0's registers, channel 4: group 1's first
    assert tile0 == info.n_bias_tiles and mfma == info.mfma_per_block
    used = pos[0] * 1024
    assert used <= info.stream_bytes and not stream[used:].any(), "stream fully consumed"
    with torch.no_grad():
        h = x
        for i, l in enumerate(fine.pts_linears):
            h = F.relu(F.linear(h, l.weight.double(), l.bias.double()))
            if i == 4:
                h = torch.cat([x, h], -1)
        if views:                                                             # rnh:284-304
            alpha = F.linear(h, fine.alpha_linear.weight.double(), fine.alpha_linear.bias.double())
            feat = F.linear(h, fine.feature_linear.weight.double(), fine.feature_linear.bias.double())
            hv = F.relu(F.linear(torch.cat([feat, xd], -1), fine.views_linears[0].weight.double(), fine.views_linears[0].bias.double()))
            ref = torch.cat([F.linear(hv, fine.rgb_linear.weight.double(), fine.rgb_linear.bias.double()), alpha], -1).numpy()
        else:
            ref = F.linear(h, fine.output_linear.weight.double(), fine.output_linear.bias.double()).numpy()
    tol = 8e-2 if precision == "bf16" else 1e-2
    err = np.abs(raw - ref).max()
    assert err <= tol * np.abs(ref).max(), f"x16 trunk + head mismatch {err} vs scale {np.abs(ref).max()}"


@pytest.mark.parametrize("cfg_kw", [dict(netwidth=192, netdepth=6, multires=8), dict(netwidth=320, netdepth=10), dict(netwidth=64, netdepth=3, skips=()),
                                    dict(netwidth=256), dict(netwidth=500, netdepth=5, skips=(1,), multires=4),
                                    dict(netwidth=192, netdepth=6, use_viewdirs=True), dict(netwidth=96, netdepth=4, skips=(1,), use_viewdirs=True, multires_views=2)],
                         ids=["w192_d6_l8", "w320_d10", "w64_d3_noskip", "w256_default", "w500_d5_skip1_l4", "w192_viewdirs", "w96_viewdirs_lv2"])
@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_width_class_stream_reproduces_any_plain_trunk(precision, cfg_kw):
    """The image of the width-class trunk kernel (csrc/nrnerf_gx16.h; nrnerf_pack_host which = 12), emulated in numpy as the kernel
    consumes it: the layers' fragment blocks back to back -- first layer (two encoding k-steps, positions p = 32 s + 8 g + e: p < 3 the
    identity columns, p = 3 zero, then (sin, cos) pairs m = (p - 4) / 2), pts_linears[1 ..] (the one after the skip index with the encoding
    k-steps first), output_linear -- each padded to a whole number of 4-unit ring periods, the width padded to a multiple of 64 with zero
    rows / columns; a copy of the first two units behind the last layer.  Any depth, skip index, width <= 512, <= 10 frequencies."""
    cfg = SceneConfig(N_importance=128, **cfg_kw)
    scene, (rb, coarse, fine), info, stream, units, bias = _pack(cfg, precision, which=12)
    assert info.frag_bytes == 1024
    W, D, L = cfg.netwidth, cfg.netdepth, cfg.multires
    skip = cfg.skips[0] if cfg.skips else -1
    WC = (W + 63) // 64 * 64
    rnd, rnd_e = rounder(precision), rounder("f16")
    u16 = stream.view(np.uint16)
    as_bf16 = (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    as_f16 = stream.view(np.float16).astype(np.float64)
    pos = [0]

    def next_A(f16):
        vals = as_f16 if (f16 or precision == "f16") else as_bf16
        f = vals[pos[0] * 512:(pos[0] + 1) * 512].reshape(64, 8)
        pos[0] += 1
        A = np.zeros((16, 32))
        for lane in range(64):
            A[lane & 15, 8 * (lane >> 4):8 * (lane >> 4) + 8] = f[lane]
        return A

    tile = [0]

    def dense(ns, nt, B, n_f16):
        start = pos[0]

        def bias_of(t):
            return np.repeat(bias[(tile[0] + t) * 16:(tile[0] + t) * 16 + 16].astype(np.float64)[:, None], B[0].shape[1], 1)
        out = [None] * nt
        for p_ in range(0, nt - 1, 2):
            D0, D1 = bias_of(p_), bias_of(p_ + 1)
            for s_ in range(ns):
                D0 = D0 + next_A(s_ < n_f16) @ B[s_]
                D1 = D1 + next_A(s_ < n_f16) @ B[s_]
            out[p_], out[p_ + 1] = D0, D1
        if nt & 1:
            Dl = bias_of(nt - 1)
            for s_ in range(ns):
                Dl = Dl + next_A(s_ < n_f16) @ B[s_]
            out[nt - 1] = Dl
        tile[0] += nt
        used_units = -(-(pos[0] - start) // 16)
        pos[0] = start + (-(-used_units // 4) * 4) * 16           # the layer's padding: on to the next whole ring period
        return out

    def hand_off(tiles):
        B = []
        for s_ in range(len(tiles) // 2):
            b = np.zeros((32, tiles[0].shape[1]))
            for g in range(4):
                for e in range(8):
                    b[8 * g + e] = tiles[2 * s_][4 * g + e] if e < 4 else tiles[2 * s_ + 1][4 * g + e - 4]
            B.append(rnd(np.maximum(b, 0.0)))
        return B

    gen = torch.Generator().manual_seed(6)
    ns_ = 24
    p = (torch.randn(ns_, 3, generator=gen) * 0.4).double()
    cols = [p]
    for k in range(L):
        cols += [torch.sin(p * 2.0 ** k), torch.cos(p * 2.0 ** k)]
    x = torch.cat(cols, -1)                                                   # [ns, 3 + 6 L], reference column order
    xn = x.numpy()

    def enc_col(s_, g, e):
        q = 32 * s_ + 8 * g + e
        if q < 3:
            return q
        if q == 3:
            return -1
        m, b = (q - 4) // 2, (q - 4) & 1
        return 3 + 6 * (m // 3) + 3 * b + (m % 3) if m < 3 * L else -1
    seen = sorted(c for s_ in range(2) for g in range(4) for e in range(8) if (c := enc_col(s_, g, e)) >= 0)
    assert seen == list(range(3 + 6 * L)), "every encoding column sits in exactly one position"
    Benc = []
    for s_ in range(2):
        b = np.zeros((32, ns_))
        for g in range(4):
            for e in range(8):
                c = enc_col(s_, g, e)
                if c >= 0:
                    b[8 * g + e] = xn[:, c]
        Benc.append(rnd_e(b))
    NT = WC // 16
    mfma = 0
    tiles = dense(2, NT, Benc, 2); mfma += 2 * NT
    for i in range(1, D):
        B = hand_off(tiles)
        n16 = 0
        if i - 1 == skip:
            B, n16 = Benc + B, 2
        tiles = dense(len(B), NT, B, n16); mfma += len(B) * NT
    B = hand_off(tiles)
    views = cfg.use_viewdirs
    if views:
        # GX_VIEWS: k-steps [direction encoding (one; the points' position map with LV), trunk output] -> WC / 32 tiles of
        # relu(views o feature) + the alpha tile (row 0); GX_RGB: WC / 64 k-steps -> one tile, rows 0..2
        LV = cfg.multires_views
        dvec = torch.randn(ns_, 3, generator=gen).double()
        dvec = dvec / dvec.norm(dim=-1, keepdim=True)
        dcols = [dvec]
        for k in range(LV):
            dcols += [torch.sin(dvec * 2.0 ** k), torch.cos(dvec * 2.0 ** k)]
        xd = torch.cat(dcols, -1)

        def dir_col(g, e):
            q = 8 * g + e
            if q < 3:
                return q
            if q == 3:
                return -1
            m, b = (q - 4) // 2, (q - 4) & 1
            return 3 + 6 * (m // 3) + 3 * b + (m % 3) if m < 3 * LV else -1
        bd = np.zeros((32, ns_))
        for g in range(4):
            for e in range(8):
                c = dir_col(g, e)
                if c >= 0:
                    bd[8 * g + e] = xd.numpy()[:, c]
        Bv = [rnd_e(bd)] + B
        NTV = WC // 32
        vt = dense(len(Bv), NTV + 1, Bv, 1); mfma += len(Bv) * (NTV + 1)
        sigma = vt[NTV][0]
        Bh = hand_off(vt[:NTV])
        Dh = dense(len(Bh), 1, Bh, 0)[0]; mfma += len(Bh)
        raw = np.concatenate([Dh[0:3].T, sigma[:, None]], 1)
    else:
        Dh = dense(len(B), 1, B, 0)[0]; mfma += len(B)
        raw = Dh[0:5].T
    assert tile[0] == info.n_bias_tiles and mfma == info.mfma_per_block
    # behind the last layer: a copy of the stream's first two units, then nothing
    assert info.stream_bytes == (pos[0] // 16 + 2) * 16384
    assert (stream[pos[0] * 1024:pos[0] * 1024 + 2 * 16384] == stream[:2 * 16384]).all()
    with torch.no_grad():
        h = x
        for i, l in enumerate(fine.pts_linears):
            h = F.relu(F.linear(h, l.weight.double(), l.bias.double()))
            if i == skip:
                h = torch.cat([x, h], -1)
        if views:
            alpha = F.linear(h, fine.alpha_linear.weight.double(), fine.alpha_linear.bias.double())
            feat = F.linear(h, fine.feature_linear.weight.double(), fine.feature_linear.bias.double())
            hvr = F.relu(F.linear(torch.cat([feat, xd], -1), fine.views_linears[0].weight.double(), fine.views_linears[0].bias.double()))
            ref = torch.cat([F.linear(hvr, fine.rgb_linear.weight.double(), fine.rgb_linear.bias.double()), alpha], -1).numpy()
        else:
            ref = F.linear(h, fine.output_linear.weight.double(), fine.output_linear.bias.double()).numpy()
    tol = 8e-2 if precision == "bf16" else 1e-2
    err = np.abs(raw - ref).max()
    assert err <= tol * np.abs(ref).max(), f"width-class trunk + head mismatch {err} vs scale {np.abs(ref).max()}"


def test_unsupported_architectures_are_rejected():
    lib = _lib.load()
    for kw in (dict(netwidth=192), dict(netwidth=128, use_viewdirs=True), dict(netwidth=128, bend_depth=7), dict(netdepth=6), dict(multires=8), dict(bend_hidden=32), dict(latent_size=16),
               dict(bend_depth=6), dict(bend_depth=7, ray_bending=True, rigidity_depth=4)):
        scene = make_scene(SceneConfig(**kw), 0)
        rb, coarse, fine = build_modules(scene)
        desc, keep = build_model_desc(coarse, fine, "bf16", 0)
        info = _lib.PackedInfo()
        rc = lib.nrnerf_pack_host(C.byref(desc), 0, C.byref(info), None, 0, C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)())
        assert rc == _lib.ERR_UNSUPPORTED, (kw, rc)
    # the deeper-bender architecture of BASELINE config 4 IS compiled (arch id 1)
    scene = make_scene(SceneConfig(bend_depth=7, use_viewdirs=True), 0)
    rb, coarse, fine = build_modules(scene)
    for prec, bender_mfmas in (("f16", 3 * (6 + 5 * 8 + 4) + 15), ("bf16", (6 + 5 * 8 + 4) + 5)):     # split product: 3 MFMAs per slab
        desc, keep = build_model_desc(coarse, fine, prec, 0)
        info = _lib.PackedInfo()
        assert lib.nrnerf_pack_host(C.byref(desc), 1, C.byref(info), None, 0, C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)()) == 0
        assert info.mfma_per_block == 976 - 16 + 16 + 72 + 8 + bender_mfmas, prec       # alpha 16, views (feature_linear folded in) 72, rgb 8
    bad = _lib.ModelDesc()
    assert lib.nrnerf_pack_host(C.byref(bad), 0, None, None, 0, C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)()) == _lib.ERR_INVALID
    out = C.c_void_p()
    assert lib.nrnerf_model_create(None, C.byref(out)) == _lib.ERR_INVALID


@pytest.mark.parametrize("precision", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("bend_depth", [5, 7])
def test_split_bender_images_are_the_two_halves_of_the_fused_stream(precision, bend_depth):
    """Split-bender path (nrnerf_bend.h): the stand-alone bender kernel and the trunk-only fine pass must use exactly the
    weights of the fused fine pass.  The fused stream is [bender + rigidity fragments | trunk + head fragments] with the
    tile pairing restarting at every layer, so the two extra images (nrnerf_pack_host which = 3 / 2) are its two halves,
    byte for byte, and so are the bias tables."""
    cfg = SceneConfig(N_importance=128, bend_depth=bend_depth)
    _, _, info_f, stream_f, _, bias_f = _pack(cfg, precision, which=1)
    _, _, info_t, stream_t, _, bias_t = _pack(cfg, precision, which=2)
    _, _, info_b, stream_b, _, bias_b = _pack(cfg, precision, which=3)
    fb = info_f.frag_bytes
    assert info_t.frag_bytes == fb and info_b.frag_bytes == fb
    nb_tiles = info_b.n_bias_tiles
    assert nb_tiles + info_t.n_bias_tiles == info_f.n_bias_tiles
    assert np.array_equal(bias_f[:nb_tiles * 32], bias_b) and np.array_equal(bias_f[nb_tiles * 32:], bias_t)
    assert info_b.mfma_per_block + info_t.mfma_per_block == info_f.mfma_per_block
    # fragment counts: bender layers stream 2 fragments per 3 MFMAs in the split-product ("f16") mode
    nfrag_b = info_b.mfma_per_block * 2 // 3 if precision == "f16" else info_b.mfma_per_block
    nfrag_t = info_t.mfma_per_block
    assert np.array_equal(stream_f[:nfrag_b * fb], stream_b[:nfrag_b * fb]) and not stream_b[nfrag_b * fb:].any()
    assert np.array_equal(stream_f[nfrag_b * fb:(nfrag_b + nfrag_t) * fb], stream_t[:nfrag_t * fb]) and not stream_t[nfrag_t * fb:].any()
    # without a bender (or with the view-dependent head) there is nothing to split
    scene = make_scene(SceneConfig(ray_bending=False), 0)
    rb, coarse, fine = build_modules(scene)
    desc, keep = build_model_desc(coarse, fine, precision, 0)
    info = _lib.PackedInfo()
    lib = _lib.load()
    assert lib.nrnerf_pack_host(C.byref(desc), 3, C.byref(info), None, 0, C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)()) == _lib.ERR_UNSUPPORTED


@pytest.mark.parametrize("width", [256, 128])
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_backward_stream_reproduces_autograd_of_the_trunk(precision, width):
    """Training (csrc/nrnerf_train.h): the backward-data kernel streams TRANSPOSED weights in PlanB's order (head^T, then
    pts_linears[7..1]^T, then pts_linears[0]^T; nrnerf_pack_host which = 5).  Emulating its register dataflow in numpy --
    d raw as the first B operand, relu masks from the forward activations, the skip layer's first two tiles and the last
    layer's two tiles being the encoding's gradient in encoding-SLOT order -- must reproduce torch.autograd's gradient
    wrt the encoded input."""
    cfg = SceneConfig(N_importance=128, netwidth=width)
    NT = width // 32
    scene, (rb, coarse, fine), info, stream, units, bias = _pack(cfg, precision, which=5)
    assert not bias.any(), "backward layers have no bias"
    KH = 1 if precision == "f32" else 8
    rnd = rounder(precision)
    fr = FragReader(stream, precision, info.frag_bytes)
    gen = torch.Generator().manual_seed(11)
    ns_ = 32
    x = (torch.randn(ns_, 63, generator=gen) * 0.5).double().requires_grad_(True)
    d_raw = torch.randn(ns_, 5, generator=gen).double()
    d_raw[:, 4] = 0.0
    # forward in fp64 torch, keeping the activations
    hs, h = [], x
    for i, l in enumerate(fine.pts_linears):
        h = F.relu(F.linear(h, l.weight.double(), l.bias.double()))
        hs.append(h)
        if i == 4:
            h = torch.cat([x, h], -1)
    raw = F.linear(h, fine.output_linear.weight.double(), fine.output_linear.bias.double())
    (g_x,) = torch.autograd.grad(raw, x, d_raw)
    hs = [t.detach().numpy() for t in hs]

    def mask_tiles(tiles, layer):          # d h_layer tiles [32, ns] -> d z_layer (rows = features 32 t + i)
        return [np.where(hs[layer][:, 32 * t:32 * t + 32].T > 0, D, 0.0) for t, D in enumerate(tiles)]

    v = np.zeros((8, ns_))
    v[:5] = d_raw.numpy().T
    slabs = vec_slabs(v, KH, rnd)
    tile0, mfma = 0, 0
    tiles = dense_emul(fr, bias, tile0, len(slabs), NT, slabs); mfma += len(slabs) * NT; tile0 += NT       # head^T -> d h_7
    denc = None
    for i in range(7, 0, -1):
        slabs = repack(mask_tiles(tiles, i), KH, False, rnd)
        nt = NT + 2 if i == 5 else NT
        out = dense_emul(fr, bias, tile0, len(slabs), nt, slabs); mfma += len(slabs) * nt; tile0 += nt
        if i == 5:
            denc, tiles = out[:2], out[2:]
        else:
            tiles = out
    slabs = repack(mask_tiles(tiles, 0), KH, False, rnd)
    out = dense_emul(fr, bias, tile0, len(slabs), 2, slabs); mfma += len(slabs) * 2; tile0 += 2             # pts_linears[0]^T
    denc = [denc[0] + out[0], denc[1] + out[1]]
    assert tile0 == info.n_bias_tiles and mfma == info.mfma_per_block
    used = fr.pos * info.frag_bytes
    assert used <= info.stream_bytes and not stream[used:].any()
    # encoding-slot order -> reference columns (nrnerf_plan.h enc_col): half hh, slot q = te * 16 + r, row = tile_row(r, hh)
    got = np.zeros((ns_, 63))
    F0 = 5
    for te in range(2):
        for hh in range(2):
            for r in range(16):
                q = te * 16 + r
                if q == 0:
                    col = 2 if hh else 0
                elif q == 1:
                    col = -1 if hh else 1
                else:
                    pi, fn = (q - 2) // 2, (q - 2) % 2
                    fl, c = pi // 3, pi % 3
                    col = 3 + 6 * (hh * F0 + fl) + 3 * fn + c
                if col >= 0:
                    got[:, col] = denc[te][tile_row(r, hh)]
    tol = 1e-9 if precision == "f32" else 6e-2
    err = np.abs(got - g_x.numpy()).max()
    assert err <= tol * np.abs(g_x.numpy()).max(), (err, np.abs(g_x.numpy()).max())


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_backward_stream_with_the_view_dependent_head_reproduces_autograd(precision):
    """Training with use_viewdirs (csrc/nrnerf_train.h, trunk_bwd<.., VIEWS>): PlanB<.., true> starts with rgb_linear^T
    (d raw's colour channels -> d hv) and ONE layer that joins both branches of the head -- k over [d raw (sigma), d z_v], rows
    over [direction-encoding slots, h_7], weights = alpha_linear^T and the transposed FOLDED views layer (views_linears[0] o
    feature_linear) -- then the trunk's transposes as before.  Emulated in numpy against torch.autograd of the reference's
    head (rnh:284-304): gradient wrt the point's encoding and wrt the direction's encoding."""
    cfg = SceneConfig(N_importance=128, use_viewdirs=True)
    NT = 8
    scene, (rb, coarse, fine), info, stream, units, bias = _pack(cfg, precision, which=5)
    assert not bias.any(), "backward layers have no bias"
    KH = 1 if precision == "f32" else 8
    rnd = rounder(precision)
    fr = FragReader(stream, precision, info.frag_bytes)
    gen = torch.Generator().manual_seed(12)
    ns_ = 32
    x = (torch.randn(ns_, 63, generator=gen) * 0.5).double().requires_grad_(True)
    ev = (torch.randn(ns_, 27, generator=gen) * 0.5).double().requires_grad_(True)
    d_raw = torch.randn(ns_, 4, generator=gen).double()
    hs, h = [], x
    for i, l in enumerate(fine.pts_linears):
        h = F.relu(F.linear(h, l.weight.double(), l.bias.double()))
        hs.append(h)
        if i == 4:
            h = torch.cat([x, h], -1)
    lin = lambda m, t: F.linear(t, m.weight.double(), m.bias.double())
    alpha = lin(fine.alpha_linear, h)                                                        # rnh:285
    hv = F.relu(lin(fine.views_linears[0], torch.cat([lin(fine.feature_linear, h), ev], -1)))  # rnh:286-301
    raw = torch.cat([lin(fine.rgb_linear, hv), alpha], -1)                                   # rnh:303-304
    g_x, g_ev = torch.autograd.grad(raw, (x, ev), d_raw)
    hs = [t.detach().numpy() for t in hs]
    hvn = hv.detach().numpy()

    def mask_tiles(tiles, act):            # d h tiles [32, ns] -> d z (rows = features 32 t + i)
        return [np.where(act[:, 32 * t:32 * t + 32].T > 0, D, 0.0) for t, D in enumerate(tiles)]

    v = np.zeros((8, ns_))
    v[:4] = d_raw.numpy().T
    dr = vec_slabs(v, KH, rnd)
    tile0, mfma = 0, 0
    tiles = dense_emul(fr, bias, tile0, len(dr), NT // 2, dr); mfma += len(dr) * (NT // 2); tile0 += NT // 2          # rgb_linear^T -> d hv
    dv = repack(mask_tiles(tiles, hvn), KH, False, rnd)
    slabs = dr + dv
    out = dense_emul(fr, bias, tile0, len(slabs), 1 + NT, slabs); mfma += len(slabs) * (1 + NT); tile0 += 1 + NT        # the joined head layer
    dencv, tiles = out[0], out[1:]
    denc = None
    for i in range(7, 0, -1):
        slabs = repack(mask_tiles(tiles, hs[i]), KH, False, rnd)
        nt = NT + 2 if i == 5 else NT
        out = dense_emul(fr, bias, tile0, len(slabs), nt, slabs); mfma += len(slabs) * nt; tile0 += nt
        if i == 5:
            denc, tiles = out[:2], out[2:]
        else:
            tiles = out
    slabs = repack(mask_tiles(tiles, hs[0]), KH, False, rnd)
    out = dense_emul(fr, bias, tile0, len(slabs), 2, slabs); mfma += len(slabs) * 2; tile0 += 2
    denc = [denc[0] + out[0], denc[1] + out[1]]
    assert tile0 == info.n_bias_tiles and mfma == info.mfma_per_block
    used = fr.pos * info.frag_bytes
    assert used <= info.stream_bytes and not stream[used:].any()

    def from_slots(tiles_, L):             # encoding-slot order -> reference columns (nrnerf_plan.h enc_col)
        F0 = (L + 1) // 2
        got = np.zeros((ns_, 3 + 6 * L))
        for te, D in enumerate(tiles_):
            for hh in range(2):
                for r in range(16):
                    q = te * 16 + r
                    if q == 0:
                        col = 2 if hh else 0
                    elif q == 1:
                        col = -1 if hh else 1
                    else:
                        pi, fn = (q - 2) // 2, (q - 2) % 2
                        fl, c = pi // 3, pi % 3
                        col = 3 + 6 * (hh * F0 + fl) + 3 * fn + c if (fl < F0 and hh * F0 + fl < L) else -1
                    if col >= 0:
                        got[:, col] = D[tile_row(r, hh)]
                    else:
                        assert not D[tile_row(r, hh)].any(), "a slot without a column must carry no gradient"
        return got

    tol = 1e-9 if precision == "f32" else 6e-2
    for got, want, what in ((from_slots(denc, 10), g_x.numpy(), "point encoding"), (from_slots([dencv], 4), g_ev.numpy(), "direction encoding")):
        err = np.abs(got - want).max()
        assert err <= tol * np.abs(want).max(), (what, err, np.abs(want).max())


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_bender_backward_stream_reproduces_autograd_of_the_bender(precision):
    """Training of the ray bender (csrc/nrnerf_train_bend.h): nrnerf_pack_host which = 6 is the fp32 stream of PlanBB --
    network[4]^T .. network[1]^T, the latent rows of network[0]^T, then rigidity_network[2]^T, rigidity_network[1]^T --
    whatever the model's precision.  Emulating the kernel's dataflow (gradient of the offsets / of the logit as first B
    operand, relu masks from the forward activations) must reproduce torch.autograd's gradients wrt the latent inputs and
    wrt every layer's pre-activation."""
    cfg = SceneConfig(N_importance=128)
    scene, (rb, coarse, fine), info, stream, units, bias = _pack(cfg, precision, which=6)
    assert info.frag_bytes == 256 and not bias.any()
    fr = FragReader(stream, "f32", info.frag_bytes)
    ident = lambda x: x
    gen = torch.Generator().manual_seed(5)
    ns_ = 32
    x = (torch.randn(ns_, 35, generator=gen) * 0.5).double().requires_grad_(True)
    g_off = torch.randn(ns_, 3, generator=gen).double()
    g_logit = torch.randn(ns_, 1, generator=gen).double()
    zs, hs, h = [], [], x
    for i, l in enumerate(rb.network):
        z = F.linear(h, l.weight.double(), None if l.bias is None else l.bias.double())
        z.retain_grad()
        zs.append(z)
        h = F.relu(z) if i != len(rb.network) - 1 else z
        hs.append(h)
    off = h
    rzs, rhs, r = [], [], x[:, :3]
    for i, l in enumerate(rb.rigidity_network):
        z = F.linear(r, l.weight.double(), l.bias.double())
        z.retain_grad()
        rzs.append(z)
        r = F.relu(z) if i != len(rb.rigidity_network) - 1 else z
        rhs.append(r)
    ((off * g_off).sum() + (r * g_logit).sum()).backward()
    BD, RD = len(rb.network), len(rb.rigidity_network)

    def masked(tiles, acts):
        return [np.where(acts[:, 32 * t:32 * t + 32].T > 0, D, 0.0) for t, D in enumerate(tiles)]

    def as_rows(tiles):
        return np.concatenate(tiles, 0).T          # [nsamp, 32 * nt]

    tile0 = mfma = 0
    v = np.zeros((8, ns_)); v[:3] = g_off.numpy().T
    slabs = vec_slabs(v, 1, ident)
    tiles = dense_emul(fr, bias, tile0, len(slabs), 2, slabs); mfma += len(slabs) * 2; tile0 += 2          # network[4]^T
    for i in range(BD - 2, -1, -1):                # tiles = d h_i
        dz = masked(tiles, hs[i].detach().numpy())
        np.testing.assert_allclose(as_rows(dz), zs[i].grad.numpy(), rtol=1e-9, atol=1e-12)
        slabs = repack(dz, 1, False)
        nt = 2 if i > 0 else 1
        tiles = dense_emul(fr, bias, tile0, len(slabs), nt, slabs); mfma += len(slabs) * nt; tile0 += nt   # network[i]^T
    np.testing.assert_allclose(as_rows(tiles), x.grad.numpy()[:, 3:35], rtol=1e-9, atol=1e-12)             # latent columns
    v = np.zeros((8, ns_)); v[0] = g_logit.numpy()[:, 0]
    slabs = vec_slabs(v, 1, ident)
    tiles = dense_emul(fr, bias, tile0, len(slabs), 1, slabs); mfma += len(slabs); tile0 += 1              # rigidity_network[2]^T
    for i in range(RD - 2, -1, -1):
        dz = masked(tiles, rhs[i].detach().numpy())
        np.testing.assert_allclose(as_rows(dz), rzs[i].grad.numpy(), rtol=1e-9, atol=1e-12)
        if i > 0:
            slabs = repack(dz, 1, False)
            tiles = dense_emul(fr, bias, tile0, len(slabs), 1, slabs); mfma += len(slabs); tile0 += 1      # rigidity_network[i]^T
    assert tile0 == info.n_bias_tiles and mfma == info.mfma_per_block
    used = fr.pos * info.frag_bytes
    assert used <= info.stream_bytes and not stream[used:].any()


# ---------------------------------------------------------------------------------------------------------------------------
# The run-time-parameterised kernel (csrc/nrnerf_generic.h): its layer PROGRAM and packed images (nrnerf_pack_host 7 / 8 / 9),
# executed in numpy exactly as the kernel walks them -- per layer and output tile the fragments (tile, k-slab) against the
# buffers E (network input) / H (hidden, overwritten in place) / V (second input), accumulators seeded with the bias table in
# the D-tile register order -- must reproduce the plain F.linear network of the reference's modules.
# ---------------------------------------------------------------------------------------------------------------------------
def _run_generic_program(info, stream, units, bias, precision, E, V):
    f32 = precision == "f32"
    KH = 4 if f32 else 8              # k per lane and fragment: 16 bytes per lane in every precision (fp32: four 32x32x2 MFMAs per fragment)
    KS, FB = 2 * KH, info.frag_bytes
    assert FB == 1024
    u = units.astype(np.int64)
    n_layers = int(u[0])
    layers = u[1:1 + 11 * n_layers].reshape(n_layers, 11)
    ke, kv, kh, lat = (int(x) for x in u[1 + 11 * n_layers:1 + 11 * n_layers + 4])
    n = E.shape[0]
    bufs = {0: np.zeros((n, ke)), 2: np.zeros((n, kv)), 1: np.zeros((n, kh)), 3: np.zeros((n, 8))}
    bufs[0][:, :E.shape[1]] = E
    bufs[2][:, :V.shape[1]] = V
    if f32:
        words = stream.view(np.float32).astype(np.float64)
    else:
        raw16 = stream.view(np.uint16)
    for (w_frag, bias_tile, nt, src0, ns0, src1, ns1, dst, relu, o_col, o_rows) in layers:
        ns = ns0 + ns1
        out = np.zeros((n, 32 * nt))
        for t in range(nt):
            acc = np.zeros((32, n))                       # D tile: row = feature, column = sample
            for h in range(2):
                for r in range(16):
                    acc[tile_row(r, h)] = bias[(bias_tile + t) * 32 + h * 16 + r]
            for sl in range(ns):
                src, s = (src0, sl) if sl < ns0 else (src1, sl - ns0)
                fi = w_frag + t * ns + sl
                if f32:
                    fr = words[fi * 256:(fi + 1) * 256].reshape(64, 4)
                else:
                    bits = raw16[fi * 512:(fi + 1) * 512].reshape(64, 8)
                    as_f16 = precision == "f16" or src != 1          # fragments against E / V are f16 (nrnerf_generic.h)
                    fr = bits.view(np.float16).astype(np.float64) if as_f16 else (bits.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
                A = np.zeros((32, KS))
                for lane in range(64):
                    A[lane & 31, KH * (lane >> 5):KH * (lane >> 5) + KH] = fr[lane]
                acc += A @ bufs[src][:, s * KS:(s + 1) * KS].T
            out[:, 32 * t:32 * t + 32] = acc.T
        if relu:
            out = np.maximum(out, 0.0)
        if dst == 1:
            bufs[1][:, :32 * nt] = out
        else:
            bufs[3][:, o_col:o_col + o_rows] = out[:, :o_rows]
    return bufs[3], (ke, kv, kh, lat)


def _posenc(x, L):
    cols = [x]
    for k in range(L):
        cols += [np.sin(x * 2.0 ** k), np.cos(x * 2.0 ** k)]
    return np.concatenate(cols, -1)


@pytest.mark.parametrize("precision", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("cfg_kw", [dict(netdepth=6, netwidth=192, netdepth_fine=10, netwidth_fine=320, multires=8, latent_size=16),
                                    dict(netdepth=7, netwidth=96, netwidth_fine=160, multires=6, multires_views=2, use_viewdirs=True),
                                    dict(netdepth=4, netwidth=72, ray_bending=False),
                                    dict(netwidth=448, multires=12, latent_size=24, ray_bending=False, time_conditioned_baseline=True,
                                         use_viewdirs=True, multires_views=6)],
                         ids=["192_320_latent16", "viewdirs_96_160", "shallow_72_no_bender", "time_conditioned_448"])
def test_generic_layer_programs_reproduce_the_networks(cfg_kw, precision):
    cfg = SceneConfig(N_importance=64, **cfg_kw)
    scene, (rb, coarse, fine), info_c, st_c, un_c, bi_c = _pack(cfg, precision, 7)
    _, _, info_f, st_f, un_f, bi_f = _pack(cfg, precision, 8)
    g = np.random.default_rng(5)
    n = 40
    pts = g.normal(size=(n, 3)) * 0.3
    dirs = g.normal(size=(n, 3))
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    lat = g.normal(size=(n, cfg.latent_size)) * 0.1
    enc = _posenc(pts, cfg.multires)
    encd = _posenc(dirs, cfg.multires_views) if cfg.use_viewdirs else np.zeros((n, 3))
    x_in = np.concatenate([enc, lat], -1) if cfg.time_conditioned_baseline else enc
    tol = 1e-5 if precision == "f32" else (0.08 if precision == "bf16" else 0.01)
    for net, info, st, un, bi in ((coarse, info_c, st_c, un_c, bi_c), (fine, info_f, st_f, un_f, bi_f)):
        O, (ke, kv, kh, latw) = _run_generic_program(info, st, un, bi, precision, x_in, encd)
        assert ke >= x_in.shape[1] and ke % 16 == 0 and kh >= int(net.W) and latw == (cfg.latent_size if cfg.time_conditioned_baseline else 0)
        # the same network with torch (float64 weights of the modules)
        with torch.no_grad():
            x = torch.from_numpy(x_in)
            h = x
            for i, lin in enumerate(net.pts_linears):
                h = F.relu(F.linear(h, lin.weight.double(), lin.bias.double()))
                if i in net.skips and i < len(net.pts_linears) - 1:
                    h = torch.cat([x, h], -1)
            if net.use_viewdirs:
                alpha = F.linear(h, net.alpha_linear.weight.double(), net.alpha_linear.bias.double())
                feat = F.linear(h, net.feature_linear.weight.double(), net.feature_linear.bias.double())
                hv = F.relu(F.linear(torch.cat([feat, torch.from_numpy(encd)], -1), net.views_linears[0].weight.double(), net.views_linears[0].bias.double()))
                want = torch.cat([F.linear(hv, net.rgb_linear.weight.double(), net.rgb_linear.bias.double()), alpha], -1).numpy()
            else:
                want = F.linear(h, net.output_linear.weight.double(), net.output_linear.bias.double()).numpy()
        got = O[:, :want.shape[1]]
        # (16-bit modes: the emulation keeps fp64 activations, only the WEIGHTS carry the 16-bit rounding)
        assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max()), (np.abs(got - want).max(), np.abs(want).max())
    if rb is not None:
        _, _, info_b, st_b, un_b, bi_b = _pack(cfg, precision, 9)
        assert info_b.frag_bytes == 1024 and info_b.stream_bytes % 1024 == 0   # (the bender program is always fp32: 4 k per lane)
        O, (ke, kv, kh, latw) = _run_generic_program(info_b, st_b, un_b, bi_b, "f32", np.concatenate([pts, lat], -1), pts)
        assert latw == cfg.latent_size and kv == 16
        with torch.no_grad():
            h = torch.from_numpy(np.concatenate([pts, lat], -1))
            for i, lin in enumerate(rb.network):
                h = F.linear(h, lin.weight.double(), lin.bias.double() if lin.bias is not None else None)
                if i < len(rb.network) - 1:
                    h = F.relu(h)
            r = torch.from_numpy(pts)
            for i, lin in enumerate(rb.rigidity_network):
                r = F.linear(r, lin.weight.double(), lin.bias.double())
                if i < len(rb.rigidity_network) - 1:
                    r = F.relu(r)
        assert np.abs(O[:, :3] - h.numpy()).max() <= 1e-5 and np.abs(O[:, 3:4] - r.numpy()).max() <= 1e-5


def _run_generic_bwd_program(info, stream, units, bias, precision, d_raw, acts):
    """The backward-data program (nrnerf_pack_host 13 / 14) as gen_kernel mode 2 walks it: H starts as zeros with the rows of d raw at the
    program's d-raw column; every layer = transposed fragments against H (from a column offset per source), outputs masked by the saved
    activation of slot mask_idx, written back to H and kept as d_pre[save_idx], or (dst 4 / 5 / 6) handed out."""
    f32 = precision == "f32"
    KH = 4 if f32 else 8
    KS = 2 * KH
    assert info.frag_bytes == 1024
    u = units.view(np.int32).astype(np.int64)           # (slots of -1 = none)
    n_layers = int(u[0])
    layers = u[1:1 + 15 * n_layers].reshape(n_layers, 15)
    ke, kv, kh, lat, draw_col = (int(x) for x in u[1 + 15 * n_layers:1 + 15 * n_layers + 5])
    n = d_raw.shape[0]
    H = np.zeros((n, kh))
    H[:, draw_col:draw_col + d_raw.shape[1]] = d_raw
    words = stream.view(np.float32).astype(np.float64) if f32 else None
    raw16 = None if f32 else stream.view(np.uint16)
    d_pre, outs = {}, {}
    for (w_frag, bias_tile, nt, src0, ns0, src1, ns1, dst, relu, o_col, o_rows, save_idx, mask_idx, boff0, boff1) in layers:
        assert src0 == 1 and (ns1 == 0 or src1 == 1) and relu == 0 and not bias[bias_tile * 32:(bias_tile + nt) * 32].any()
        ns = ns0 + ns1
        out = np.zeros((n, 32 * nt))
        for t in range(nt):
            acc = np.zeros((32, n))
            for sl in range(ns):
                s, boff = (sl, boff0) if sl < ns0 else (sl - ns0, boff1)
                fi = w_frag + t * ns + sl
                if f32:
                    fr = words[fi * 256:(fi + 1) * 256].reshape(64, 4)
                else:
                    bits = raw16[fi * 512:(fi + 1) * 512].reshape(64, 8)
                    fr = bits.view(np.float16).astype(np.float64) if precision == "f16" else (bits.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
                A = np.zeros((32, KS))
                for lane in range(64):
                    A[lane & 31, KH * (lane >> 5):KH * (lane >> 5) + KH] = fr[lane]
                acc += A @ H[:, boff + s * KS:boff + (s + 1) * KS].T
            out[:, 32 * t:32 * t + 32] = acc.T
        if dst == 1:
            if mask_idx >= 0:
                w = acts[mask_idx].shape[1]
                out[:, :w] *= acts[mask_idx] > 0
                out[:, w:] = 0.0
            H[:, :32 * nt] = out
            if save_idx >= 0:
                d_pre[save_idx] = out.copy()
        else:
            assert dst in (4, 5, 6)
            outs[dst] = out[:, :o_rows]
    return d_pre, outs


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("cfg_kw", [dict(netdepth=6, netwidth=192, netdepth_fine=5, netwidth_fine=320, skips=(2,), multires=8, latent_size=16),
                                    dict(netdepth=4, netwidth=96, netwidth_fine=160, multires=6, multires_views=2, use_viewdirs=True, skips=(1,)),
                                    dict(netdepth=3, netwidth=72, skips=(), ray_bending=False),
                                    dict(netdepth=5, netwidth=132, skips=(2,), multires=5, latent_size=24, ray_bending=False, time_conditioned_baseline=True,
                                         use_viewdirs=True, multires_views=3)],
                         ids=["192_320_skip2", "viewdirs_96_160", "no_skip_72", "time_conditioned_viewdirs_132"])
def test_generic_backward_data_programs_reproduce_autograd(cfg_kw, precision):
    """Training of a non-compiled architecture (csrc/nrnerf_api.cpp::gen_pack_mlp_bwd, gen_kernel mode 2): the transposed layer program,
    run in numpy as the kernel walks it on the activations torch saved, must give torch's (float64) gradient of every pre-activation, of the
    network input [encoding | latent code] (both layers that read it) and of the direction encoding -- plain and view-dependent heads
    (rnh:284-304: d sigma joins the colour branch's gradient in ONE layer [feature_linear^T | alpha_linear^T]), skip at any layer / none."""
    cfg = SceneConfig(N_importance=64, **cfg_kw)
    scene, (rb, coarse, fine), info_c, st_c, un_c, bi_c = _pack(cfg, precision, 13)
    _, _, info_f, st_f, un_f, bi_f = _pack(cfg, precision, 14)
    g = np.random.default_rng(7)
    n = 24
    pts = g.normal(size=(n, 3)) * 0.3
    dirs = g.normal(size=(n, 3))
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    lat = g.normal(size=(n, cfg.latent_size)) * 0.1
    enc = _posenc(pts, cfg.multires)
    encd = _posenc(dirs, cfg.multires_views) if cfg.use_viewdirs else np.zeros((n, 3))
    x_in = np.concatenate([enc, lat], -1) if cfg.time_conditioned_baseline else enc
    d_raw = g.normal(size=(n, 4))
    tol = 1e-5 if precision == "f32" else 0.06
    for net, info, st, un, bi in ((coarse, info_c, st_c, un_c, bi_c), (fine, info_f, st_f, un_f, bi_f)):
        D = len(net.pts_linears)
        x = torch.from_numpy(x_in).requires_grad_(True)
        ev = torch.from_numpy(encd).requires_grad_(True)
        zs, hs, h = [], [], x
        for i, lin in enumerate(net.pts_linears):
            z = F.linear(h, lin.weight.double(), lin.bias.double())
            z.retain_grad()
            zs.append(z)
            h = F.relu(z)
            hs.append(h)
            if i in net.skips and i < D - 1:
                h = torch.cat([x, h], -1)
        if net.use_viewdirs:
            alpha = F.linear(h, net.alpha_linear.weight.double(), net.alpha_linear.bias.double())
            feat = F.linear(h, net.feature_linear.weight.double(), net.feature_linear.bias.double())
            feat.retain_grad()
            zv = F.linear(torch.cat([feat, ev], -1), net.views_linears[0].weight.double(), net.views_linears[0].bias.double())
            zv.retain_grad()
            hv = F.relu(zv)
            out = torch.cat([F.linear(hv, net.rgb_linear.weight.double(), net.rgb_linear.bias.double()), alpha], -1)
        else:
            out = F.linear(h, net.output_linear.weight.double(), net.output_linear.bias.double())[:, :4]
        (out * torch.from_numpy(d_raw)).sum().backward()
        acts = {i: hs[i].detach().numpy() for i in range(D)}
        if net.use_viewdirs:
            acts[D] = feat.detach().numpy()
            acts[D + 1] = hv.detach().numpy()
        d_pre, outs = _run_generic_bwd_program(info, st, un, bi, precision, d_raw, acts)

        def close(got, want, what):
            assert np.abs(got - want).max() <= tol * max(1e-3, np.abs(want).max()), (what, np.abs(got - want).max(), np.abs(want).max())
        for i in range(D):
            close(d_pre[i][:, :zs[i].shape[1]], zs[i].grad.numpy(), f"d_pre[{i}]")
        d_in = outs[4] + (outs[5] if 5 in outs else 0.0)
        assert (5 in outs) == any(0 <= int(k) <= D - 2 for k in net.skips)
        close(d_in, x.grad.numpy(), "d input")
        if net.use_viewdirs:
            close(d_pre[D][:, :feat.shape[1]], feat.grad.numpy(), "d feature")
            close(d_pre[D + 1][:, :zv.shape[1]], zv.grad.numpy(), "d pre of the colour branch")
            close(outs[6], ev.grad.numpy(), "d direction encoding")


def test_which_generic_shapes_render_with_exact_jacobian_directions():
    """nrnerf_pack_host 7 works for any supported shape (also a compiled one).  Exact Jacobian view directions (rnh:358-385) on a
    non-compiled trunk (round 6): rendered natively when the ray BENDER has one of the two compiled shapes (its divergence kernel
    supplies J d) and the handle is not an "f16" one (no training images); any other bender / precision stays unsupported there, i.e.
    goes to the reference."""
    lib = _lib.load()
    info = _lib.PackedInfo()

    def status(cfg_kw, precision):
        rb, coarse, fine = build_modules(make_scene(SceneConfig(N_importance=64, **cfg_kw), 3))
        desc, keep = build_model_desc(coarse, fine, precision, 0)
        return lib.nrnerf_pack_host(C.byref(desc), 7, C.byref(info), None, 0, C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)())

    exact = dict(netwidth=192, use_viewdirs=True, approx_nonrigid_viewdirs=False)
    assert status(exact, "f32") == 0 and status(exact, "bf16") == 0 and status(dict(exact, bend_depth=7), "f32") == 0
    assert status(exact, "f16") == _lib.ERR_UNSUPPORTED
    assert status(dict(exact, bend_hidden=96), "f32") == _lib.ERR_UNSUPPORTED
    assert status(dict(exact, latent_size=16), "f32") == _lib.ERR_UNSUPPORTED
    coarse = fine = None
    # a width beyond the generic kernel's limit
    cfg = SceneConfig(N_importance=64, netwidth=640)
    rb, coarse, fine = build_modules(make_scene(cfg, 3))
    desc, keep = build_model_desc(coarse, fine, "f32", 0)
    assert lib.nrnerf_pack_host(C.byref(desc), 7, C.byref(info), None, 0, C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)()) == _lib.ERR_UNSUPPORTED


def test_which_architectures_have_a_backward_data_program():
    """gen_trainable (csrc/nrnerf_api.cpp) through nrnerf_pack_host 13 / 14: fp32 and bf16 handles of anything the run-time-parameterised kernel
    renders, the view-dependent head up to width 480 (the rows of d raw sit beside the activations in a 512-column buffer); not f16 handles
    (they train through a bf16 one), not the time-conditioned baseline next to a bender (train.py:574-576 rules that out anyway), and with the
    training handle's description (no exact_viewdirs, render_rays_train computes the directions) also the exact-direction models."""
    lib = _lib.load()
    info = _lib.PackedInfo()

    def status(cfg_kw, precision, which=13, flags=0):
        rb, coarse, fine = build_modules(make_scene(SceneConfig(N_importance=64, **cfg_kw), 3))
        desc, keep = build_model_desc(coarse, fine, precision, 0, flags)
        return lib.nrnerf_pack_host(C.byref(desc), which, C.byref(info), None, 0, C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)())

    for kw in (dict(), dict(netwidth=192), dict(netwidth=512, netdepth=16, skips=(7,)), dict(netwidth=480, use_viewdirs=True),
               dict(netwidth=132, ray_bending=False, time_conditioned_baseline=True, latent_size=24)):
        for prec in ("f32", "bf16"):
            assert status(kw, prec) == 0 and status(kw, prec, 14) == 0, (kw, prec)
    assert status(dict(netwidth=192), "f16") == _lib.ERR_UNSUPPORTED
    assert status(dict(netwidth=512, use_viewdirs=True), "bf16") == _lib.ERR_UNSUPPORTED
    assert status(dict(netwidth=190), "f32") == _lib.ERR_UNSUPPORTED                     # (rows of 4 elements)
    exact = dict(netwidth=192, use_viewdirs=True, approx_nonrigid_viewdirs=False)
    assert status(exact, "f32") == 0                                   # (round 6: the bender has a compiled shape, nrnerf_render computes J d itself)
    assert status(dict(exact, bend_hidden=96), "f32") == _lib.ERR_UNSUPPORTED
    assert status(dict(exact, bend_hidden=96), "f32", flags=_lib.MODEL_PY_TRAINING_HANDLE) == 0
