// nrnerf_plan.h -- the single source of truth shared by the host weight packer and the
// device network kernel: which MFMA fragments exist, in which order they are streamed, how
// they are grouped into LDS staging units, and which reference weight element every fragment
// element holds.
//
// Dataflow the plan describes: every wave owns one "block" of 32
// consecutive samples of one ray.  Each layer is computed TRANSPOSED, D^T = W * H^T, with the
// weights as the MFMA A operand and the activations as the B operand
// (v_mfma_f32_32x32x16_{bf16,f16} or v_mfma_f32_32x32x2_f32).  The D tile (32 output
// features x 32 samples) lands with lane = sample and registers = features, which -- up to a
// fixed permutation of the k index -- is exactly the B-operand layout of the next layer.  The
// permutation is folded into the packed weights here, so activations never leave registers.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NRN_HD __host__ __device__
#else
#define NRN_HD
#endif

// build-time tuning knobs of the LDS weight ring (Makefile: TUNE="-DNRN_UNIT_BYTES=32768 -DNRN_RING=3")
#ifndef NRN_UNIT_BYTES
#define NRN_UNIT_BYTES 16384
#endif
#ifndef NRN_RING
#define NRN_RING 4
#endif

namespace nrn {

constexpr NRN_HD int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr NRN_HD int imax(int a, int b) { return a > b ? a : b; }

// Row (0..31) of a 32x32 MFMA D tile held by accumulator register r (0..15) of a lane in
// half h (= lane >> 5).  CDNA4 C/D layout: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*h.
constexpr NRN_HD int tile_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Operand shape of one precision policy.  KH = k elements per lane per MFMA.
//   bf16/f16: 32x32x16, a lane holds 8 consecutive k (KH = 8), lanes 32..63 hold k 8..15
//   f32     : 32x32x2,  a lane holds 1 k         (KH = 1), lanes 32..63 hold k = 1
template <int KH_, bool SPLIT_ = (KH_ != 1)>
struct Shape {
    static constexpr int KH = KH_;
    static constexpr int KS = 2 * KH_;             // k per MFMA
    static constexpr int SP = 16 / KH_;            // B-operand slabs produced by one 32-feature D tile
    static constexpr int ELEM_BYTES = (KH_ == 1) ? 4 : 2;
    static constexpr int FRAG_BYTES = 64 * KH_ * ELEM_BYTES;   // one A fragment: 1024 (16-bit) / 256 (f32)
    // SPLIT: the bender / rigidity MLPs are evaluated with a 3-term split product on the f16 matrix pipe
    //   W x ~= Whi xhi + 2^-11 (Whi xlo + Wlo xhi)     hi = f16(v), lo = f16((v - hi) * 2^11)
    // which is fp32-equivalent (bent-point rmse 2e-8): the offsets feed a 2^9-frequency encoding.  The lo parts are kept
    // pre-scaled by 2^11 (exact) and accumulated separately: unscaled they are f16 subnormals (w*2^-12 < 6.1e-5), which
    // the matrix pipe does not keep (measured: the Wlo term vanished).  Split layers stream two A fragments (hi, lo) per
    // (tile, slab) and issue three MFMAs, and packing every activation into hi + lo is VALU-heavy (8.8 VALU per MFMA in
    // the stand-alone bender kernel).  The precision ladder of the library:
    //   "f32"   exact fp32 everywhere;
    //   "f16"   f16 trunk + SPLIT bender: the accurate 16-bit mode (74 dB vs the fp32 render on the fitted model);
    //   "bf16"  bf16 trunk + SINGLE-product f16 bender (SPLIT = false): a third of the bender's MFMAs and less than half
    //           of its VALU work; measured cost on the fitted model: the bender's own error floor is 71-74 dB, below the
    //           bf16 trunk's 66 dB (tools/experiments/bender_precision_probe.py; with 5x larger offsets 60-67 dB).
    static constexpr bool SPLIT = SPLIT_;
    static constexpr float LO_SCALE = 2048.0f;
    static constexpr int UNIT_BYTES = NRN_UNIT_BYTES;             // staging granularity of the LDS ring
    static constexpr int UNIT_FRAGS = UNIT_BYTES / FRAG_BYTES;    // 16 (16-bit) / 64 (f32) fragments per unit
};

template <int W_, int D_, int SKIP_, int L_, int BW_, int BD_, int RW_, int RD_, int LAT_, int LV_ = 4, int TCB_ = 0>
struct ArchT {
    static constexpr int W = W_, D = D_, SKIP = SKIP_, L = L_, LV = LV_;     // LV: frequencies of the direction encoding
    // TCB: "time-conditioned baseline" (run_nerf_helpers.py:207-209, 273-282): no bender, the latent code is appended
    // to the network input and again at the skip connection
    static constexpr int TCB = TCB_;
    static constexpr int BW = BW_, BD = BD_, RW = RW_, RD = RD_, LAT = LAT_;
    static_assert(W_ % 32 == 0 && BW_ % 32 == 0 && RW_ % 32 == 0, "widths must be multiples of 32");
    static_assert(LAT_ % 8 == 0, "latent size must be a multiple of 8");
};

// ---------------------------------------------------------------------------------------
// Input vectors of the three first layers, as per-lane-half slot lists.
// ---------------------------------------------------------------------------------------
// Positional encoding (reference Embedder, run_nerf_helpers.py:120-150; column order
// [x y z, sin(2^0 xyz), cos(2^0 xyz), sin(2^1 xyz), ...]).  The two lane halves of a sample
// split the frequencies: half 0 evaluates f in [0, F0), half 1 f in [F0, L), so both run the
// same code with a different scale.  Per-half slot q: 0,1 = identity coords (x,y | z,0), then
// for each local frequency fl and coordinate c the pair (sin, cos).
constexpr NRN_HD int enc_F0(int L) { return (L + 1) / 2; }
constexpr NRN_HD int enc_slots(int L) { return 2 + 6 * enc_F0(L); }
constexpr NRN_HD int enc_col(int L, int h, int q) {
    if (q == 0) return h ? 2 : 0;
    if (q == 1) return h ? -1 : 1;
    int pi = (q - 2) / 2, fn = (q - 2) % 2, fl = pi / 3, c = pi % 3;
    int f = h * enc_F0(L) + fl;
    if (fl >= enc_F0(L) || f >= L) return -1;
    return 3 + 6 * f + 3 * fn + c;
}
// Slabs of the trunk's input: the xyz encoding, plus (TCB) the latent code as extra slabs, element (s,h,e) of the
// latent part = latent[(2s+h)*KH + e].
template <class SH, class A> constexpr NRN_HD int ns_enc_xyz() { return cdiv(enc_slots(A::L), SH::KH); }
template <class SH, class A> constexpr NRN_HD int ns_trunk_in() { return ns_enc_xyz<SH, A>() + (A::TCB ? cdiv(A::LAT, 2 * SH::KH) : 0); }

// Bender input [xyz, latent] (run_nerf_helpers.py:525): logical vector
//   v[0..2] = xyz, v[3..7] = 0 (keeps the latent 8-aligned), v[8 .. 8+LAT) = latent.
// Lane half h, slab s, element e holds v[(2s+h)*KH + e].
constexpr NRN_HD int bin_len(int LAT) { return 8 + LAT; }
constexpr NRN_HD int bin_col(int idx, int LAT) {
    if (idx < 3) return idx;
    if (idx < 8) return -1;
    return (idx - 8 < LAT) ? 3 + (idx - 8) : -1;
}
// Rigidity input = xyz only (run_nerf_helpers.py:546): v[0..2] = xyz, rest 0.
constexpr NRN_HD int rin_len() { return 8; }
constexpr NRN_HD int rin_col(int idx) { return (idx < 3) ? idx : -1; }

// ---------------------------------------------------------------------------------------
// Layer list
// ---------------------------------------------------------------------------------------
enum LayerKind : int {
    LK_BEND_IN = 0, LK_BEND_HID, LK_BEND_OUT, LK_RIG_IN, LK_RIG_HID, LK_RIG_OUT,
    LK_TR_IN, LK_TR_HID, LK_TR_SKIP, LK_HEAD,
    // view-dependent head (reference NeRF.forward, run_nerf_helpers.py:284-304)
    LK_ALPHA,    // alpha_linear:   trunk output -> 1   (rows duplicated for both lane halves)
    LK_FEAT,     // feature_linear: trunk output -> W   (no activation) -- not a layer of any plan any more: folded into LK_VIEWS
    LK_VIEWS,    // views_linears[0] o feature_linear: [trunk output W, direction encoding] -> W/2, relu; slabs: direction encoding first
    LK_RGB,      // rgb_linear: W/2 -> 3
    // backward-data layers of the trunk (training, nrnerf_train.h): dX = W^T dY, the A operand holds W^T
    LK_B_HEAD,   // output_linear^T:   d raw (C channels) -> d h_{D-1}
    LK_B_HID,    // pts_linears[i]^T:  W -> W
    LK_B_SKIP,   // pts_linears[SKIP+1]^T: W -> [encoding slots (2 tiles), W]
    LK_B_IN,     // pts_linears[0]^T:  W -> encoding slots (2 tiles)
    // backward-data layers of the bender / rigidity MLPs (training, nrnerf_train_bend.h)
    LK_BB_OUT,   // network[BD-1]^T:          d offsets (3) -> BW
    LK_BB_HID,   // network[i]^T:             BW -> BW
    LK_BB_IN,    // network[0]^T:             BW -> the LAT latent inputs (the xyz inputs carry no gradient)
    LK_BR_OUT,   // rigidity_network[RD-1]^T: d logit (1) -> RW
    LK_BR_HID,   // rigidity_network[i]^T:    RW -> RW   (rigidity_network[0]^T is not needed: its input is xyz)
    // backward-data layers of the view-dependent head (training; they replace LK_B_HEAD at the front of the trunk's plan)
    LK_B_RGB,    // rgb_linear^T: d raw (channels 0..2) -> d hv (W/2: the hidden layer of the colour branch)
    LK_B_VHEAD   // [alpha_linear; views_linears[0] o feature_linear]^T: [d raw (channel 3 = sigma), d z_v (W/2)] ->
                 //   [direction-encoding slots (1 tile), d h_{D-1} (W)]: both branches of the head land in the same accumulators
};

struct LayerSpec {
    int kind;     // LayerKind
    int index;    // index into the reference ModuleList (network[i] / rigidity_network[i] / pts_linears[i])
    int ns;       // input slabs per output tile
    int nt;       // output tiles of 32 rows
    int tile0;    // global index of the layer's first tile
    int split;    // 1: 3-term split product (2 A fragments and 3 MFMAs per slab)
};

struct TileInfo {
    int layer;        // index into layers[]
    int t;            // tile within the layer
    int gbase;        // index, in the whole stream, of this tile's first fragment
    int gstride;      // distance between the fragments of consecutive slabs of this tile
};
// Fragment (tile t, slab s, part) of a layer sits at  gbase(t) + s * gstride(t) + part   (part = 1: the lo half of a
// split layer).  Tiles are packed in PAIRS with their slabs interleaved -- (2p,0) (2p+1,0) (2p,1) (2p+1,1) ... -- so
// that a wave runs two independent accumulator chains while still consuming the stream front to back: a dependent
// MFMA that is not issued strictly back-to-back waits ~43 extra cycles for its accumulator (MI355X_MICROARCH.md),
// alternating between two chains hides that.  An odd last tile is packed alone.

// The stream is cut into fixed units of UNIT_BYTES regardless of tile boundaries; fragment g lives in
// unit g / UNIT_FRAGS, which the kernel stages in LDS ring slot (g / UNIT_FRAGS) % RING.  Everything is
// compile-time, so the kernel needs no table: it knows statically when a fragment index crosses into a
// new unit.  The unit count is padded to a multiple of RING so the slot of unit 0 is the same every pass.
constexpr int RING = NRN_RING;

constexpr int MAX_LAYERS = 40;
constexpr int MAX_TILES = 256;

struct Tables {
    LayerSpec layers[MAX_LAYERS];
    TileInfo tiles[MAX_TILES];
    int nlayers, ntiles, nfrags, nunits, nunits_padded, mfma_per_block;
};

// stream positions of every tile's fragments (pairs of tiles with interleaved slabs, see TileInfo) and the totals
template <class SH>
constexpr void place_fragments(Tables& T) {
    int g = 0, mf = 0;
    for (int l = 0; l < T.nlayers; ++l) {
        const int fps = 1 + T.layers[l].split;          // fragments per (tile, slab)
        const int ns = T.layers[l].ns, nt = T.layers[l].nt, t0 = T.layers[l].tile0;
        for (int p = 0; p + 1 < nt; p += 2) {
            T.tiles[t0 + p] = TileInfo{l, p, g, 2 * fps};
            T.tiles[t0 + p + 1] = TileInfo{l, p + 1, g + fps, 2 * fps};
            g += 2 * ns * fps;
        }
        if (nt & 1) {
            T.tiles[t0 + nt - 1] = TileInfo{l, nt - 1, g, fps};
            g += ns * fps;
        }
        mf += nt * ns * (T.layers[l].split ? 3 : 1);
    }
    T.nfrags = g;
    T.nunits = cdiv(g, SH::UNIT_FRAGS);
    T.nunits_padded = cdiv(T.nunits, RING) * RING;
    T.mfma_per_block = mf;
}

// TRUNK = false: only the bender / rigidity layers (the stand-alone bender kernel, nrnerf_bend.h; needs HAS_BEND)
template <class SH, class A, bool HAS_BEND, bool VIEWS = false, bool TRUNK = true>
constexpr Tables build_tables() {
    constexpr int KH = SH::KH, SP = SH::SP;
    constexpr int NS_ENC = ns_trunk_in<SH, A>();
    constexpr int NS_ENCV = cdiv(enc_slots(A::LV), KH);
    constexpr int NS_BIN = cdiv(bin_len(A::LAT), 2 * KH);
    constexpr int NS_RIN = cdiv(rin_len(), 2 * KH);
    constexpr int NT_W = A::W / 32, NT_BW = A::BW / 32, NT_RW = A::RW / 32;
    Tables T{};
    int nl = 0, tile0 = 0;
    auto add = [&](int kind, int index, int ns, int nt) {
        const int split = (SH::SPLIT && kind <= LK_RIG_OUT) ? 1 : 0;
        T.layers[nl] = LayerSpec{kind, index, ns, nt, tile0, split};
        tile0 += nt;
        ++nl;
    };
    if (HAS_BEND) {
        add(LK_BEND_IN, 0, NS_BIN, NT_BW);
        for (int i = 1; i < A::BD - 1; ++i) add(LK_BEND_HID, i, NT_BW * SP, NT_BW);
        add(LK_BEND_OUT, A::BD - 1, NT_BW * SP, 1);
        add(LK_RIG_IN, 0, NS_RIN, NT_RW);
        for (int i = 1; i < A::RD - 1; ++i) add(LK_RIG_HID, i, NT_RW * SP, NT_RW);
        add(LK_RIG_OUT, A::RD - 1, NT_RW * SP, 1);
    }
    if (TRUNK) {
        add(LK_TR_IN, 0, NS_ENC, NT_W);
        for (int i = 1; i < A::D; ++i) {
            if (i - 1 == A::SKIP) add(LK_TR_SKIP, i, NS_ENC + NT_W * SP, NT_W);
            else add(LK_TR_HID, i, NT_W * SP, NT_W);
        }
        if (VIEWS) {
            add(LK_ALPHA, 0, NT_W * SP, 1);
            // feature_linear has no layer of its own: no nonlinearity sits between it and views_linears[0] (rnh:286-301), so the
            // packer folds it into that layer's hidden columns, W_v[:, :W] W_f (and W_v[:, :W] b_f into the bias): 65 536 of
            // the 593 408 MAC per sample less, and the trunk output feeds the views layer directly
            add(LK_VIEWS, 0, NS_ENCV + NT_W * SP, NT_W / 2);
            add(LK_RGB, 0, (NT_W / 2) * SP, 1);
        } else {
            add(LK_HEAD, 0, NT_W * SP, 1);
        }
    }
    T.nlayers = nl;
    T.ntiles = tile0;
    place_fragments<SH>(T);
    return T;
}

// Backward-data plan of the trunk (no bender, no view-dependent head, no time conditioning): the layers of the forward
// pass in reverse order with transposed weights.  Activations flow exactly as in the forward pass (D tile -> B slabs
// of the next layer), so `in_col`'s hidden() map describes the k index here too.  The encoding part of the gradient
// comes out in ENCODING-SLOT order: tile te, accumulator register r of lane half h = slot te*16 + r of that half
// (enc_col), which is how the forward kernel builds the encoding, so each lane ends up holding the gradient of its own
// sin/cos values.
constexpr NRN_HD int enc_tiles(int L) { return cdiv(enc_slots(L), 16); }
constexpr int DRAW_LEN = 8;                         // d raw: C <= 5 channels, padded
// VIEWS: the view-dependent head (rnh:284-304) in front -- rgb_linear^T, then ONE layer for alpha_linear^T and the transposed
// (views_linears[0] o feature_linear) whose first output tile is the gradient of the direction encoding (in encoding-slot
// order, like the point encoding's) and whose other tiles are d h_{D-1}.
template <class SH, class A, bool VIEWS = false>
constexpr Tables build_tables_bwd() {
    constexpr int KH = SH::KH, SP = SH::SP;
    constexpr int NT_W = A::W / 32, NT_E = enc_tiles(A::L), NT_EV = enc_tiles(A::LV);
    constexpr int NS_DR = cdiv(DRAW_LEN, 2 * KH);
    Tables T{};
    int nl = 0, tile0 = 0;
    auto add = [&](int kind, int index, int ns, int nt) {
        T.layers[nl] = LayerSpec{kind, index, ns, nt, tile0, 0};
        tile0 += nt;
        ++nl;
    };
    if (VIEWS) {
        add(LK_B_RGB, 0, NS_DR, NT_W / 2);
        add(LK_B_VHEAD, 0, NS_DR + (NT_W / 2) * SP, NT_EV + NT_W);
    } else {
        add(LK_B_HEAD, 0, NS_DR, NT_W);
    }
    for (int i = A::D - 1; i >= 1; --i) {
        if (i - 1 == A::SKIP) add(LK_B_SKIP, i, NT_W * SP, NT_E + NT_W);
        else add(LK_B_HID, i, NT_W * SP, NT_W);
    }
    add(LK_B_IN, 0, NT_W * SP, NT_E);
    T.nlayers = nl;
    T.ntiles = tile0;
    place_fragments<SH>(T);
    return T;
}

template <class SH, class A, bool VIEWS = false>
struct PlanB {
    static_assert(!A::TCB, "no training support for the time-conditioned baseline");
    static constexpr int KH = SH::KH, SP = SH::SP;
    static constexpr int NT_W = A::W / 32, NT_E = enc_tiles(A::L), NT_EV = enc_tiles(A::LV);
    static constexpr int NS_DR = cdiv(DRAW_LEN, 2 * KH);
    static constexpr Tables TB = build_tables_bwd<SH, A, VIEWS>();
    static constexpr int NLAYERS = TB.nlayers, NTILES = TB.ntiles, NFRAGS = TB.nfrags;
    static constexpr int NUNITS = TB.nunits, NUP = TB.nunits_padded, UF = SH::UNIT_FRAGS;
    static constexpr int MFMA_PER_BLOCK = TB.mfma_per_block;
    static_assert(TB.ntiles <= MAX_TILES && TB.nlayers <= MAX_LAYERS, "plan too large");
    // layers[0] = head^T (VIEWS: rgb_linear^T, [alpha; views o feature]^T); then pts_linears[i]^T for i = D-1 .. 1; last: pts_linears[0]^T
    static constexpr int H = VIEWS ? 1 : 0;
    static constexpr int layer_of(int i) { return H + (i == 0 ? A::D : 1 + (A::D - 1 - i)); }
};

// Backward-data plan of the bender and rigidity MLPs: network[BD-1..0]^T, then rigidity_network[RD-1..1]^T.
template <class SH, class A>
constexpr Tables build_tables_bwd_bender() {
    constexpr int KH = SH::KH, SP = SH::SP;
    constexpr int NT_BW = A::BW / 32, NT_RW = A::RW / 32, NT_LAT = cdiv(A::LAT, 32);
    constexpr int NS_DR = cdiv(DRAW_LEN, 2 * KH);
    Tables T{};
    int nl = 0, tile0 = 0;
    auto add = [&](int kind, int index, int ns, int nt) {
        T.layers[nl] = LayerSpec{kind, index, ns, nt, tile0, 0};
        tile0 += nt;
        ++nl;
    };
    add(LK_BB_OUT, A::BD - 1, NS_DR, NT_BW);
    for (int i = A::BD - 2; i >= 1; --i) add(LK_BB_HID, i, NT_BW * SP, NT_BW);
    add(LK_BB_IN, 0, NT_BW * SP, NT_LAT);
    add(LK_BR_OUT, A::RD - 1, NS_DR, NT_RW);
    for (int i = A::RD - 2; i >= 1; --i) add(LK_BR_HID, i, NT_RW * SP, NT_RW);
    T.nlayers = nl;
    T.ntiles = tile0;
    place_fragments<SH>(T);
    return T;
}
template <class SH, class A>
struct PlanBB {
    static constexpr int KH = SH::KH, SP = SH::SP;
    static constexpr int NT_BW = A::BW / 32, NT_RW = A::RW / 32, NT_LAT = cdiv(A::LAT, 32);
    static constexpr int NS_DR = cdiv(DRAW_LEN, 2 * KH);
    static constexpr Tables TB = build_tables_bwd_bender<SH, A>();
    static constexpr int NLAYERS = TB.nlayers, NTILES = TB.ntiles, NFRAGS = TB.nfrags;
    static_assert(TB.ntiles <= MAX_TILES && TB.nlayers <= MAX_LAYERS, "plan too large");
    // layers[0] = network[BD-1]^T, layers[k] = network[BD-1-k]^T (k <= BD-1), then rigidity_network[RD-1]^T, ...
    static constexpr int L_BEND(int i) { return A::BD - 1 - i; }            // network[i]^T
    static constexpr int L_RIG(int i) { return A::BD + (A::RD - 1 - i); }   // rigidity_network[i]^T, i >= 1
};

// Which element W[y][x] of the reference weight the A-fragment element (tile t, row i, slab s, half h, element e) of a
// backward layer holds (A = W^T: row <-> input feature x of W, k <-> output feature y of W); -1 = zero.
template <class SH, class A>
constexpr NRN_HD int bwd_y(int kind, int s, int h, int e, int out_features) {
    constexpr int KH = SH::KH, SP = SH::SP;
    if (kind == LK_B_HEAD || kind == LK_BB_OUT || kind == LK_BR_OUT || kind == LK_B_RGB) {
        const int ch = (2 * s + h) * KH + e;
        return ch < out_features ? ch : -1;
    }
    const int tp = s / SP, u = s % SP, r = u * KH + e;
    const int y = 32 * tp + tile_row(r, h);
    return y < out_features ? y : -1;
}
template <class SH, class A>
constexpr NRN_HD int bwd_x(int kind, int t, int i, int in_features) {
    constexpr int NT_E = enc_tiles(A::L), IN_CH = 3 + 6 * A::L;
    auto slot = [&](int te) {
        const int hh = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3);     // inverse of tile_row
        const int q = te * 16 + r;
        return q < enc_slots(A::L) ? enc_col(A::L, hh, q) : -1;
    };
    if (kind == LK_B_IN) return slot(t);
    if (kind == LK_B_SKIP) return t < NT_E ? slot(t) : IN_CH + 32 * (t - NT_E) + i;
    if (kind == LK_BB_IN) return (32 * t + i < A::LAT) ? 3 + 32 * t + i : -1;       // bender input [xyz, latent]: latent columns only
    const int x = 32 * t + i;
    return x < in_features ? x : -1;
}

template <class SH, class A, bool HAS_BEND, bool VIEWS = false, bool TRUNK = true>
struct Plan {
    static_assert(TRUNK || (HAS_BEND && !VIEWS), "a plan without trunk is the bender alone");
    static constexpr int KH = SH::KH, SP = SH::SP;
    static constexpr int NS_ENC = ns_trunk_in<SH, A>();          // xyz encoding slabs (+ latent slabs when A::TCB)
    static constexpr int NS_ENC_XYZ = ns_enc_xyz<SH, A>();
    static constexpr int NS_ENCV = cdiv(enc_slots(A::LV), KH);
    static constexpr int NS_BIN = cdiv(bin_len(A::LAT), 2 * KH);
    static constexpr int NS_RIN = cdiv(rin_len(), 2 * KH);
    static constexpr int NT_W = A::W / 32, NT_BW = A::BW / 32, NT_RW = A::RW / 32;
    static constexpr Tables TB = build_tables<SH, A, HAS_BEND, VIEWS, TRUNK>();
    static constexpr int NLAYERS = TB.nlayers;
    static constexpr int NTILES = TB.ntiles;
    static constexpr int NFRAGS = TB.nfrags;
    static constexpr int NUNITS = TB.nunits;                 // units holding real fragments
    static constexpr int NUP = TB.nunits_padded;             // units streamed per pass (multiple of RING)
    static constexpr int UF = SH::UNIT_FRAGS;
    static constexpr int MFMA_PER_BLOCK = TB.mfma_per_block;
    static_assert(TB.ntiles <= MAX_TILES && TB.nlayers <= MAX_LAYERS, "plan too large");
    // indices into layers[]
    static constexpr int L_BEND0 = 0;
    static constexpr int L_RIG0 = HAS_BEND ? A::BD : 0;
    static constexpr int L_TRUNK0 = HAS_BEND ? (A::BD + A::RD) : 0;
    static constexpr int L_HEAD = NLAYERS - 1;                 // output_linear (no view dependence)
    static constexpr int L_ALPHA = L_TRUNK0 + A::D;            // view-dependent head: alpha, feature, views, rgb
    static constexpr int L_VIEWS = L_ALPHA + 1, L_RGB = L_ALPHA + 2;       // (feature_linear is folded into L_VIEWS)
};

// ---------------------------------------------------------------------------------------
// Element maps: which reference weight element a fragment element holds.
// ---------------------------------------------------------------------------------------
// A fragment (layer, tile t, slab s) holds, for lane l (row i = l & 31, half h = l >> 5) and
// element e < KH:   W_ref[ out_row(t, i) ][ in_col(s, h, e) ]   (0 where either is -1).
template <class SH, class A>
constexpr NRN_HD int in_col(int kind, int s, int h, int e, int in_features) {
    constexpr int KH = SH::KH, SP = SH::SP;
    constexpr int NS_ENC = ns_trunk_in<SH, A>(), NS_XYZ = ns_enc_xyz<SH, A>();
    constexpr int IN_CH = 3 + 6 * A::L + (A::TCB ? A::LAT : 0);       // width of the trunk's input vector
    auto hidden = [&](int s2, int base) {
        int tp = s2 / SP, u = s2 % SP, r = u * KH + e;
        int col = 32 * tp + tile_row(r, h);
        return (base + col < in_features) ? base + col : -1;
    };
    auto enc = [&](int s2) {
        if (s2 >= NS_XYZ) {                      // TCB latent slabs: reference column = 63 + latent index
            int li = (2 * (s2 - NS_XYZ) + h) * KH + e;
            return (li < A::LAT) ? 3 + 6 * A::L + li : -1;
        }
        int q = s2 * KH + e;
        return (q < enc_slots(A::L)) ? enc_col(A::L, h, q) : -1;
    };
    constexpr int NS_ENCV = cdiv(enc_slots(A::LV), KH);
    switch (kind) {
        case LK_BEND_IN: return bin_col((2 * s + h) * KH + e, A::LAT);
        case LK_RIG_IN:  return rin_col((2 * s + h) * KH + e);
        case LK_TR_IN:   return enc(s);
        case LK_TR_SKIP: return (s < NS_ENC) ? enc(s) : hidden(s - NS_ENC, IN_CH);
        case LK_VIEWS: {     // reference column order is [feature W, direction encoding]; our slab order is the reverse
            if (s >= NS_ENCV) return hidden(s - NS_ENCV, 0);
            int q = s * KH + e;
            int c = (q < enc_slots(A::LV)) ? enc_col(A::LV, h, q) : -1;
            return c < 0 ? -1 : A::W + c;
        }
        default:         return hidden(s, 0);
    }
}
// 16-bit modes: which fragments are f16 regardless of the hidden type -- everything whose B operand
// is bounded by construction (coordinates, latent codes, sin/cos) or feeds the encoding (the bender).
template <class SH, class A>
constexpr NRN_HD bool frag_is_f16(int kind, int s) {
    constexpr int NS_ENC = ns_trunk_in<SH, A>();
    constexpr int NS_ENCV = cdiv(enc_slots(A::LV), SH::KH);
    if (kind <= LK_RIG_OUT || kind == LK_TR_IN) return true;
    if (kind == LK_TR_SKIP) return s < NS_ENC;
    if (kind == LK_VIEWS) return s < NS_ENCV;
    return false;
}
template <class A>
constexpr NRN_HD int out_row(int kind, int t, int i, int out_features) {
    switch (kind) {
        case LK_BEND_OUT: return (i < 8 && (i & 3) < 3) ? (i & 3) : -1;     // xyz offsets duplicated for both lane halves
        case LK_RIG_OUT:  return (i == 0 || i == 4) ? 0 : -1;               // rigidity logit duplicated likewise
        case LK_ALPHA:    return (i == 0 || i == 4) ? 0 : -1;
        case LK_RGB:      return (i < 8 && (i & 3) < 3) ? (i & 3) : -1;
        case LK_HEAD:     return (i < 4) ? i : ((i == 8 && out_features > 4) ? 4 : -1);   // rgb,sigma in acc[0..3], ch 4 in acc[4]
        default:          return (32 * t + i < out_features) ? 32 * t + i : -1;
    }
}

// ---------------------------------------------------------------------------------------
// The trunk on v_mfma_f32_16x16x32_{bf16,f16} (nrnerf_net_x16.h): 16-row tiles, 32 k per MFMA.
// ---------------------------------------------------------------------------------------
// Under the socket's power cap the 16x16x32 shape sustains 10-18 % more flops than 32x32x16 (tools/probes/mfma_shape_power.hip:
// the accumulator is a quarter the size; per 32 x 32 x 256 of work the register file sees 512 instead of 640 accesses).
// Dataflow: a wave owns blocks of 16 consecutive samples; lane = (n = lane & 15: sample, g = lane >> 4: k group).  A fragment
// (tile t, k-step s) holds W[16 t + (lane & 15)][k(s, g, e)], e < 8; the B operand holds act[k(s, g, e)][sample n]; the D tile
// gives lane (n, g) the four features 16 t + 4 g + i of sample n.  TWO consecutive D tiles (features 32 s .. 32 s + 31) make the
// B operand of k-step s of the next layer without leaving the lane: element e < 4 = feature 32 s + 4 g + e (tile 2 s), e >= 4 =
// feature 32 s + 16 + 4 g + (e - 4) (tile 2 s + 1) -- the k order the packer gives the weights.
// Encoding (K = 64 = two k-steps; Embedder's 3 + 6 L <= 63 columns): slot q = 8 s + e of group g holds, for q < 14 or g < 2,
// the (sin, cos) pair number m = 4 (q / 2) + g of the list (frequency f = m / 3, coordinate c = m % 3), and the four spare slots
// (q = 14, 15 of groups 2, 3, when L = 10) the identity columns x, y | z, 0: every group runs the same code.
constexpr NRN_HD int x16_hidden_feature(int s, int g, int e) { return 32 * s + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4)); }
// (NSLOT = 8 x the k-steps the encoding fills: 16 for the points' (L = 10), 8 for the directions' (LV = 4: pairs m = g, 4 + g, 8 + g in
//  slots 0..5 of every group, the identity columns x, y | z, 0 in slots 6, 7 of groups 0, 1))
constexpr NRN_HD int x16_enc_col_n(int L, int NSLOT, int s, int g, int e) {       // reference column of encoding slot (s, g, e), -1: zero
    const int q = 8 * s + e, m = 4 * (q / 2) + g;
    if (m < 3 * L) return 3 + 6 * (m / 3) + 3 * (q & 1) + (m % 3);
    // spare slots of this group, in order: the identity columns, two per group starting at the first group that has spares
    // (L = 10: exactly q = 14, 15 of groups 2, 3)
    int spare = 0;
    for (int qq = 0; qq < q; ++qq) spare += (4 * (qq / 2) + g >= 3 * L) ? 1 : 0;
    int before = 0;                                                    // spare slots of lower groups
    for (int gg = 0; gg < g; ++gg)
        for (int qq = 0; qq < NSLOT; ++qq) before += (4 * (qq / 2) + gg >= 3 * L) ? 1 : 0;
    const int id = before + spare;
    return id < 3 ? id : -1;
}
constexpr NRN_HD int x16_enc_col(int L, int s, int g, int e) { return x16_enc_col_n(L, 16, s, g, e); }
constexpr NRN_HD int x16_dir_col(int LV, int g, int e) { return x16_enc_col_n(LV, 8, 0, g, e); }
// VIEWS (view-dependent head, rnh:284-304) behind the trunk, two more layers:
//   LK_VIEWS  k-steps [direction encoding (one: 3 + 6 LV <= 32 columns, x16_dir_col(LV, g, e)), trunk output (W / 32)] ->
//             W / 32 tiles of relu(views_linears[0] o feature_linear) (feature_linear folded into the weights by the packer, as
//             in the 32x32x16 plans) + ONE last tile whose row 0 is alpha_linear (zero weights in the direction k-step; no relu:
//             the kernel takes it from the accumulator) -- an odd tile count: the alpha tile runs alone, like the plain head;
//   LK_RGB    W / 64 k-steps -> one tile, rows 0..2 = rgb_linear.
template <class SH, class A, bool VIEWS = false>
constexpr Tables build_tables_x16() {
    static_assert(SH::KH == 8, "16-bit operands");
    constexpr int NT = A::W / 16, NS_H = A::W / 32, NS_E = 2;
    static_assert(3 + 6 * A::L <= 64 && A::W % 32 == 0, "the encoding fills two k-steps");
    static_assert(!VIEWS || (3 + 6 * A::LV <= 32 && A::W % 64 == 0), "the direction encoding fills one k-step");
    Tables T{};
    int nl = 0, tile0 = 0;
    auto add = [&](int kind, int index, int ns, int nt) {
        T.layers[nl] = LayerSpec{kind, index, ns, nt, tile0, 0};
        tile0 += nt;
        ++nl;
    };
    add(LK_TR_IN, 0, NS_E, NT);
    for (int i = 1; i < A::D; ++i) {
        if (i - 1 == A::SKIP) add(LK_TR_SKIP, i, NS_E + NS_H, NT);
        else add(LK_TR_HID, i, NS_H, NT);
    }
    if (VIEWS) {
        add(LK_VIEWS, 0, 1 + NS_H, NT / 2 + 1);
        add(LK_RGB, 0, NS_H / 2, 1);
    } else {
        add(LK_HEAD, 0, NS_H, 1);
    }
    T.nlayers = nl;
    T.ntiles = tile0;
    place_fragments<SH>(T);
    return T;
}
template <class SH, class A, bool VIEWS = false>
struct PlanX16 {
    static constexpr int NT = A::W / 16, NS_H = A::W / 32, NS_E = 2;
    static constexpr int NT_V = NT / 2, NS_V = NS_H / 2;       // VIEWS: feature tiles of the views layer (+ 1 alpha tile), k-steps of rgb_linear
    static constexpr Tables TB = build_tables_x16<SH, A, VIEWS>();
    static constexpr int NLAYERS = TB.nlayers, NTILES = TB.ntiles, NFRAGS = TB.nfrags;
    static constexpr int NUNITS = TB.nunits, NUP = TB.nunits_padded, UF = SH::UNIT_FRAGS;
    static constexpr int MFMA_PER_BLOCK = TB.mfma_per_block;         // per 16-sample block
    static_assert(TB.ntiles <= MAX_TILES && TB.nlayers <= MAX_LAYERS, "plan too large");
    static constexpr int L_HEAD = NLAYERS - 1;                 // output_linear, or (VIEWS) rgb_linear
    static constexpr int L_VIEWS = NLAYERS - 2;                // (VIEWS) [views_linears[0] o feature_linear; alpha_linear]
};
// reference element of fragment (layer kind, tile t, lane row r, k-step s, group g, element e): (row, column), -1 = zero
// (LK_VIEWS: rows / columns of the FOLDED layer -- hidden columns first, then the direction encoding's; its last tile, the alpha
//  row, is the packer's business: another nn.Linear)
template <class A>
constexpr NRN_HD int x16_out_row(int kind, int t, int r, int out_features) {
    if (kind == LK_HEAD) return r < out_features ? r : -1;           // channels 0..3 in group 0's four registers, channel 4 in group 1's first
    if (kind == LK_RGB) return r < 3 ? r : -1;
    return (16 * t + r < out_features) ? 16 * t + r : -1;
}
template <class A>
constexpr NRN_HD int x16_in_col(int kind, int s, int g, int e, int in_features) {
    constexpr int IN_CH = 3 + 6 * A::L;
    if (kind == LK_TR_IN) return x16_enc_col(A::L, s, g, e);
    if (kind == LK_VIEWS) {                                          // in_features = hidden width + direction-encoding columns
        constexpr int EV = 3 + 6 * A::LV;
        if (s == 0) { const int c = x16_dir_col(A::LV, g, e); return c < 0 ? -1 : (in_features - EV) + c; }
        const int c = x16_hidden_feature(s - 1, g, e);
        return c < in_features - EV ? c : -1;
    }
    if (kind == LK_TR_SKIP) {
        if (s < 2) return x16_enc_col(A::L, s, g, e);
        const int c = IN_CH + x16_hidden_feature(s - 2, g, e);
        return c < in_features ? c : -1;
    }
    const int c = x16_hidden_feature(s, g, e);
    return c < in_features ? c : -1;
}

using ArchDefault = ArchT<256, 8, 4, 10, 64, 5, 32, 3, 32>;      // arch id 0: the reference's shipped configuration
using ArchDeepBend = ArchT<256, 8, 4, 10, 64, 7, 32, 3, 32>;     // arch id 1: deeper ray-bending MLP (BASELINE config 4)
using ArchTimeCond = ArchT<256, 8, 4, 10, 64, 5, 32, 3, 32, 4, 1>;  // arch id 2: time-conditioned baseline (no bender)
// arch id 5: --netwidth 128 --netwidth_fine 128 (train.py:1004-1010), with the 5-layer bender or without one, no
// view-dependent head (ids 3 and 4 are the dispatch rows of architectures 0 and 1 with exact view directions)
using ArchNarrow = ArchT<128, 8, 4, 10, 64, 5, 32, 3, 32>;
constexpr int NUM_ARCHS = 4;
template <int ID> struct ArchById { using type = ArchDefault; };
template <> struct ArchById<1> { using type = ArchDeepBend; };
template <> struct ArchById<2> { using type = ArchTimeCond; };
template <> struct ArchById<5> { using type = ArchNarrow; };
using ShapeF32 = Shape<1>;
using Shape16 = Shape<8, true>;        // "f16" mode: split-product bender
using Shape16Fast = Shape<8, false>;   // "bf16" mode: single-product f16 bender

}  // namespace nrn
