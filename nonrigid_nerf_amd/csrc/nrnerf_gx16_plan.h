// nrnerf_gx16_plan.h -- layer tables of the WIDTH-CLASS trunk kernel for architectures outside the compiled set (nrnerf_gx16.h; packer:
// nrnerf_api.cpp).
//
// The run-time-parameterised kernel of nrnerf_generic.h keeps activations in LDS and lets the waves split the OUTPUT tiles, so every
// wave pulls its own weights from L2 for 64 samples: 0.17-0.24 of the bf16 peak.  The compiled kernels are 3 x faster because the
// activations stay in registers and all four waves share one weight stream -- which ties them to compile-time shapes.  Here only the
// WIDTH CLASS (the trunk width rounded up to a multiple of 64, zero-padded) is compile-time: a layer of a given kind is a compile-time
// code block (dense_x16 of nrnerf_net_x16.h on a one-layer plan), and the DEPTH, the skip index and the encoding size are run-time:
// the kernel loops over the layers, the weight stream is the layers' blocks back to back, each padded to a whole number of ring
// periods so that every layer starts at ring slot 0, and the ring's source pointer advances at run time.
//   GX_IN    k-steps [encoding (2)]                      -> WC / 16 tiles            pts_linears[0]
//   GX_HID   k-steps [hidden (WC / 32)]                  -> WC / 16 tiles            pts_linears[i]
//   GX_SKIP  k-steps [encoding (2), hidden (WC / 32)]    -> WC / 16 tiles            pts_linears[skip + 1]  (reference order [input, h], rnh:277-282)
//   GX_HEAD  k-steps [hidden (WC / 32)]                  -> one tile                 output_linear (4 / 5 channels)
// and for the view-dependent head (rnh:284-304; the views layer is W // 2 wide, padded to WC / 2) instead of GX_HEAD:
//   GX_VIEWS k-steps [direction encoding (1), hidden (WC / 32)] -> WC / 32 tiles of relu(views_linears[0] o feature_linear) + one last
//            tile whose row 0 is alpha_linear (no relu; zero weights in the direction k-step)     (as PlanX16<VIEWS>, nrnerf_plan.h)
//   GX_RGB   k-steps [views hidden (WC / 64)]            -> one tile, rows 0..2               rgb_linear
// Direction encoding (one k-step, LV <= 4): the same position map as the points' (gx_enc_col(LV, 0, g, e)).
// Encoding slots (two k-steps = 64 positions, L <= 10): position p = 32 s + 8 g + e; p < 3: the identity column p; p = 3: zero; p >= 4:
// pair m = (p - 4) / 2 (frequency m / 3, coordinate m % 3), sin for even p, cos for odd -- a lane's 8 slots hold 4 whole (sin, cos) pairs.
#pragma once
#include "nrnerf_plan.h"

namespace nrn {

// BACKWARD-DATA of a plain-head trunk on the same dataflow (round 6; training of a non-compiled architecture, gx16_bwd_kernel): the layers in
// reverse with transposed weights, no biases.  A layer's input is d z (the gradient wrt its pre-activations, WC / 32 k-steps in operand
// order), its output the gradient wrt its INPUT -- hidden features (WC / 16 tiles), and for the layers that read the encoding four more
// tiles IN FRONT of them whose 64 rows are the encoding's slot positions (gx_enc_col):
//   GX_BHEAD k-steps [d raw (1: position 8 g + e = channel, < 4)] -> WC / 16 tiles        output_linear^T
//   GX_BHID  k-steps [d z (WC / 32)]                           -> WC / 16 tiles           pts_linears[i]^T
//   GX_BSKIP k-steps [d z (WC / 32)]                           -> 4 + WC / 16 tiles       pts_linears[skip + 1]^T: [encoding | hidden]
//   GX_BIN   k-steps [d z (WC / 32)]                           -> 4 tiles                 pts_linears[0]^T: the encoding's gradient
enum GxKind : int { GX_IN = 0, GX_HID = 1, GX_SKIP = 2, GX_HEAD = 3, GX_VIEWS = 4, GX_RGB = 5, GX_BHEAD = 6, GX_BHID = 7, GX_BSKIP = 8, GX_BIN = 9 };
constexpr int GX_NS_E = 2;                      // encoding k-steps (3 + 6 L + 1 <= 64: L <= 10)
constexpr int GX_MAX_L = 10;

constexpr NRN_HD int gx_enc_col(int L, int s, int g, int e) {       // reference column of encoding position (s, g, e), -1: zero
    const int p = 32 * s + 8 * g + e;
    if (p < 3) return p;
    if (p == 3) return -1;
    const int m = (p - 4) / 2, b = (p - 4) & 1;
    if (m >= 3 * L) return -1;
    return 3 + 6 * (m / 3) + 3 * b + (m % 3);
}
constexpr NRN_HD int gx_width_class(int W) { return ((W + 63) / 64) * 64; }

constexpr int GX_MAX_LV = 4;                    // direction frequencies (3 + 6 LV + 1 <= 32)
constexpr int gx_layer_ns(int wc, int kind) {
    if (kind == GX_BHEAD) return 1;
    return (kind == GX_IN) ? GX_NS_E : (kind == GX_SKIP ? GX_NS_E + wc / 32 : (kind == GX_VIEWS ? 1 + wc / 32 : (kind == GX_RGB ? wc / 64 : wc / 32)));
}
constexpr int gx_layer_tiles(int wc, int kind) {
    if (kind == GX_BSKIP) return 4 + wc / 16;
    if (kind == GX_BIN) return 4;
    return (kind == GX_HEAD || kind == GX_RGB) ? 1 : (kind == GX_VIEWS ? wc / 32 + 1 : wc / 16);
}
// reference column of encoding position p (0 .. 63: the row of a backward encoding tile, the slot of a forward encoding k-step), -1: none
constexpr NRN_HD int gx_enc_col_of_pos(int L, int p) { return gx_enc_col(L, p / 32, (p % 32) / 8, p % 8); }
// (a plain constexpr function of (width class, kind): the kernel instantiates it per template argument, the packer calls it at run time)
constexpr Tables build_tables_gx(int wc, int kind) {
    Tables T{};
    const int ns = gx_layer_ns(wc, kind);
    const int nt = gx_layer_tiles(wc, kind);
    T.layers[0] = LayerSpec{kind, 0, ns, nt, 0, 0};
    T.nlayers = 1;
    T.ntiles = nt;
    place_fragments<Shape16Fast>(T);
    return T;
}
template <int WC, int KIND>
struct PlanGX {
    static_assert(WC % 64 == 0 && WC >= 64 && WC <= 512, "width classes are multiples of 64 up to 512");
    static constexpr Tables TB = build_tables_gx(WC, KIND);
    static constexpr int NT = TB.ntiles, NFRAGS = TB.nfrags;
    static constexpr int NUNITS = TB.nunits;                     // units holding fragments
    static constexpr int NUP = TB.nunits_padded;                 // units streamed for this layer: a whole number of ring periods
    static_assert(NUP % RING == 0, "every layer starts at ring slot 0");
};
// units of one layer of kind k at width class wc (host-side mirror for the packer)
constexpr int gx_layer_units(int wc, int kind) {
    const int units = cdiv(gx_layer_ns(wc, kind) * gx_layer_tiles(wc, kind), Shape16Fast::UNIT_FRAGS);
    return cdiv(units, RING) * RING;
}
// run-time shape of one packed trunk (the kernel's GxArgs fields the packer decides)
struct GxMeta { int wc = 0, depth = 0, skip = -1, L = 0, n_bias_tiles = 0, views = 0, LV = 0; };

}  // namespace nrn
