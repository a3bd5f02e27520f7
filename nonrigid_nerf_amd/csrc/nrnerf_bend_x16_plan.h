// nrnerf_bend_x16_plan.h -- layer tables of the ray bender on v_mfma_f32_16x16x32_f16 (kernel: nrnerf_bend_x16.h; packer: nrnerf_api.cpp).
//
// The bender's two MLPs (reference ray_bending.forward, run_nerf_helpers.py:507-577) in the dataflow of the 16x16x32 trunk kernel
// (nrnerf_plan.h, "The trunk on v_mfma_f32_16x16x32"): a wave owns blocks of 16 consecutive samples; lane = (n = lane & 15: sample,
// g = lane >> 4: k group); a fragment (tile t, k-step s) holds W[16 t + (lane & 15)][k(s, g, e)], e < 8; two consecutive D tiles make
// the next layer's B operand of one k-step without leaving the lane (x16_hidden_feature).  Layers:
//   offsets  LK_BEND_IN   k-steps [xyz (position 8 g + e: g = 0, e < 3; rest zero), latent code (position 8 g + e = latent[8 g + e])] -> BW / 16 tiles
//            LK_BEND_HID  BW / 32 k-steps -> BW / 16 tiles            (BD - 2 of them)
//            LK_BEND_OUT  BW / 32 k-steps -> one tile, rows 0..2 = the offsets (group 0's registers)
//   rigidity LK_RIG_IN    the xyz k-step -> RW / 16 tiles;  LK_RIG_HID RW / 32 k-steps -> RW / 16 tiles (RD - 2);  LK_RIG_OUT -> one tile, row 0
// All operands f16 (the single-product bender of "bf16" mode, Shape::SPLIT = false; the 3-term split product of "f16" mode keeps the
// 32x32x16 kernel of nrnerf_bend.h).  Kept out of nrnerf_plan.h: every translation unit depends on that header.
#pragma once
#include "nrnerf_plan.h"

namespace nrn {

template <class A>
constexpr Tables build_tables_x16_bend() {
    static_assert(A::LAT == 32 && A::BW % 32 == 0 && A::RW % 32 == 0, "the latent code fills exactly one k-step");
    Tables T{};
    int nl = 0, tile0 = 0;
    auto add = [&](int kind, int index, int ns, int nt) {
        T.layers[nl] = LayerSpec{kind, index, ns, nt, tile0, 0};
        tile0 += nt;
        ++nl;
    };
    add(LK_BEND_IN, 0, 2, A::BW / 16);
    for (int i = 1; i < A::BD - 1; ++i) add(LK_BEND_HID, i, A::BW / 32, A::BW / 16);
    add(LK_BEND_OUT, A::BD - 1, A::BW / 32, 1);
    add(LK_RIG_IN, 0, 1, A::RW / 16);
    for (int i = 1; i < A::RD - 1; ++i) add(LK_RIG_HID, i, A::RW / 32, A::RW / 16);
    add(LK_RIG_OUT, A::RD - 1, A::RW / 32, 1);
    T.nlayers = nl;
    T.ntiles = tile0;
    place_fragments<Shape16Fast>(T);
    return T;
}

template <class A>
struct PlanX16Bend {
    static constexpr int NS_B = A::BW / 32, NS_R = A::RW / 32;          // k-steps of a hidden layer's input
    static constexpr Tables TB = build_tables_x16_bend<A>();
    static constexpr int NLAYERS = TB.nlayers, NTILES = TB.ntiles, NFRAGS = TB.nfrags;
    static constexpr int MFMA_PER_BLOCK = TB.mfma_per_block;            // per 16-sample block
    static constexpr int L_BEND0 = 0, L_RIG0 = A::BD;
    static_assert(TB.ntiles <= MAX_TILES && TB.nlayers <= MAX_LAYERS, "plan too large");
};

// reference element (row, column) of fragment (layer kind, tile t, lane row r, k-step s, group g, element e); -1 = zero
constexpr NRN_HD int x16b_out_row(int kind, int t, int r, int out_features) {
    if (kind == LK_BEND_OUT) return r < 3 ? r : -1;
    if (kind == LK_RIG_OUT) return r == 0 ? 0 : -1;
    return (16 * t + r < out_features) ? 16 * t + r : -1;
}
constexpr NRN_HD int x16b_in_col(int kind, int s, int g, int e, int in_features) {
    if (kind == LK_BEND_IN) {                       // reference columns [xyz, latent] (rnh:525)
        if (s == 0) return (g == 0 && e < 3) ? e : -1;
        const int c = 3 + 8 * g + e;
        return c < in_features ? c : -1;
    }
    if (kind == LK_RIG_IN) return (g == 0 && e < 3) ? e : -1;          // xyz only (rnh:546)
    const int c = x16_hidden_feature(s, g, e);
    return c < in_features ? c : -1;
}

}  // namespace nrn
