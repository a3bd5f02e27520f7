// nrnerf_api.cpp -- C ABI of libnrnerf_hip.so (include/nrnerf.h): weight packing into the MFMA
// fragment stream described by nrnerf_plan.h, model lifetime, workspace carving, kernel sequencing.
//
// Kernel sequence of one nrnerf_render call (reference render_rays, train.py:792-980):
//   K0  network kernel, coarse weights, z = linspace(near, far, S)        -> raw_c [N,S,4]
//   K1  composite (+ sample_pdf + merge + z_std when I > 0)               -> rgb0/disp0/acc0 or final; z_fine [N,S+I]
//   K2  network kernel, fine weights, z = z_fine                          -> raw_f [N,S+I,4]
//   K3  composite                                                         -> rgb/disp/acc
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <vector>

#include "nrnerf.h"
#include "nrnerf_kernels.h"
#include "nrnerf_aux.h"
#include "nrnerf_x16_api.h"
#include "nrnerf_bend_x16_plan.h"
#include "nrnerf_loss.h"
#include "nrnerf_optim.h"
#include "nrnerf_gen_train.h"
#include "nrnerf_gx16_bwd_api.h"
#include "nrnerf_gx16_plan.h"
#include "nrnerf_plan.h"

// the run-time-parameterised kernel's TRAINING instantiations (nrnerf_generic.hip; the rendering ones: launch_generic, nrnerf_kernels.h)
namespace nrn { hipError_t launch_generic_train(int precision, const GenArgs& a, int num_cus, hipStream_t stream); }
using namespace nrn;

#ifndef NRN_WGRAD_SYNC_DEFAULT
// pairs of blocks between workgroup barriers in trunk_wgrad (WgradArgs::sync_every; env NRNERF_WGRAD_SYNC, 0 = never).  Measured
// at 16 384 rays (tools/experiments/wgrad_sync_sweep.sh): never 3.30 ms per launch, every 2 pairs 2.76, 8: 2.54, 32: 2.55, 128: 2.82,
// 512: 3.09 -- the waves that share a fragment stay within L2's reach of each other, the barrier itself costs nothing because
// the loads already requested stay in flight across it.
#define NRN_WGRAD_SYNC_DEFAULT 16
#endif

// Nothing throws across the C ABI (include/nrnerf.h): every extern "C" body is a function-try-block that turns
// std::bad_alloc (the packer's std::vector growth) into NRNERF_ERR_NOMEM and anything else -- the packer's
// plan-consistency checks throw std::logic_error -- into NRNERF_ERR_INTERNAL.
#define NRN_CATCH catch (const std::bad_alloc&) { return NRNERF_ERR_NOMEM; } catch (...) { return NRNERF_ERR_INTERNAL; }

namespace {

// ------------------------------------------------------------------------------------------
// host-side element conversion
// ------------------------------------------------------------------------------------------
inline uint16_t f32_to_bf16(float f) {       // round to nearest even, NaN preserved
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline uint16_t f32_to_f16(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}

struct PackedPass {
    std::vector<uint8_t> stream;
    std::vector<uint32_t> unit_off;   // nunits + 1, in 16-byte words
    std::vector<float> bias;          // ntiles * 32
    int ntiles = 0, nunits = 0, frag_bytes = 0, slot_bytes = 0, mfma_per_block = 0;
    // where every stream element / bias entry comes from in the flat parameter vector (nrnerf_model_update_device):
    // index (-1 = constant zero) and target format (RepackFmt); filled when a FlatLayout is given to the packer
    std::vector<int32_t> src, bias_src;
    std::vector<uint8_t> fmt;
};

// Flat parameter vector of a model (documented in nrnerf.h at nrnerf_model_update_device): every nn.Linear as weight
// [out, in] row-major then bias [out] (if it has one), in the order bender.network[0..], bender.rigidity_network[0..],
// coarse (pts_linears[0..], then output_linear | alpha, feature, views, rgb), fine likewise.
struct FlatLayout {
    std::vector<std::pair<const float*, int64_t>> base;      // host weight / bias pointer of the description -> offset
    int64_t total = 0;
    void add(const nrnerf_linear& l) {
        if (!l.weight) return;
        base.push_back({l.weight, total}); total += (int64_t)l.out_features * l.in_features;
        if (l.bias) { base.push_back({l.bias, total}); total += l.out_features; }
    }
    int64_t of(const float* p) const {
        for (auto& b : base) if (b.first == p) return b.second;
        return -1;
    }
    // DERIVED entries, after every real parameter: per network with the view-dependent head (coarse, then fine) the views
    // layer with feature_linear folded in -- weight [W/2][W + direction encoding] = [W_v[:, :W] W_f | W_v[:, W:]], then bias
    // [W/2] = W_v[:, :W] b_f + b_v -- keyed by the views weight pointer of the description
    std::vector<std::pair<const float*, int64_t>> folded;
    void add_folded(const nrnerf_mlp_desc& m) {
        if (!m.use_viewdirs || !m.views_linear.weight) return;
        folded.push_back({m.views_linear.weight, total});
        total += (int64_t)m.views_linear.out_features * m.views_linear.in_features + m.views_linear.out_features;
    }
    int64_t folded_of(const float* views_weight) const {
        for (auto& b : folded) if (b.first == views_weight) return b.second;
        return -1;
    }
};
// views_linears[0] o feature_linear as one layer (see LK_VIEWS in nrnerf_plan.h): fp64 products rounded to fp32 once
struct FoldedViews {
    std::vector<float> w, b;
    nrnerf_linear lin{};
    explicit FoldedViews(const nrnerf_mlp_desc& m) {
        const nrnerf_linear& v = m.views_linear;
        const nrnerf_linear& f = m.feature_linear;
        const int O = v.out_features, K = v.in_features, W = f.out_features, Wi = f.in_features;
        if (!v.weight || !f.weight || K < W) throw std::logic_error("view-dependent head: views layer narrower than the feature vector");
        w.assign((size_t)O * (size_t)(K - W + Wi), 0.0f);
        b.assign((size_t)O, 0.0f);
        const int Kf = K - W + Wi;                 // (Wi == W for the reference's head: same row length as the views layer)
        for (int r = 0; r < O; ++r) {
            for (int c = 0; c < Wi; ++c) {
                double acc = 0.0;
                for (int k = 0; k < W; ++k) acc += (double)v.weight[(size_t)r * K + k] * (double)f.weight[(size_t)k * Wi + c];
                w[(size_t)r * Kf + c] = (float)acc;
            }
            for (int c = W; c < K; ++c) w[(size_t)r * Kf + Wi + (c - W)] = v.weight[(size_t)r * K + c];
            double acc = v.bias ? (double)v.bias[r] : 0.0;
            if (f.bias) for (int k = 0; k < W; ++k) acc += (double)v.weight[(size_t)r * K + k] * (double)f.bias[k];
            b[r] = (float)acc;
        }
        lin.weight = w.data(); lin.bias = b.data(); lin.out_features = O; lin.in_features = Kf;
    }
};
void add_mlp(FlatLayout& f, const nrnerf_mlp_desc& m) {
    for (int i = 0; i < m.depth; ++i) f.add(m.pts_linears[i]);
    if (m.use_viewdirs) { f.add(m.alpha_linear); f.add(m.feature_linear); f.add(m.views_linear); f.add(m.rgb_linear); }
    else f.add(m.output_linear);
}
FlatLayout flat_layout(const nrnerf_model_desc& d) {
    FlatLayout f;
    if (d.bender) {
        for (int i = 0; i < d.bender->depth; ++i) f.add(d.bender->network[i]);
        for (int i = 0; i < d.bender->rigidity_depth; ++i) f.add(d.bender->rigidity_network[i]);
    }
    add_mlp(f, *d.coarse);
    if (d.fine) add_mlp(f, *d.fine);
    f.add_folded(*d.coarse);
    if (d.fine) f.add_folded(*d.fine);
    return f;
}

const nrnerf_linear* layer_source(const nrnerf_model_desc& d, const nrnerf_mlp_desc& mlp, const LayerSpec& sp) {
    switch (sp.kind) {
        case LK_BEND_IN: case LK_BEND_HID: case LK_BEND_OUT: return &d.bender->network[sp.index];
        case LK_RIG_IN: case LK_RIG_HID: case LK_RIG_OUT: return &d.bender->rigidity_network[sp.index];
        case LK_TR_IN: case LK_TR_HID: case LK_TR_SKIP: return &mlp.pts_linears[sp.index];
        case LK_HEAD: return &mlp.output_linear;
        case LK_ALPHA: return &mlp.alpha_linear;
        case LK_FEAT: return &mlp.feature_linear;
        case LK_VIEWS: return &mlp.views_linear;
        case LK_RGB: return &mlp.rgb_linear;
    }
    return nullptr;
}

// tcb_shift (training of the time-conditioned baseline with architecture A = the plain trunk): the module's first layer
// reads [encoding, latent] and its skip layer [encoding, latent, h] (rnh:207-209, 273-282); the latent columns act as a
// per-ray bias (the latent is constant along a ray) that the caller supplies (nrnerf_trunk_args.ray_bias), so the images
// hold the encoding and the hidden columns only: hidden column c of the skip layer sits tcb_shift columns further right.
template <class SH, class A, bool HAS_BEND, bool VIEWS, bool TRUNK = true>
void pack_pass(const nrnerf_model_desc& d, const nrnerf_mlp_desc& mlp, int precision, PackedPass& out, const FlatLayout* lay = nullptr,
               int tcb_shift = 0) {
    using PL = Plan<SH, A, HAS_BEND, VIEWS, TRUNK>;
    constexpr int KH = SH::KH;
    const Tables& T = PL::TB;
    out.ntiles = T.ntiles; out.nunits = T.nunits_padded;
    out.frag_bytes = SH::FRAG_BYTES; out.slot_bytes = SH::UNIT_BYTES; out.mfma_per_block = T.mfma_per_block;
    out.stream.assign((size_t)T.nunits_padded * SH::UNIT_BYTES, 0);       // zero padded to whole units
    out.unit_off.assign(T.nunits_padded + 1, 0);
    for (int u = 0; u <= T.nunits_padded; ++u) out.unit_off[u] = (uint32_t)((size_t)u * SH::UNIT_BYTES / 16);
    out.bias.assign((size_t)T.ntiles * 32, 0.0f);
    if (lay) {
        out.src.assign(out.stream.size() / SH::ELEM_BYTES, -1);
        out.fmt.assign(out.stream.size() / SH::ELEM_BYTES, KH == 1 ? 0 : 1);
        out.bias_src.assign(out.bias.size(), -1);
    }
    size_t written = 0;
    std::unique_ptr<FoldedViews> folded;           // the views layer's weights with feature_linear folded in (VIEWS plans)
    for (int l = 0; l < T.nlayers; ++l) {
        const LayerSpec& sp = T.layers[l];
        const nrnerf_linear* lin = layer_source(d, mlp, sp);
        int64_t wbase = lay ? lay->of(lin->weight) : -1, bbase = (lay && lin->bias) ? lay->of(lin->bias) : -1;
        if (sp.kind == LK_VIEWS) {                 // source: the derived entries of the flat vector (FlatLayout::add_folded)
            folded.reset(new FoldedViews(mlp));
            const int64_t fb = lay ? lay->folded_of(mlp.views_linear.weight) : -1;
            lin = &folded->lin;
            wbase = fb;
            bbase = fb < 0 ? -1 : fb + (int64_t)lin->out_features * lin->in_features;
        }
        auto orow = [&](int t, int i) { return out_row<A>(sp.kind, t, i, lin->out_features); };
        for (int t = 0; t < sp.nt; ++t) {
            const TileInfo& ti = T.tiles[sp.tile0 + t];
            for (int s = 0; s < sp.ns; ++s) {
                // split layers: fragment pair (hi, lo) with lo = f16((w - f16(w)) * 2^11); others: one fragment
                for (int part = 0; part <= sp.split; ++part) {
                    const size_t fi = (size_t)ti.gbase + (size_t)s * ti.gstride + part;
                    if (fi >= (size_t)T.nfrags) throw std::logic_error("plan / packer drift");
                    uint8_t* fr = out.stream.data() + fi * SH::FRAG_BYTES;
                    const bool as_f16 = (precision == NRNERF_PREC_F16) || frag_is_f16<SH, A>(sp.kind, s);
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i = lane & 31, h = lane >> 5;
                        const int row = orow(t, i);
                        for (int e = 0; e < KH; ++e) {
                            int col = in_col<SH, A>(sp.kind, s, h, e, lin->in_features - ((sp.kind == LK_TR_IN || sp.kind == LK_TR_SKIP) ? tcb_shift : 0));
                            if (tcb_shift && sp.kind == LK_TR_SKIP && col >= 3 + 6 * A::L) col += tcb_shift;
                            const float w = (row < 0 || col < 0) ? 0.0f : lin->weight[(size_t)row * lin->in_features + col];
                            if (lay) {
                                const size_t el = fi * (SH::FRAG_BYTES / SH::ELEM_BYTES) + (size_t)lane * KH + e;
                                out.src[el] = (row < 0 || col < 0 || wbase < 0) ? -1 : (int32_t)(wbase + (int64_t)row * lin->in_features + col);
                                out.fmt[el] = (KH == 1) ? 0 : (part == 1 ? 3 : (as_f16 ? 2 : 1));
                            }
                            if (KH == 1) {
                                std::memcpy(fr + lane * 4, &w, 4);
                            } else {
                                float v = w;
                                if (part == 1) v = (w - (float)(_Float16)w) * SH::LO_SCALE;
                                const uint16_t q = as_f16 ? f32_to_f16(v) : f32_to_bf16(v);
                                std::memcpy(fr + (lane * KH + e) * 2, &q, 2);
                            }
                        }
                    }
                    ++written;
                }
            }
            for (int h = 0; h < 2; ++h)
                for (int r = 0; r < 16; ++r) {
                    const int row = orow(t, tile_row(r, h));
                    out.bias[(size_t)(sp.tile0 + t) * 32 + h * 16 + r] = (row >= 0 && lin->bias) ? lin->bias[row] : 0.0f;
                    if (lay && row >= 0 && bbase >= 0) out.bias_src[(size_t)(sp.tile0 + t) * 32 + h * 16 + r] = (int32_t)(bbase + row);
                }
        }
    }
    if (written != (size_t)T.nfrags) throw std::logic_error("plan / packer drift");
}

// The trunk-only image of the 16x16x32 kernel (nrnerf_net_x16.h, PlanX16): fragment (tile t, k-step s) holds, for lane (r = lane & 15,
// g = lane >> 4) and element e < 8,  W[x16_out_row(t, r)][x16_in_col(s, g, e)]; encoding k-steps f16, hidden ones the model's type;
// bias table [tile][16 rows].
// VIEWS: the view-dependent head as PlanX16 lays it out -- [views_linears[0] o feature_linear (FoldedViews) | alpha_linear in the last tile],
// then rgb_linear.
template <class SH, class A, bool VIEWS = false>
void pack_pass_x16(const nrnerf_mlp_desc& mlp, int precision, PackedPass& out, const FlatLayout* lay = nullptr) {
    using PL = PlanX16<SH, A, VIEWS>;
    const Tables& T = PL::TB;
    out.ntiles = T.ntiles; out.nunits = T.nunits_padded;
    out.frag_bytes = SH::FRAG_BYTES; out.slot_bytes = SH::UNIT_BYTES; out.mfma_per_block = T.mfma_per_block;
    out.stream.assign((size_t)T.nunits_padded * SH::UNIT_BYTES, 0);
    out.unit_off.assign(T.nunits_padded + 1, 0);
    for (int u = 0; u <= T.nunits_padded; ++u) out.unit_off[u] = (uint32_t)((size_t)u * SH::UNIT_BYTES / 16);
    out.bias.assign((size_t)T.ntiles * 16, 0.0f);
    if (lay) {
        out.src.assign(out.stream.size() / 2, -1);
        out.fmt.assign(out.stream.size() / 2, 1);
        out.bias_src.assign(out.bias.size(), -1);
    }
    size_t written = 0;
    std::unique_ptr<FoldedViews> folded;
    for (int l = 0; l < T.nlayers; ++l) {
        const LayerSpec& sp = T.layers[l];
        const nrnerf_linear* lin0 = (sp.kind == LK_HEAD) ? &mlp.output_linear : (sp.kind == LK_RGB ? &mlp.rgb_linear : &mlp.pts_linears[sp.index]);
        int64_t wbase0 = lay ? lay->of(lin0->weight) : -1, bbase0 = (lay && lin0->bias) ? lay->of(lin0->bias) : -1;
        if (sp.kind == LK_VIEWS) {                 // source: the derived entries of the flat vector (FlatLayout::add_folded), as pack_pass
            folded.reset(new FoldedViews(mlp));
            lin0 = &folded->lin;
            wbase0 = lay ? lay->folded_of(mlp.views_linear.weight) : -1;
            bbase0 = wbase0 < 0 ? -1 : wbase0 + (int64_t)lin0->out_features * lin0->in_features;
        }
        for (int t = 0; t < sp.nt; ++t) {
            // the views layer's last tile: alpha_linear (row 0), over the hidden k-steps only
            const bool alpha_tile = sp.kind == LK_VIEWS && t == sp.nt - 1;
            const nrnerf_linear* lin = alpha_tile ? &mlp.alpha_linear : lin0;
            const int64_t wbase = alpha_tile ? (lay ? lay->of(lin->weight) : -1) : wbase0;
            const int64_t bbase = alpha_tile ? ((lay && lin->bias) ? lay->of(lin->bias) : -1) : bbase0;
            const TileInfo& ti = T.tiles[sp.tile0 + t];
            for (int s = 0; s < sp.ns; ++s) {
                const size_t fi = (size_t)ti.gbase + (size_t)s * ti.gstride;
                if (fi >= (size_t)T.nfrags) throw std::logic_error("plan / packer drift");
                uint8_t* fr = out.stream.data() + fi * SH::FRAG_BYTES;
                const bool enc_step = ((sp.kind == LK_TR_IN || sp.kind == LK_TR_SKIP) && s < PL::NS_E) || (sp.kind == LK_VIEWS && s == 0);
                const bool as_f16 = precision == NRNERF_PREC_F16 || enc_step;
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 15, g = lane >> 4;
                    const int row = alpha_tile ? (r == 0 ? 0 : -1) : x16_out_row<A>(sp.kind, t, r, lin->out_features);
                    for (int e = 0; e < 8; ++e) {
                        int col;
                        if (alpha_tile) { col = (s == 0) ? -1 : x16_hidden_feature(s - 1, g, e); if (col >= lin->in_features) col = -1; }
                        else col = x16_in_col<A>(sp.kind, s, g, e, lin->in_features);
                        const float w = (row < 0 || col < 0) ? 0.0f : lin->weight[(size_t)row * lin->in_features + col];
                        const size_t el = fi * (SH::FRAG_BYTES / 2) + (size_t)lane * 8 + e;
                        if (lay) {
                            out.src[el] = (row < 0 || col < 0 || wbase < 0) ? -1 : (int32_t)(wbase + (int64_t)row * lin->in_features + col);
                            out.fmt[el] = as_f16 ? 2 : 1;
                        }
                        const uint16_t q = as_f16 ? f32_to_f16(w) : f32_to_bf16(w);
                        std::memcpy(fr + (lane * 8 + e) * 2, &q, 2);
                    }
                }
                ++written;
            }
            for (int r = 0; r < 16; ++r) {
                const int row = alpha_tile ? (r == 0 ? 0 : -1) : x16_out_row<A>(sp.kind, t, r, lin->out_features);
                out.bias[(size_t)(sp.tile0 + t) * 16 + r] = (row >= 0 && lin->bias) ? lin->bias[row] : 0.0f;
                if (lay && row >= 0 && bbase >= 0) out.bias_src[(size_t)(sp.tile0 + t) * 16 + r] = (int32_t)(bbase + row);
            }
        }
    }
    if (written != (size_t)T.nfrags) throw std::logic_error("plan / packer drift");
}
// does the 16x16x32 trunk kernel have this network?  (compiled architecture 0's trunk, output_linear head, 16-bit precision)
bool x16_eligible(const nrnerf_model_desc& d, const nrnerf_mlp_desc& m, bool any_16bit = false) {
    using A = ArchDefault;
    const bool width_ok = m.width == ArchDefault::W || m.width == ArchNarrow::W;        // the two compiled trunk widths
    // (both 16-bit modes; nrnerf_model_desc::flags & NRNERF_MODEL_NO_X16_F16 keeps "f16" mode on the 32x32x16 kernels only.  At render
    //  time NRNERF_RENDER_NO_X16 selects the 32x32x16 trunk-only kernel per call: the split path is then bit-identical to the
    //  fused-bender fine pass in "f16" mode, which tests/test_gpu_parity.py asserts)
    const bool f16_too = !(d.flags & NRNERF_MODEL_NO_X16_F16);
    if (d.precision != NRNERF_PREC_BF16 && !(d.precision == NRNERF_PREC_F16 && (f16_too || any_16bit))) return false;
    if (m.time_conditioned || d.multires != A::L || m.depth != A::D || !width_ok || m.skip != A::SKIP) return false;
    if (m.use_viewdirs) {            // view-dependent head: width 256, 4 direction frequencies, finite-difference (not exact Jacobian) directions
        if (m.width != A::W || d.multires_views != A::LV || (d.exact_viewdirs && d.bender)) return false;
        if (m.views_linear.out_features != A::W / 2 || m.feature_linear.out_features != A::W) return false;
        return true;
    }
    if (m.output_ch != 4 && m.output_ch != 5) return false;
    return true;
}
void pack_x16(const nrnerf_model_desc& d, const nrnerf_mlp_desc& m, PackedPass& out, const FlatLayout* lay = nullptr) {
    if (m.use_viewdirs) {
        if (d.precision == NRNERF_PREC_F16) pack_pass_x16<Shape16, ArchDefault, true>(m, d.precision, out, lay);
        else pack_pass_x16<Shape16Fast, ArchDefault, true>(m, d.precision, out, lay);
        return;
    }
    if (m.width == ArchNarrow::W) {
        if (d.precision == NRNERF_PREC_F16) pack_pass_x16<Shape16, ArchNarrow>(m, d.precision, out, lay);
        else pack_pass_x16<Shape16Fast, ArchNarrow>(m, d.precision, out, lay);
        return;
    }
    if (d.precision == NRNERF_PREC_F16) pack_pass_x16<Shape16, ArchDefault>(m, d.precision, out, lay);
    else pack_pass_x16<Shape16Fast, ArchDefault>(m, d.precision, out, lay);
}

// The bender + rigidity MLPs for the 16x16x32 stand-alone bender (nrnerf_bend_x16.h, PlanX16Bend): f16 fragments of 16 rows x 32 k,
// element (lane (r, g), e) = W[x16b_out_row(t, r)][x16b_in_col(s, g, e)]; bias table [tile][16 rows].
template <class A>
void pack_pass_x16_bend(const nrnerf_bender_desc& bd, PackedPass& out, const FlatLayout* lay = nullptr) {
    using PL = PlanX16Bend<A>;
    using SH = Shape16Fast;
    const Tables& T = PL::TB;
    out.ntiles = T.ntiles; out.nunits = cdiv(T.nfrags, SH::UNIT_FRAGS);
    out.frag_bytes = SH::FRAG_BYTES; out.slot_bytes = SH::UNIT_BYTES; out.mfma_per_block = T.mfma_per_block;
    out.stream.assign((size_t)T.nfrags * SH::FRAG_BYTES, 0);
    out.unit_off.assign(1, 0);
    out.bias.assign((size_t)T.ntiles * 16, 0.0f);
    if (lay) {
        out.src.assign(out.stream.size() / 2, -1);
        out.fmt.assign(out.stream.size() / 2, 2);              // f16
        out.bias_src.assign(out.bias.size(), -1);
    }
    size_t written = 0;
    for (int l = 0; l < T.nlayers; ++l) {
        const LayerSpec& sp = T.layers[l];
        const nrnerf_linear* lin = (sp.kind <= LK_BEND_OUT) ? &bd.network[sp.index] : &bd.rigidity_network[sp.index];
        const int64_t wbase = lay ? lay->of(lin->weight) : -1, bbase = (lay && lin->bias) ? lay->of(lin->bias) : -1;
        for (int t = 0; t < sp.nt; ++t) {
            const TileInfo& ti = T.tiles[sp.tile0 + t];
            for (int s = 0; s < sp.ns; ++s) {
                const size_t fi = (size_t)ti.gbase + (size_t)s * ti.gstride;
                if (fi >= (size_t)T.nfrags) throw std::logic_error("plan / packer drift");
                uint8_t* fr = out.stream.data() + fi * SH::FRAG_BYTES;
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 15, g = lane >> 4;
                    const int row = x16b_out_row(sp.kind, t, r, lin->out_features);
                    for (int e = 0; e < 8; ++e) {
                        const int col = x16b_in_col(sp.kind, s, g, e, lin->in_features);
                        const float w = (row < 0 || col < 0) ? 0.0f : lin->weight[(size_t)row * lin->in_features + col];
                        if (lay) out.src[fi * (SH::FRAG_BYTES / 2) + (size_t)lane * 8 + e] =
                            (row < 0 || col < 0 || wbase < 0) ? -1 : (int32_t)(wbase + (int64_t)row * lin->in_features + col);
                        const uint16_t q = f32_to_f16(w);
                        std::memcpy(fr + (lane * 8 + e) * 2, &q, 2);
                    }
                }
                ++written;
            }
            for (int r = 0; r < 16; ++r) {
                const int row = x16b_out_row(sp.kind, t, r, lin->out_features);
                out.bias[(size_t)(sp.tile0 + t) * 16 + r] = (row >= 0 && lin->bias) ? lin->bias[row] : 0.0f;
                if (lay && row >= 0 && bbase >= 0) out.bias_src[(size_t)(sp.tile0 + t) * 16 + r] = (int32_t)(bbase + row);
            }
        }
    }
    if (written != (size_t)T.nfrags) throw std::logic_error("plan / packer drift");
}
// does the 16x16x32 bender kernel have this bender?  (one of the two compiled shapes; both 16-bit modes: the single-product f16 bender.
// "f16" mode's fp32-equivalent three-product bender -- 3 x the MFMAs, 11.7 % of a 1080p frame in round 5 -- stays what the FUSED-bender kernels
// and the 32x32x16 stand-alone bender compute; NRNERF_MODEL_NO_X16_F16 keeps an "f16" handle on those alone.  That the single-product bender
// meets "f16" mode's stated bar (>= 40 dB vs the fp32 oracle, <= 0.1 dB vs ground truth) on all four fitted checkpoints:
// tests/test_fitted_checkpoint.py, profiles/r06_fitted_accuracy.txt.)
bool bend_x16_eligible(const nrnerf_model_desc& d) {
    if (!d.bender) return false;
    if (d.precision != NRNERF_PREC_BF16 && !(d.precision == NRNERF_PREC_F16 && !(d.flags & NRNERF_MODEL_NO_X16_F16))) return false;
    const nrnerf_bender_desc& b = *d.bender;
    using A = ArchDefault;
    return b.latent_size == A::LAT && b.hidden == A::BW && (b.depth == ArchDefault::BD || b.depth == ArchDeepBend::BD) &&
           b.rigidity_hidden == A::RW && b.rigidity_depth == A::RD;
}
void pack_bend_x16(const nrnerf_model_desc& d, PackedPass& out, const FlatLayout* lay = nullptr) {
    if (d.bender->depth == ArchDeepBend::BD) pack_pass_x16_bend<ArchDeepBend>(*d.bender, out, lay);
    else pack_pass_x16_bend<ArchDefault>(*d.bender, out, lay);
}

// Transposed weights for the backward-data kernel (nrnerf_train.h): PlanB's layer list, fragment element
// (tile t, row i, slab s, half h, element e) = W[y][x] with (y, x) from bwd_y / bwd_x.  No biases.
// VIEWS (view-dependent head, rnh:284-304): rgb_linear^T, then the layer that joins both branches of the head -- its k index
// runs over [d raw (only channel 3, sigma, is used: alpha_linear), d z_v (W/2)], its rows over [direction-encoding slots (one
// tile, enc_col order), h_{D-1} (W)]; the weights of the d z_v part are the FOLDED views layer's (FoldedViews: hidden columns
// first, then the direction encoding's), transposed.
template <class SH, class A, bool VIEWS = false>
void pack_pass_bwd(const nrnerf_mlp_desc& mlp, int precision, PackedPass& out, const FlatLayout* lay = nullptr, int tcb_shift = 0) {
    using PL = PlanB<SH, A, VIEWS>;
    constexpr int KH = SH::KH, SP = SH::SP;
    const Tables& T = PL::TB;
    out.ntiles = T.ntiles; out.nunits = T.nunits_padded;
    out.frag_bytes = SH::FRAG_BYTES; out.slot_bytes = SH::UNIT_BYTES; out.mfma_per_block = T.mfma_per_block;
    out.stream.assign((size_t)T.nunits_padded * SH::UNIT_BYTES, 0);
    out.unit_off.assign(T.nunits_padded + 1, 0);
    for (int u = 0; u <= T.nunits_padded; ++u) out.unit_off[u] = (uint32_t)((size_t)u * SH::UNIT_BYTES / 16);
    out.bias.assign((size_t)T.ntiles * 32, 0.0f);
    if (lay) {
        out.src.assign(out.stream.size() / SH::ELEM_BYTES, -1);
        out.fmt.assign(out.stream.size() / SH::ELEM_BYTES, KH == 1 ? 0 : (precision == NRNERF_PREC_F16 ? 2 : 1));
        out.bias_src.assign(out.bias.size(), -1);
    }
    std::unique_ptr<FoldedViews> folded;
    if (VIEWS) folded.reset(new FoldedViews(mlp));
    const int64_t fbase = (VIEWS && lay) ? lay->folded_of(mlp.views_linear.weight) : -1;
    const int64_t abase = (VIEWS && lay) ? lay->of(mlp.alpha_linear.weight) : -1;
    // LK_B_VHEAD: value and flat-vector position of element (tile t, row i, slab s, half h, element e)
    auto vhead = [&](int t, int i, int s, int h, int e, int64_t* src) -> float {
        constexpr int NT_EV = PL::NT_EV, NS_DR = PL::NS_DR;
        const int Kf = folded->lin.in_features, Wi = mlp.feature_linear.in_features;
        *src = -1;
        int col;                                    // column of the folded layer = output row of its transpose
        if (t < NT_EV) {
            const int hh = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3), q = t * 16 + r;      // inverse of tile_row
            const int c = q < enc_slots(A::LV) ? enc_col(A::LV, hh, q) : -1;
            if (c < 0) return 0.0f;
            col = Wi + c;
        } else {
            col = 32 * (t - NT_EV) + i;
            if (col >= Wi) return 0.0f;
        }
        if (s < NS_DR) {                            // d raw: only sigma (channel 3) enters here, through alpha_linear
            const int ch = (2 * s + h) * KH + e;
            if (ch != 3 || t < NT_EV) return 0.0f;
            if (abase >= 0) *src = abase + col;
            return mlp.alpha_linear.weight[col];
        }
        const int s2 = s - NS_DR, tp = s2 / SP, u = s2 % SP, r = u * KH + e;
        const int y = 32 * tp + tile_row(r, h);
        if (y >= folded->lin.out_features) return 0.0f;
        if (fbase >= 0) *src = fbase + (int64_t)y * Kf + col;
        return folded->w[(size_t)y * Kf + col];
    };
    size_t written = 0;
    for (int l = 0; l < T.nlayers; ++l) {
        const LayerSpec& sp = T.layers[l];
        const nrnerf_linear* lin = (sp.kind == LK_B_HEAD) ? &mlp.output_linear
                                 : (sp.kind == LK_B_RGB) ? &mlp.rgb_linear : (sp.kind == LK_B_VHEAD) ? &mlp.alpha_linear : &mlp.pts_linears[sp.index];
        const int64_t wbase = lay ? lay->of(lin->weight) : -1;
        for (int t = 0; t < sp.nt; ++t) {
            const TileInfo& ti = T.tiles[sp.tile0 + t];
            for (int s = 0; s < sp.ns; ++s) {
                const size_t fi = (size_t)ti.gbase + (size_t)s * ti.gstride;
                if (fi >= (size_t)T.nfrags) throw std::logic_error("plan / packer drift");
                uint8_t* fr = out.stream.data() + fi * SH::FRAG_BYTES;
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    int x = bwd_x<SH, A>(sp.kind, t, i, lin->in_features - ((sp.kind == LK_B_IN || sp.kind == LK_B_SKIP) ? tcb_shift : 0));
                    if (tcb_shift && sp.kind == LK_B_SKIP && x >= 3 + 6 * A::L) x += tcb_shift;      // see pack_pass
                    for (int e = 0; e < KH; ++e) {
                        float w;
                        int64_t src = -1;
                        if (sp.kind == LK_B_VHEAD) {
                            w = vhead(t, i, s, h, e, &src);
                        } else {
                            const int y = bwd_y<SH, A>(sp.kind, s, h, e, lin->out_features);
                            w = (x < 0 || y < 0) ? 0.0f : lin->weight[(size_t)y * lin->in_features + x];
                            if (x >= 0 && y >= 0 && wbase >= 0) src = wbase + (int64_t)y * lin->in_features + x;
                        }
                        if (lay && src >= 0) out.src[fi * (SH::FRAG_BYTES / SH::ELEM_BYTES) + (size_t)lane * KH + e] = (int32_t)src;
                        if (KH == 1) {
                            std::memcpy(fr + lane * 4, &w, 4);
                        } else {
                            const uint16_t q = (precision == NRNERF_PREC_F16) ? f32_to_f16(w) : f32_to_bf16(w);
                            std::memcpy(fr + (lane * KH + e) * 2, &q, 2);
                        }
                    }
                }
                ++written;
            }
        }
    }
    if (written != (size_t)T.nfrags) throw std::logic_error("plan / packer drift");
}

// Transposed weights of the bender / rigidity MLPs for their backward-data kernel (nrnerf_train_bend.h): PlanBB's layer
// list, always fp32.  Same element rule as pack_pass_bwd.
template <class A>
void pack_pass_bwd_bender(const nrnerf_bender_desc& b, PackedPass& out, const FlatLayout* lay = nullptr) {
    using SH = ShapeF32;
    using PL = PlanBB<SH, A>;
    const Tables& T = PL::TB;
    out.ntiles = T.ntiles; out.nunits = T.nunits_padded;
    out.frag_bytes = SH::FRAG_BYTES; out.slot_bytes = SH::UNIT_BYTES; out.mfma_per_block = T.mfma_per_block;
    out.stream.assign((size_t)T.nunits_padded * SH::UNIT_BYTES, 0);
    out.unit_off.assign(T.nunits_padded + 1, 0);
    for (int u = 0; u <= T.nunits_padded; ++u) out.unit_off[u] = (uint32_t)((size_t)u * SH::UNIT_BYTES / 16);
    out.bias.assign((size_t)T.ntiles * 32, 0.0f);
    if (lay) {
        out.src.assign(out.stream.size() / SH::ELEM_BYTES, -1);
        out.fmt.assign(out.stream.size() / SH::ELEM_BYTES, 0);
        out.bias_src.assign(out.bias.size(), -1);
    }
    size_t written = 0;
    for (int l = 0; l < T.nlayers; ++l) {
        const LayerSpec& sp = T.layers[l];
        const bool rig = sp.kind == LK_BR_OUT || sp.kind == LK_BR_HID;
        const nrnerf_linear* lin = rig ? &b.rigidity_network[sp.index] : &b.network[sp.index];
        const int64_t wbase = lay ? lay->of(lin->weight) : -1;
        for (int t = 0; t < sp.nt; ++t) {
            const TileInfo& ti = T.tiles[sp.tile0 + t];
            for (int s = 0; s < sp.ns; ++s) {
                const size_t fi = (size_t)ti.gbase + (size_t)s * ti.gstride;
                if (fi >= (size_t)T.nfrags) throw std::logic_error("plan / packer drift");
                uint8_t* fr = out.stream.data() + fi * SH::FRAG_BYTES;
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    const int x = bwd_x<SH, A>(sp.kind, t, i, lin->in_features);
                    const int y = bwd_y<SH, A>(sp.kind, s, h, 0, lin->out_features);
                    const float w = (x < 0 || y < 0) ? 0.0f : lin->weight[(size_t)y * lin->in_features + x];
                    if (lay && x >= 0 && y >= 0 && wbase >= 0)
                        out.src[fi * (SH::FRAG_BYTES / SH::ELEM_BYTES) + (size_t)lane] = (int32_t)(wbase + (int64_t)y * lin->in_features + x);
                    std::memcpy(fr + lane * 4, &w, 4);
                }
                ++written;
            }
        }
    }
    if (written != (size_t)T.nfrags) throw std::logic_error("plan / packer drift");
}

bool linear_is(const nrnerf_linear& l, int out_f, int in_f, bool need_bias) {
    return l.weight && l.out_features == out_f && l.in_features == in_f && (!need_bias || l.bias);
}

// Does (desc, mlp) match compiled architecture A?  (ArchDefault: 8x256 trunk with skip after layer 4, L = 10, bender
// 5x64, rigidity 3x32, latent 32, optional view-dependent head with L = 4; ArchDeepBend: the same with a 7-layer bender.)
template <class A>
int check_arch_t(const nrnerf_model_desc& d, const nrnerf_mlp_desc& m) {
    if (d.precision < 0 || d.precision > 2) return NRNERF_ERR_INVALID;
    if (d.multires != A::L) return NRNERF_ERR_UNSUPPORTED;
    if ((m.time_conditioned != 0) != (A::TCB != 0)) return NRNERF_ERR_UNSUPPORTED;
    if (A::TCB && d.bender) return NRNERF_ERR_UNSUPPORTED;      // the reference forbids the combination (train.py:574-576)
    if (m.use_viewdirs && d.multires_views != A::LV) return NRNERF_ERR_UNSUPPORTED;
    if (m.depth != A::D || m.width != A::W || m.skip != A::SKIP) return NRNERF_ERR_UNSUPPORTED;
    if (m.output_ch != 4 && m.output_ch != 5) return NRNERF_ERR_UNSUPPORTED;
    if (!m.pts_linears) return NRNERF_ERR_INVALID;
    const int enc = 3 + 6 * A::L + (A::TCB ? A::LAT : 0);
    if (A::TCB && m.pts_linears[0].in_features != enc) return NRNERF_ERR_UNSUPPORTED;     // other latent size
    for (int i = 0; i < A::D; ++i) {
        const int in_f = (i == 0) ? enc : ((i - 1 == A::SKIP) ? A::W + enc : A::W);
        if (!linear_is(m.pts_linears[i], A::W, in_f, true)) return NRNERF_ERR_INVALID;
    }
    if (m.use_viewdirs) {
        if (m.output_ch != 4) return NRNERF_ERR_INVALID;
        if (!linear_is(m.alpha_linear, 1, A::W, true) || !linear_is(m.feature_linear, A::W, A::W, true) ||
            !linear_is(m.views_linear, A::W / 2, A::W + 3 + 6 * A::LV, true) || !linear_is(m.rgb_linear, 3, A::W / 2, true))
            return NRNERF_ERR_INVALID;
    } else if (!linear_is(m.output_linear, m.output_ch, A::W, true)) {
        return NRNERF_ERR_INVALID;
    }
    if (d.bender) {
        const nrnerf_bender_desc& b = *d.bender;
        if (b.latent_size != A::LAT || b.depth != A::BD || b.hidden != A::BW || b.rigidity_depth != A::RD ||
            b.rigidity_hidden != A::RW)
            return NRNERF_ERR_UNSUPPORTED;
        if (!b.network || !b.rigidity_network) return NRNERF_ERR_INVALID;
        for (int i = 0; i < A::BD; ++i) {
            const int in_f = (i == 0) ? 3 + A::LAT : A::BW, out_f = (i == A::BD - 1) ? 3 : A::BW;
            if (!linear_is(b.network[i], out_f, in_f, i != A::BD - 1)) return NRNERF_ERR_INVALID;
        }
        for (int i = 0; i < A::RD; ++i) {
            const int in_f = (i == 0) ? 3 : A::RW, out_f = (i == A::RD - 1) ? 1 : A::RW;
            if (!linear_is(b.rigidity_network[i], out_f, in_f, true)) return NRNERF_ERR_INVALID;
        }
    }
    return NRNERF_OK;
}

template <class A>
void pack_arch(const nrnerf_model_desc& d, const nrnerf_mlp_desc& m, PackedPass& out, bool bender_only = false,
               const FlatLayout* lay = nullptr) {
    const bool bend = d.bender != nullptr, views = m.use_viewdirs != 0;
    auto go = [&](auto sh) {
        using SH = decltype(sh);
        if (bender_only) pack_pass<SH, A, true, false, false>(d, m, d.precision, out, lay);       // nrnerf_bend.h
        else if (bend && views) pack_pass<SH, A, true, true>(d, m, d.precision, out, lay);
        else if (bend) pack_pass<SH, A, true, false>(d, m, d.precision, out, lay);
        else if (views) pack_pass<SH, A, false, true>(d, m, d.precision, out, lay);
        else pack_pass<SH, A, false, false>(d, m, d.precision, out, lay);
    };
    if (d.precision == NRNERF_PREC_F32) go(ShapeF32{});
    else if (d.precision == NRNERF_PREC_BF16) go(Shape16Fast{});      // single-product bender (nrnerf_plan.h Shape::SPLIT)
    else go(Shape16{});
}

// picks the compiled architecture (nrnerf_plan.h ArchById) the description matches; *arch_id receives its id
int pack_dispatch(const nrnerf_model_desc& d, const nrnerf_mlp_desc& m, PackedPass& out, int* arch_id = nullptr,
                  bool bender_only = false, const FlatLayout* lay = nullptr) {
    int rc = check_arch_t<ArchDefault>(d, m);
    if (rc == NRNERF_OK) {
        pack_arch<ArchDefault>(d, m, out, bender_only, lay);
        if (arch_id) *arch_id = 0;
        return NRNERF_OK;
    }
    if (rc == NRNERF_ERR_UNSUPPORTED && !d.bender && m.time_conditioned) {
        const int rc2 = check_arch_t<ArchTimeCond>(d, m);
        if (rc2 == NRNERF_OK) {
            pack_arch<ArchTimeCond>(d, m, out, false, lay);
            if (arch_id) *arch_id = 2;
            return NRNERF_OK;
        }
        return rc2;
    }
    if (rc == NRNERF_ERR_UNSUPPORTED && d.bender) {     // arch 1 is a bender variant: only compiled with a bender
        const int rc1 = check_arch_t<ArchDeepBend>(d, m);
        if (rc1 == NRNERF_OK) {
            pack_arch<ArchDeepBend>(d, m, out, bender_only, lay);
            if (arch_id) *arch_id = 1;
            return NRNERF_OK;
        }
        if (rc1 != NRNERF_ERR_UNSUPPORTED) return rc1;
    }
    if (rc == NRNERF_ERR_UNSUPPORTED && !m.time_conditioned && !m.use_viewdirs) {      // arch 5: netwidth 128, no view-dependent head
        const int rc5 = check_arch_t<ArchNarrow>(d, m);
        if (rc5 == NRNERF_OK) {
            pack_arch<ArchNarrow>(d, m, out, bender_only, lay);
            if (arch_id) *arch_id = 5;
            return NRNERF_OK;
        }
        if (rc5 != NRNERF_ERR_UNSUPPORTED) return rc5;
    }
    return rc;
}
// ------------------------------------------------------------------------------------------
// Any other architecture: the run-time-parameterised kernel of nrnerf_generic.h.  The reference builds NeRF(D, W) for any
// --netdepth / --netwidth (and _fine), any --multires / --multires_views, a bender for any --ray_bending_latent_size
// (train.py:1004-1010, 1060, 1133-1139, 564-630); what is not one of the compiled shapes gets a layer PROGRAM here: per layer
// the fragment offset of its weights, its sources among the LDS buffers E (network input) / H (hidden) / V (second input) and
// its destination.  Columns keep the reference's order, so a fragment element is W[32 t + i][first column of the source + k].
// ------------------------------------------------------------------------------------------
struct GenProgram {
    GenArgs proto{};          // mode, L, LV, lat, layers, ke / kv / kh filled in; pointers are per launch
    PackedPass pk;
};
int pad16(int v) { return (v + 15) / 16 * 16; }
struct GenSource { int buf, col0, n; };

// one layer into the program and the packed images.  16-bit precisions: fragments read against E / V are f16, against H the
// model's type (nrnerf_generic.h).
// (t_wbase >= 0: `lin` is a TRANSPOSED copy of rows [t_row0, t_row0 + lin.in_features ... ) -- element (row r, column k) of `lin` is the
//  original layer's W[k][t_col0 + r], whose flat-vector position is t_wbase + k * t_orig_in + t_col0 + r: the device-side refresh
//  (nrnerf_model_update_device) then fills the backward-data images of a non-compiled architecture like every other image)
// (a transposed layer's place in the flat parameter buffer: element (output row r, source column k) of the layer is element
//  (k - kshift, col0 + r) of the ORIGINAL matrix at wbase, whose rows have orig_in elements; columns k < kshift are constant zeros)
struct GenTSrc { int64_t wbase = -1; int orig_in = 0, col0 = 0, kshift = 0; };
void gen_add_layer(GenProgram& g, int precision, const nrnerf_linear& lin, GenSource a, GenSource b, int dst, int relu, int o_col, const FlatLayout* lay,
                   const GenTSrc* ta = nullptr, const GenTSrc* tb = nullptr) {
    if (g.proto.n_layers >= GEN_MAX_LAYERS) throw std::logic_error("generic program too long");
    const bool f32 = precision == NRNERF_PREC_F32;
    // a fragment = 64 lanes x 16 bytes in every precision: 8 16-bit k per lane (one MFMA), or 4 fp32 k per lane (four 32x32x2 MFMAs:
    // lane half h holds k = 8 s + 4 h + e), see nrnerf_generic.h::GenTypes
    const int KH = f32 ? 4 : 8, KS = 2 * KH, FB = 1024, EB = f32 ? 4 : 2;
    GenLayer& ly = g.proto.layer[g.proto.n_layers++];
    ly.w_frag = (int)(g.pk.stream.size() / FB);
    ly.bias_tile = (int)(g.pk.bias.size() / 32);
    ly.nt = (lin.out_features + 31) / 32;
    ly.src0 = a.buf; ly.ns0 = (a.n + KS - 1) / KS;
    ly.src1 = b.buf; ly.ns1 = (b.n + KS - 1) / KS;
    ly.dst = dst; ly.relu = relu; ly.o_col = o_col; ly.o_rows = lin.out_features;
    ly.save_idx = -1; ly.mask_idx = -1; ly.boff0 = 0; ly.boff1 = 0;
    if (ly.nt > GEN_WAVES * GEN_MAXT || a.col0 + a.n > lin.in_features || b.col0 + b.n > lin.in_features) throw std::logic_error("generic layer out of range");
    const int ns = ly.ns0 + ly.ns1;
    const size_t f0 = g.pk.stream.size();
    g.pk.stream.resize(f0 + (size_t)ly.nt * ns * FB, 0);
    const bool transposed = ta != nullptr;
    const int64_t wbase = (lay && !transposed) ? lay->of(lin.weight) : -1, bbase = (lay && lin.bias && !transposed) ? lay->of(lin.bias) : -1;
    if (lay) { g.pk.src.resize(g.pk.stream.size() / EB, -1); g.pk.fmt.resize(g.pk.stream.size() / EB, 0); }
    for (int t = 0; t < ly.nt; ++t)
        for (int sl = 0; sl < ns; ++sl) {
            const GenSource& src = (sl < ly.ns0) ? a : b;
            const int s = (sl < ly.ns0) ? sl : sl - ly.ns0;
            const bool as_f16 = precision == NRNERF_PREC_F16 || src.buf != GB_H;
            uint8_t* fr = g.pk.stream.data() + f0 + ((size_t)t * ns + sl) * FB;
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, h = lane >> 5, row = 32 * t + i;
                for (int e = 0; e < KH; ++e) {
                    const int k = s * KS + h * KH + e;
                    const bool live = row < lin.out_features && k < src.n;
                    const float w = live ? lin.weight[(size_t)row * lin.in_features + src.col0 + k] : 0.0f;
                    const size_t el = (f0 + ((size_t)t * ns + sl) * FB) / EB + (size_t)lane * KH + e;
                    if (lay) {
                        if (transposed) {
                            const GenTSrc* ts = (sl < ly.ns0) ? ta : tb;
                            g.pk.src[el] = (live && ts && ts->wbase >= 0 && k >= ts->kshift) ? (int32_t)(ts->wbase + (int64_t)(k - ts->kshift) * ts->orig_in + ts->col0 + row) : -1;
                        } else g.pk.src[el] = (live && wbase >= 0) ? (int32_t)(wbase + (int64_t)row * lin.in_features + src.col0 + k) : -1;
                        g.pk.fmt[el] = f32 ? 0 : (as_f16 ? 2 : 1);
                    }
                    if (f32) std::memcpy(fr + (lane * KH + e) * 4, &w, 4);
                    else { const uint16_t q = as_f16 ? f32_to_f16(w) : f32_to_bf16(w); std::memcpy(fr + (lane * KH + e) * 2, &q, 2); }
                }
            }
        }
    const size_t b0 = g.pk.bias.size();
    g.pk.bias.resize(b0 + (size_t)ly.nt * 32, 0.0f);
    if (lay) g.pk.bias_src.resize(g.pk.bias.size(), -1);
    for (int t = 0; t < ly.nt; ++t)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * t + tile_row(r, h);
                if (row < lin.out_features && lin.bias) {
                    g.pk.bias[b0 + (size_t)t * 32 + h * 16 + r] = lin.bias[row];
                    if (lay && bbase >= 0) g.pk.bias_src[b0 + (size_t)t * 32 + h * 16 + r] = (int32_t)(bbase + row);
                }
            }
}
void gen_finish(GenProgram& g, int precision) {
    (void)precision;
    const int FB = 1024;
    g.pk.frag_bytes = FB; g.pk.slot_bytes = FB;
    g.pk.ntiles = (int)(g.pk.bias.size() / 32);
    g.proto.n_bias_tiles = g.pk.ntiles;
    g.pk.nunits = (int)(g.pk.stream.size() / FB);
    g.pk.unit_off.assign(1, 0);
}

int gen_skip(const nrnerf_mlp_desc& m) { return (m.skip >= 0 && m.skip <= m.depth - 2) ? m.skip : -1; }
// what the generic kernel takes (everything else: NRNERF_ERR_UNSUPPORTED, i.e. the reference's own function)
int gen_check_mlp(const nrnerf_model_desc& d, const nrnerf_mlp_desc& m) {
    if (d.precision < 0 || d.precision > 2) return NRNERF_ERR_INVALID;
    if (!m.pts_linears || m.depth < 1 || m.width < 1) return NRNERF_ERR_INVALID;
    if (d.multires < 0 || d.multires > 16 || m.width > GEN_MAX_W || m.depth > 16) return NRNERF_ERR_UNSUPPORTED;
    if (m.time_conditioned && d.bender) return NRNERF_ERR_UNSUPPORTED;                       // train.py:574-576
    const int enc = 3 + 6 * d.multires;
    const int lat = m.time_conditioned ? m.pts_linears[0].in_features - enc : 0;
    if (lat < 0 || lat > 64 || pad16(enc + lat) > GEN_MAX_E) return NRNERF_ERR_UNSUPPORTED;
    // NeRF.forward concatenates [input, h] after layer `skip` whatever follows (rnh:277-282): after the LAST layer the
    // reference's own head fails on the wider vector; a skip index beyond the depth never triggers
    if (m.skip == m.depth - 1) return NRNERF_ERR_UNSUPPORTED;
    const int skip = gen_skip(m);
    for (int i = 0; i < m.depth; ++i) {
        const int in_f = (i == 0) ? enc + lat : ((skip >= 0 && i - 1 == skip) ? m.width + enc + lat : m.width);
        if (!linear_is(m.pts_linears[i], m.width, in_f, true)) return NRNERF_ERR_INVALID;
    }
    if (m.use_viewdirs) {
        if (d.multires_views < 0 || d.multires_views > 10 || m.output_ch != 4) return NRNERF_ERR_UNSUPPORTED;
        const int half = m.views_linear.out_features;
        if (!linear_is(m.alpha_linear, 1, m.width, true) || !linear_is(m.feature_linear, m.width, m.width, true) || half < 1 || half > GEN_MAX_W ||
            !linear_is(m.views_linear, half, m.width + 3 + 6 * d.multires_views, true) || !linear_is(m.rgb_linear, 3, half, true))
            return NRNERF_ERR_INVALID;
    } else {
        if (m.output_ch < 4 || m.output_ch > 5) return NRNERF_ERR_UNSUPPORTED;
        if (!linear_is(m.output_linear, m.output_ch, m.width, true)) return NRNERF_ERR_INVALID;
    }
    return NRNERF_OK;
}
int gen_check_bender(const nrnerf_bender_desc& b) {
    if (!b.network || !b.rigidity_network || b.depth < 2 || b.rigidity_depth < 2) return NRNERF_ERR_INVALID;
    if (b.latent_size < 0 || b.latent_size > 64 || b.hidden < 1 || b.hidden > GEN_MAX_W || b.rigidity_hidden < 1 || b.rigidity_hidden > GEN_MAX_W ||
        b.depth + b.rigidity_depth > GEN_MAX_LAYERS)
        return NRNERF_ERR_UNSUPPORTED;
    for (int i = 0; i < b.depth; ++i) {
        const int in_f = (i == 0) ? 3 + b.latent_size : b.hidden, out_f = (i == b.depth - 1) ? 3 : b.hidden;
        if (!linear_is(b.network[i], out_f, in_f, i != b.depth - 1)) return NRNERF_ERR_INVALID;
    }
    for (int i = 0; i < b.rigidity_depth; ++i) {
        const int in_f = (i == 0) ? 3 : b.rigidity_hidden, out_f = (i == b.rigidity_depth - 1) ? 1 : b.rigidity_hidden;
        if (!linear_is(b.rigidity_network[i], out_f, in_f, true)) return NRNERF_ERR_INVALID;
    }
    return NRNERF_OK;
}
// NeRF.forward (rnh:240-314) as a layer program
void gen_pack_mlp(const nrnerf_model_desc& d, const nrnerf_mlp_desc& m, GenProgram& g, const FlatLayout* lay) {
    const int enc = 3 + 6 * d.multires, lat = m.time_conditioned ? m.pts_linears[0].in_features - enc : 0, in_w = enc + lat, W = m.width;
    g.proto.mode = 1; g.proto.L = d.multires; g.proto.LV = m.use_viewdirs ? d.multires_views : -1; g.proto.lat = lat;
    g.proto.ke = pad16(in_w);
    g.proto.kv = m.use_viewdirs ? pad16(3 + 6 * d.multires_views) : 16;
    int widest = W;
    const int skip = gen_skip(m);
    const GenSource none{GB_H, 0, 0};
    for (int i = 0; i < m.depth; ++i) {
        if (i == 0) gen_add_layer(g, d.precision, m.pts_linears[0], GenSource{GB_E, 0, in_w}, none, GB_H, 1, 0, lay);
        else if (skip >= 0 && i - 1 == skip)            // h = cat([input_pts, h]) (rnh:280-282): columns [input | hidden]
            gen_add_layer(g, d.precision, m.pts_linears[i], GenSource{GB_E, 0, in_w}, GenSource{GB_H, in_w, W}, GB_H, 1, 0, lay);
        else gen_add_layer(g, d.precision, m.pts_linears[i], GenSource{GB_H, 0, W}, none, GB_H, 1, 0, lay);
    }
    if (m.use_viewdirs) {                              // rnh:284-304
        const int half = m.views_linear.out_features, dv = 3 + 6 * d.multires_views;
        widest = imax(widest, half);
        gen_add_layer(g, d.precision, m.alpha_linear, GenSource{GB_H, 0, W}, none, GB_O, 0, 3, lay);
        gen_add_layer(g, d.precision, m.feature_linear, GenSource{GB_H, 0, W}, none, GB_H, 0, 0, lay);
        gen_add_layer(g, d.precision, m.views_linear, GenSource{GB_H, 0, W}, GenSource{GB_V, W, dv}, GB_H, 1, 0, lay);   // cat([feature, input_views])
        gen_add_layer(g, d.precision, m.rgb_linear, GenSource{GB_H, 0, half}, none, GB_O, 0, 0, lay);
    } else {
        gen_add_layer(g, d.precision, m.output_linear, GenSource{GB_H, 0, W}, none, GB_O, 0, 0, lay);
    }
    g.proto.kh = (widest + 31) / 32 * 32;
    for (int i = 0; i < m.depth; ++i) g.proto.layer[i].save_idx = i;        // (training: layer i's activations, only with GenArgs::save set)
    if (m.use_viewdirs) {                                                   // + feature_linear's outputs (slot D) and the colour branch's activations (slot D + 1)
        g.proto.layer[m.depth + 1].save_idx = m.depth;
        g.proto.layer[m.depth + 2].save_idx = m.depth + 1;
    }
    gen_finish(g, d.precision);
}
// Backward-data of a plain-headed NeRF (training of a non-compiled architecture): the forward's layers in reverse order with TRANSPOSED
// weights, no biases, run by the same kernel (GenArgs::mode 2).  H starts as the rows of d raw; output_linear^T gives d h_{D-1}, masked
// by the forward activation of layer D - 1 = d pre_{D-1} (saved as index D - 1); pts_linears[i]^T takes d pre_i to d pre_{i-1} (mask:
// activation i - 1); the layer behind the skip connection has [input | hidden] columns: its input part goes straight to memory
// (GB_OUT1: the encoding's gradient), its hidden part goes on; pts_linears[0]^T ends in the encoding's gradient (GB_OUT0).
bool gen_trainable(const nrnerf_model_desc& d, const nrnerf_mlp_desc& m) {
    if (d.precision == NRNERF_PREC_F16 || m.width % 4 != 0 || m.depth < 1) return false;
    if (m.time_conditioned && (d.bender || m.pts_linears[0].in_features > 512)) return false;
    if (m.use_viewdirs) return m.feature_linear.out_features == m.width && m.views_linear.in_features == m.width + 3 + 6 * d.multires_views &&
                               m.views_linear.out_features <= m.width && m.depth + 5 <= GEN_MAX_LAYERS && (m.width + 31) / 32 * 32 + 32 <= GEN_MAX_W;
    return m.output_ch >= 4 && m.output_ch <= 5 && m.depth + 2 <= GEN_MAX_LAYERS;          // (layers of the backward-data program)
}
// column of H where mode 2 parks the rows of d raw (plain head: 0 -- they are the first layer's only input; view-dependent head: behind the
// activations, because d sigma is needed again when the colour branch's gradient has come down to h_{D-1})
int gen_draw_col(const nrnerf_mlp_desc& m) { return m.use_viewdirs ? (m.width + 31) / 32 * 32 : 0; }
void gen_pack_mlp_bwd(const nrnerf_model_desc& d, const nrnerf_mlp_desc& m, GenProgram& g, const FlatLayout* lay) {
    // (time-conditioned baseline, rnh:207-209, 273-282: the latent code's columns follow the encoding's in both layers that read the input,
    //  and so do their gradients in the two outputs)
    const int enc = m.time_conditioned ? m.pts_linears[0].in_features : 3 + 6 * d.multires, W = m.width, D = m.depth, skip = gen_skip(m);
    g.proto.mode = 2; g.proto.L = d.multires; g.proto.LV = -1; g.proto.lat = 0;
    g.proto.ke = 16; g.proto.kv = 16;
    const GenSource none{GB_H, 0, 0};
    std::vector<std::vector<float>> keep;           // transposed copies (alive until the fragments are written: gen_add_layer copies)
    auto transposed = [&](const nrnerf_linear& lin, int col0, int ncols) {       // rows = the layer's input columns [col0, col0 + ncols), columns = its outputs
        keep.emplace_back((size_t)ncols * lin.out_features);
        std::vector<float>& t = keep.back();
        for (int r = 0; r < ncols; ++r)
            for (int k = 0; k < lin.out_features; ++k) t[(size_t)r * lin.out_features + k] = lin.weight[(size_t)k * lin.in_features + col0 + r];
        nrnerf_linear lt{};
        lt.weight = t.data(); lt.bias = nullptr; lt.out_features = ncols; lt.in_features = lin.out_features;
        return lt;
    };
    auto add = [&](const nrnerf_linear& lin, int col0, int ncols, int k_in, int dst, int mask_idx, int save_idx, int boff = 0) {
        const nrnerf_linear lt = transposed(lin, col0, ncols);
        const GenTSrc ts{lay ? lay->of(lin.weight) : -1, lin.in_features, col0, 0};
        gen_add_layer(g, d.precision, lt, GenSource{GB_H, 0, k_in}, none, dst, 0, 0, lay, &ts, nullptr);
        GenLayer& ly = g.proto.layer[g.proto.n_layers - 1];
        ly.mask_idx = mask_idx; ly.save_idx = save_idx; ly.boff0 = boff;
    };
    if (m.use_viewdirs) {
        // rgb = rgb_linear(hv), hv = relu(views_linear([feature, enc(dir)])), feature = feature_linear(h), sigma = alpha_linear(h)  (rnh:284-304)
        const int half = m.views_linear.out_features, dv = 3 + 6 * d.multires_views, dc = gen_draw_col(m);
        add(m.rgb_linear, 0, half, 3, GB_H, D + 1, D + 1, dc);                        // d rgb (columns dc .. dc + 2) -> d pre of the colour branch (slot D + 1)
        add(m.views_linear, W, dv, half, GB_OUT2, -1, -1);                            // its direction columns: the direction encoding's gradient, to memory
        add(m.views_linear, 0, W, half, GB_H, -1, D);                                 // its feature columns: d feature (slot D; feature_linear has no relu)
        // d h_{D-1} = feature_linear^T d feature + alpha_linear^T d sigma: ONE layer, columns [d feature | the d raw block, only its column 3 live]
        keep.emplace_back((size_t)W * (W + 4), 0.0f);
        std::vector<float>& t = keep.back();
        for (int r = 0; r < W; ++r) {
            for (int k = 0; k < W; ++k) t[(size_t)r * (W + 4) + k] = m.feature_linear.weight[(size_t)k * W + r];
            t[(size_t)r * (W + 4) + W + 3] = m.alpha_linear.weight[r];
        }
        nrnerf_linear lt{};
        lt.weight = t.data(); lt.bias = nullptr; lt.out_features = W; lt.in_features = W + 4;
        const GenTSrc tf{lay ? lay->of(m.feature_linear.weight) : -1, W, 0, 0}, tal{lay ? lay->of(m.alpha_linear.weight) : -1, W, 0, 3};
        gen_add_layer(g, d.precision, lt, GenSource{GB_H, 0, W}, GenSource{GB_H, W, 4}, GB_H, 0, 0, lay, &tf, &tal);
        GenLayer& ly = g.proto.layer[g.proto.n_layers - 1];
        ly.mask_idx = D - 1; ly.save_idx = D - 1; ly.boff1 = dc;
    } else {
        add(m.output_linear, 0, W, 4, GB_H, D - 1, D - 1);                             // d raw (rgb, sigma; a 5th channel never reaches a loss) -> d pre_{D-1}
    }
    for (int i = D - 1; i >= 1; --i) {
        if (skip >= 0 && i - 1 == skip) {
            add(m.pts_linears[i], 0, enc, W, GB_OUT1, -1, -1);                         // input part: gradient of the encoding, to memory
            add(m.pts_linears[i], enc, W, W, GB_H, i - 1, i - 1);                      // hidden part: d pre_{i-1}
        } else {
            add(m.pts_linears[i], 0, W, W, GB_H, i - 1, i - 1);
        }
    }
    add(m.pts_linears[0], 0, enc, W, GB_OUT0, -1, -1);
    g.proto.kh = m.use_viewdirs ? gen_draw_col(m) + 32 : (imax(W, 16) + 31) / 32 * 32;
    gen_finish(g, d.precision);
}
// ray_bending.forward (rnh:507-577): always packed (and run) in fp32
void gen_pack_bender(const nrnerf_bender_desc& b, GenProgram& g, const FlatLayout* lay) {
    g.proto.mode = 0; g.proto.L = 0; g.proto.LV = -1; g.proto.lat = b.latent_size;
    g.proto.ke = pad16(3 + b.latent_size); g.proto.kv = 16;
    const GenSource none{GB_H, 0, 0};
    for (int i = 0; i < b.depth; ++i) {
        const bool last = i == b.depth - 1;
        gen_add_layer(g, NRNERF_PREC_F32, b.network[i], i == 0 ? GenSource{GB_E, 0, 3 + b.latent_size} : GenSource{GB_H, 0, b.hidden}, none,
                      last ? GB_O : GB_H, last ? 0 : 1, 0, lay);
    }
    for (int i = 0; i < b.rigidity_depth; ++i) {
        const bool last = i == b.rigidity_depth - 1;
        gen_add_layer(g, NRNERF_PREC_F32, b.rigidity_network[i], i == 0 ? GenSource{GB_V, 0, 3} : GenSource{GB_H, 0, b.rigidity_hidden}, none,
                      last ? GB_O : GB_H, last ? 0 : 1, 3, lay);
    }
    g.proto.kh = (imax(b.hidden, b.rigidity_hidden) + 31) / 32 * 32;
    gen_finish(g, NRNERF_PREC_F32);
}

// the trunk-only / bender-only kernels of the split-bender path: the trunk is the architecture's without bender (the 5- and
// the 7-layer bender share architecture 0's), the bender kernel is compiled per bender shape (narrow trunk: the 5-layer one)
int trunk_arch(int arch_id) { return arch_id == 5 ? 5 : 0; }
int bender_arch(int arch_id) { return arch_id == 5 ? 0 : arch_id; }

// algorithmic MACs per sample, unpadded
double algo_macs(const nrnerf_model_desc& d, const nrnerf_mlp_desc& m) {
    double macs = 0;
    for (int i = 0; i < m.depth; ++i) macs += (double)m.pts_linears[i].in_features * m.pts_linears[i].out_features;
    if (m.use_viewdirs) {
        for (const nrnerf_linear* l : {&m.alpha_linear, &m.feature_linear, &m.views_linear, &m.rgb_linear})
            macs += (double)l->in_features * l->out_features;
    } else {
        macs += (double)m.output_linear.in_features * m.output_linear.out_features;
    }
    if (d.bender) {
        for (int i = 0; i < d.bender->depth; ++i) macs += (double)d.bender->network[i].in_features * d.bender->network[i].out_features;
        for (int i = 0; i < d.bender->rigidity_depth; ++i)
            macs += (double)d.bender->rigidity_network[i].in_features * d.bender->rigidity_network[i].out_features;
    }
    return macs;
}

struct PassDev {
    void* stream = nullptr;
    float* bias = nullptr;
    size_t stream_bytes = 0, bias_floats = 0;
    double algo_flops_per_sample = 0;      // 2 * MAC
    double mfma_flops_per_sample = 0;      // issued, incl. padding
    int output_ch = 4;
    // device copies of the packer's source maps (nrnerf_model_update_device); null when not recorded
    int32_t* src = nullptr; int32_t* bias_src = nullptr; uint8_t* fmt = nullptr;
    size_t n_elems = 0;
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// makes `want` the calling thread's current device for the lifetime of the guard (restored on every exit path)
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int want) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; prev = -1; return; }
        if (prev != want && hipSetDevice(want) != hipSuccess) ok = false;
        if (prev == want) prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

struct nrnerf_model {
    int device = 0, precision = 0, has_bend = 0, views = 0, arch_id = 0, needs_latents = 0, num_cus = 0, latent_size = 0, exact = 0;
    PassDev coarse, fine;
    bool fine_is_coarse = false;
    // split-bender path (bender; finite-difference view directions if any): the fine network WITHOUT the bender layers (its input points
    // come from the stand-alone bender kernel) and the bender + rigidity layers alone
    PassDev fine_trunk, coarse_trunk, bend_only;
    bool split_ok = false;
    // the fine network's trunk once more, packed for the 16x16x32 kernel (nrnerf_net_x16.h): what the split-bender path's fine pass
    // runs when the call asks for no detail outputs
    PassDev fine_trunk_x16, coarse_trunk_x16;
    PassDev bend_x16;             // the bender + rigidity MLPs packed for the 16x16x32 stand-alone bender ("bf16" mode)
    // training (nrnerf_train.h): transposed trunk weights of both networks; train_ok: see training_eligible, fp32 or bf16
    PassDev coarse_bwd, fine_bwd;
    bool train_ok = false;
    // view-dependent head / time-conditioned baseline: bender-less forward images for trunk_fwd_train (with both branches of the head /
    // without the latent columns)
    PassDev coarse_train, fine_train;
    // training of the ray bender (nrnerf_train_bend.h): its layers alone in fp32 (whatever the model's precision) and
    // their transposes; bend_train_ok: train_ok and a bender
    PassDev bend_train_fwd, bend_train_bwd;
    bool bend_train_ok = false;
    // generic architecture (nrnerf_generic.h): layer programs instead of compiled plans; gen_* hold their packed images
    bool generic = false;
    int gen_compiled_bender = -1;   // >= 0: the bender has one of the compiled shapes (0: 5 x 64, 1: 7 x 64; latent 32, rigidity 3 x 32) and the
                                    // stand-alone bender kernel (nrnerf_bend.h, image `bend_only`) takes the passes without detail outputs
    GenArgs gen_bend_prog{}, gen_coarse_prog{}, gen_fine_prog{};
    PassDev gen_bend, gen_coarse, gen_fine;
    // the trunks of a generic model packed for the width-class 16x16x32 kernel (nrnerf_gx16.h): 16-bit modes, no view-dependent head
    PassDev gx_coarse, gx_fine;
    GxMeta gx_meta_coarse, gx_meta_fine;
    // training of a generic model with a plain head (fp32 / bf16): the backward-data programs (transposed weights); the forward is
    // gen_coarse / gen_fine run with GenArgs::save set
    bool gen_train_ok = false;
    GenArgs gen_coarse_bwd_prog{}, gen_fine_bwd_prog{};
    PassDev gen_coarse_bwd, gen_fine_bwd;
    struct GenTrainNet { int W = 0, D = 0, dv = 0, draw_col = 0, in_w = 0, lat = 0; bool skip = false, views = false; } gen_tn[2];     // [coarse, fine]
    bool gen_fine_is_coarse = false;
    int64_t flat_floats = 0;      // length of the flat parameter vector nrnerf_model_update_device expects
    PassDev gx_coarse_bwd, gx_fine_bwd;          // backward-data programs of the width-class trunks (nrnerf_gx16_bwd.h), when gx16_trainable
    GxMeta gx_meta_coarse_bwd, gx_meta_fine_bwd;
    unsigned* adam_barrier = nullptr;   // two words of device memory: the grid barrier of nrnerf_adam_step (nrnerf_optim.hip)
    // profiling (guarded; the render path itself is otherwise read-only on the handle)
    mutable std::mutex prof_mu;
    mutable bool prof_on = false;
    struct Ev { int kernel; hipEvent_t a, b; double flops, mfma; const char* name; };
    mutable std::vector<Ev> prof_events;
};

namespace {
constexpr int NRN_RING_LAG_HOST = 2;      // NRN_RING_LAG of nrnerf_net_impl.h (device header): units of DMA lead behind the ring's read position

int upload_pass(const PackedPass& pk, PassDev& dev) {
    dev.stream_bytes = pk.stream.size();
    dev.bias_floats = pk.bias.size();
    if (hipMalloc(&dev.stream, pk.stream.size()) != hipSuccess) return NRNERF_ERR_NOMEM;
    if (hipMalloc((void**)&dev.bias, pk.bias.size() * 4) != hipSuccess) return NRNERF_ERR_NOMEM;
    if (hipMemcpy(dev.stream, pk.stream.data(), pk.stream.size(), hipMemcpyHostToDevice) != hipSuccess) return NRNERF_ERR_HIP;
    if (hipMemcpy(dev.bias, pk.bias.data(), pk.bias.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return NRNERF_ERR_HIP;
    if (!pk.src.empty()) {
        dev.n_elems = pk.src.size();
        if (hipMalloc((void**)&dev.src, pk.src.size() * 4) != hipSuccess || hipMalloc((void**)&dev.fmt, pk.fmt.size()) != hipSuccess ||
            hipMalloc((void**)&dev.bias_src, pk.bias_src.size() * 4) != hipSuccess) return NRNERF_ERR_NOMEM;
        if (hipMemcpy(dev.src, pk.src.data(), pk.src.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(dev.fmt, pk.fmt.data(), pk.fmt.size(), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(dev.bias_src, pk.bias_src.data(), pk.bias_src.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return NRNERF_ERR_HIP;
    }
    return NRNERF_OK;
}
// new weights of the same architecture into the buffers the kernels already read (stream-ordered)
int refresh_pass(const PackedPass& pk, PassDev& dev, hipStream_t stream) {
    if (pk.stream.size() != dev.stream_bytes || pk.bias.size() != dev.bias_floats) return NRNERF_ERR_INVALID;
    if (hipMemcpyAsync(dev.stream, pk.stream.data(), pk.stream.size(), hipMemcpyHostToDevice, stream) != hipSuccess) return NRNERF_ERR_HIP;
    if (hipMemcpyAsync(dev.bias, pk.bias.data(), pk.bias.size() * 4, hipMemcpyHostToDevice, stream) != hipSuccess) return NRNERF_ERR_HIP;
    return NRNERF_OK;
}
void free_pass(PassDev& dev) {
    if (dev.stream) (void)hipFree(dev.stream);
    if (dev.bias) (void)hipFree(dev.bias);
    if (dev.src) (void)hipFree(dev.src);
    if (dev.fmt) (void)hipFree(dev.fmt);
    if (dev.bias_src) (void)hipFree(dev.bias_src);
    dev = PassDev{};
}

// The two extra weight images of the split-bender path: the fine network without its bender layers (same trunk for the
// 5- and the 7-layer bender: compiled architecture 0 without bender) and the bender + rigidity layers alone.
int pack_split(const nrnerf_model_desc& d, PackedPass& trunk, PackedPass& bend, PackedPass* coarse_trunk = nullptr,
               const FlatLayout* lay = nullptr) {
    const nrnerf_mlp_desc& fm = d.fine ? *d.fine : *d.coarse;
    nrnerf_model_desc d2 = d;
    d2.bender = nullptr;
    int rc = pack_dispatch(d2, fm, trunk, nullptr, false, lay);
    if (rc != NRNERF_OK) return rc;
    if (coarse_trunk) {
        rc = pack_dispatch(d2, *d.coarse, *coarse_trunk, nullptr, false, lay);
        if (rc != NRNERF_OK) return rc;
    }
    return pack_dispatch(d, fm, bend, nullptr, /*bender_only=*/true, lay);
}

// transposed trunk weights for the backward-data kernel; eligible models only (see nrnerf_model::train_ok)
bool training_eligible(const nrnerf_model_desc& d, const nrnerf_model* m) {
    // (with the view-dependent head: the directions are an input of the trunk's training kernels -- finite differences, the rays' own, or
    //  the exact Jacobian directions, whose tangent and its gradient come from nrnerf_bender_divergence_* (tangent / g_tangent))
    // (time-conditioned baseline, architecture 2: trained through the plain trunk's kernels, the latent columns of its two
    //  input layers as per-ray biases -- pack_pass, tcb_shift)
    return (m->arch_id <= 2 || m->arch_id == 5) && d.precision != NRNERF_PREC_F16;
}
int tcb_shift_of(const nrnerf_mlp_desc& mlp) {      // latent columns of a time-conditioned trunk (0: plain trunk)
    return mlp.time_conditioned ? mlp.pts_linears[0].in_features - (3 + 6 * ArchDefault::L) : 0;
}
void pack_bwd(const nrnerf_model_desc& d, const nrnerf_mlp_desc& mlp, PackedPass& out, const FlatLayout* lay = nullptr) {
    const bool narrow = mlp.width == ArchNarrow::W, views = mlp.use_viewdirs != 0;      // (no view-dependent head at width 128)
    const int ts = tcb_shift_of(mlp);
    if (d.precision == NRNERF_PREC_F32) {
        if (narrow) pack_pass_bwd<ShapeF32, ArchNarrow>(mlp, d.precision, out, lay, ts);
        else if (views) pack_pass_bwd<ShapeF32, ArchDefault, true>(mlp, d.precision, out, lay, ts);
        else pack_pass_bwd<ShapeF32, ArchDefault>(mlp, d.precision, out, lay, ts);
    } else {
        if (narrow) pack_pass_bwd<Shape16, ArchNarrow>(mlp, d.precision, out, lay, ts);
        else if (views) pack_pass_bwd<Shape16, ArchDefault, true>(mlp, d.precision, out, lay, ts);
        else pack_pass_bwd<Shape16, ArchDefault>(mlp, d.precision, out, lay, ts);
    }
}
int upload_training(const nrnerf_model_desc& d, nrnerf_model* m, hipStream_t refresh_stream = nullptr, bool refresh = false,
                    const FlatLayout* lay = nullptr) {
    if (!training_eligible(d, m)) return NRNERF_OK;
    PackedPass bc, bf;
    pack_bwd(d, *d.coarse, bc, lay);
    int rc = refresh ? refresh_pass(bc, m->coarse_bwd, refresh_stream) : upload_pass(bc, m->coarse_bwd);
    if (rc == NRNERF_OK && d.fine) {
        pack_bwd(d, *d.fine, bf, lay);
        rc = refresh ? refresh_pass(bf, m->fine_bwd, refresh_stream) : upload_pass(bf, m->fine_bwd);
    }
    if (refresh && rc == NRNERF_OK && hipStreamSynchronize(refresh_stream) != hipSuccess) rc = NRNERF_ERR_HIP;   // host images die here
    if (rc == NRNERF_OK && (m->views || d.coarse->time_conditioned)) {       // bender-less forward images: with the view-dependent head / without the latent columns
        auto pack_train = [&](const nrnerf_mlp_desc& mlp, PackedPass& out) {
            nrnerf_model_desc d2 = d;
            d2.bender = nullptr;
            const int ts = tcb_shift_of(mlp);
            if (mlp.use_viewdirs) {             // trunk + both branches of the view-dependent head (trunk_fwd_train<.., VIEWS>)
                if (d.precision == NRNERF_PREC_F32) pack_pass<ShapeF32, ArchDefault, false, true>(d2, mlp, d.precision, out, lay, ts);
                else pack_pass<Shape16Fast, ArchDefault, false, true>(d2, mlp, d.precision, out, lay, ts);
            } else if (d.precision == NRNERF_PREC_F32) pack_pass<ShapeF32, ArchDefault, false, false>(d2, mlp, d.precision, out, lay, ts);
            else pack_pass<Shape16Fast, ArchDefault, false, false>(d2, mlp, d.precision, out, lay, ts);
        };
        PackedPass tc, tf;
        pack_train(*d.coarse, tc);
        rc = refresh ? refresh_pass(tc, m->coarse_train, refresh_stream) : upload_pass(tc, m->coarse_train);
        if (rc == NRNERF_OK && d.fine) {
            pack_train(*d.fine, tf);
            rc = refresh ? refresh_pass(tf, m->fine_train, refresh_stream) : upload_pass(tf, m->fine_train);
        }
        if (refresh && rc == NRNERF_OK && hipStreamSynchronize(refresh_stream) != hipSuccess) rc = NRNERF_ERR_HIP;
    }
    const double mfma_flop = 2.0 * 32 * 32 * (d.precision == NRNERF_PREC_F32 ? 2 : 16);
    m->coarse_bwd.mfma_flops_per_sample = bc.mfma_per_block * mfma_flop / 32.0;
    m->fine_bwd.mfma_flops_per_sample = (d.fine ? bf.mfma_per_block : bc.mfma_per_block) * mfma_flop / 32.0;
    m->train_ok = (rc == NRNERF_OK);
    if (rc == NRNERF_OK && d.bender) {
        PackedPass bfw, bbw;
        nrnerf_model_desc d32 = d;
        d32.precision = NRNERF_PREC_F32;
        if (bender_arch(m->arch_id) == 0) {       // the bender-only plan does not depend on the trunk's width
            pack_pass<ShapeF32, ArchDefault, true, false, false>(d32, *d.coarse, NRNERF_PREC_F32, bfw, lay);
            pack_pass_bwd_bender<ArchDefault>(*d.bender, bbw, lay);
        } else {
            pack_pass<ShapeF32, ArchDeepBend, true, false, false>(d32, *d.coarse, NRNERF_PREC_F32, bfw, lay);
            pack_pass_bwd_bender<ArchDeepBend>(*d.bender, bbw, lay);
        }
        rc = refresh ? refresh_pass(bfw, m->bend_train_fwd, refresh_stream) : upload_pass(bfw, m->bend_train_fwd);
        if (rc == NRNERF_OK) rc = refresh ? refresh_pass(bbw, m->bend_train_bwd, refresh_stream) : upload_pass(bbw, m->bend_train_bwd);
        if (refresh && rc == NRNERF_OK && hipStreamSynchronize(refresh_stream) != hipSuccess) rc = NRNERF_ERR_HIP;
        m->bend_train_ok = (rc == NRNERF_OK);
    }
    return rc;
}

// does the bender have compiled architecture A's bender shape?  (check_arch_t's bender part)
template <class A>
bool bender_matches(const nrnerf_bender_desc& b) {
    if (b.latent_size != A::LAT || b.depth != A::BD || b.hidden != A::BW || b.rigidity_depth != A::RD || b.rigidity_hidden != A::RW) return false;
    if (!b.network || !b.rigidity_network) return false;
    for (int i = 0; i < A::BD; ++i)
        if (!linear_is(b.network[i], (i == A::BD - 1) ? 3 : A::BW, (i == 0) ? 3 + A::LAT : A::BW, i != A::BD - 1)) return false;
    for (int i = 0; i < A::RD; ++i)
        if (!linear_is(b.rigidity_network[i], (i == A::RD - 1) ? 1 : A::RW, (i == 0) ? 3 : A::RW, true)) return false;
    return true;
}

// ---- the width-class trunk kernel for architectures outside the compiled set (nrnerf_gx16.h, nrnerf_gx16_plan.h)
// does it take this network?  16-bit modes, no view-dependent head, no time conditioning, <= 10 encoding frequencies, 4 / 5 output channels
bool gx16_eligible(const nrnerf_model_desc& d, const nrnerf_mlp_desc& m) {
    if (d.precision != NRNERF_PREC_BF16 && d.precision != NRNERF_PREC_F16) return false;
    if (m.time_conditioned || d.multires < 0 || d.multires > GX_MAX_L) return false;
    if (m.width < 1 || m.width > 512 || m.depth < 1 || m.depth > 16) return false;
    if (m.use_viewdirs) {            // view-dependent head: <= 4 direction frequencies, finite-difference directions, views layer <= half the class
        if (d.multires_views < 0 || d.multires_views > GX_MAX_LV || (d.exact_viewdirs && d.bender)) return false;
        if (m.views_linear.out_features > gx_width_class(m.width) / 2 || m.feature_linear.out_features != m.width) return false;
        return true;
    }
    if (m.output_ch != 4 && m.output_ch != 5) return false;
    return true;
}
// The stream: the layers' fragment blocks back to back in evaluation order (IN, then HID / SKIP per pts_linears[i], then HEAD), each
// padded to gx_layer_units() units, + a copy of the stream's first RING - LAG units behind the last (the ring runs on into the next
// iteration's first units).  Fragment element (lane (r, g), e) of (tile t, k-step s): W[row][col] with the maps of nrnerf_gx16_plan.h.
void pack_gx16(const nrnerf_model_desc& d, const nrnerf_mlp_desc& mlp, PackedPass& out, GxMeta& meta, const FlatLayout* lay = nullptr) {
    using SH = Shape16Fast;
    const int wc = gx_width_class(mlp.width), D = mlp.depth, skip = gen_skip(mlp), L = d.multires, enc = 3 + 6 * L, W = mlp.width;
    std::vector<int> kinds;
    kinds.push_back(GX_IN);
    for (int i = 1; i < D; ++i) kinds.push_back((i - 1 == skip) ? GX_SKIP : GX_HID);
    const bool views = mlp.use_viewdirs != 0;
    if (views) { kinds.push_back(GX_VIEWS); kinds.push_back(GX_RGB); }
    else kinds.push_back(GX_HEAD);
    std::unique_ptr<FoldedViews> folded;
    if (views) folded.reset(new FoldedViews(mlp));
    const int EV = 3 + 6 * d.multires_views;
    int units = 0, tiles = 0, mfma = 0;
    for (int k : kinds) { units += gx_layer_units(wc, k); tiles += gx_layer_tiles(wc, k); }
    constexpr int TAIL = RING - NRN_RING_LAG_HOST;
    out.ntiles = tiles; out.nunits = units + TAIL;
    out.frag_bytes = SH::FRAG_BYTES; out.slot_bytes = SH::UNIT_BYTES;
    out.stream.assign((size_t)(units + TAIL) * SH::UNIT_BYTES, 0);
    out.unit_off.assign(units + TAIL + 1, 0);
    for (int u = 0; u <= units + TAIL; ++u) out.unit_off[u] = (uint32_t)((size_t)u * SH::UNIT_BYTES / 16);
    out.bias.assign((size_t)tiles * 16, 0.0f);
    if (lay) {
        out.src.assign(out.stream.size() / 2, -1);
        out.fmt.assign(out.stream.size() / 2, 1);
        out.bias_src.assign(out.bias.size(), -1);
    }
    size_t unit0 = 0, tile0 = 0;
    for (size_t li = 0; li < kinds.size(); ++li) {
        const int kind = kinds[li];
        const Tables T = build_tables_gx(wc, kind);
        const LayerSpec& sp = T.layers[0];
        const nrnerf_linear* lin0 = (kind == GX_HEAD) ? &mlp.output_linear : (kind == GX_RGB ? &mlp.rgb_linear : (kind == GX_VIEWS ? &folded->lin : &mlp.pts_linears[li]));
        int64_t wbase0 = (lay && kind != GX_VIEWS) ? lay->of(lin0->weight) : -1, bbase0 = (lay && kind != GX_VIEWS && lin0->bias) ? lay->of(lin0->bias) : -1;
        if (kind == GX_VIEWS && lay) {             // the derived entries of the flat vector (FlatLayout::add_folded), as pack_pass
            wbase0 = lay->folded_of(mlp.views_linear.weight);
            bbase0 = wbase0 < 0 ? -1 : wbase0 + (int64_t)lin0->out_features * lin0->in_features;
        }
        for (int t = 0; t < sp.nt; ++t) {
            const bool alpha_tile = kind == GX_VIEWS && t == sp.nt - 1;          // the views layer's last tile: alpha_linear (row 0)
            const nrnerf_linear* lin = alpha_tile ? &mlp.alpha_linear : lin0;
            const int64_t wbase = alpha_tile ? (lay ? lay->of(lin->weight) : -1) : wbase0;
            const int64_t bbase = alpha_tile ? ((lay && lin->bias) ? lay->of(lin->bias) : -1) : bbase0;
            const TileInfo& ti = T.tiles[t];
            auto row_of = [&](int r) {
                if (alpha_tile) return r == 0 ? 0 : -1;
                if (kind == GX_HEAD) return r < lin->out_features ? r : -1;
                if (kind == GX_RGB) return r < 3 ? r : -1;
                return (16 * t + r < lin->out_features) ? 16 * t + r : -1;
            };
            for (int s = 0; s < sp.ns; ++s) {
                const size_t fi = unit0 * SH::UNIT_FRAGS + (size_t)ti.gbase + (size_t)s * ti.gstride;
                uint8_t* fr = out.stream.data() + fi * SH::FRAG_BYTES;
                const bool enc_step = ((kind == GX_IN || kind == GX_SKIP) && s < GX_NS_E) || (kind == GX_VIEWS && s == 0);
                const bool as_f16 = d.precision == NRNERF_PREC_F16 || enc_step;
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 15, g = lane >> 4;
                    const int row = row_of(r);
                    for (int e = 0; e < 8; ++e) {
                        int col;
                        if (kind == GX_VIEWS) {            // folded columns: hidden (W) first, then the direction encoding's (EV); alpha: hidden only
                            if (s == 0) { const int c = gx_enc_col(d.multires_views, 0, g, e); col = (c < 0 || alpha_tile) ? -1 : (lin->in_features - EV) + c; }
                            else { const int c = x16_hidden_feature(s - 1, g, e); col = c < W ? c : -1; }
                        } else if (kind == GX_RGB) {
                            const int c = x16_hidden_feature(s, g, e);
                            col = c < lin->in_features ? c : -1;
                        } else if (enc_step) col = gx_enc_col(L, s, g, e);
                        else {
                            const int c = x16_hidden_feature(kind == GX_SKIP ? s - GX_NS_E : s, g, e);
                            col = c < W ? (kind == GX_SKIP ? enc + c : c) : -1;
                        }
                        if (col >= lin->in_features) col = -1;
                        const float w = (row < 0 || col < 0) ? 0.0f : lin->weight[(size_t)row * lin->in_features + col];
                        const size_t el = fi * (SH::FRAG_BYTES / 2) + (size_t)lane * 8 + e;
                        if (lay) {
                            out.src[el] = (row < 0 || col < 0 || wbase < 0) ? -1 : (int32_t)(wbase + (int64_t)row * lin->in_features + col);
                            out.fmt[el] = as_f16 ? 2 : 1;
                        }
                        const uint16_t q = as_f16 ? f32_to_f16(w) : f32_to_bf16(w);
                        std::memcpy(fr + (lane * 8 + e) * 2, &q, 2);
                    }
                }
            }
            for (int r = 0; r < 16; ++r) {
                const int row = row_of(r);
                out.bias[(tile0 + t) * 16 + r] = (row >= 0 && lin->bias) ? lin->bias[row] : 0.0f;
                if (lay && row >= 0 && bbase >= 0) out.bias_src[(tile0 + t) * 16 + r] = (int32_t)(bbase + row);
            }
        }
        mfma += sp.ns * sp.nt;
        unit0 += gx_layer_units(wc, kind);
        tile0 += sp.nt;
    }
    // the copy of the first units behind the last layer
    std::memcpy(out.stream.data() + (size_t)units * SH::UNIT_BYTES, out.stream.data(), (size_t)TAIL * SH::UNIT_BYTES);
    if (lay) {
        const size_t n = (size_t)TAIL * SH::UNIT_BYTES / 2, o = (size_t)units * SH::UNIT_BYTES / 2;
        std::copy(out.src.begin(), out.src.begin() + n, out.src.begin() + o);
        std::copy(out.fmt.begin(), out.fmt.begin() + n, out.fmt.begin() + o);
    }
    out.mfma_per_block = mfma;
    meta.wc = wc; meta.depth = D; meta.skip = skip; meta.L = L; meta.n_bias_tiles = tiles; meta.views = views ? 1 : 0; meta.LV = d.multires_views;
}

// The backward-data program of the same trunk (nrnerf_gx16_bwd.h; plain head, bf16): output_linear^T, pts_linears[D-1 .. 1]^T, pts_linears[0]^T,
// every fragment element (tile t, row r, k-slot (s, g, e)) = W[k feature][column of output row] -- the k feature of a hidden k-step is
// x16_hidden_feature(s, g, e) (the operand order d z is handed on in), of the d raw k-step the channel 8 g + e < 4; an output row is a hidden
// feature 16 t + r, or -- the four tiles in front of them in the two layers that read the encoding -- the encoding's slot position.
// No biases (the table is zeros).  Same stream conventions as pack_gx16.
void pack_gx16_bwd(const nrnerf_model_desc& d, const nrnerf_mlp_desc& mlp, PackedPass& out, GxMeta& meta, const FlatLayout* lay = nullptr) {
    using SH = Shape16Fast;
    const int wc = gx_width_class(mlp.width), D = mlp.depth, skip = gen_skip(mlp), L = d.multires, enc = 3 + 6 * L, W = mlp.width;
    std::vector<int> kinds, layer_of;
    kinds.push_back(GX_BHEAD); layer_of.push_back(-1);
    for (int i = D - 1; i >= 1; --i) { kinds.push_back((i - 1 == skip) ? GX_BSKIP : GX_BHID); layer_of.push_back(i); }
    kinds.push_back(GX_BIN); layer_of.push_back(0);
    int units = 0, tiles = 0, mfma = 0;
    for (int k : kinds) { units += gx_layer_units(wc, k); tiles += gx_layer_tiles(wc, k); }
    constexpr int TAIL = RING - NRN_RING_LAG_HOST;
    out.ntiles = tiles; out.nunits = units + TAIL;
    out.frag_bytes = SH::FRAG_BYTES; out.slot_bytes = SH::UNIT_BYTES;
    out.stream.assign((size_t)(units + TAIL) * SH::UNIT_BYTES, 0);
    out.unit_off.assign(units + TAIL + 1, 0);
    for (int u = 0; u <= units + TAIL; ++u) out.unit_off[u] = (uint32_t)((size_t)u * SH::UNIT_BYTES / 16);
    out.bias.assign((size_t)tiles * 16, 0.0f);
    if (lay) {
        out.src.assign(out.stream.size() / 2, -1);
        out.fmt.assign(out.stream.size() / 2, 1);
        out.bias_src.assign(out.bias.size(), -1);
    }
    size_t unit0 = 0;
    for (size_t li = 0; li < kinds.size(); ++li) {
        const int kind = kinds[li];
        const Tables T = build_tables_gx(wc, kind);
        const LayerSpec& sp = T.layers[0];
        const nrnerf_linear* lin = (kind == GX_BHEAD) ? &mlp.output_linear : &mlp.pts_linears[layer_of[li]];
        const int64_t wbase = lay ? lay->of(lin->weight) : -1;
        const int enc_tiles = (kind == GX_BSKIP || kind == GX_BIN) ? 4 : 0;
        for (int t = 0; t < sp.nt; ++t) {
            const TileInfo& ti = T.tiles[t];
            // the column of the layer's weight this output row is the gradient of (-1: none)
            auto col_of = [&](int r) {
                if (t < enc_tiles) return gx_enc_col_of_pos(L, 16 * t + r);
                const int f = 16 * (t - enc_tiles) + r;
                if (f >= W) return -1;
                return kind == GX_BSKIP ? enc + f : f;
            };
            for (int s = 0; s < sp.ns; ++s) {
                const size_t fi = unit0 * SH::UNIT_FRAGS + (size_t)ti.gbase + (size_t)s * ti.gstride;
                uint8_t* fr = out.stream.data() + fi * SH::FRAG_BYTES;
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 15, g = lane >> 4;
                    const int col = (kind == GX_BIN && t >= enc_tiles) ? -1 : col_of(r);
                    for (int e = 0; e < 8; ++e) {
                        int row;                     // the weight's ROW = the k feature
                        if (kind == GX_BHEAD) row = (g == 0 && e < 4) ? e : -1;
                        else { const int c = x16_hidden_feature(s, g, e); row = c < W ? c : -1; }
                        if (row >= lin->out_features) row = -1;
                        const bool live = row >= 0 && col >= 0 && col < lin->in_features;
                        const float w = live ? lin->weight[(size_t)row * lin->in_features + col] : 0.0f;
                        const size_t el = fi * (SH::FRAG_BYTES / 2) + (size_t)lane * 8 + e;
                        if (lay) {
                            out.src[el] = (live && wbase >= 0) ? (int32_t)(wbase + (int64_t)row * lin->in_features + col) : -1;
                            out.fmt[el] = 1;
                        }
                        const uint16_t q = f32_to_bf16(w);
                        std::memcpy(fr + (lane * 8 + e) * 2, &q, 2);
                    }
                }
            }
        }
        mfma += sp.ns * sp.nt;
        unit0 += gx_layer_units(wc, kind);
    }
    std::memcpy(out.stream.data() + (size_t)units * SH::UNIT_BYTES, out.stream.data(), (size_t)TAIL * SH::UNIT_BYTES);
    if (lay) {
        const size_t n = (size_t)TAIL * SH::UNIT_BYTES / 2, o = (size_t)units * SH::UNIT_BYTES / 2;
        std::copy(out.src.begin(), out.src.begin() + n, out.src.begin() + o);
        std::copy(out.fmt.begin(), out.fmt.begin() + n, out.fmt.begin() + o);
    }
    out.mfma_per_block = mfma;
    meta.wc = wc; meta.depth = D; meta.skip = skip; meta.L = L; meta.n_bias_tiles = tiles; meta.views = 0; meta.LV = 0;
}
// which trunks the x16 training kernels take (forward with saves: gx16_kernel<.., SAVE>; backward-data: gx16_bwd_kernel): bf16, plain head, no
// latent input columns, width % 4 == 0 (rows of whole 8-byte pieces)
bool gx16_trainable(const nrnerf_model_desc& d, const nrnerf_mlp_desc& m) {
    return d.precision == NRNERF_PREC_BF16 && gx16_eligible(d, m) && !m.use_viewdirs && !m.time_conditioned && m.width % 4 == 0;
}

// ---- models of an architecture outside the compiled set (nrnerf_generic.h)
int gen_pack_all(const nrnerf_model_desc& d, const FlatLayout* lay, GenProgram& gb, GenProgram& gc, GenProgram& gf) {
    // exact Jacobian view directions (rnh:358-385) off the compiled set: the tangent J d comes from the bender's compiled divergence kernel
    // (ray mode of bend_div_fwd), so the BENDER must have one of the two compiled shapes, and the handle its training images (not "f16")
    if (d.exact_viewdirs && d.bender && d.coarse->use_viewdirs &&
        (d.precision == NRNERF_PREC_F16 || !(bender_matches<ArchDefault>(*d.bender) || bender_matches<ArchDeepBend>(*d.bender))))
        return NRNERF_ERR_UNSUPPORTED;
    int rc = gen_check_mlp(d, *d.coarse);
    if (rc == NRNERF_OK && d.fine) rc = gen_check_mlp(d, *d.fine);
    if (rc == NRNERF_OK && d.fine && (d.fine->time_conditioned != 0) != (d.coarse->time_conditioned != 0)) rc = NRNERF_ERR_INVALID;
    if (rc == NRNERF_OK && d.bender) rc = gen_check_bender(*d.bender);
    if (rc != NRNERF_OK) return rc;
    if (d.bender) gen_pack_bender(*d.bender, gb, lay);
    gen_pack_mlp(d, *d.coarse, gc, lay);
    if (d.fine) gen_pack_mlp(d, *d.fine, gf, lay);
    return NRNERF_OK;
}
// issued MFMA flops per sample of a layer program (padding included): every (tile, k-slab) is one 32 x 32 x KS MFMA per 32 samples
double gen_mfma_flops_per_sample(const GenArgs& g, bool f32) {
    double f = 0;
    for (int l = 0; l < g.n_layers; ++l) f += (double)g.layer[l].nt * (g.layer[l].ns0 + g.layer[l].ns1) * 2.0 * 32 * (f32 ? 8 : 16);
    return f;
}
// the compiled bender's training images for a generic handle (as pack_training's: the fp32 forward plan and the backward plan)
void gen_pack_bender_train(const nrnerf_model_desc& d, int cb, PackedPass& bfw, PackedPass& bbw, const FlatLayout* lay) {
    nrnerf_model_desc d32 = d;
    d32.precision = NRNERF_PREC_F32;
    if (cb == 0) {
        pack_pass<ShapeF32, ArchDefault, true, false, false>(d32, *d.coarse, NRNERF_PREC_F32, bfw, lay);
        pack_pass_bwd_bender<ArchDefault>(*d.bender, bbw, lay);
    } else {
        pack_pass<ShapeF32, ArchDeepBend, true, false, false>(d32, *d.coarse, NRNERF_PREC_F32, bfw, lay);
        pack_pass_bwd_bender<ArchDeepBend>(*d.bender, bbw, lay);
    }
}
int create_generic(const nrnerf_model_desc& d, const FlatLayout& lay, nrnerf_model** out) {
    GenProgram gb, gc, gf;
    int rc = gen_pack_all(d, &lay, gb, gc, gf);
    if (rc != NRNERF_OK) return rc;
    DeviceGuard guard(d.device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    struct Owner {
        nrnerf_model* m;
        ~Owner() { if (m) nrnerf_model_destroy(m); }
    } own{new (std::nothrow) nrnerf_model()};
    nrnerf_model* m = own.m;
    if (!m) return NRNERF_ERR_NOMEM;
    m->generic = true;
    m->device = d.device; m->precision = d.precision;
    m->has_bend = d.bender != nullptr;
    m->views = d.coarse->use_viewdirs != 0;
    m->arch_id = -1; m->exact = 0;
    m->needs_latents = m->has_bend || d.coarse->time_conditioned;
    m->latent_size = d.bender ? d.bender->latent_size : gc.proto.lat;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d.device) != hipSuccess) return NRNERF_ERR_HIP;
    m->num_cus = prop.multiProcessorCount;
    if (hipMalloc((void**)&m->adam_barrier, 2 * sizeof(unsigned)) != hipSuccess || hipMemset(m->adam_barrier, 0, 2 * sizeof(unsigned)) != hipSuccess) return NRNERF_ERR_NOMEM;
    m->flat_floats = lay.total;
    nrnerf_model_desc d2 = d;
    d2.bender = nullptr;
    if (d.bender) {
        rc = upload_pass(gb.pk, m->gen_bend);
        if (rc != NRNERF_OK) return rc;
        m->gen_bend_prog = gb.proto;
        m->gen_bend.algo_flops_per_sample = 2.0 * (algo_macs(d, *d.coarse) - algo_macs(d2, *d.coarse));
        // the usual case -- an odd TRUNK with the reference's hard-coded bender (rnh:406-407): its compiled stand-alone kernel
        const int cb = bender_matches<ArchDefault>(*d.bender) ? 0 : (bender_matches<ArchDeepBend>(*d.bender) ? 1 : -1);
        if (cb >= 0) {
            PackedPass pb;
            auto go = [&](auto sh) {
                using SH = decltype(sh);
                if (cb == 0) pack_pass<SH, ArchDefault, true, false, false>(d, *d.coarse, d.precision, pb, &lay);
                else pack_pass<SH, ArchDeepBend, true, false, false>(d, *d.coarse, d.precision, pb, &lay);
            };
            if (d.precision == NRNERF_PREC_F32) go(ShapeF32{});
            else if (d.precision == NRNERF_PREC_BF16) go(Shape16Fast{});
            else go(Shape16{});
            rc = upload_pass(pb, m->bend_only);
            if (rc != NRNERF_OK) return rc;
            m->bend_only.algo_flops_per_sample = m->gen_bend.algo_flops_per_sample;
            m->gen_compiled_bender = cb;
            if (d.precision != NRNERF_PREC_F16) {    // ... and its training kernels (forward with saved activations, backward, divergence chains)
                PackedPass bfw, bbw;
                gen_pack_bender_train(d, cb, bfw, bbw, &lay);
                rc = upload_pass(bfw, m->bend_train_fwd);
                if (rc == NRNERF_OK) rc = upload_pass(bbw, m->bend_train_bwd);
                if (rc != NRNERF_OK) return rc;
                m->bend_train_ok = true;
            }
            if (bend_x16_eligible(d)) {              // "bf16" mode: the 16x16x32 stand-alone bender (nrnerf_bend_x16.h)
                PackedPass pbx;
                pack_bend_x16(d, pbx, &lay);
                rc = upload_pass(pbx, m->bend_x16);
                if (rc != NRNERF_OK) return rc;
                m->bend_x16.algo_flops_per_sample = m->gen_bend.algo_flops_per_sample;
                m->bend_x16.mfma_flops_per_sample = pbx.mfma_per_block * (2.0 * 16 * 16 * 32) / 16.0;
            }
        }
    }
    rc = upload_pass(gc.pk, m->gen_coarse);
    if (rc != NRNERF_OK) return rc;
    m->gen_coarse_prog = gc.proto;
    m->gen_coarse.algo_flops_per_sample = 2.0 * algo_macs(d2, *d.coarse);
    m->gen_coarse.output_ch = d.coarse->output_ch;
    m->gen_coarse.mfma_flops_per_sample = gen_mfma_flops_per_sample(gc.proto, d.precision == NRNERF_PREC_F32);
    m->coarse.output_ch = d.coarse->output_ch;
    if (d.fine) {
        rc = upload_pass(gf.pk, m->gen_fine);
        if (rc != NRNERF_OK) return rc;
        m->gen_fine_prog = gf.proto;
        m->gen_fine.algo_flops_per_sample = 2.0 * algo_macs(d2, *d.fine);
        m->gen_fine.output_ch = d.fine->output_ch;
        m->gen_fine.mfma_flops_per_sample = gen_mfma_flops_per_sample(gf.proto, d.precision == NRNERF_PREC_F32);
    } else {
        m->gen_fine = m->gen_coarse; m->gen_fine_prog = m->gen_coarse_prog; m->gen_fine_is_coarse = true;
    }
    m->fine.output_ch = m->gen_fine.output_ch;
    m->fine_is_coarse = !d.fine;
    if (gen_trainable(d, *d.coarse) && (!d.fine || gen_trainable(d, *d.fine))) {
        GenProgram bc, bf;
        gen_pack_mlp_bwd(d, *d.coarse, bc, &lay);
        rc = upload_pass(bc.pk, m->gen_coarse_bwd);
        if (rc != NRNERF_OK) return rc;
        m->gen_coarse_bwd_prog = bc.proto;
        auto describe = [&](const nrnerf_mlp_desc& n) {
            nrnerf_model::GenTrainNet t;
            t.W = n.width; t.D = n.depth; t.skip = gen_skip(n) >= 0; t.views = n.use_viewdirs != 0;
            t.dv = t.views ? 3 + 6 * d.multires_views : 0; t.draw_col = gen_draw_col(n);
            t.in_w = n.time_conditioned ? n.pts_linears[0].in_features : 3 + 6 * d.multires; t.lat = t.in_w - (3 + 6 * d.multires);
            return t;
        };
        m->gen_tn[0] = describe(*d.coarse);
        if (d.fine) {
            gen_pack_mlp_bwd(d, *d.fine, bf, &lay);
            rc = upload_pass(bf.pk, m->gen_fine_bwd);
            if (rc != NRNERF_OK) return rc;
            m->gen_fine_bwd_prog = bf.proto;
            m->gen_tn[1] = describe(*d.fine);
        } else {
            m->gen_tn[1] = m->gen_tn[0];
        }
        m->gen_train_ok = true;
    }
    m->exact = d.exact_viewdirs != 0 && m->has_bend && m->views;
    if (m->exact && !(m->bend_train_ok && m->gen_train_ok)) return NRNERF_ERR_UNSUPPORTED;      // (needs bend_div_fwd and the per-sample-direction instantiation)
    // the trunks also for the width-class x16 kernel: rendering passes that run on ready-made points (a bender in front), and the training
    // forward of any such trunk (its points are always handed in)
    if (gx16_eligible(d, *d.coarse) && (!d.fine || gx16_eligible(d, *d.fine))) {
        PackedPass pgc, pgf;
        pack_gx16(d, *d.coarse, pgc, m->gx_meta_coarse, &lay);
        rc = upload_pass(pgc, m->gx_coarse);
        if (rc != NRNERF_OK) return rc;
        m->gx_coarse.algo_flops_per_sample = m->gen_coarse.algo_flops_per_sample;
        m->gx_coarse.mfma_flops_per_sample = pgc.mfma_per_block * (2.0 * 16 * 16 * 32) / 16.0;
        m->gx_coarse.output_ch = d.coarse->output_ch;
        if (d.fine) {
            pack_gx16(d, *d.fine, pgf, m->gx_meta_fine, &lay);
            rc = upload_pass(pgf, m->gx_fine);
            if (rc != NRNERF_OK) return rc;
            m->gx_fine.algo_flops_per_sample = m->gen_fine.algo_flops_per_sample;
            m->gx_fine.mfma_flops_per_sample = pgf.mfma_per_block * (2.0 * 16 * 16 * 32) / 16.0;
            m->gx_fine.output_ch = d.fine->output_ch;
        }
        // ... and their backward-data programs (training: nrnerf_generic_trunk_backward on gx16_bwd_kernel)
        if (m->gen_train_ok && gx16_trainable(d, *d.coarse) && (!d.fine || gx16_trainable(d, *d.fine))) {
            PackedPass pbc, pbf;
            pack_gx16_bwd(d, *d.coarse, pbc, m->gx_meta_coarse_bwd, &lay);
            rc = upload_pass(pbc, m->gx_coarse_bwd);
            if (rc != NRNERF_OK) return rc;
            if (d.fine) {
                pack_gx16_bwd(d, *d.fine, pbf, m->gx_meta_fine_bwd, &lay);
                rc = upload_pass(pbf, m->gx_fine_bwd);
                if (rc != NRNERF_OK) return rc;
            }
        }
    }
    own.m = nullptr;
    *out = m;
    return NRNERF_OK;
}
int update_generic(nrnerf_model* m, const nrnerf_model_desc& d, hipStream_t stream) {
    GenProgram gb, gc, gf;
    int rc = gen_pack_all(d, nullptr, gb, gc, gf);
    if (rc != NRNERF_OK) return rc == NRNERF_ERR_UNSUPPORTED ? NRNERF_ERR_INVALID : rc;
    DeviceGuard guard(m->device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    PackedPass pb;
    if (d.bender && m->gen_compiled_bender >= 0) {
        auto go = [&](auto sh) {
            using SH = decltype(sh);
            if (m->gen_compiled_bender == 0) pack_pass<SH, ArchDefault, true, false, false>(d, *d.coarse, d.precision, pb, nullptr);
            else pack_pass<SH, ArchDeepBend, true, false, false>(d, *d.coarse, d.precision, pb, nullptr);
        };
        if (!(m->gen_compiled_bender == 0 ? bender_matches<ArchDefault>(*d.bender) : bender_matches<ArchDeepBend>(*d.bender))) return NRNERF_ERR_INVALID;
        if (d.precision == NRNERF_PREC_F32) go(ShapeF32{});
        else if (d.precision == NRNERF_PREC_BF16) go(Shape16Fast{});
        else go(Shape16{});
        rc = refresh_pass(pb, m->bend_only, stream);
        if (rc != NRNERF_OK) return rc;
    }
    PackedPass pbx;
    if (d.bender && m->bend_x16.stream) {
        pack_bend_x16(d, pbx);
        rc = refresh_pass(pbx, m->bend_x16, stream);
        if (rc != NRNERF_OK) return rc;
    }
    PackedPass tbf, tbb;
    if (d.bender && m->bend_train_ok) {
        gen_pack_bender_train(d, m->gen_compiled_bender, tbf, tbb, nullptr);
        rc = refresh_pass(tbf, m->bend_train_fwd, stream);
        if (rc == NRNERF_OK) rc = refresh_pass(tbb, m->bend_train_bwd, stream);
        if (rc != NRNERF_OK) return rc;
    }
    if (d.bender) rc = refresh_pass(gb.pk, m->gen_bend, stream);         // (sizes differ for another architecture: NRNERF_ERR_INVALID)
    if (rc == NRNERF_OK) rc = refresh_pass(gc.pk, m->gen_coarse, stream);
    if (rc == NRNERF_OK && d.fine) rc = refresh_pass(gf.pk, m->gen_fine, stream);
    GenProgram bwc, bwf;
    if (rc == NRNERF_OK && m->gen_train_ok) {
        gen_pack_mlp_bwd(d, *d.coarse, bwc, nullptr);
        rc = refresh_pass(bwc.pk, m->gen_coarse_bwd, stream);
        if (rc == NRNERF_OK && d.fine) {
            gen_pack_mlp_bwd(d, *d.fine, bwf, nullptr);
            rc = refresh_pass(bwf.pk, m->gen_fine_bwd, stream);
        }
    }
    PackedPass pgc, pgf;
    if (rc == NRNERF_OK && m->gx_coarse.stream) {
        GxMeta mc, mf;
        pack_gx16(d, *d.coarse, pgc, mc);
        rc = refresh_pass(pgc, m->gx_coarse, stream);
        if (rc == NRNERF_OK && d.fine && m->gx_fine.stream) {
            pack_gx16(d, *d.fine, pgf, mf);
            rc = refresh_pass(pgf, m->gx_fine, stream);
        }
    }
    PackedPass pbc, pbf;
    if (rc == NRNERF_OK && m->gx_coarse_bwd.stream) {
        GxMeta mc, mf;
        pack_gx16_bwd(d, *d.coarse, pbc, mc);
        rc = refresh_pass(pbc, m->gx_coarse_bwd, stream);
        if (rc == NRNERF_OK && d.fine && m->gx_fine_bwd.stream) {
            pack_gx16_bwd(d, *d.fine, pbf, mf);
            rc = refresh_pass(pbf, m->gx_fine_bwd, stream);
        }
    }
    if (hipStreamSynchronize(stream) != hipSuccess && rc == NRNERF_OK) rc = NRNERF_ERR_HIP;      // the packed host images die with this call
    return rc;
}

}  // namespace

extern "C" {

int nrnerf_abi_version(void) { return NRNERF_ABI_VERSION; }

const char* nrnerf_strerror(int status) {
    switch (status) {
        case NRNERF_OK: return "ok";
        case NRNERF_ERR_INVALID: return "invalid argument";
        case NRNERF_ERR_UNSUPPORTED: return "unsupported architecture or flag combination (no kernel compiled for it)";
        case NRNERF_ERR_HIP: return "HIP runtime error (no device, or a launch/copy failed)";
        case NRNERF_ERR_WORKSPACE: return "workspace too small or misaligned";
        case NRNERF_ERR_NOMEM: return "out of memory (device, or host while packing weights)";
        case NRNERF_ERR_INTERNAL: return "internal error (an exception was caught at the C ABI)";
    }
    return "unknown status";
}

int nrnerf_pack_host(const nrnerf_model_desc* desc, int which, nrnerf_packed_info* info, void* stream_out,
                     size_t stream_cap, uint32_t* unit_table_out, float* bias_table_out) try {
    if (!desc || desc->struct_size != sizeof(nrnerf_model_desc) || !desc->coarse) return NRNERF_ERR_INVALID;
    const nrnerf_mlp_desc* m = (which == 1 && desc->fine) ? desc->fine : desc->coarse;
    PackedPass pk;
    int rc;
    if (which == 2 || which == 3) {
        if (!desc->bender || (desc->coarse->use_viewdirs && desc->exact_viewdirs)) return NRNERF_ERR_UNSUPPORTED;
        PackedPass other;
        rc = (which == 2) ? pack_split(*desc, pk, other) : pack_split(*desc, other, pk);
    } else if (which == 4 || which == 5) {       // transposed trunk weights of the backward-data kernel (training)
        if (desc->coarse->time_conditioned || desc->precision == NRNERF_PREC_F16) return NRNERF_ERR_UNSUPPORTED;
        PackedPass fwd;
        nrnerf_model_desc d2 = *desc;
        d2.bender = nullptr;
        const nrnerf_mlp_desc* mm = (which == 5 && desc->fine) ? desc->fine : desc->coarse;
        int arch_id = 0;
        rc = pack_dispatch(d2, *mm, fwd, &arch_id);   // validates the architecture
        if (rc == NRNERF_OK && arch_id != 0 && arch_id != 5) rc = NRNERF_ERR_UNSUPPORTED;      // training kernels: trunks of width 256 / 128
        if (rc == NRNERF_OK) pack_bwd(*desc, *mm, pk);
    } else if (which == 6) {                     // transposed bender / rigidity weights of its backward-data kernel (fp32)
        if (!desc->bender) return NRNERF_ERR_UNSUPPORTED;
        PackedPass fwd;
        int arch_id = 0;
        rc = pack_dispatch(*desc, *desc->coarse, fwd, &arch_id);
        if (rc == NRNERF_OK && arch_id > 1 && arch_id != 5) rc = NRNERF_ERR_UNSUPPORTED;
        if (rc == NRNERF_OK) {
            if (bender_arch(arch_id) == 0) pack_pass_bwd_bender<ArchDefault>(*desc->bender, pk);
            else pack_pass_bwd_bender<ArchDeepBend>(*desc->bender, pk);
        }
    } else if (which == 11 || which == 12) {     // the coarse / fine trunk packed for the width-class kernel (nrnerf_gx16.h)
        const nrnerf_mlp_desc* mm = (which == 12 && desc->fine) ? desc->fine : desc->coarse;
        if (!gx16_eligible(*desc, *mm)) return NRNERF_ERR_UNSUPPORTED;
        GxMeta gm;
        rc = NRNERF_OK;
        pack_gx16(*desc, *mm, pk, gm);
    } else if (which == 10) {                    // the fine network's trunk packed for the 16x16x32 kernel (nrnerf_net_x16.h)
        const nrnerf_mlp_desc* mm = desc->fine ? desc->fine : desc->coarse;
        if (!x16_eligible(*desc, *mm, /*any_16bit=*/true)) return NRNERF_ERR_UNSUPPORTED;
        rc = NRNERF_OK;
        pack_x16(*desc, *mm, pk);
    } else if (which == 13 || which == 14) {    // backward-data programs of the run-time-parameterised kernel (training): 13 = coarse, 14 = fine
        GenProgram gb, gc, gf;
        rc = gen_pack_all(*desc, nullptr, gb, gc, gf);
        const nrnerf_mlp_desc* mm = (which == 14) ? desc->fine : desc->coarse;
        if (rc == NRNERF_OK && !mm) rc = NRNERF_ERR_INVALID;
        if (rc == NRNERF_OK && !gen_trainable(*desc, *mm)) rc = NRNERF_ERR_UNSUPPORTED;
        if (rc == NRNERF_OK) {
            GenProgram g;
            gen_pack_mlp_bwd(*desc, *mm, g, nullptr);
            pk = g.pk;
            // unit table: n_layers, per layer 15 integers (GenLayer incl. save / mask slot and source offsets), then ke, kv, kh, lat, the d raw column
            pk.unit_off.assign(1, (uint32_t)g.proto.n_layers);
            for (int l = 0; l < g.proto.n_layers; ++l) {
                const GenLayer& y = g.proto.layer[l];
                for (int v : {y.w_frag, y.bias_tile, y.nt, y.src0, y.ns0, y.src1, y.ns1, y.dst, y.relu, y.o_col, y.o_rows, y.save_idx, y.mask_idx, y.boff0, y.boff1})
                    pk.unit_off.push_back((uint32_t)v);
            }
            for (int v : {g.proto.ke, g.proto.kv, g.proto.kh, g.proto.lat, gen_draw_col(*mm)}) pk.unit_off.push_back((uint32_t)v);
            pk.nunits = (int)pk.unit_off.size() - 1;
        }
    } else if (which >= 7 && which <= 9) {      // layer programs of the run-time-parameterised kernel: 7 = coarse, 8 = fine, 9 = ray bender
        GenProgram gb, gc, gf;
        rc = gen_pack_all(*desc, nullptr, gb, gc, gf);
        if (rc == NRNERF_OK && ((which == 8 && !desc->fine) || (which == 9 && !desc->bender))) rc = NRNERF_ERR_INVALID;
        if (rc == NRNERF_OK) {
            GenProgram& g = which == 7 ? gc : (which == 8 ? gf : gb);
            pk = g.pk;
            // the "unit table" of a program: n_layers, then per layer the 11 integers of GenLayer; followed by ke, kv, kh, lat
            pk.unit_off.assign(1, (uint32_t)g.proto.n_layers);
            for (int l = 0; l < g.proto.n_layers; ++l) {
                const GenLayer& y = g.proto.layer[l];
                for (int v : {y.w_frag, y.bias_tile, y.nt, y.src0, y.ns0, y.src1, y.ns1, y.dst, y.relu, y.o_col, y.o_rows}) pk.unit_off.push_back((uint32_t)v);
            }
            for (int v : {g.proto.ke, g.proto.kv, g.proto.kh, g.proto.lat}) pk.unit_off.push_back((uint32_t)v);
            pk.nunits = (int)pk.unit_off.size() - 1;
        }
    } else {
        rc = pack_dispatch(*desc, *m, pk);
    }
    if (rc != NRNERF_OK) return rc;
    if (info) {
        info->stream_bytes = pk.stream.size();
        info->n_units = (uint32_t)pk.nunits;
        info->n_bias_tiles = (uint32_t)pk.ntiles;
        info->frag_bytes = (uint32_t)pk.frag_bytes;
        info->slot_bytes = (uint32_t)pk.slot_bytes;
        info->mfma_per_block = (uint32_t)pk.mfma_per_block;
    }
    if (stream_out) {
        if (stream_cap < pk.stream.size()) return NRNERF_ERR_INVALID;
        std::memcpy(stream_out, pk.stream.data(), pk.stream.size());
    }
    if (unit_table_out) std::memcpy(unit_table_out, pk.unit_off.data(), pk.unit_off.size() * 4);
    if (bias_table_out) std::memcpy(bias_table_out, pk.bias.data(), pk.bias.size() * 4);
    return NRNERF_OK;
} NRN_CATCH

int nrnerf_model_create(const nrnerf_model_desc* desc, nrnerf_model** out) try {
    if (!out) return NRNERF_ERR_INVALID;
    *out = nullptr;
    if (!desc || desc->struct_size != sizeof(nrnerf_model_desc) || !desc->coarse) return NRNERF_ERR_INVALID;
    PackedPass pc, pf;
    int arch_id = 0, arch_f = 0;
    const FlatLayout lay = flat_layout(*desc);
    if (desc->fine && (desc->fine->use_viewdirs != 0) != (desc->coarse->use_viewdirs != 0)) return NRNERF_ERR_INVALID;
    int rc = pack_dispatch(*desc, *desc->coarse, pc, &arch_id, false, &lay);
    if (rc == NRNERF_OK && desc->fine) {
        rc = pack_dispatch(*desc, *desc->fine, pf, &arch_f, false, &lay);
        if (rc == NRNERF_OK && arch_f != arch_id) rc = NRNERF_ERR_UNSUPPORTED;      // e.g. --netwidth_fine != --netwidth: generic below
    }
    // NRNERF_MODEL_FORCE_GENERIC: the generic kernel also for the compiled shapes (tests: the two routes against each other)
    const bool force_generic = (desc->flags & NRNERF_MODEL_FORCE_GENERIC) != 0;
    if (rc == NRNERF_OK && force_generic && !(desc->exact_viewdirs && desc->bender && desc->coarse->use_viewdirs)) rc = NRNERF_ERR_UNSUPPORTED;
    if (rc == NRNERF_ERR_UNSUPPORTED) return create_generic(*desc, lay, out);
    if (rc != NRNERF_OK) return rc;
    DeviceGuard guard(desc->device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    // owns the handle until it is handed to the caller: every early return (and an exception caught by NRN_CATCH) frees
    // what was uploaded so far
    struct Owner {
        nrnerf_model* m;
        ~Owner() { if (m) nrnerf_model_destroy(m); }
    } own{new (std::nothrow) nrnerf_model()};
    nrnerf_model* m = own.m;
    if (!m) return NRNERF_ERR_NOMEM;
    m->device = desc->device;
    m->precision = desc->precision;
    m->has_bend = desc->bender != nullptr;
    m->views = desc->coarse->use_viewdirs != 0;
    m->arch_id = arch_id;
    m->exact = desc->exact_viewdirs != 0 && m->has_bend && m->views;     // only meaningful with bender + view-dependent head
    if (m->exact && arch_id > 1) return NRNERF_ERR_UNSUPPORTED;
    m->needs_latents = m->has_bend || desc->coarse->time_conditioned;
    m->latent_size = desc->bender ? desc->bender->latent_size : 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, desc->device) != hipSuccess) return NRNERF_ERR_HIP;
    m->num_cus = prop.multiProcessorCount;
    if (hipMalloc((void**)&m->adam_barrier, 2 * sizeof(unsigned)) != hipSuccess || hipMemset(m->adam_barrier, 0, 2 * sizeof(unsigned)) != hipSuccess) return NRNERF_ERR_NOMEM;
    m->flat_floats = lay.total;
    const double mfma_flop = 2.0 * 32 * 32 * (desc->precision == NRNERF_PREC_F32 ? 2 : 16);
    rc = upload_pass(pc, m->coarse);
    m->coarse.algo_flops_per_sample = 2.0 * algo_macs(*desc, *desc->coarse);
    m->coarse.mfma_flops_per_sample = pc.mfma_per_block * mfma_flop / 32.0;
    m->coarse.output_ch = desc->coarse->output_ch;
    if (rc == NRNERF_OK && desc->fine) {
        rc = upload_pass(pf, m->fine);
        m->fine.algo_flops_per_sample = 2.0 * algo_macs(*desc, *desc->fine);
        m->fine.mfma_flops_per_sample = pf.mfma_per_block * mfma_flop / 32.0;
        m->fine.output_ch = desc->fine->output_ch;
    } else {
        m->fine = m->coarse;
        m->fine_is_coarse = true;
    }
    if (rc == NRNERF_OK && m->has_bend && !m->exact) {       // (exact view directions need the bender's Jacobian: fused only)
        PackedPass pt, pb, pct;
        if (pack_split(*desc, pt, pb, &pct, &lay) == NRNERF_OK) {
            rc = upload_pass(pt, m->fine_trunk);
            if (rc == NRNERF_OK) rc = upload_pass(pb, m->bend_only);
            if (rc == NRNERF_OK) rc = upload_pass(pct, m->coarse_trunk);
            m->coarse_trunk.mfma_flops_per_sample = pct.mfma_per_block * mfma_flop / 32.0;
            m->coarse_trunk.output_ch = m->coarse.output_ch;
            nrnerf_model_desc d2 = *desc;
            d2.bender = nullptr;
            m->fine_trunk.algo_flops_per_sample = 2.0 * algo_macs(d2, desc->fine ? *desc->fine : *desc->coarse);
            m->fine_trunk.mfma_flops_per_sample = pt.mfma_per_block * mfma_flop / 32.0;
            m->fine_trunk.output_ch = m->fine.output_ch;
            m->coarse_trunk.algo_flops_per_sample = 2.0 * algo_macs(d2, *desc->coarse);
            m->bend_only.algo_flops_per_sample = m->fine.algo_flops_per_sample - m->fine_trunk.algo_flops_per_sample;
            m->bend_only.mfma_flops_per_sample = pb.mfma_per_block * mfma_flop / 32.0;
            m->split_ok = (rc == NRNERF_OK);
            const nrnerf_mlp_desc& fm = desc->fine ? *desc->fine : *desc->coarse;
            if (rc == NRNERF_OK && x16_eligible(*desc, fm)) {
                PackedPass px;
                pack_x16(*desc, fm, px, &lay);
                rc = upload_pass(px, m->fine_trunk_x16);
                m->fine_trunk_x16.algo_flops_per_sample = m->fine_trunk.algo_flops_per_sample;
                m->fine_trunk_x16.mfma_flops_per_sample = px.mfma_per_block * (2.0 * 16 * 16 * 32) / 16.0;
                m->fine_trunk_x16.output_ch = m->fine.output_ch;
            }
            if (rc == NRNERF_OK && bend_x16_eligible(*desc)) {
                PackedPass pbx;
                pack_bend_x16(*desc, pbx, &lay);
                rc = upload_pass(pbx, m->bend_x16);
                m->bend_x16.algo_flops_per_sample = m->bend_only.algo_flops_per_sample;
                m->bend_x16.mfma_flops_per_sample = pbx.mfma_per_block * (2.0 * 16 * 16 * 32) / 16.0;
            }
            // the coarse network's trunk in the same packing: the coarse pass of the split path on the 16x16x32 kernel too
            if (rc == NRNERF_OK && desc->fine && x16_eligible(*desc, *desc->coarse)) {
                PackedPass pxc;
                pack_x16(*desc, *desc->coarse, pxc, &lay);
                rc = upload_pass(pxc, m->coarse_trunk_x16);
                m->coarse_trunk_x16.algo_flops_per_sample = m->coarse_trunk.algo_flops_per_sample;
                m->coarse_trunk_x16.mfma_flops_per_sample = pxc.mfma_per_block * (2.0 * 16 * 16 * 32) / 16.0;
                m->coarse_trunk_x16.output_ch = m->coarse.output_ch;
            }
        }
    }
    if (rc == NRNERF_OK) rc = upload_training(*desc, m, nullptr, false, &lay);
    if (rc != NRNERF_OK) return rc;
    own.m = nullptr;
    *out = m;
    return NRNERF_OK;
} NRN_CATCH

int nrnerf_model_update(nrnerf_model* m, const nrnerf_model_desc* desc, void* hip_stream) try {
    if (!m || !desc || desc->struct_size != sizeof(nrnerf_model_desc) || !desc->coarse) return NRNERF_ERR_INVALID;
    if (desc->device != m->device || desc->precision != m->precision || (desc->bender != nullptr) != (m->has_bend != 0) ||
        (desc->coarse->use_viewdirs != 0) != (m->views != 0) || (desc->fine != nullptr) == m->fine_is_coarse ||
        ((desc->exact_viewdirs != 0 && m->has_bend && m->views) != (m->exact != 0)))
        return NRNERF_ERR_INVALID;                     // a different model: create a new handle instead
    if (m->generic) return update_generic(m, *desc, (hipStream_t)hip_stream);
    PackedPass pc, pf;
    int arch_id = 0, arch_f = 0;
    int rc = pack_dispatch(*desc, *desc->coarse, pc, &arch_id);
    if (rc != NRNERF_OK) return rc;
    if (desc->fine) {
        rc = pack_dispatch(*desc, *desc->fine, pf, &arch_f);
        if (rc != NRNERF_OK) return rc;
    }
    if (arch_id != m->arch_id || (desc->fine && arch_f != m->arch_id)) return NRNERF_ERR_INVALID;
    DeviceGuard guard(m->device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    hipStream_t stream = (hipStream_t)hip_stream;
    rc = refresh_pass(pc, m->coarse, stream);
    if (rc == NRNERF_OK && desc->fine) rc = refresh_pass(pf, m->fine, stream);
    PackedPass pt, pb, pct;
    if (rc == NRNERF_OK && m->split_ok) {
        rc = pack_split(*desc, pt, pb, &pct);
        if (rc == NRNERF_OK) rc = refresh_pass(pt, m->fine_trunk, stream);
        if (rc == NRNERF_OK) rc = refresh_pass(pb, m->bend_only, stream);
        if (rc == NRNERF_OK) rc = refresh_pass(pct, m->coarse_trunk, stream);
    }
    PackedPass px;
    if (rc == NRNERF_OK && m->fine_trunk_x16.stream) {
        pack_x16(*desc, desc->fine ? *desc->fine : *desc->coarse, px);
        rc = refresh_pass(px, m->fine_trunk_x16, stream);
    }
    PackedPass pbx;
    if (rc == NRNERF_OK && m->bend_x16.stream) {
        pack_bend_x16(*desc, pbx);
        rc = refresh_pass(pbx, m->bend_x16, stream);
    }
    PackedPass pxc;
    if (rc == NRNERF_OK && m->coarse_trunk_x16.stream) {
        pack_x16(*desc, *desc->coarse, pxc);
        rc = refresh_pass(pxc, m->coarse_trunk_x16, stream);
    }
    // the packed host images die with this call: wait until the copies have consumed them
    if (hipStreamSynchronize(stream) != hipSuccess && rc == NRNERF_OK) rc = NRNERF_ERR_HIP;
    if (rc == NRNERF_OK && m->train_ok) rc = upload_training(*desc, m, stream, /*refresh=*/true);
    return rc;
} NRN_CATCH

int64_t nrnerf_model_flat_size(const nrnerf_model* m) { return m ? m->flat_floats : -1; }

int nrnerf_model_precision(const nrnerf_model* m) { return m ? m->precision : NRNERF_ERR_INVALID; }

namespace {
// which compiled bender shape (0: 5 x 64, 1: 7 x 64) the bender's training kernels run: the handle's architecture, or -- a generic handle --
// the compiled shape its bender happens to have (the reference's hard-coded one next to an odd trunk, rnh:406-407)
int bender_arch_of(const nrnerf_model* m) { return m->generic ? m->gen_compiled_bender : bender_arch(m->arch_id); }
}  // namespace
int nrnerf_model_trains_bender(const nrnerf_model* m) { return m ? (m->bend_train_ok ? 1 : 0) : NRNERF_ERR_INVALID; }
int nrnerf_model_is_generic(const nrnerf_model* m) { return m ? (m->generic ? 1 : 0) : NRNERF_ERR_INVALID; }

}  // extern "C"
namespace {
int device_of(const void* ptr, int& dev);
// every packed image of the handle (weight stream + bias table) as segments of repack launches over `flat_params`; `emit(batch, last)` is
// called per full batch of REPACK_MAX_SEGMENTS and once for the last (possibly empty) one
template <class EMIT>
int repack_batches(nrnerf_model* m, const float* flat_params, EMIT&& emit) {
    PassDev* passes[] = {&m->coarse, m->fine_is_coarse ? nullptr : &m->fine, &m->fine_trunk, &m->coarse_trunk, &m->bend_only, &m->fine_trunk_x16, &m->coarse_trunk_x16, &m->bend_x16,
                         &m->coarse_bwd, &m->fine_bwd, &m->bend_train_fwd, &m->bend_train_bwd, &m->coarse_train, &m->fine_train,
                         &m->gen_bend, &m->gen_coarse, m->gen_fine_is_coarse ? nullptr : &m->gen_fine, &m->gx_coarse, &m->gx_fine, &m->gen_coarse_bwd, &m->gen_fine_bwd, &m->gx_coarse_bwd, &m->gx_fine_bwd};
    for (PassDev* p : passes)
        if (p && p->stream && !p->src) return NRNERF_ERR_UNSUPPORTED;          // (before anything is launched)
    RepackBatchArgs b{};
    b.flat = flat_params;
    auto add = [&](const int32_t* src, const uint8_t* fmt, void* dst, long long n) -> bool {
        if (n <= 0) return true;
        if (b.n_segments == REPACK_MAX_SEGMENTS) {
            if (!emit(b, false)) return false;
            b.n_segments = 0;
        }
        const int k = b.n_segments++;
        if (k == 0) b.block0[0] = 0;
        b.src[k] = src; b.fmt[k] = fmt; b.dst[k] = dst; b.n[k] = n;
        b.block0[k + 1] = b.block0[k] + (unsigned)((n + 255) / 256);
        return true;
    };
    for (PassDev* p : passes) {
        if (!p || !p->stream) continue;
        if (!add(p->src, p->fmt, p->stream, (long long)p->n_elems) || !add(p->bias_src, nullptr, p->bias, (long long)p->bias_floats)) return NRNERF_ERR_HIP;
    }
    return emit(b, true) ? NRNERF_OK : NRNERF_ERR_HIP;
}
}  // namespace
extern "C" {

int nrnerf_model_update_device(nrnerf_model* m, const float* flat_params, int64_t n_floats, void* hip_stream) try {
    if (!m || !flat_params || n_floats != m->flat_floats) return NRNERF_ERR_INVALID;
    DeviceGuard guard(m->device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    hipStream_t stream = (hipStream_t)hip_stream;
    // every image (weight stream + bias table) as one segment of ONE launch
    return repack_batches(m, flat_params, [&](const RepackBatchArgs& b, bool) { return launch_repack_batch(b, stream) == hipSuccess; });
} NRN_CATCH

int nrnerf_adam_step(nrnerf_model* m, const nrnerf_adam_args* a, void* hip_stream) try {
    if (!a || a->struct_size != sizeof(nrnerf_adam_args) || a->n_segments < 0 || a->n_segments > NRNERF_ADAM_MAX_SEGMENTS || !a->step) return NRNERF_ERR_INVALID;
    if (!(a->beta1 >= 0.0f && a->beta1 < 1.0f && a->beta2 >= 0.0f && a->beta2 < 1.0f && a->eps >= 0.0f)) return NRNERF_ERR_INVALID;
    const bool repack = m && a->flat_params;
    if (repack && a->n_floats != m->flat_floats) return NRNERF_ERR_INVALID;
    if (!m && !a->barrier) return NRNERF_ERR_INVALID;
    AdamKernelArgs k{};
    for (int i = 0; i < a->n_segments; ++i) {
        const nrnerf_adam_segment& s = a->segments[i];
        if (s.n > 0 && (!s.param || !s.grad || !s.exp_avg || !s.exp_avg_sq)) return NRNERF_ERR_INVALID;
        k.seg[i] = AdamSegment{s.param, s.grad, s.exp_avg, s.exp_avg_sq, (unsigned long long)s.n};
        k.gran0[i + 1] = k.gran0[i] + (long long)((s.n + 3) / 4);
    }
    for (int i = a->n_segments; i < ADAM_MAX_SEGMENTS; ++i) k.gran0[i + 1] = k.gran0[i];
    k.n_segments = a->n_segments;
    k.lr = a->lr; k.beta1 = a->beta1; k.beta2 = a->beta2; k.eps = a->eps; k.lr_device = a->lr_device; k.step = a->step;
    int dev = 0, num_cus = 0;
    if (m) { dev = m->device; num_cus = m->num_cus; k.barrier = m->adam_barrier; }
    else {
        if (device_of(a->step, dev) != NRNERF_OK) return NRNERF_ERR_INVALID;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return NRNERF_ERR_HIP;
        num_cus = prop.multiProcessorCount; k.barrier = a->barrier;
    }
    if (!k.barrier) return NRNERF_ERR_INVALID;
    DeviceGuard guard(dev);
    if (!guard.ok) return NRNERF_ERR_HIP;
    hipStream_t stream = (hipStream_t)hip_stream;
    if (launch_adam(k, num_cus, stream) != hipSuccess) return NRNERF_ERR_HIP;
    if (!repack) return NRNERF_OK;
    // ... and every packed image from the updated parameters, right behind it on the same stream
    return repack_batches(m, a->flat_params, [&](const RepackBatchArgs& b, bool) { return launch_repack_batch(b, stream) == hipSuccess; });
} NRN_CATCH

namespace {
int cus_of_device(int dev) {
    static int cache[64] = {};
    if (dev >= 0 && dev < 64 && cache[dev] > 0) return cache[dev];
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 64) cache[dev] = prop.multiProcessorCount;
    return prop.multiProcessorCount;
}
// panels of all jobs and the number of sample chunks (= records of partial sums) a call is cut into: ~2 workgroups per CU, >= 1024 samples each
bool tn_plan(const nrnerf_tn_args* a, int num_cus, long long& n_sub, int& kch) {
    if (!a || a->struct_size != sizeof(nrnerf_tn_args) || a->n_jobs < 1 || !a->jobs || a->n_rows < 1 || a->out_floats < 1) return false;
    n_sub = 0;
    for (int j = 0; j < a->n_jobs; ++j) {
        const nrnerf_tn_job& jb = a->jobs[j];
        if (!jb.a || !jb.b || jb.wo < 1 || jb.wi < 1 || jb.lda < jb.wo || jb.ldb < jb.wi || jb.ldo < jb.wi || jb.out_offset < 0) return false;
        if (jb.out_offset + (long long)(jb.wo - 1) * jb.ldo + jb.wi > a->out_floats) return false;
        if (jb.bias_offset >= 0 && jb.bias_offset + jb.wo > a->out_floats) return false;
        n_sub += (long long)((jb.wo + 255) / 256) * ((jb.wi + 255) / 256);
    }
    long long k = (2ll * (num_cus > 0 ? num_cus : 256) + n_sub - 1) / n_sub;          // (one workgroup per CU at a time: two rounds even out the jobs' sizes)
    const long long by_rows = a->n_rows / 1024 > 1 ? a->n_rows / 1024 : 1;
    if (k > by_rows) k = by_rows;
    if (k > 64) k = 64;
    if (k < 1) k = 1;
    kch = (int)k;
    return true;
}
}  // namespace

size_t nrnerf_tn_workspace_bytes(const nrnerf_tn_args* a) {
    long long n_sub; int kch;
    int dev = 0;
    if (!a || !a->out || device_of(a->out, dev) != NRNERF_OK) { (void)hipGetDevice(&dev); }
    if (!tn_plan(a, cus_of_device(dev), n_sub, kch)) return 0;
    return (size_t)kch * (size_t)a->out_floats * sizeof(float);
}

int nrnerf_tn_products(const nrnerf_tn_args* a, void* hip_stream) try {
    if (!a || a->struct_size != sizeof(nrnerf_tn_args) || !a->out || !a->workspace) return NRNERF_ERR_INVALID;
    int dev = 0;
    if (device_of(a->out, dev) != NRNERF_OK) return NRNERF_ERR_INVALID;
    long long n_sub; int kch;
    if (!tn_plan(a, cus_of_device(dev), n_sub, kch)) return NRNERF_ERR_INVALID;
    if (a->workspace_bytes < (size_t)kch * (size_t)a->out_floats * sizeof(float) || ((uintptr_t)a->workspace & 15)) return NRNERF_ERR_WORKSPACE;
    DeviceGuard guard(dev);
    if (!guard.ok) return NRNERF_ERR_HIP;
    hipStream_t stream = (hipStream_t)hip_stream;
    float* parts = (float*)a->workspace;
    if (launch_tn_clear(parts, a->out_floats, kch, stream) != hipSuccess) return NRNERF_ERR_HIP;
    bool misaligned = false;
    TnKernelArgs k{};
    k.kch = kch; k.n_rows = a->n_rows; k.total = a->out_floats; k.partials = parts;
    auto flush = [&]() -> bool {
        if (k.n_sub == 0) return true;
        const hipError_t rc = launch_tn_products(k, a->is_bf16 == 0, stream);
        k.n_sub = 0;
        if (rc == hipErrorInvalidValue) misaligned = true;
        return rc == hipSuccess;
    };
    for (int j = 0; j < a->n_jobs; ++j) {
        const nrnerf_tn_job& jb = a->jobs[j];
        for (int o0 = 0; o0 < jb.wo; o0 += 256)
            for (int k0 = 0; k0 < jb.wi; k0 += 256) {
                if (k.n_sub == TN_MAX_SUBJOBS && !flush()) return misaligned ? NRNERF_ERR_INVALID : NRNERF_ERR_HIP;
                k.sub[k.n_sub++] = TnSubJob{jb.a, jb.b, jb.lda, jb.ldb, jb.wo, jb.wi, o0, k0, jb.ldo, (long long)jb.out_offset, (long long)jb.bias_offset};
            }
    }
    if (!flush()) return misaligned ? NRNERF_ERR_INVALID : NRNERF_ERR_HIP;
    return launch_tn_reduce(parts, a->out_floats, kch, a->out, stream) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

namespace {
int encoding_call(const nrnerf_encoding_args* a, bool backward, void* hip_stream) {
    if (!a || a->struct_size != sizeof(nrnerf_encoding_args) || a->n_rows < 0 || !a->src) return NRNERF_ERR_INVALID;
    if (a->n_rows == 0) return NRNERF_OK;
    int dev = 0;
    if (device_of(a->src, dev) != NRNERF_OK) return NRNERF_ERR_INVALID;
    DeviceGuard guard(dev);
    if (!guard.ok) return NRNERF_ERR_HIP;
    EncodingArgs e{a->src, a->src_stride, (long long)a->n_rows, a->n_freqs, a->enc, a->enc_cols, a->enc_is_bf16, a->codes, a->n_lat, a->rows_per_code,
                   a->d_enc0, a->d_enc1, a->d_enc_stride, a->d_src, a->d_src_stride};
    const hipError_t rc = launch_encoding_rows(e, backward, (hipStream_t)hip_stream);
    return rc == hipSuccess ? NRNERF_OK : (rc == hipErrorInvalidValue ? NRNERF_ERR_INVALID : NRNERF_ERR_HIP);
}
}  // namespace
int nrnerf_encoding_forward(const nrnerf_encoding_args* a, void* hip_stream) try { return encoding_call(a, false, hip_stream); } NRN_CATCH
int nrnerf_encoding_backward(const nrnerf_encoding_args* a, void* hip_stream) try { return encoding_call(a, true, hip_stream); } NRN_CATCH

void nrnerf_model_destroy(nrnerf_model* m) {
    if (!m) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(m->device);
    for (auto& e : m->prof_events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (!m->fine_is_coarse) free_pass(m->fine);
    free_pass(m->coarse);
    free_pass(m->fine_trunk);
    free_pass(m->fine_trunk_x16);
    free_pass(m->coarse_trunk_x16);
    free_pass(m->bend_x16);
    free_pass(m->gx_coarse);
    free_pass(m->gx_fine);
    free_pass(m->gx_coarse_bwd);
    free_pass(m->gx_fine_bwd);
    free_pass(m->gen_coarse_bwd);
    free_pass(m->gen_fine_bwd);
    free_pass(m->coarse_trunk);
    free_pass(m->bend_only);
    free_pass(m->coarse_bwd);
    free_pass(m->fine_bwd);
    free_pass(m->bend_train_fwd);
    free_pass(m->bend_train_bwd);
    free_pass(m->coarse_train);
    free_pass(m->fine_train);
    free_pass(m->gen_bend);
    if (!m->gen_fine_is_coarse) free_pass(m->gen_fine);
    free_pass(m->gen_coarse);
    if (m->adam_barrier) (void)hipFree(m->adam_barrier);
    (void)hipSetDevice(prev);
    delete m;
}

// work counters of the 16x16x32 stand-alone bender (BendArgs::work_counter): two launches per call, up to 512 counters each, 64 bytes apart
constexpr int BEND_COUNTERS_PER_LAUNCH = 512;
// ... and behind them one counter per 16x16x32 trunk launch (NetArgs::work_counter), 64 bytes apart
constexpr size_t BEND_COUNTER_BYTES = (size_t)2 * BEND_COUNTERS_PER_LAUNCH * 64 + 256;
size_t nrnerf_workspace_bytes(const nrnerf_model* model, int32_t n_rays, int32_t n_samples, int32_t n_importance) {
    if (n_rays <= 0 || n_samples <= 0 || n_importance < 0) return 0;
    const size_t N = (size_t)n_rays, S = (size_t)n_samples, SF = S + (size_t)n_importance;
    size_t b = align_up(N * S * 4 * sizeof(float), 256);
    if (n_importance > 0) b += align_up(N * SF * sizeof(float), 256) + align_up(N * SF * 4 * sizeof(float), 256);
    b += align_up(N * SF * 4 * sizeof(float), 256);      // bent point + rigidity of the final pass (surface reduction, split-bender path)
    b += align_up(N * S * sizeof(float), 256);           // jittered coarse depths (perturb > 0)
    if (n_importance > 0) {                              // split-bender path: coarse bent points, depths + rows of the new samples
        b += align_up(N * S * 4 * sizeof(float), 256);
        b += align_up(N * (SF - S) * sizeof(float), 256) + align_up(N * (SF - S), 256);
    }
    if (model && model->generic && model->exact) b += align_up(N * SF * 3 * sizeof(float), 256);      // per-sample Jacobian directions of a pass
    b += BEND_COUNTER_BYTES;                             // work counters of the stand-alone bender launches (BendArgs::work_counter)
    return b;
}

int nrnerf_render(const nrnerf_model* m, const nrnerf_render_args* a, void* hip_stream) try {
    if (!m || !a || a->struct_size != sizeof(nrnerf_render_args)) return NRNERF_ERR_INVALID;
    if (a->n_rays < 0 || a->n_samples < 2 || a->n_importance < 0) return NRNERF_ERR_INVALID;
    if (a->n_samples > NRNERF_MAX_SAMPLES || a->n_samples + a->n_importance > NRNERF_MAX_SAMPLES) return NRNERF_ERR_UNSUPPORTED;
    if (a->n_rays == 0) return NRNERF_OK;
    if (!a->rays || a->ray_stride < 8 || !a->rgb_map || !a->disp_map || !a->acc_map) return NRNERF_ERR_INVALID;
    if (m->needs_latents && (!a->latents || a->latent_stride < 0)) return NRNERF_ERR_INVALID;
    if (m->views && (!m->has_bend || m->exact) && a->ray_stride < 11) return NRNERF_ERR_INVALID;   // needs the unit view directions
    const size_t need = nrnerf_workspace_bytes(m, a->n_rays, a->n_samples, a->n_importance);
    if (!a->workspace || a->workspace_bytes < need || ((uintptr_t)a->workspace & 255)) return NRNERF_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)hip_stream;
    const int N = a->n_rays, S = a->n_samples, I = a->n_importance, SF = S + I;
    // launches go to the model's device whatever the calling thread's current device is (restored on every exit path)
    DeviceGuard guard(m->device);
    if (!guard.ok) return NRNERF_ERR_HIP;

    char* ws = (char*)a->workspace;
    float* raw_c = (float*)ws;
    ws += align_up((size_t)N * S * 4 * sizeof(float), 256);
    float* z_fine = nullptr; float* raw_f = nullptr;
    if (I > 0) {
        z_fine = (float*)ws; ws += align_up((size_t)N * SF * sizeof(float), 256);
        raw_f = (float*)ws; ws += align_up((size_t)N * SF * 4 * sizeof(float), 256);
    }
    const bool surface = a->surface_pts || a->surface_rigidity || a->median_index;
    float* bent4 = (float*)ws;
    float* const bent4_ws = bent4;       // (the slot itself: `bent4` is nulled below when the compiled path does not need it)
    ws += align_up((size_t)N * SF * 4 * sizeof(float), 256);
    float* z_coarse = (float*)ws;
    ws += align_up((size_t)N * S * sizeof(float), 256);
    // Split-bender path: bender, no view-dependent head, a fine pass, no per-sample detail outputs (those are written by
    // the fused kernels).  NRNERF_RENDER_FUSED_FINE_BENDER keeps the fused fine pass (A/B and bit-identity tests).
    auto any_detail = [](const nrnerf_sample_outputs& o) {
        return o.visibility_weights || o.opacity_alpha || o.initial_input_pts || o.unmasked_offsets || o.masked_offsets ||
               o.input_pts || o.rigidity_mask;
    };
    const bool force_fused = (a->flags & NRNERF_RENDER_FUSED_FINE_BENDER) != 0;
    // (the stand-alone bender kernel indexes its 32-sample blocks with 32 bits: beyond 2^31 blocks stay on the fused kernels)
    // (8-bit ranks among the merged depths: beyond 256 samples per ray the fused-bender fine pass renders)
    const bool split = m->split_ok && I > 0 && !a->detailed_output && !any_detail(a->coarse) && !any_detail(a->fine) && !force_fused &&
                       (long long)N * ((imax(S, I) + 31) / 32) < (1ll << 31) && SF <= 256;
    float* bent_c = nullptr; float* z_new = nullptr; uint8_t* rank_new = nullptr;
    if (I > 0) {
        bent_c = (float*)ws; ws += align_up((size_t)N * S * 4 * sizeof(float), 256);
        z_new = (float*)ws; ws += align_up((size_t)N * I * sizeof(float), 256);
        rank_new = (uint8_t*)ws; ws += align_up((size_t)N * I, 256);
    }
    float* const jdirs = (m->generic && m->exact) ? (float*)ws : nullptr;      // [N, S + I | S, 3]: exact Jacobian directions of the pass in flight
    if (m->generic && m->exact) ws += align_up((size_t)N * SF * 3 * sizeof(float), 256);
    // one counter per stand-alone bender launch of the call (64 bytes apart), zeroed by ONE memset node ahead of the first launch
    unsigned* const bend_counters = (unsigned*)ws;
    bool counters_zeroed = false;
    auto bend_counter = [&](int which) -> unsigned* {      // (per launch: one counter per pair of co-resident workgroups, 64 bytes apart)
        if (m->num_cus > BEND_COUNTERS_PER_LAUNCH) return nullptr;                                  // (nullptr: the kernel's fixed shares)
        if (!counters_zeroed) {
            if (hipMemsetAsync(bend_counters, 0, BEND_COUNTER_BYTES, stream) != hipSuccess) return nullptr;
            counters_zeroed = true;
        }
        return bend_counters + (size_t)16 * BEND_COUNTERS_PER_LAUNCH * which;
    };
    auto trunk_counter = [&](int which) -> unsigned* {     // (0: coarse pass, 1: fine pass; nullptr: fixed shares)
        if (a->flags & NRNERF_RENDER_FIXED_SHARES) return nullptr;
        if (!counters_zeroed) {
            if (hipMemsetAsync(bend_counters, 0, BEND_COUNTER_BYTES, stream) != hipSuccess) return nullptr;
            counters_zeroed = true;
        }
        return bend_counters + (size_t)16 * BEND_COUNTERS_PER_LAUNCH * 2 + 16 * which;
    };
    if (!surface && !split) bent4 = nullptr;
    if ((a->u_fine || a->noise_fine) && I == 0) return NRNERF_ERR_INVALID;

    Knobs kn{};
    kn.has_cutoff = a->has_rigidity_cutoff; kn.cutoff = a->rigidity_cutoff;
    kn.has_scaling = a->has_test_time_scaling; kn.scaling = a->test_time_scaling;
    kn.has_removal = a->has_removal_threshold; kn.removal = a->removal_threshold;
    kn.detailed = a->detailed_output;

    bool prof;
    { std::lock_guard<std::mutex> g(m->prof_mu); prof = m->prof_on; }
    auto timed = [&](int kernel, const char* name, double flops, double mfma, auto&& launch) -> hipError_t {
        if (!prof) return launch();
        nrnerf_model::Ev ev{kernel, nullptr, nullptr, flops, mfma, name};
        // device-scope events (no system-scope cache write-back with every record).  Measured A/B against default events on
        // one box: no difference (36.00 / 36.04 vs 36.14 / 35.99 ms per step) -- the kernels' event times add up to the step
        // time within 0.05 ms either way, i.e. there are no launch gaps to recover between the five kernels of a render
        if (hipEventCreateWithFlags(&ev.a, hipEventDisableSystemFence) != hipSuccess ||
            hipEventCreateWithFlags(&ev.b, hipEventDisableSystemFence) != hipSuccess) return hipErrorUnknown;
        (void)hipEventRecord(ev.a, stream);
        hipError_t e = launch();
        (void)hipEventRecord(ev.b, stream);
        std::lock_guard<std::mutex> g(m->prof_mu);
        m->prof_events.push_back(ev);
        return e;
    };

    auto sample_out = [](const nrnerf_sample_outputs& o) {
        return SampleOut{o.visibility_weights, o.opacity_alpha, o.initial_input_pts, o.unmasked_offsets,
                         o.masked_offsets, o.input_pts, o.rigidity_mask};
    };

    // the stand-alone bender of the split path: the 16x16x32 kernel (nrnerf_bend_x16.h; "bf16" mode's single-product bender) unless the call
    // asks for the 32x32x16 one (NRNERF_RENDER_BENDER_32X32: the bit-identity tests against the fused-bender kernels)
    const bool bend_x16 = m->bend_x16.stream && !(a->flags & (NRNERF_RENDER_BENDER_32X32 | NRNERF_RENDER_NO_X16));
    if (bend_x16 && !(a->flags & NRNERF_RENDER_FIXED_SHARES)) (void)bend_counter(0);       // (the memset node ahead of every timed launch)
    auto run_bender = [&](BendArgs& b, int slot, int n_samples) -> hipError_t {
        if (bend_x16) {
            b.wstream = m->bend_x16.stream; b.bias = m->bend_x16.bias;
            b.work_counter = (a->flags & NRNERF_RENDER_FIXED_SHARES) ? nullptr : bend_counter(slot == 5 ? 0 : 1);
            return timed(slot, "bend_kernel_x16", (double)N * n_samples * m->bend_x16.algo_flops_per_sample, (double)N * n_samples * m->bend_x16.mfma_flops_per_sample,
                         [&] { return launch_bend_x16(bender_arch(m->arch_id), b, m->num_cus, stream); });
        }
        b.wstream = m->bend_only.stream; b.bias = m->bend_only.bias;
        return timed(slot, "bend_kernel", (double)N * n_samples * m->bend_only.algo_flops_per_sample, (double)N * n_samples * m->bend_only.mfma_flops_per_sample,
                     [&] { return launch_bend(m->precision, bender_arch(m->arch_id), b, m->num_cus, stream); });
    };

    // ---- stratified jitter of the coarse depths (perturb > 0): both coarse kernels then read explicit depths
    const float* zc = nullptr;
    if (a->u_coarse) {
        JitterArgs ja{a->rays, a->ray_stride, a->u_coarse, N, S, a->lindisp, z_coarse};
        if (launch_zjitter(ja, stream) != hipSuccess) return NRNERF_ERR_HIP;
        zc = z_coarse;
    }

    if (m->generic) {
        // ---- architecture outside the compiled set (nrnerf_generic.h): per pass the bender over all samples of the pass (no
        //      split-bender trick: the generic path trades speed for generality), the canonical network on the bent points,
        //      the composite kernel.  Kernel slots of the profile: 5 / 4 = bender of the coarse / fine pass.
        const bool bend = m->has_bend != 0, views = m->views != 0;
        auto bender_pass = [&](const float* zv, int nS, float* out4, const nrnerf_sample_outputs& so, int slot) -> hipError_t {
            if (m->gen_compiled_bender >= 0 && !any_detail(so) && (long long)N * ((nS + 31) / 32) < (1ll << 31)) {
                // the reference's own bender shape: the compiled stand-alone kernel (weights resident in LDS), all nS samples of a ray
                BendArgs b{};
                b.rays = a->rays; b.ray_stride = a->ray_stride; b.latents = a->latents; b.lat_stride = a->latent_stride;
                b.z = zv; b.lindisp = a->lindisp; b.rank = nullptr; b.n_rays = N; b.n_per_ray = nS; b.out_stride = nS;
                b.bent4 = out4; b.knobs = kn;
                if (bend_x16) {
                    b.wstream = m->bend_x16.stream; b.bias = m->bend_x16.bias;
                    // (fixed shares here: with all nS samples of a ray in one launch the counters measured SLOWER -- width 192: 1.72 -> 1.82 ms
                    //  per fine pass, profiles/r06_dynamic_shares_ab.txt -- where the split path's launches gain 8 %)
                    b.work_counter = nullptr;
                    return timed(slot, "bend_kernel_x16", (double)N * nS * m->bend_x16.algo_flops_per_sample, (double)N * nS * m->bend_x16.mfma_flops_per_sample,
                                 [&] { return launch_bend_x16(m->gen_compiled_bender, b, m->num_cus, stream); });
                }
                b.wstream = m->bend_only.stream; b.bias = m->bend_only.bias;
                return timed(slot, "bend_kernel", (double)N * nS * m->bend_only.algo_flops_per_sample, 0,
                             [&] { return launch_bend(m->precision, m->gen_compiled_bender, b, m->num_cus, stream); });
            }
            GenArgs g = m->gen_bend_prog;
            g.rays = a->rays; g.ray_stride = a->ray_stride; g.latents = a->latents; g.lat_stride = a->latent_stride;
            g.z = zv; g.lindisp = a->lindisp; g.n_rays = N; g.S = nS;
            g.wstream = m->gen_bend.stream; g.bias = m->gen_bend.bias;
            g.bent4 = out4; g.ex = sample_out(so); g.knobs = kn;
            return timed(slot, "gen_kernel (bender program)", (double)N * nS * m->gen_bend.algo_flops_per_sample, 0, [&] { return launch_generic(m->precision, g, m->num_cus, stream); });
        };
        // `fuse` (a FINAL pass): its compositing arguments -- taken as the kernel's epilogue when the pass runs on the width-class kernel
        // (then *fused = true and the caller skips the composite launch); NRNERF_RENDER_UNFUSED_COMPOSITE keeps the launch (bit-identity tests)
        auto network_pass = [&](const GenArgs& prog, const PassDev& pd, const float* zv, int nS, const float* pts, float* raw4, float* raw_user,
                                float* bent_out, const nrnerf_sample_outputs& so, int slot, const CompositeArgs* fuse = nullptr,
                                bool* fused = nullptr) -> hipError_t {
            // the trunk on the width-class 16x16x32 kernel (nrnerf_gx16.h) when the pass runs on ready-made points and wants no detail outputs
            const PassDev& gx = (&pd == &m->gen_coarse) ? m->gx_coarse : m->gx_fine;
            const GxMeta& gm = (&pd == &m->gen_coarse) ? m->gx_meta_coarse : m->gx_meta_fine;
            const PassDev& gxu = (m->gen_fine_is_coarse && &pd == &m->gen_fine) ? m->gx_coarse : gx;      // (one network for both passes)
            const GxMeta& gmu = (m->gen_fine_is_coarse && &pd == &m->gen_fine) ? m->gx_meta_coarse : gm;
            if (m->exact) {
                // exact Jacobian view directions (rnh:291-294, 358-385): J d of every sample from the bender's divergence kernel in ray mode
                // (value + tangent chain, fp32), then the network program on the per-sample directions (its training instantiation takes
                // them; nothing is saved)
                BendDivArgs t{};
                t.latents = a->latents; t.lat_stride = a->latent_stride; t.m = (long long)N * nS;
                t.wstream = m->bend_train_fwd.stream; t.bias = m->bend_train_fwd.bias;
                t.knobs.has_cutoff = kn.has_cutoff; t.knobs.cutoff = kn.cutoff; t.knobs.has_scaling = kn.has_scaling; t.knobs.scaling = kn.scaling;
                t.rays = a->rays; t.ray_stride = a->ray_stride; t.zr = zv; t.S = nS; t.lindisp = a->lindisp; t.dirs_out = jdirs;
                const bool b16 = m->precision != NRNERF_PREC_F32;
                hipError_t je = (m->gen_compiled_bender == 0) ? launch_bend_div_fwd_a0(t, m->num_cus, stream, b16) : launch_bend_div_fwd_a1(t, m->num_cus, stream, b16);
                if (je != hipSuccess) return je;
                GenArgs g = prog;
                g.mode = 1;
                g.rays = pts; g.ray_stride = 0; g.latents = nullptr; g.lat_stride = 0;
                g.z = nullptr; g.lindisp = 0; g.n_rays = N; g.S = nS;
                g.pts4 = pts; g.dirs_from_pts = 0; g.dirs = jdirs;
                g.wstream = pd.stream; g.bias = pd.bias;
                g.raw4 = raw4; g.raw_out = raw_user; g.raw_ch = pd.output_ch; g.bent4 = const_cast<float*>(pts);        // (read: the removal knob, rnh:308-311)
                g.save = nullptr; g.mask = nullptr; g.save_stride = 0; g.save_w = 0;
                g.knobs = kn;
                return timed(slot, "gen_kernel (exact Jacobian directions)", (double)N * nS * pd.algo_flops_per_sample, (double)N * nS * pd.mfma_flops_per_sample,
                             [&] { return launch_generic_train(m->precision, g, m->num_cus, stream); });
            }
            if (pts && gxu.stream && !(a->flags & NRNERF_RENDER_NO_X16) && !any_detail(so) && !kn.detailed &&
                (long long)N * nS < (1ll << 32)) {      // (the kernel's 32-bit sample rows; beyond: the run-time-parameterised kernel below)
                GxArgs x{};
                x.pts4 = pts; x.raw4 = raw4; x.raw_out = raw_user; x.raw_ch = pd.output_ch;
                x.n_rays = N; x.S = nS; x.wstream = gxu.stream; x.bias = gxu.bias;
                x.depth = gmu.depth; x.skip = gmu.skip; x.L = gmu.L; x.n_bias_tiles = gmu.n_bias_tiles; x.LV = gmu.LV;
                if (fuse && !(a->flags & NRNERF_RENDER_UNFUSED_COMPOSITE) && nS <= 256 &&
                    (long long)N >= gx16_rays_per_group(gmu.wc, nS) * m->num_cus) {
                    x.fuse_on = 1; x.fuse = *fuse; x.fuse.raw4 = nullptr; x.raw4 = nullptr;
                    *fused = true;
                }
                return timed(slot, (x.fuse_on ? "gx16_kernel + fused compositing" : "gx16_kernel"), (double)N * nS * gxu.algo_flops_per_sample, (double)N * nS * gxu.mfma_flops_per_sample,
                             [&] { return launch_gx16(m->precision, gmu.wc, gmu.views != 0, x, m->num_cus, stream); });
            }
            GenArgs g = prog;
            g.rays = a->rays; g.ray_stride = a->ray_stride; g.latents = a->latents; g.lat_stride = a->latent_stride;
            g.z = zv; g.lindisp = a->lindisp; g.n_rays = N; g.S = nS;
            g.pts4 = pts; g.dirs_from_pts = (pts && views) ? 1 : 0;
            g.wstream = pd.stream; g.bias = pd.bias;
            g.raw4 = raw4; g.raw_out = raw_user; g.raw_ch = pd.output_ch;
            g.bent4 = pts ? const_cast<float*>(pts) : bent_out;          // with a bender: read (removal knob); without: written (points of the pass)
            g.knobs = kn;
            if (!pts) g.ex = sample_out(so);                             // without a bender the network kernel reports the points
            return timed(slot, "gen_kernel", (double)N * nS * pd.algo_flops_per_sample, (double)N * nS * pd.mfma_flops_per_sample,
                         [&] { return launch_generic(m->precision, g, m->num_cus, stream); });
        };
        float* const bent_final = bent4_ws;                         // points of the final pass [N, S + I | S, 4]
        float* const bentA = (I > 0) ? bent_c : bent4_ws;           // points of the coarse pass: its own array when a fine pass follows
        hipError_t ge = hipSuccess;
        if (bend) ge = bender_pass(zc, S, bentA, a->coarse, 5);
        if (ge != hipSuccess) return NRNERF_ERR_HIP;
        CompositeArgs gc{};
        gc.rays = a->rays; gc.ray_stride = a->ray_stride;
        gc.raw4 = raw_c; gc.z = zc; gc.n_rays = N; gc.S = S; gc.n_importance = I;
        gc.lindisp = a->lindisp; gc.white_bkgd = a->white_bkgd; gc.noise = a->noise_coarse; gc.u = a->u_fine;
        gc.vis = a->coarse.visibility_weights; gc.alpha = a->coarse.opacity_alpha;
        if (I > 0) {
            gc.rgb = a->rgb0 ? a->rgb0 : raw_f; gc.disp = a->disp0 ? a->disp0 : raw_f + (size_t)N * 3; gc.acc = a->acc0 ? a->acc0 : raw_f + (size_t)N * 4;
            gc.z_std = a->z_std; gc.z_out = z_fine;
        } else {
            gc.rgb = a->rgb_map; gc.disp = a->disp_map; gc.acc = a->acc_map; gc.z_user = a->z_vals;
            if (surface) { gc.bent4 = bent_final; gc.surf_pts = a->surface_pts; gc.surf_rig = a->surface_rigidity; gc.med_idx = a->median_index; }
        }
        bool fused_c = false, fused_f = false;
        ge = network_pass(m->gen_coarse_prog, m->gen_coarse, zc, S, bend ? bentA : nullptr, raw_c, (I == 0) ? a->raw : nullptr,
                          (I == 0 && surface) ? bent_final : nullptr, a->coarse, 0, (I == 0) ? &gc : nullptr, &fused_c);
        if (ge != hipSuccess) return NRNERF_ERR_HIP;
        if (!fused_c && timed(1, "composite_kernel", 0, 0, [&] { return launch_composite(gc, stream); }) != hipSuccess) return NRNERF_ERR_HIP;
        if (I == 0) return NRNERF_OK;
        if (bend) ge = bender_pass(z_fine, SF, bent_final, a->fine, 4);
        if (ge != hipSuccess) return NRNERF_ERR_HIP;
        CompositeArgs gf{};
        gf.rays = a->rays; gf.ray_stride = a->ray_stride;
        gf.raw4 = raw_f; gf.z = z_fine; gf.n_rays = N; gf.S = SF; gf.n_importance = 0;
        gf.white_bkgd = a->white_bkgd; gf.noise = a->noise_fine;
        gf.rgb = a->rgb_map; gf.disp = a->disp_map; gf.acc = a->acc_map; gf.z_user = a->z_vals;
        gf.vis = a->fine.visibility_weights; gf.alpha = a->fine.opacity_alpha;
        if (surface) { gf.bent4 = bent_final; gf.surf_pts = a->surface_pts; gf.surf_rig = a->surface_rigidity; gf.med_idx = a->median_index; }
        ge = network_pass(m->gen_fine_prog, m->gen_fine, z_fine, SF, bend ? bent_final : nullptr, raw_f, a->raw, surface ? bent_final : nullptr, a->fine, 2,
                          &gf, &fused_f);
        if (ge != hipSuccess) return NRNERF_ERR_HIP;
        if (!fused_f && timed(3, "composite_kernel", 0, 0, [&] { return launch_composite(gf, stream); }) != hipSuccess) return NRNERF_ERR_HIP;
        return NRNERF_OK;
    }

    // ---- K0: coarse network
    NetArgs na{};
    na.rays = a->rays; na.ray_stride = a->ray_stride;
    na.latents = a->latents; na.lat_stride = a->latent_stride;
    na.z = zc; na.lindisp = a->lindisp; na.n_rays = N; na.S = S;
    na.wstream = m->coarse.stream; na.bias = m->coarse.bias;
    na.raw4 = raw_c;
    na.raw_out = (I == 0) ? a->raw : nullptr;
    na.raw_ch = m->coarse.output_ch;
    na.bent4 = (I == 0) ? bent4 : (split ? bent_c : nullptr);
    na.ex = sample_out(a->coarse);
    na.knobs = kn;
    // The coarse pass stays fused by default: measured on MI355X (round 2), bender kernel 1.56 ms + trunk-only coarse
    // kernel 8.72 ms = 10.28 ms against 10.22 ms fused -- nothing is saved there, unlike in the fine pass where a third of
    // the samples skips the bender.  NRNERF_RENDER_SPLIT_COARSE splits it as well (A/B).
    const bool split_coarse_on = (a->flags & NRNERF_RENDER_SPLIT_COARSE) != 0;
    // The 16x16x32 trunk-only kernel (nrnerf_net_x16.h) for the passes of the split path when the call wants no detail outputs.
    // Per call (nrnerf_render_args::flags; the parity tests run the kernels side by side in one process): NRNERF_RENDER_NO_X16 = the
    // 32x32x16 kernels of nrnerf_net_mb.h, NRNERF_RENDER_X16_FINE_ONLY = the fine pass only, default = the coarse pass too --
    // stand-alone bender over the S coarse samples + 16x16x32 trunk instead of the fused-bender 32x32x16 kernel.
    const int x16_mode = (a->flags & NRNERF_RENDER_NO_X16) ? 0 : ((a->flags & NRNERF_RENDER_X16_FINE_ONLY) ? 1 : 2);
    const bool x16_coarse = split && x16_mode >= 2 && m->coarse_trunk_x16.stream && !a->detailed_output && !kn.detailed;
    const bool split_coarse = split && (split_coarse_on || x16_coarse);
    // Compositing fused into the FINAL pass' network kernel (north_star: "compositing fused into the ray loop"; the
    // reference calls raw2outputs inline, train.py:943-950): the kernel variants without a fused bender -- the trunk-only fine
    // pass of the split-bender path, every pass of a model without bender -- let each wave own whole rays, keep their raw
    // outputs in LDS and composite them itself (nrnerf_composite_ray.h: the composite kernel's own code, so the same bits).
    // The pass' raw array (16 B per sample written and read back) never exists and one launch goes.  The coarse pass of a
    // hierarchical render keeps its composite kernel: sample_pdf and the merge follow it there.
    // NRNERF_RENDER_UNFUSED_COMPOSITE keeps the separate launch (A/B and bit-identity tests).
    const bool unfused_composite = (a->flags & NRNERF_RENDER_UNFUSED_COMPOSITE) != 0;
    auto final_composite = [&](int pass_S, const float* zv, const float* noise, const nrnerf_sample_outputs& so, const float* raw4) {
        CompositeArgs c{};
        c.rays = a->rays; c.ray_stride = a->ray_stride;
        c.raw4 = raw4; c.z = zv; c.n_rays = N; c.S = pass_S; c.n_importance = 0;
        c.lindisp = a->lindisp; c.white_bkgd = a->white_bkgd; c.noise = noise;
        c.rgb = a->rgb_map; c.disp = a->disp_map; c.acc = a->acc_map;
        c.z_std = nullptr; c.z_out = nullptr; c.z_user = a->z_vals;
        c.vis = so.visibility_weights; c.alpha = so.opacity_alpha;
        if (surface) { c.bent4 = bent4; c.surf_pts = a->surface_pts; c.surf_rig = a->surface_rigidity; c.med_idx = a->median_index; }
        return c;
    };
    // (small batches keep the separate launch: a fused pass hands out whole GROUPS of rays -- 4 rays = 24 blocks at 192 samples --
    //  where the plain mapping hands out 8-block tiles, so below one group per CU the plain mapping fills more of the chip)
    auto enough_rays_to_fuse = [&](int pass_S) {
        const int bpr = (pass_S + 31) / 32;
        // rays per group = waves per workgroup x rays per wave: fp32 kernels 4 x 1; 16-bit two-blocks-per-wave kernels 4 x (1 or 2);
        // 16-bit one-block-per-wave kernels (architecture 5) 8 x 1
        const long long rays_per_group = (m->precision == NRNERF_PREC_F32) ? 4 : (m->arch_id == 5 ? 8 : ((bpr & 1) ? 8 : 4));
        return (long long)N >= rays_per_group * m->num_cus;
    };
    const bool fuse_coarse_only = I == 0 && !m->has_bend && !unfused_composite && S <= 256 && enough_rays_to_fuse(S);
    if (fuse_coarse_only) { na.fuse_on = 1; na.fuse = final_composite(S, zc, a->noise_coarse, a->coarse, nullptr); na.raw4 = nullptr; }
    // ---- K1's arguments: coarse composite (+ sampling + merge when a fine pass follows)
    CompositeArgs ca{};
    ca.rays = a->rays; ca.ray_stride = a->ray_stride;
    ca.raw4 = raw_c; ca.z = zc; ca.n_rays = N; ca.S = S; ca.n_importance = I;
    ca.lindisp = a->lindisp; ca.white_bkgd = a->white_bkgd;
    ca.noise = a->noise_coarse; ca.u = a->u_fine;
    if (I > 0) {
        // rgb0/disp0/acc0 are optional for the caller but the kernel always writes them: park them in raw_f
        // (not yet written) when the caller passed NULL.
        ca.rgb = a->rgb0 ? a->rgb0 : raw_f;
        ca.disp = a->disp0 ? a->disp0 : raw_f + (size_t)N * 3;
        ca.acc = a->acc0 ? a->acc0 : raw_f + (size_t)N * 4;
        ca.z_std = a->z_std; ca.z_out = z_fine; ca.z_user = nullptr;
        if (split) { ca.split_bent_in = bent_c; ca.split_bent_out = bent4; ca.z_new = z_new; ca.rank_new = rank_new; }
    } else {
        ca.rgb = a->rgb_map; ca.disp = a->disp_map; ca.acc = a->acc_map;
        ca.z_std = nullptr; ca.z_out = nullptr; ca.z_user = a->z_vals;
    }
    ca.vis = a->coarse.visibility_weights; ca.alpha = a->coarse.opacity_alpha;
    if (I == 0 && surface) { ca.bent4 = bent4; ca.surf_pts = a->surface_pts; ca.surf_rig = a->surface_rigidity; ca.med_idx = a->median_index; }
    // K1 inside K0 (north_star: "compositing fused into the ray loop"; train.py:889-920): on the split path's 16x16x32 coarse trunk a wave
    // owns whole rays, so compositing, sample_pdf and the merge run as its epilogue (net_kernel_x16<.., SAMPLE>: composite_kernel's own
    // code, same bits) and raw_c never reaches HBM.  NRNERF_RENDER_COARSE_EPILOGUE_ON / _OFF select per call; the default follows the
    // A/B on one box.
#ifndef NRN_COARSE_EPILOGUE_DEFAULT
#define NRN_COARSE_EPILOGUE_DEFAULT 0
#endif
    const bool epilogue_wanted = (a->flags & NRNERF_RENDER_COARSE_EPILOGUE_OFF) ? false :
                                 ((a->flags & NRNERF_RENDER_COARSE_EPILOGUE_ON) ? true : NRN_COARSE_EPILOGUE_DEFAULT != 0);
    const bool fuse_coarse_epilogue = x16_coarse && epilogue_wanted && !unfused_composite && S <= x16_coarse_epilogue_max_samples() &&
                                      (long long)N >= x16_rays_per_group(trunk_arch(m->arch_id), S) * m->num_cus;
    if (fuse_coarse_epilogue) { na.fuse_on = 1; na.fuse = ca; na.fuse.raw4 = nullptr; na.raw4 = nullptr; }
    hipError_t e;
    if (split_coarse) {
        // KBc: stand-alone bender over the S coarse samples, then the coarse trunk on the bent points
        BendArgs bc{};
        bc.rays = a->rays; bc.ray_stride = a->ray_stride;
        bc.latents = a->latents; bc.lat_stride = a->latent_stride;
        bc.z = zc; bc.lindisp = a->lindisp; bc.rank = nullptr; bc.n_rays = N; bc.n_per_ray = S; bc.out_stride = S;
        bc.bent4 = bent_c; bc.knobs = kn;
        e = run_bender(bc, 5, S);
        if (e != hipSuccess) return NRNERF_ERR_HIP;
        na.pts4 = bent_c; na.bent4 = nullptr;
        if (x16_coarse) {
            na.wstream = m->coarse_trunk_x16.stream; na.bias = m->coarse_trunk_x16.bias;
            e = timed(0, (fuse_coarse_epilogue ? "net_kernel_x16 + fused compositing, sample_pdf, merge" : "net_kernel_x16"), (double)N * S * m->coarse_trunk_x16.algo_flops_per_sample, (double)N * S * m->coarse_trunk_x16.mfma_flops_per_sample,
                      [&] { na.work_counter = trunk_counter(0); return launch_net_x16(m->precision, trunk_arch(m->arch_id), m->views, na, m->num_cus, stream); });
        } else {
            na.wstream = m->coarse_trunk.stream; na.bias = m->coarse_trunk.bias;
            e = timed(0, "net_kernel (trunk only)", (double)N * S * m->coarse_trunk.algo_flops_per_sample, (double)N * S * m->coarse_trunk.mfma_flops_per_sample,
                      [&] { return launch_net(m->precision, false, m->views, trunk_arch(m->arch_id), na, m->num_cus, stream); });
        }
    } else {
        e = timed(0, (m->has_bend ? "net_kernel (fused bender)" : (fuse_coarse_only ? "net_kernel + fused compositing" : "net_kernel")), (double)N * S * m->coarse.algo_flops_per_sample, (double)N * S * m->coarse.mfma_flops_per_sample,
                  [&] { return launch_net(m->precision, m->has_bend, m->views, m->exact ? 3 + m->arch_id : m->arch_id, na, m->num_cus, stream); });
    }
    if (e != hipSuccess) return NRNERF_ERR_HIP;
    if (fuse_coarse_only) return NRNERF_OK;

    // ---- K1: coarse composite (+ sampling), unless it ran as K0's epilogue
    if (!fuse_coarse_epilogue)
        e = timed(1, "composite_kernel", 0, 0, [&] { return launch_composite(ca, stream); });
    if (e != hipSuccess) return NRNERF_ERR_HIP;
    if (I == 0) return NRNERF_OK;

    // ---- K2: fine network on the merged depths
    NetArgs nf = na;
    nf.fuse_on = 0; nf.fuse = CompositeArgs{};           // (the coarse pass' epilogue, if any, was its own)
    nf.pts4 = nullptr;
    nf.z = z_fine; nf.S = SF;
    nf.raw4 = raw_f; nf.raw_out = a->raw; nf.raw_ch = m->fine.output_ch;
    nf.ex = sample_out(a->fine);
    // K3 inside K2 (see final_composite above) whenever K2 is a kernel without a fused bender
    // the split path's trunk-only pass on the 16x16x32 kernel (nrnerf_net_x16.h) when the call wants no detail outputs
    // (NRNERF_RENDER_NO_X16: the 32x32x16 kernel of nrnerf_net_mb.h)
    // (read per call, like NRNERF_UNFUSED_COMPOSITE: the parity tests run both kernels in one process)
    const bool x16 = split && x16_mode != 0 && m->fine_trunk_x16.stream && !a->detailed_output && !kn.detailed;
    const long long x16_group = x16_rays_per_group(trunk_arch(m->arch_id), SF);          // (nrnerf_net_x16.hip: the kernel's own ray-group size)
    const bool fuse_fine = (split || !m->has_bend) && !unfused_composite && SF <= 256 &&
                           (x16 ? (long long)N >= x16_group * m->num_cus : enough_rays_to_fuse(SF));
    if (fuse_fine) { nf.fuse_on = 1; nf.fuse = final_composite(SF, z_fine, a->noise_fine, a->fine, nullptr); nf.raw4 = nullptr; }
    if (split) {
        // KB: only the I importance samples go through the bender; the coarse samples' bent points are already in place
        BendArgs ba{};
        ba.rays = a->rays; ba.ray_stride = a->ray_stride;
        ba.latents = a->latents; ba.lat_stride = a->latent_stride;
        ba.z = z_new; ba.rank = rank_new; ba.n_rays = N; ba.n_per_ray = I; ba.out_stride = SF;
        ba.bent4 = bent4; ba.knobs = kn;
        e = run_bender(ba, 4, I);
        if (e != hipSuccess) return NRNERF_ERR_HIP;
        // K2: trunk + head on ready-made points (compiled architecture 0 without bender)
        nf.pts4 = bent4; nf.bent4 = nullptr;
        if (x16) {
            nf.wstream = m->fine_trunk_x16.stream; nf.bias = m->fine_trunk_x16.bias;
            e = timed(2, (fuse_fine ? "net_kernel_x16 + fused compositing" : "net_kernel_x16"), (double)N * SF * m->fine_trunk_x16.algo_flops_per_sample, (double)N * SF * m->fine_trunk_x16.mfma_flops_per_sample,
                      [&] { nf.work_counter = trunk_counter(1); return launch_net_x16(m->precision, trunk_arch(m->arch_id), m->views, nf, m->num_cus, stream); });
        } else {
            nf.wstream = m->fine_trunk.stream; nf.bias = m->fine_trunk.bias;
            e = timed(2, (fuse_fine ? "net_kernel (trunk only) + fused compositing" : "net_kernel (trunk only)"), (double)N * SF * m->fine_trunk.algo_flops_per_sample, (double)N * SF * m->fine_trunk.mfma_flops_per_sample,
                      [&] { return launch_net(m->precision, false, m->views, trunk_arch(m->arch_id), nf, m->num_cus, stream); });
        }
    } else {
        nf.wstream = m->fine.stream; nf.bias = m->fine.bias;
        nf.bent4 = bent4;
        e = timed(2, (m->has_bend ? "net_kernel (fused bender)" : (fuse_fine ? "net_kernel + fused compositing" : "net_kernel")), (double)N * SF * m->fine.algo_flops_per_sample, (double)N * SF * m->fine.mfma_flops_per_sample,
                  [&] { return launch_net(m->precision, m->has_bend, m->views, m->exact ? 3 + m->arch_id : m->arch_id, nf, m->num_cus, stream); });
    }
    if (e != hipSuccess) return NRNERF_ERR_HIP;

    if (fuse_fine) return NRNERF_OK;

    // ---- K3: fine composite
    const CompositeArgs cf = final_composite(SF, z_fine, a->noise_fine, a->fine, raw_f);
    e = timed(3, "composite_kernel", 0, 0, [&] { return launch_composite(cf, stream); });
    if (e != hipSuccess) return NRNERF_ERR_HIP;
    return NRNERF_OK;
} NRN_CATCH

int nrnerf_generate_rays(const nrnerf_camera* cam, float near_plane, float far_plane, float* rays_out,
                         int32_t ray_stride, void* hip_stream) try {
    if (!cam || !rays_out || (ray_stride != 8 && ray_stride != 11)) return NRNERF_ERR_INVALID;
    if (cam->height <= 0 || cam->width <= 0 || cam->focal_x == 0.0f || cam->focal_y == 0.0f) return NRNERF_ERR_INVALID;
    RayGenArgs a{};
    std::memcpy(a.c2w, cam->c2w, sizeof(a.c2w));
    a.fx = cam->focal_x; a.fy = cam->focal_y; a.cx = cam->center_x; a.cy = cam->center_y;
    a.H = cam->height; a.W = cam->width; a.near = near_plane; a.far = far_plane;
    a.rays = rays_out; a.ray_stride = ray_stride;
    // the launch goes to the device that owns rays_out, whatever the calling thread's current device is (as nrnerf_render)
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, rays_out) != hipSuccess) { (void)hipGetLastError(); return NRNERF_ERR_INVALID; }
    if (attr.type != hipMemoryTypeDevice) return NRNERF_ERR_INVALID;
    DeviceGuard guard(attr.device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    return launch_raygen(a, (hipStream_t)hip_stream) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

int nrnerf_sample_depths(const float* rays, int32_t ray_stride, const float* uniforms, int32_t n_rays, int32_t n_samples,
                         int32_t lindisp, float* z_out, void* hip_stream) try {
    if (!rays || !z_out || ray_stride < 8 || n_rays < 0 || n_samples < 2 || n_samples > NRNERF_MAX_SAMPLES) return NRNERF_ERR_INVALID;
    if (n_rays == 0) return NRNERF_OK;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, z_out) != hipSuccess) { (void)hipGetLastError(); return NRNERF_ERR_INVALID; }
    if (attr.type != hipMemoryTypeDevice) return NRNERF_ERR_INVALID;
    DeviceGuard guard(attr.device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    JitterArgs j{rays, ray_stride, uniforms, n_rays, n_samples, lindisp, z_out};
    return launch_zjitter(j, (hipStream_t)hip_stream) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH
int nrnerf_sample_depths_points(const float* rays, int32_t ray_stride, const float* uniforms, int32_t n_rays, int32_t n_samples,
                                int32_t lindisp, float* z_out, float* points_out, void* hip_stream) try {
    if (!rays || !z_out || !points_out || ray_stride < 8 || n_rays < 0 || n_samples < 2 || n_samples > NRNERF_MAX_SAMPLES) return NRNERF_ERR_INVALID;
    if (n_rays == 0) return NRNERF_OK;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, z_out) != hipSuccess) { (void)hipGetLastError(); return NRNERF_ERR_INVALID; }
    if (attr.type != hipMemoryTypeDevice) return NRNERF_ERR_INVALID;
    DeviceGuard guard(attr.device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    SamplePointsArgs j{rays, ray_stride, uniforms, n_rays, n_samples, lindisp, z_out, points_out};
    return launch_sample_points(j, (hipStream_t)hip_stream) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

namespace {
// the device that owns `ptr` (device memory): NRNERF_OK and `dev`, or NRNERF_ERR_INVALID
int device_of(const void* ptr, int& dev) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) { (void)hipGetLastError(); return NRNERF_ERR_INVALID; }
    if (attr.type != hipMemoryTypeDevice) return NRNERF_ERR_INVALID;
    dev = attr.device;
    return NRNERF_OK;
}
}  // namespace

namespace {
int generic_trunk_call(const nrnerf_model* m, const nrnerf_generic_trunk_args* a, bool backward, void* hip_stream) {
    if (!m || !a || a->struct_size != sizeof(nrnerf_generic_trunk_args)) return NRNERF_ERR_INVALID;
    if (!m->generic || !m->gen_train_ok) return NRNERF_ERR_UNSUPPORTED;
    if (a->which < 0 || a->which > 1 || a->n_rays < 0 || a->n_samples < 1 || a->n_samples > NRNERF_MAX_SAMPLES || !a->acts) return NRNERF_ERR_INVALID;
    const bool fine = a->which == 1 && !m->gen_fine_is_coarse;
    const nrnerf_model::GenTrainNet& tn = m->gen_tn[fine ? 1 : 0];
    if (!backward && (!a->pts4 || !a->raw4 || (tn.views && !a->dirs) || (tn.lat > 0 && !a->latents))) return NRNERF_ERR_INVALID;
    // the epilogue writes channels 0..3 of a row of `raw`, and channel 4 when raw_ch > 4: the row must hold them and the network must have them
    if (!backward && a->raw && (a->raw_ch < 4 || a->raw_ch > (fine ? m->gen_fine : m->gen_coarse).output_ch)) return NRNERF_ERR_INVALID;
    if (backward && (!a->d_raw4 || !a->d_pre || !a->d_enc0 || (tn.skip && !a->d_enc1) || (tn.views && !a->d_encv))) return NRNERF_ERR_INVALID;
    if (a->n_rays == 0) return NRNERF_OK;
    DeviceGuard guard(m->device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    const long long M = (long long)a->n_rays * a->n_samples;
    GenArgs g = backward ? (fine ? m->gen_fine_bwd_prog : m->gen_coarse_bwd_prog) : (fine ? m->gen_fine_prog : m->gen_coarse_prog);
    const PassDev& pd = backward ? (fine ? m->gen_fine_bwd : m->gen_coarse_bwd) : (fine ? m->gen_fine : m->gen_coarse);
    g.wstream = pd.stream; g.bias = pd.bias;
    g.n_rays = a->n_rays; g.S = a->n_samples;
    g.save_stride = M * tn.W; g.save_w = tn.W;
    if (!backward) {
        // the forward pass on the width-class 16x16x32 kernel (nrnerf_gx16.h, SAVE: activations written from the registers) when it has this
        // trunk: bf16, plain head, no latent input columns; 0.55 of the matrix pipe's peak instead of the run-time-parameterised kernel's 0.07
        const PassDev& gx = (fine ? m->gx_fine : m->gx_coarse);
        const GxMeta& gm = (fine ? m->gx_meta_fine : m->gx_meta_coarse);
        if (gx.stream && m->precision == NRNERF_PREC_BF16 && !tn.views && tn.lat == 0 && tn.W % 4 == 0 && M < (1ll << 32) &&
            (long long)a->n_rays * ((a->n_samples + 15) / 16) < (1ll << 31)) {
            GxArgs x{};
            x.pts4 = a->pts4; x.raw4 = a->raw4; x.raw_out = a->raw; x.raw_ch = a->raw ? a->raw_ch : 4;
            x.n_rays = a->n_rays; x.S = a->n_samples; x.wstream = gx.stream; x.bias = gx.bias;
            x.depth = gm.depth; x.skip = gm.skip; x.L = gm.L; x.n_bias_tiles = gm.n_bias_tiles; x.LV = gm.LV;
            x.save = a->acts; x.save_stride = M * tn.W; x.save_w = tn.W; x.relu_bits = a->relu_bits;
            return launch_gx16(m->precision, gm.wc, false, x, m->num_cus, (hipStream_t)hip_stream) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
        }
        g.mode = 1;
        g.rays = a->pts4; g.ray_stride = 0;                                              // (points are handed in: the ray record is never read)
        g.latents = tn.lat > 0 ? a->latents : nullptr; g.lat_stride = tn.lat;
        g.z = nullptr; g.lindisp = 0; g.pts4 = a->pts4; g.dirs_from_pts = 0; g.dirs = tn.views ? a->dirs : nullptr;
        g.raw4 = a->raw4; g.raw_out = a->raw; g.raw_ch = a->raw ? a->raw_ch : 4; g.bent4 = nullptr;
        g.save = a->acts; g.mask = nullptr;
    } else {
        // backward-data on the width-class kernel's dataflow (nrnerf_gx16_bwd.h) when the forward call left its relu bits
        const PassDev& gxb = (fine ? m->gx_fine_bwd : m->gx_coarse_bwd);
        const GxMeta& gmb = (fine ? m->gx_meta_fine_bwd : m->gx_meta_coarse_bwd);
        if (gxb.stream && a->relu_bits && !tn.views && tn.lat == 0 && M < (1ll << 32) && (long long)a->n_rays * ((a->n_samples + 15) / 16) < (1ll << 31)) {
            GxBwdArgs b{};
            b.d_raw4 = a->d_raw4; b.relu_bits = a->relu_bits; b.d_pre = a->d_pre; b.save_stride = M * tn.W; b.save_w = tn.W;
            b.d_enc0 = a->d_enc0; b.d_enc1 = a->d_enc1; b.enc_w = tn.in_w;
            b.n_rays = a->n_rays; b.S = a->n_samples; b.wstream = gxb.stream; b.bias = gxb.bias;
            b.depth = gmb.depth; b.skip = gmb.skip; b.L = gmb.L; b.n_bias_tiles = gmb.n_bias_tiles;
            return launch_gx16_bwd(gmb.wc, b, m->num_cus, (hipStream_t)hip_stream) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
        }
        g.mode = 2;
        g.rays = a->d_raw4; g.ray_stride = 0;
        g.draw = a->d_raw4; g.draw_ch = 4; g.draw_col = tn.draw_col;
        g.mask = a->acts; g.save = a->d_pre;
        g.gout[0] = a->d_enc0; g.gout[1] = a->d_enc1; g.gout[2] = a->d_encv; g.gout_w = tn.in_w; g.gout_w2 = tn.dv;
    }
    return launch_generic_train(m->precision, g, m->num_cus, (hipStream_t)hip_stream) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
}
int loss_call(const nrnerf_loss_args* a, bool backward, void* hip_stream) {
    if (!a || a->struct_size != sizeof(nrnerf_loss_args) || a->n_rays < 0 || a->n_samples < 0 || !a->rgb_map || !a->target) return NRNERF_ERR_INVALID;
    if (a->weights && (!a->offsets || !a->rigidity)) return NRNERF_ERR_INVALID;
    if (a->divergence && !a->alpha) return NRNERF_ERR_INVALID;
    if ((a->weights || a->divergence) && a->n_samples < 1) return NRNERF_ERR_INVALID;
    if (!backward && !a->loss) return NRNERF_ERR_INVALID;
    if (a->offsets_stride < 0 || a->rigidity_stride < 0 || (a->offsets_stride != 0 && a->offsets_stride < 3)) return NRNERF_ERR_INVALID;
    if (backward && ((!a->g_loss && !a->g_mean) || !a->g_rgb_map || (a->rgb0 && !a->g_rgb0) || (a->weights && (!a->g_offsets || !a->g_rigidity)) ||
                     (a->divergence && !a->g_divergence))) return NRNERF_ERR_INVALID;
    if (a->n_rays == 0) return NRNERF_OK;
    int dev = 0;
    if (device_of(backward ? (const void*)a->g_rgb_map : (const void*)a->loss, dev) != NRNERF_OK) return NRNERF_ERR_INVALID;
    DeviceGuard guard(dev);
    if (!guard.ok) return NRNERF_ERR_HIP;
    LossArgs l{a->n_rays, a->n_samples, a->rgb_map, a->rgb0, a->target, a->weights, a->offsets, a->rigidity, a->alpha, a->divergence,
               a->offsets_weight, a->rigidity_weight, a->divergence_weight, a->schedule, a->loss, a->g_loss, a->g_rgb_map, a->g_rgb0, a->g_offsets,
               a->g_rigidity, a->g_divergence, a->offsets_stride ? a->offsets_stride : 3, a->rigidity_stride ? a->rigidity_stride : 1,
               backward ? a->g_mean : nullptr};
    return launch_loss(l, backward, (hipStream_t)hip_stream) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
}
}  // namespace
int nrnerf_generic_trunk_forward(const nrnerf_model* m, const nrnerf_generic_trunk_args* a, void* hip_stream) try { return generic_trunk_call(m, a, false, hip_stream); } NRN_CATCH
int nrnerf_generic_trunk_backward(const nrnerf_model* m, const nrnerf_generic_trunk_args* a, void* hip_stream) try { return generic_trunk_call(m, a, true, hip_stream); } NRN_CATCH
int nrnerf_model_trains_generic(const nrnerf_model* m) { return m ? ((m->generic && m->gen_train_ok) ? 1 : 0) : NRNERF_ERR_INVALID; }
size_t nrnerf_generic_trunk_bits_bytes(const nrnerf_model* m, int32_t which, int32_t n_rays, int32_t n_samples) {
    if (!m || !m->generic || which < 0 || which > 1 || n_rays < 1 || n_samples < 1) return 0;
    const bool fine = which == 1 && !m->gen_fine_is_coarse;
    const PassDev& gxb = fine ? m->gx_fine_bwd : m->gx_coarse_bwd;
    const GxMeta& gmb = fine ? m->gx_meta_fine_bwd : m->gx_meta_coarse_bwd;
    if (!gxb.stream) return 0;
    return (size_t)gmb.depth * (size_t)n_rays * (size_t)((n_samples + 15) / 16) * 64 * (size_t)gx16_bits_bytes_per_lane(gmb.wc);
}
int nrnerf_loss_forward(const nrnerf_loss_args* a, void* hip_stream) try { return loss_call(a, false, hip_stream); } NRN_CATCH
int nrnerf_code_gradients(const int64_t* index, const float* g, int32_t n_rays, int32_t latent_size, int32_t n_codes, float* out, void* hip_stream) try {
    if (!index || !g || !out || n_rays < 0 || latent_size < 1 || latent_size > 256 || n_codes < 0) return NRNERF_ERR_INVALID;
    if (n_codes == 0) return NRNERF_OK;
    int dev = 0;
    if (device_of(out, dev) != NRNERF_OK) return NRNERF_ERR_INVALID;
    DeviceGuard guard(dev);
    if (!guard.ok) return NRNERF_ERR_HIP;
    CodeGradArgs c{(const long long*)index, g, n_rays, latent_size, n_codes, out};
    return launch_code_gradients(c, (hipStream_t)hip_stream) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH
int nrnerf_loss_backward(const nrnerf_loss_args* a, void* hip_stream) try { return loss_call(a, true, hip_stream); } NRN_CATCH

int nrnerf_merge_rows(const uint8_t* rank_new, int32_t n_rays, int32_t n_samples, int32_t n_importance, float* coarse_a, float* coarse_b,
                      float* new_a, float* new_b, float* merged_a, float* merged_b, int32_t inverse, void* hip_stream) try {
    if (!rank_new || !coarse_a || !new_a || !merged_a || n_rays < 0 || n_samples < 1 || n_importance < 1 || n_samples + n_importance > 256) return NRNERF_ERR_INVALID;
    if ((coarse_b != nullptr) != (merged_b != nullptr) || (new_b != nullptr) != (merged_b != nullptr)) return NRNERF_ERR_INVALID;
    if (n_rays == 0) return NRNERF_OK;
    int dev = 0;
    if (device_of(merged_a, dev) != NRNERF_OK) return NRNERF_ERR_INVALID;
    DeviceGuard guard(dev);
    if (!guard.ok) return NRNERF_ERR_HIP;
    MergeRowsArgs a{n_rays, n_samples, n_importance, rank_new, coarse_a, coarse_b, new_a, new_b, merged_a, merged_b, inverse ? 1 : 0};
    return launch_merge_rows(a, (hipStream_t)hip_stream) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

namespace {
int reduce_partials_call(const float* partials, int64_t record_stride, int32_t n_partials, int32_t n_short, const int32_t* index,
                         int64_t n_out, float* out, const float* aux, int32_t n_aux, const int64_t* aux_pos, void* hip_stream) {
    if (!partials || !index || !out || n_out < 0 || n_partials < 1 || n_short < 0 || n_short > n_partials || record_stride < 1 ||
        record_stride >= NRNERF_REDUCE_SHORT) return NRNERF_ERR_INVALID;
    if (aux && (n_aux < 0 || !aux_pos)) return NRNERF_ERR_INVALID;
    if (n_out == 0) return NRNERF_OK;
    int dev = 0;
    if (device_of(out, dev) != NRNERF_OK) return NRNERF_ERR_INVALID;
    DeviceGuard guard(dev);
    if (!guard.ok) return NRNERF_ERR_HIP;
    ReducePartialsArgs a{partials, record_stride, n_partials, n_short, index, n_out, out, aux, aux ? n_aux : 0, {-1, -1, -1, -1}};
    if (aux)
        for (int c = 0; c < 4; ++c) {
            if (aux_pos[c] >= n_out) return NRNERF_ERR_INVALID;
            a.aux_pos[c] = aux_pos[c];
        }
    return launch_reduce_partials(a, (hipStream_t)hip_stream) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
}
}  // namespace
int nrnerf_reduce_partials(const float* partials, int64_t record_stride, int32_t n_partials, int32_t n_short, const int32_t* index,
                           int64_t n_out, float* out, void* hip_stream) try {
    return reduce_partials_call(partials, record_stride, n_partials, n_short, index, n_out, out, nullptr, 0, nullptr, hip_stream);
} NRN_CATCH
int nrnerf_reduce_partials_aux(const float* partials, int64_t record_stride, int32_t n_partials, int32_t n_short, const int32_t* index,
                               int64_t n_out, float* out, const float* aux, int32_t n_aux, const int64_t* aux_pos, void* hip_stream) try {
    return reduce_partials_call(partials, record_stride, n_partials, n_short, index, n_out, out, aux, n_aux, aux_pos, hip_stream);
} NRN_CATCH

int nrnerf_tile_row_sums(const void* tiles, int64_t n_rows, float* out, void* hip_stream) try {
    if (!tiles || !out || n_rows < 0) return NRNERF_ERR_INVALID;
    if (n_rows == 0) return NRNERF_OK;
    int dev = 0;
    if (device_of(out, dev) != NRNERF_OK) return NRNERF_ERR_INVALID;
    DeviceGuard guard(dev);
    if (!guard.ok) return NRNERF_ERR_HIP;
    return launch_tile_row_sums(tiles, n_rows, out, (hipStream_t)hip_stream) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

int nrnerf_tiles_to_rows(const void* tiles, int32_t n_rays, int32_t n_samples, int32_t width, void* rows, void* hip_stream) try {
    if (!tiles || !rows || n_rays < 0 || n_samples < 1 || n_samples > 256 || (width != 256 && width != 128)) return NRNERF_ERR_INVALID;
    if (n_rays == 0) return NRNERF_OK;
    int dev = 0;
    if (device_of(rows, dev) != NRNERF_OK) return NRNERF_ERR_INVALID;
    DeviceGuard guard(dev);
    if (!guard.ok) return NRNERF_ERR_HIP;
    return launch_tiles_to_rows(tiles, n_rays, n_samples, width, rows, (hipStream_t)hip_stream) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

int nrnerf_direction_encoding(const float* bent4, int32_t n_rays, int32_t n_samples, int32_t n_freqs, void* enc, int32_t enc_is_bf16,
                              float* g_bent4, void* hip_stream) try {
    if (!bent4 || !enc || n_rays < 0 || n_samples < 2 || n_samples > NRNERF_MAX_SAMPLES || n_freqs < 0 || n_freqs > 10) return NRNERF_ERR_INVALID;
    if (n_rays == 0) return NRNERF_OK;
    int dev = 0;
    if (device_of(enc, dev) != NRNERF_OK) return NRNERF_ERR_INVALID;
    DeviceGuard guard(dev);
    if (!guard.ok) return NRNERF_ERR_HIP;
    DirEncodingArgs d{bent4, n_rays, n_samples, n_freqs, enc, enc_is_bf16 ? 1 : 0, g_bent4};
    return launch_dir_encoding(d, g_bent4 != nullptr, (hipStream_t)hip_stream) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

// ---- training entry points (nrnerf_train.h, composite_bwd_kernel) -------------------------------------------------
namespace {
int trunk_common(const nrnerf_model* m, const nrnerf_trunk_args* a, bool bwd, TrunkArgs& t) {
    if (!m || !a || a->struct_size != sizeof(nrnerf_trunk_args)) return NRNERF_ERR_INVALID;
    if (!m->train_ok) return NRNERF_ERR_UNSUPPORTED;
    if (a->n_rays < 0 || a->n_samples < 1 || a->n_samples > NRNERF_MAX_SAMPLES || (a->which != 0 && a->which != 1)) return NRNERF_ERR_INVALID;
    if (!a->pts4 || !a->acts) return NRNERF_ERR_INVALID;
    if (!bwd && (!a->raw4 || (a->raw && a->raw_ch != 4 && a->raw_ch != 5))) return NRNERF_ERR_INVALID;
    if (bwd && (!a->d_raw4 || !a->d_pre || !a->d_pts4)) return NRNERF_ERR_INVALID;
    const bool fine = a->which == 1;
    const PassDev& fwd = m->coarse_train.stream ? ((fine && !m->fine_is_coarse) ? m->fine_train : m->coarse_train)
                                  : (m->has_bend ? (fine ? m->fine_trunk : m->coarse_trunk) : (fine ? m->fine : m->coarse));
    const PassDev& bw = (fine && !m->fine_is_coarse) ? m->fine_bwd : m->coarse_bwd;
    t = TrunkArgs{};
    t.pts4 = a->pts4; t.n_rays = a->n_rays; t.S = a->n_samples;
    t.wstream = bwd ? bw.stream : fwd.stream; t.bias = fwd.bias;
    t.raw4 = a->raw4; t.raw_out = a->raw; t.raw_ch = a->raw_ch;
    t.acts = a->acts; t.d_raw4 = a->d_raw4; t.d_pre = a->d_pre; t.d_pts4 = a->d_pts4; t.ray_bias = a->ray_bias;
    t.mask = (unsigned short*)a->relu_mask;
    if (m->precision != NRNERF_PREC_F32 && !t.mask) return NRNERF_ERR_INVALID;
    if (m->views) {             // the colour branch behind the trunk (the *_views kernels)
        if (!a->dirs || !a->hv || (bwd && !a->d_pre_v) || (m->precision != NRNERF_PREC_F32 && !a->hv_mask)) return NRNERF_ERR_INVALID;
        t.dirs = a->dirs; t.hv = a->hv; t.hv_mask = (unsigned short*)a->hv_mask; t.d_pre_v = a->d_pre_v; t.d_dirs = a->d_dirs;
    }
    return NRNERF_OK;
}
}  // namespace

int nrnerf_trunk_forward(const nrnerf_model* m, const nrnerf_trunk_args* a, void* hip_stream) try {
    TrunkArgs t;
    const int rc = trunk_common(m, a, false, t);
    if (rc != NRNERF_OK) return rc;
    if (a->n_rays == 0) return NRNERF_OK;
    DeviceGuard guard(m->device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    const bool f32 = m->precision == NRNERF_PREC_F32;
    const hipStream_t s = (hipStream_t)hip_stream;
    const hipError_t e = (m->arch_id == 5) ? (f32 ? launch_trunk_fwd_train_f32_a5(t, m->num_cus, s) : launch_trunk_fwd_train_bf16_a5(t, m->num_cus, s))
                       : m->views ? (f32 ? launch_trunk_fwd_train_f32_views(t, m->num_cus, s) : launch_trunk_fwd_train_bf16_views(t, m->num_cus, s))
                                  : (f32 ? launch_trunk_fwd_train_f32(t, m->num_cus, s) : launch_trunk_fwd_train_bf16(t, m->num_cus, s));
    return e == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

int nrnerf_trunk_backward(const nrnerf_model* m, const nrnerf_trunk_args* a, void* hip_stream) try {
    TrunkArgs t;
    const int rc = trunk_common(m, a, true, t);
    if (rc != NRNERF_OK) return rc;
    if (a->n_rays == 0) return NRNERF_OK;
    DeviceGuard guard(m->device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    const bool f32 = m->precision == NRNERF_PREC_F32;
    const hipStream_t s = (hipStream_t)hip_stream;
    const hipError_t e = (m->arch_id == 5) ? (f32 ? launch_trunk_bwd_f32_a5(t, m->num_cus, s) : launch_trunk_bwd_bf16_a5(t, m->num_cus, s))
                       : m->views ? (f32 ? launch_trunk_bwd_f32_views(t, m->num_cus, s) : launch_trunk_bwd_bf16_views(t, m->num_cus, s))
                                  : (f32 ? launch_trunk_bwd_f32(t, m->num_cus, s) : launch_trunk_bwd_bf16(t, m->num_cus, s));
    return e == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

int nrnerf_trunk_wgrad(const nrnerf_model* m, const nrnerf_wgrad_args* a, void* hip_stream) try {
    if (!m || !a || a->struct_size != sizeof(nrnerf_wgrad_args)) return NRNERF_ERR_INVALID;
    if (!m->train_ok) return NRNERF_ERR_UNSUPPORTED;
    if (a->n_rays < 0 || a->n_samples < 1 || a->n_samples > NRNERF_MAX_SAMPLES || a->n_partials < 1 || a->n_partials > 4096) return NRNERF_ERR_INVALID;
    if (!a->acts || !a->d_pre || !a->pts4 || !a->d_raw4 || !a->enc || !a->g_head || !a->partials) return NRNERF_ERR_INVALID;
    if (a->n_rays == 0) return NRNERF_OK;
    const bool f32 = m->precision == NRNERF_PREC_F32;
    const int W = (m->arch_id == 5) ? ArchNarrow::W : ArchDefault::W, D = ArchDefault::D, SKIP = ArchDefault::SKIP;
    const long long nblocks = (long long)a->n_rays * ((a->n_samples + 31) / 32);
    const long long M = (long long)a->n_rays * a->n_samples;
    // elements of one layer of acts / d_pre: bf16 [block][W][32 samples] tiles, or (fp32 mode) rows [sample][W]
    const size_t layer = f32 ? (size_t)M * W : (size_t)nblocks * W * 32;
    const size_t esz = f32 ? 4 : 2;
    float* const dwh = a->partials;                                 // record layout: NRNERF_WGRAD_STRIDE
    float* const dwe = dwh + (size_t)(D - 1) * W * W;
    float* const dwo = dwe + (size_t)2 * W * 64;
    float* const db = dwo + (size_t)W * 64;
    const char* acts = (const char*)a->acts;
    const char* dpre = (const char*)a->d_pre;
    WgradArgs w{};
    w.nblocks = f32 ? M : nblocks; w.pstride = NRNERF_WGRAD_STRIDE(D, W);
    w.sync_every = NRN_WGRAD_SYNC_DEFAULT;       // (swept in round 3, tools/experiments/README.md; a build-time constant: the library reads no environment)
    // a 64-column job (encoding, head) loads 2 TR + 2 fragments per block and wave, a hidden-to-hidden one 2 TR + 2 TCW:
    // give it that share of the workgroups, so that all workgroups of the launch finish together
    // (fp32 mode: the same split; its 64-column jobs issue a quarter / half of a hidden-to-hidden job's MFMAs per sample and
    //  finish early -- 1.9 of 8.9 n_partials workgroups)
    const int kh = a->n_partials;
    int kl = NRNERF_WGRAD_SHORT_PARTIALS(kh, W);
    kl = kl > kh ? kh : kl;
    int n = 0;
    for (int i = 1; i < D; ++i)                                     // hidden-to-hidden layers: the bulk, first in the grid
        w.job[n++] = WgradJob{dpre + i * layer * esz, acts + (i - 1) * layer * esz, W, dwh + (size_t)(i - 1) * W * W, db + (size_t)i * W, kh, 0, W};
    const bool views = m->views != 0;
    float* const dwf = db + (size_t)(D + 1) * W;                     // view-dependent head: NRNERF_WGRAD_STRIDE_VIEWS
    float* const dwd = dwf + (size_t)(W / 2) * W;
    float* const dwr = dwd + (size_t)(W / 2) * 64;
    float* const dbv = dwr + (size_t)(W / 2) * 64;
    if (views) {
        if (!a->dirs || !a->hv || !a->d_pre_v || !a->encv) return NRNERF_ERR_INVALID;
        w.pstride = NRNERF_WGRAD_STRIDE_VIEWS(D, W);
        // (half the rows of a hidden-to-hidden product per block: its workgroups finish early; kept at kh records so that the
        //  caller's reduction knows two record counts only)
        w.job[n++] = WgradJob{a->d_pre_v, acts + (D - 1) * layer * esz, W, dwf, dbv, kh, 0, W / 2};
    }
    w.job[n++] = WgradJob{dpre, a->enc, 64, dwe, db, kl, 0, W};
    w.job[n++] = WgradJob{dpre + (SKIP + 1) * layer * esz, a->enc, 64, dwe + (size_t)W * 64, db + (size_t)D * W, kl, 0, W};
    w.job[n++] = WgradJob{acts + (D - 1) * layer * esz, a->g_head, 64, dwo, db + (size_t)D * W, kl, 0, W};
    if (views) {        // (their row sums -- of d_pre_v again, of hv -- land in the scratch row db[depth])
        w.job[n++] = WgradJob{a->d_pre_v, a->encv, 64, dwd, db + (size_t)D * W, kl, 0, W / 2};
        w.job[n++] = WgradJob{a->hv, a->g_head, 64, dwr, db + (size_t)D * W, kl, 0, W / 2};
    }
    w.njobs = n;
    for (int j = 0, wg = 0; j < n; ++j) { w.job[j].wg0 = wg; wg += w.job[j].kch; w.nwg = wg; }
    DeviceGuard guard(m->device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    const hipStream_t s = (hipStream_t)hip_stream;
    const WgradOperandArgs ops{a->pts4, a->d_raw4, a->n_rays, a->n_samples, ArchDefault::L, a->enc, a->g_head, f32 ? nullptr : a->head_sums,
                               views ? a->dirs : nullptr, ArchDefault::LV, views ? a->encv : nullptr};
    if ((f32 ? launch_wgrad_operands_f32(ops, s) : launch_wgrad_operands(ops, s)) != hipSuccess) return NRNERF_ERR_HIP;
    const hipError_t e = (m->arch_id == 5) ? (f32 ? launch_trunk_wgrad_f32_a5(w, s) : launch_trunk_wgrad_bf16_a5(w, s))
                            : views ? (f32 ? launch_trunk_wgrad_f32_views(w, s) : launch_trunk_wgrad_bf16_views(w, s))
                                    : (f32 ? launch_trunk_wgrad_f32(w, s) : launch_trunk_wgrad_bf16(w, s));
    return e == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

namespace {
int bender_common(const nrnerf_model* m, const nrnerf_bender_args* a, bool bwd, BendTrainArgs& t) {
    if (!m || !a || a->struct_size != sizeof(nrnerf_bender_args)) return NRNERF_ERR_INVALID;
    if (!m->bend_train_ok) return NRNERF_ERR_UNSUPPORTED;
    if (a->n_rays < 0 || a->n_samples < 1 || a->n_samples > NRNERF_MAX_SAMPLES) return NRNERF_ERR_INVALID;
    if (!a->rays || a->ray_stride < 6 || !a->latents || a->latent_stride < m->latent_size || !a->z) return NRNERF_ERR_INVALID;
    if (!a->bent4 || !a->off4 || !a->acts_offsets || !a->acts_rigidity) return NRNERF_ERR_INVALID;
    if (bwd && (!a->g_bent4 || !a->dz_offsets || !a->dz_rigidity || !a->dz_out4 || !a->d_latents)) return NRNERF_ERR_INVALID;
    t = BendTrainArgs{};
    t.rays = a->rays; t.ray_stride = a->ray_stride; t.latents = a->latents; t.lat_stride = a->latent_stride; t.z = a->z;
    t.n_rays = a->n_rays; t.S = a->n_samples;
    const PassDev& p = bwd ? m->bend_train_bwd : m->bend_train_fwd;
    t.wstream = p.stream; t.bias = p.bias;
    t.knobs.has_cutoff = a->has_rigidity_cutoff; t.knobs.cutoff = a->rigidity_cutoff;
    t.knobs.has_scaling = a->has_test_time_scaling; t.knobs.scaling = a->test_time_scaling;
    t.bent4 = a->bent4; t.off4 = a->off4; t.acts_b = a->acts_offsets; t.acts_r = a->acts_rigidity;
    t.g_bent4 = a->g_bent4; t.g_bent4_b = a->g_bent4_b; t.g_unmasked = a->g_unmasked_offsets; t.g_mask = a->g_rigidity_mask;
    t.dz_b = a->dz_offsets; t.dz_r = a->dz_rigidity; t.dz_out4 = a->dz_out4; t.d_lat = a->d_latents;
    return NRNERF_OK;
}
}  // namespace

int nrnerf_bender_forward(const nrnerf_model* m, const nrnerf_bender_args* a, void* hip_stream) try {
    BendTrainArgs t;
    const int rc = bender_common(m, a, false, t);
    if (rc != NRNERF_OK) return rc;
    if (a->n_rays == 0) return NRNERF_OK;
    DeviceGuard guard(m->device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    const bool b16 = m->precision != NRNERF_PREC_F32;        // element type of the saved arrays (nrnerf_bender_args)
    const hipError_t e = (bender_arch_of(m) == 0) ? launch_bend_fwd_train_a0(t, m->num_cus, (hipStream_t)hip_stream, b16)
                                           : launch_bend_fwd_train_a1(t, m->num_cus, (hipStream_t)hip_stream, b16);
    return e == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

int nrnerf_bender_backward(const nrnerf_model* m, const nrnerf_bender_args* a, void* hip_stream) try {
    BendTrainArgs t;
    const int rc = bender_common(m, a, true, t);
    if (rc != NRNERF_OK) return rc;
    if (a->n_rays == 0) return NRNERF_OK;
    DeviceGuard guard(m->device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    const bool b16 = m->precision != NRNERF_PREC_F32;
    const hipError_t e = (bender_arch_of(m) == 0) ? launch_bend_bwd_a0(t, m->num_cus, (hipStream_t)hip_stream, b16)
                                           : launch_bend_bwd_a1(t, m->num_cus, (hipStream_t)hip_stream, b16);
    return e == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

int nrnerf_bender_wgrad(const nrnerf_model* m, const nrnerf_bender_wgrad_args* a, void* hip_stream) try {
    if (!m || !a || a->struct_size != sizeof(nrnerf_bender_wgrad_args)) return NRNERF_ERR_INVALID;
    if (!m->bend_train_ok) return NRNERF_ERR_UNSUPPORTED;
    if (a->n_rays < 0 || a->n_samples < 1 || a->n_samples > NRNERF_MAX_SAMPLES || a->n_partials < 4 || a->n_partials > 4096 || a->n_partials % 4) return NRNERF_ERR_INVALID;
    if (!a->rays || a->ray_stride < 6 || !a->latents || a->latent_stride < m->latent_size || !a->z) return NRNERF_ERR_INVALID;
    if (!a->acts_offsets || !a->acts_rigidity || !a->dz_offsets || !a->dz_rigidity || !a->dz_out4 || !a->partials) return NRNERF_ERR_INVALID;
    if (a->n_rays == 0) return NRNERF_OK;
    const int BD = (bender_arch_of(m) == 0) ? ArchDefault::BD : ArchDeepBend::BD;
    const int BW = ArchDefault::BW, RD = ArchDefault::RD, RW = ArchDefault::RW, X0 = 3 + ArchDefault::LAT;
    const size_t M = (size_t)a->n_rays * a->n_samples;
    if (m->precision != NRNERF_PREC_F32 && M * 64 * 4 >= 0xffffff00ull) return NRNERF_ERR_INVALID;      // 32-bit offsets in bend_wgrad16
    BendWgradArgs w{};
    int n = 0;
    const int b16 = m->precision != NRNERF_PREC_F32;        // the saved arrays' element type; dz_out4 is fp32 in every mode
    const size_t esz = b16 ? 2 : 4;
    auto at = [&](const void* base, size_t elems) { return (const void*)((const char*)base + elems * esz); };
    w.job[n++] = BendWgradJob{a->dz_offsets, BW, BW, nullptr, X0, X0, nullptr, nullptr, b16, 0};            // network[0]: input = [point, latent]
    for (int i = 1; i <= BD - 2; ++i)
        w.job[n++] = BendWgradJob{at(a->dz_offsets, (size_t)i * M * BW), BW, BW, at(a->acts_offsets, (size_t)(i - 1) * M * BW), BW, BW, nullptr, nullptr, b16, b16};
    w.job[n++] = BendWgradJob{a->dz_out4, 4, 3, at(a->acts_offsets, (size_t)(BD - 2) * M * BW), BW, BW, nullptr, nullptr, 0, b16};   // network[BD-1]: 3 x BW
    w.job[n++] = BendWgradJob{a->dz_rigidity, RW, RW, nullptr, X0, 3, nullptr, nullptr, b16, 0};             // rigidity_network[0]: input = the point
    for (int i = 1; i <= RD - 2; ++i)
        w.job[n++] = BendWgradJob{at(a->dz_rigidity, (size_t)i * M * RW), RW, RW, at(a->acts_rigidity, (size_t)(i - 1) * M * RW), RW, RW, nullptr, nullptr, b16, b16};
    w.job[n++] = BendWgradJob{a->dz_out4 + 3, 4, 1, at(a->acts_rigidity, (size_t)(RD - 2) * M * RW), RW, RW, nullptr, nullptr, 0, b16};   // the logit's layer: 1 x RW
    w.njobs = n; w.nparts = a->n_partials; w.m = (long long)M; w.out = a->partials;
    w.rays = a->rays; w.ray_stride = a->ray_stride; w.latents = a->latents; w.lat_stride = a->latent_stride; w.lat = m->latent_size;
    w.z = a->z; w.S = a->n_samples;
    DeviceGuard guard(m->device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    return launch_bend_wgrad(w, (hipStream_t)hip_stream, m->precision != NRNERF_PREC_F32) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

namespace {
int divergence_common(const nrnerf_model* m, const nrnerf_divergence_args* a, bool bwd, BendDivArgs& t) {
    if (!m || !a || a->struct_size != sizeof(nrnerf_divergence_args)) return NRNERF_ERR_INVALID;
    if (!m->bend_train_ok) return NRNERF_ERR_UNSUPPORTED;
    if (a->n_points < 0 || a->n_points >= (1ll << 36)) return NRNERF_ERR_INVALID;
    if (!a->points || !a->probe || !a->latents || (a->latent_stride != 0 && a->latent_stride < m->latent_size)) return NRNERF_ERR_INVALID;
    if (!a->divergence || !a->off4 || !a->toff4 || !a->acts_offsets || !a->tacts_offsets || !a->acts_rigidity || !a->tacts_rigidity)
        return NRNERF_ERR_INVALID;
    if (bwd && ((!a->g_divergence && !a->g_tangent) || !a->dz_offsets || !a->dtz_offsets || !a->dz_rigidity || !a->dtz_rigidity || !a->dz_out4 ||
                !a->dtz_out4 || !a->d_latents || !a->partials || a->n_partials < 4 || a->n_partials > 4096 || a->n_partials % 4))
        return NRNERF_ERR_INVALID;
    t = BendDivArgs{};
    t.pts = a->points; t.latents = a->latents; t.lat_stride = a->latent_stride; t.e = a->probe; t.m = a->n_points;
    const PassDev& p = bwd ? m->bend_train_bwd : m->bend_train_fwd;
    t.wstream = p.stream; t.bias = p.bias;
    t.knobs.has_cutoff = a->has_rigidity_cutoff; t.knobs.cutoff = a->rigidity_cutoff;
    t.knobs.has_scaling = a->has_test_time_scaling; t.knobs.scaling = a->test_time_scaling;
    t.div = a->divergence; t.off4 = a->off4; t.toff4 = a->toff4; t.tvec = a->tangent; t.g_tvec = a->g_tangent;
    t.r_g_bent4 = a->render_g_bent4; t.r_g_bent4_b = a->render_g_bent4_b; t.r_g_unmasked = a->render_g_unmasked_offsets; t.r_g_mask = a->render_g_rigidity_mask;
    t.bent4 = bwd ? nullptr : a->bent4;
    t.acts_b = a->acts_offsets; t.tacts_b = a->tacts_offsets; t.acts_r = a->acts_rigidity; t.tacts_r = a->tacts_rigidity;
    t.g_div = a->g_divergence; t.dz_b = a->dz_offsets; t.dtz_b = a->dtz_offsets; t.dz_r = a->dz_rigidity; t.dtz_r = a->dtz_rigidity;
    t.dz_out4 = a->dz_out4; t.dtz_out4 = a->dtz_out4; t.d_lat = a->d_latents;
    return NRNERF_OK;
}
}  // namespace

int nrnerf_bender_divergence_forward(const nrnerf_model* m, const nrnerf_divergence_args* a, void* hip_stream) try {
    BendDivArgs t;
    const int rc = divergence_common(m, a, false, t);
    if (rc != NRNERF_OK) return rc;
    if (a->n_points == 0) return NRNERF_OK;
    DeviceGuard guard(m->device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    const bool b16 = m->precision != NRNERF_PREC_F32;
    const hipError_t e = (bender_arch_of(m) == 0) ? launch_bend_div_fwd_a0(t, m->num_cus, (hipStream_t)hip_stream, b16)
                                                        : launch_bend_div_fwd_a1(t, m->num_cus, (hipStream_t)hip_stream, b16);
    return e == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

int nrnerf_bender_divergence_backward(const nrnerf_model* m, const nrnerf_divergence_args* a, void* hip_stream) try {
    BendDivArgs t;
    const int rc = divergence_common(m, a, true, t);
    if (rc != NRNERF_OK) return rc;
    if (a->n_points == 0) return NRNERF_OK;
    const bool b16 = m->precision != NRNERF_PREC_F32;
    if (b16 && (size_t)a->n_points * 64 * 4 >= 0xffffff00ull) return NRNERF_ERR_INVALID;      // 32-bit offsets in bend_wgrad16: nothing is launched
    DeviceGuard guard(m->device);
    if (!guard.ok) return NRNERF_ERR_HIP;
    hipError_t e = (bender_arch_of(m) == 0) ? launch_bend_div_bwd_a0(t, m->num_cus, (hipStream_t)hip_stream, b16)
                                                  : launch_bend_div_bwd_a1(t, m->num_cus, (hipStream_t)hip_stream, b16);
    if (e != hipSuccess) return NRNERF_ERR_HIP;
    // weight / bias gradients: dW_i = dz_i^T h_{i-1} + dtz_i^T th_{i-1} (two products per job), db_i = column sums of dz_i
    const int BD = (bender_arch_of(m) == 0) ? ArchDefault::BD : ArchDeepBend::BD;
    const int BW = ArchDefault::BW, RD = ArchDefault::RD, RW = ArchDefault::RW, LAT = m->latent_size;
    const size_t M = (size_t)a->n_points;
    BendWgradArgs w{};
    int n = 0;
    const int s16 = b16 ? 1 : 0;                             // the saved arrays' element type; points / probes / latents / dz_out4: fp32
    const size_t esz = b16 ? 2 : 4;
    auto at = [&](const void* base, size_t elems) { return (const void*)((const char*)base + elems * esz); };
    w.job[n++] = BendWgradJob{a->dz_offsets, BW, BW, a->points, 3, 3, a->dtz_offsets, a->probe, s16, 0};                 // network[0][:, 0:3]: th_0 = e
    w.job[n++] = BendWgradJob{a->dz_offsets, BW, BW, a->latents, a->latent_stride, LAT, nullptr, nullptr, s16, 0};        // network[0][:, 3:]
    for (int i = 1; i <= BD - 2; ++i)
        w.job[n++] = BendWgradJob{at(a->dz_offsets, (size_t)i * M * BW), BW, BW, at(a->acts_offsets, (size_t)(i - 1) * M * BW), BW, BW,
                                  at(a->dtz_offsets, (size_t)i * M * BW), at(a->tacts_offsets, (size_t)(i - 1) * M * BW), s16, s16};
    w.job[n++] = BendWgradJob{a->dz_out4, 4, 3, at(a->acts_offsets, (size_t)(BD - 2) * M * BW), BW, BW,
                              a->dtz_out4, at(a->tacts_offsets, (size_t)(BD - 2) * M * BW), 0, s16};
    w.job[n++] = BendWgradJob{a->dz_rigidity, RW, RW, a->points, 3, 3, a->dtz_rigidity, a->probe, s16, 0};               // rigidity_network[0]
    for (int i = 1; i <= RD - 2; ++i)
        w.job[n++] = BendWgradJob{at(a->dz_rigidity, (size_t)i * M * RW), RW, RW, at(a->acts_rigidity, (size_t)(i - 1) * M * RW), RW, RW,
                                  at(a->dtz_rigidity, (size_t)i * M * RW), at(a->tacts_rigidity, (size_t)(i - 1) * M * RW), s16, s16};
    w.job[n++] = BendWgradJob{a->dz_out4 + 3, 4, 1, at(a->acts_rigidity, (size_t)(RD - 2) * M * RW), RW, RW,
                              a->dtz_out4 + 3, at(a->tacts_rigidity, (size_t)(RD - 2) * M * RW), 0, s16};
    w.njobs = n; w.nparts = a->n_partials; w.m = (long long)M; w.out = a->partials;
    w.S = 1;
    e = launch_bend_wgrad(w, (hipStream_t)hip_stream, m->precision != NRNERF_PREC_F32);
    return e == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

namespace {
int composite_device(const nrnerf_composite_args* a, int* dev) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, a->raw4) != hipSuccess) { (void)hipGetLastError(); return NRNERF_ERR_INVALID; }
    if (attr.type != hipMemoryTypeDevice) return NRNERF_ERR_INVALID;
    *dev = attr.device;
    return NRNERF_OK;
}
}  // namespace

int nrnerf_composite_forward(const nrnerf_composite_args* a, void* hip_stream) try {
    if (!a || a->struct_size != sizeof(nrnerf_composite_args)) return NRNERF_ERR_INVALID;
    if (a->n_rays < 0 || a->n_samples < 2 || a->n_importance < 0) return NRNERF_ERR_INVALID;
    if (a->n_samples > NRNERF_MAX_SAMPLES || a->n_samples + a->n_importance > NRNERF_MAX_SAMPLES) return NRNERF_ERR_UNSUPPORTED;
    if (a->rank_new && a->n_samples + a->n_importance > 256) return NRNERF_ERR_UNSUPPORTED;       // 8-bit ranks (the split fine bender)
    if (a->n_rays == 0) return NRNERF_OK;
    if (!a->rays || a->ray_stride < 8 || !a->raw4 || !a->rgb || !a->disp || !a->acc) return NRNERF_ERR_INVALID;
    if (a->n_importance > 0 && !a->z_merged) return NRNERF_ERR_INVALID;
    if ((a->z_new != nullptr) != (a->rank_new != nullptr)) return NRNERF_ERR_INVALID;
    int dev = 0;
    int rc = composite_device(a, &dev);
    if (rc != NRNERF_OK) return rc;
    DeviceGuard guard(dev);
    if (!guard.ok) return NRNERF_ERR_HIP;
    CompositeArgs c{};
    c.rays = a->rays; c.ray_stride = a->ray_stride; c.raw4 = a->raw4; c.z = a->z; c.lindisp = a->lindisp;
    c.white_bkgd = a->white_bkgd; c.noise = a->noise; c.u = a->u; c.n_rays = a->n_rays; c.S = a->n_samples;
    c.n_importance = a->n_importance; c.rgb = a->rgb; c.disp = a->disp; c.acc = a->acc; c.z_std = a->z_std;
    c.z_out = a->z_merged; c.vis = a->weights; c.alpha = a->alpha;
    if (a->n_importance > 0) { c.z_new = a->z_new; c.rank_new = a->rank_new; }
    return launch_composite(c, (hipStream_t)hip_stream) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

int nrnerf_composite_backward(const nrnerf_composite_args* a, void* hip_stream) try {
    if (!a || a->struct_size != sizeof(nrnerf_composite_args)) return NRNERF_ERR_INVALID;
    if (a->n_rays < 0 || a->n_samples < 2 || a->n_samples > NRNERF_MAX_SAMPLES) return NRNERF_ERR_INVALID;
    if (a->n_rays == 0) return NRNERF_OK;
    if (!a->rays || a->ray_stride < 8 || !a->raw4 || !a->g_rgb || !a->d_raw4) return NRNERF_ERR_INVALID;
    int dev = 0;
    int rc = composite_device(a, &dev);
    if (rc != NRNERF_OK) return rc;
    DeviceGuard guard(dev);
    if (!guard.ok) return NRNERF_ERR_HIP;
    CompositeBwdArgs c{};
    c.rays = a->rays; c.ray_stride = a->ray_stride; c.raw4 = a->raw4; c.z = a->z; c.lindisp = a->lindisp;
    c.white_bkgd = a->white_bkgd; c.noise = a->noise; c.n_rays = a->n_rays; c.S = a->n_samples;
    c.g_rgb = a->g_rgb; c.g_disp = a->g_disp; c.g_acc = a->g_acc; c.g_w = a->g_weights; c.d_raw4 = a->d_raw4;
    return launch_composite_bwd(c, (hipStream_t)hip_stream) == hipSuccess ? NRNERF_OK : NRNERF_ERR_HIP;
} NRN_CATCH

int nrnerf_profile_begin(nrnerf_model* m) try {
    if (!m) return NRNERF_ERR_INVALID;
    std::lock_guard<std::mutex> g(m->prof_mu);
    for (auto& e : m->prof_events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    m->prof_events.clear();
    m->prof_on = true;
    return NRNERF_OK;
} NRN_CATCH

int nrnerf_profile_end(nrnerf_model* m, nrnerf_profile* out) try {
    if (!m || !out) return NRNERF_ERR_INVALID;
    std::lock_guard<std::mutex> g(m->prof_mu);
    m->prof_on = false;
    std::memset(out, 0, sizeof(*out));
    int rc = NRNERF_OK;
    for (auto& e : m->prof_events) {
        float ms = 0.f;
        if (hipEventSynchronize(e.b) != hipSuccess || hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess) rc = NRNERF_ERR_HIP;
        out->ms[e.kernel] += ms;
        out->launches[e.kernel] += 1;
        out->flops[e.kernel] += e.flops;
        out->mfma_flops[e.kernel] += e.mfma;
        if (e.name) std::snprintf(out->kernel_name[e.kernel], sizeof(out->kernel_name[e.kernel]), "%s", e.name);
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    m->prof_events.clear();
    return rc;
} NRN_CATCH

}  // extern "C"
