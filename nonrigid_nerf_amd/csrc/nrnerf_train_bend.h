// nrnerf_train_bend.h -- the ray bender for TRAINING (reference ray_bending.forward, run_nerf_helpers.py:507-577, under
// autograd): forward with saved activations and backward-data, both in exact fp32 on v_mfma_f32_32x32x2_f32.
//
// The deformation MLPs are 3 % of the model's flops, but as eager PyTorch ops their autograd graph streams
// [samples x 64] floats through ~100 elementwise / GEMM launches per step and bounds the training step at scale
// (DESIGN.md section 3.5).  Here they are two kernels built from the stand-alone bender's parts (weights resident in LDS,
// `dense_b`, in-register hand-off):
//   bend_fwd_train   bent point, rigidity mask, unmasked offsets; every hidden activation written to HBM in true feature
//                    order (as nrnerf_train.h does for the trunk);
//   bend_bwd         from the gradients of (bent point, unmasked offsets, rigidity mask): the gradient wrt every layer's
//                    pre-activation (stored for the library weight-gradient GEMMs) and wrt each sample's latent inputs
//                    (summed per ray by the caller); second plan PlanBB = the layers reversed with transposed weights;
//                    rigidity_network[0]^T and the xyz rows of network[0]^T are never formed (the sample positions carry
//                    no gradient: the reference detaches z_samples and the rays are data).
// Always fp32, whatever the trunk's precision: offsets feed a 2^9-frequency encoding.  These two are first order; the
// divergence regulariser (second order in autograd's terms) is bend_div_fwd / bend_div_bwd further down.
#pragma once
#include <type_traits>
#include "nrnerf_bend.h"
#include "nrnerf_train.h"

namespace nrn {

template <class PE, int T, class ACT>
__device__ __forceinline__ void pack_lin(const f32x16& acc, ACT& out) {      // D tile -> next B operand, no activation
    static_for<0, PE::SP>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        static_for<0, PE::KH>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            out.template set<T * PE::SP + u, e>(acc[u * PE::KH + e]);
        });
    });
}

// SA: element type of the saved arrays (PolF32: fp32, exact; PolBF16: bf16 -- see BendTrainArgs)
template <class A, class SA>
__global__ void __launch_bounds__(256, 2) bend_fwd_train(const BendTrainArgs a) {
    using P = PolF32;
    using PL = Plan<P, A, true, false, false>;
    constexpr int KH = P::KH, SP = P::SP, WAVES = 4;
    constexpr int NS_BIN = PL::NS_BIN, NS_RIN = PL::NS_RIN;
    constexpr int NB = PL::NT_BW * SP, NR = PL::NT_RW * SP;
    using ST = WResident<P, PL::NFRAGS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bias_lds = (float*)(smem + ST::BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    ST st;
    st.init(a.wstream, smem, tid, WAVES * 64, lane);
    for (int i = tid; i < PL::NTILES * 32; i += WAVES * 64) bias_lds[i] = a.bias[i];
    __syncthreads();
    const BiasPtr bias_lane = bias_lane_ptr(bias_lds, h);

    const int S = a.S;
    const int bpr = (S + 31) >> 5;
    const int nblocks = a.n_rays * bpr;
    const size_t M = (size_t)a.n_rays * S;
    for (int blk = (int)blockIdx.x * WAVES + wave; blk < nblocks; blk += (int)gridDim.x * WAVES) {
        const int ray = blk / bpr;
        const int sidx = (blk - ray * bpr) * 32 + j;
        const bool ok = sidx < S;
        const size_t so = (size_t)ray * S + (ok ? sidx : S - 1);
        const float* rp = a.rays + (size_t)ray * a.ray_stride;
        const float z = a.z[so];
        float p[3] = {__fadd_rn(rp[0], __fmul_rn(rp[3], z)), __fadd_rn(rp[1], __fmul_rn(rp[4], z)), __fadd_rn(rp[2], __fmul_rn(rp[5], z))};
        const float* lat = a.latents + (size_t)ray * a.lat_stride;
        auto binval = [&](auto idxc) -> float {
            constexpr int idx = decltype(idxc)::value;
            if constexpr (idx < 3) return p[idx];
            else if constexpr (idx < 8) return 0.0f;
            else if constexpr (idx - 8 < A::LAT) return lat[idx - 8];
            else return 0.0f;
        };
        Act<P, NS_BIN, false> bin;
        static_for<0, NS_BIN>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            const float v0 = binval(std::integral_constant<int, 2 * s>{}), v1 = binval(std::integral_constant<int, 2 * s + 1>{});
            bin.template set<s, 0>(h ? v1 : v0);
        });
        // hidden activation of layer `layer`: keep (true feature order) and hand on
        auto keep = [&](void* base, int width, auto lc, auto tc, const f32x16& acc, auto& out) {
            constexpr int layer = decltype(lc)::value, t = decltype(tc)::value;
            if (ok) {
                const size_t row = ((size_t)layer * M + so) * width + 32 * t + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    store4<SA>(base, row + 8 * q, relu_bits(acc[4 * q]), relu_bits(acc[4 * q + 1]), relu_bits(acc[4 * q + 2]), relu_bits(acc[4 * q + 3]));
            }
            pack_act<P, t>(acc, out);
        };
        Act<P, NB, false> ba, bb;
        float off[3];
        dense_b<P, false, PL, PL::L_BEND0, NS_BIN>(st, bias_lane, bin, [&](auto tc, const f32x16& acc) {
            keep(a.acts_b, A::BW, std::integral_constant<int, 0>{}, tc, acc, ba); });
        static_for<1, A::BD - 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i % 2 == 1) dense_b<P, false, PL, PL::L_BEND0 + i, NB>(st, bias_lane, ba, [&](auto tc, const f32x16& acc) { keep(a.acts_b, A::BW, ic, tc, acc, bb); });
            else dense_b<P, false, PL, PL::L_BEND0 + i, NB>(st, bias_lane, bb, [&](auto tc, const f32x16& acc) { keep(a.acts_b, A::BW, ic, tc, acc, ba); });
        });
        auto take_off = [&](auto, const f32x16& acc) { off[0] = acc[0]; off[1] = acc[1]; off[2] = acc[2]; };
        if constexpr ((A::BD - 2) % 2 == 1) dense_b<P, false, PL, PL::L_BEND0 + A::BD - 1, NB>(st, bias_lane, bb, take_off);
        else dense_b<P, false, PL, PL::L_BEND0 + A::BD - 1, NB>(st, bias_lane, ba, take_off);
        // rigidity MLP (input = xyz only)
        Act<P, NS_RIN, false> rin;
        static_for<0, NS_RIN>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            const float v0 = (2 * s < 3) ? p[2 * s < 3 ? 2 * s : 0] : 0.0f, v1 = (2 * s + 1 < 3) ? p[2 * s + 1 < 3 ? 2 * s + 1 : 0] : 0.0f;
            rin.template set<s, 0>(h ? v1 : v0);
        });
        Act<P, NR, false> ra, rb;
        float logit;
        dense_b<P, false, PL, PL::L_RIG0, NS_RIN>(st, bias_lane, rin, [&](auto tc, const f32x16& acc) {
            keep(a.acts_r, A::RW, std::integral_constant<int, 0>{}, tc, acc, ra); });
        static_for<1, A::RD - 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i % 2 == 1) dense_b<P, false, PL, PL::L_RIG0 + i, NR>(st, bias_lane, ra, [&](auto tc, const f32x16& acc) { keep(a.acts_r, A::RW, ic, tc, acc, rb); });
            else dense_b<P, false, PL, PL::L_RIG0 + i, NR>(st, bias_lane, rb, [&](auto tc, const f32x16& acc) { keep(a.acts_r, A::RW, ic, tc, acc, ra); });
        });
        auto take_logit = [&](auto, const f32x16& acc) { logit = acc[0]; };
        if constexpr ((A::RD - 2) % 2 == 1) dense_b<P, false, PL, PL::L_RIG0 + A::RD - 1, NR>(st, bias_lane, rb, take_logit);
        else dense_b<P, false, PL, PL::L_RIG0 + A::RD - 1, NR>(st, bias_lane, ra, take_logit);

        const float th = tanhf(logit);
        float mask = (th + 1.0f) / 2.0f;                                                   // rnh:559-561
        if (a.knobs.has_cutoff && mask <= a.knobs.cutoff) mask = 0.0f;                     // rnh:563-564
        float bent[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float mo = __fmul_rn(mask, off[c]);                                            // rnh:567
            if (a.knobs.has_scaling) mo = __fmul_rn(mo, a.knobs.scaling);                  // rnh:568-569
            bent[c] = __fadd_rn(p[c], mo);                                                 // rnh:570
        }
        if (ok && h == 0) {
            *(f32x4*)(a.bent4 + so * 4) = f32x4{bent[0], bent[1], bent[2], mask};
            *(f32x4*)(a.off4 + so * 4) = f32x4{off[0], off[1], off[2], th};
        }
    }
}

template <class A, class SA>
__global__ void __launch_bounds__(256, 2) bend_bwd(const BendTrainArgs a) {
    using P = PolF32;
    using PL = PlanBB<P, A>;
    constexpr int KH = P::KH, SP = P::SP, WAVES = 4;
    constexpr int NS_DR = PL::NS_DR, NB = PL::NT_BW * SP, NR = PL::NT_RW * SP;
    using ST = WResident<P, PL::NFRAGS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bias_lds = (float*)(smem + ST::BYTES);          // zero: backward layers have no bias
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    ST st;
    st.init(a.wstream, smem, tid, WAVES * 64, lane);
    for (int i = tid; i < PL::NTILES * 32; i += WAVES * 64) bias_lds[i] = 0.0f;
    __syncthreads();
    const BiasPtr bias_lane = bias_lane_ptr(bias_lds, h);

    const int S = a.S;
    const int bpr = (S + 31) >> 5;
    const int nblocks = a.n_rays * bpr;
    const size_t M = (size_t)a.n_rays * S;
    for (int blk = (int)blockIdx.x * WAVES + wave; blk < nblocks; blk += (int)gridDim.x * WAVES) {
        const int ray = blk / bpr;
        const int sidx = (blk - ray * bpr) * 32 + j;
        const bool ok = sidx < S;
        const size_t so = (size_t)ray * S + (ok ? sidx : S - 1);
        // bent = p + s * mask * off (rnh:567-570): gradients of its three differentiable outputs -> d off, d logit
        f32x4 gb = *(const f32x4*)(a.g_bent4 + so * 4);
        if (a.g_bent4_b) gb += *(const f32x4*)(a.g_bent4_b + so * 4);
        const f32x4 bm = *(const f32x4*)(a.bent4 + so * 4);        // .w = mask after the cutoff knob
        const f32x4 ot = *(const f32x4*)(a.off4 + so * 4);         // offsets xyz, tanh(logit)
        const float sc = a.knobs.has_scaling ? a.knobs.scaling : 1.0f;
        const float mask = bm[3], th = ot[3];
        float g_off[3], g_mask = a.g_mask ? a.g_mask[so] : 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g_off[c] = gb[c] * mask * sc + (a.g_unmasked ? a.g_unmasked[so * 3 + c] : 0.0f);
            g_mask += gb[c] * ot[c] * sc;
        }
        const bool cut = a.knobs.has_cutoff && (th + 1.0f) / 2.0f <= a.knobs.cutoff;       // the assignment mask[...] = 0 has no gradient
        float g_logit = cut ? 0.0f : g_mask * 0.5f * (1.0f - th * th);
        if (!ok) { g_off[0] = g_off[1] = g_off[2] = 0.0f; g_logit = 0.0f; }
        if (ok && h == 0) *(f32x4*)(a.dz_out4 + so * 4) = f32x4{g_off[0], g_off[1], g_off[2], g_logit};

        // d h_layer tile t -> d z_layer (mask with the saved activation), stored for the weight gradients, handed on
        // the saved activations of a layer (its relu mask), requested BEFORE the transposed layer whose epilogue applies them:
        // loaded inside the epilogue, every tile waited a memory latency for them (same finding as trunk_bwd's masks)
        f32x4 hv[2][4];                                    // [tile][q]: features 32 t + 8 q + 4 h .. + 3 of this lane's sample
        auto fetch_acts = [&](const void* acts, int width, auto lc, auto ntc) {
            constexpr int layer = decltype(lc)::value, nt = decltype(ntc)::value;
#pragma unroll
            for (int t = 0; t < nt; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) hv[t][q] = load4<SA>(acts, ((size_t)layer * M + so) * width + 32 * t + 4 * h + 8 * q);
        };
        auto mask_store = [&](const void*, void* dz, int width, auto lc, auto tc, const f32x16& acc, auto& out) {
            constexpr int layer = decltype(lc)::value, t = decltype(tc)::value;
            const size_t row = ((size_t)layer * M + so) * width + 32 * t + 4 * h;
            f32x16 g = acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int k = 0; k < 4; ++k) g[4 * q + k] = (hv[t][q][k] > 0.0f) ? acc[4 * q + k] : 0.0f;
                if (ok) store4<SA>(dz, row + 8 * q, g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]);
            }
            pack_lin<P, t>(g, out);
        };
        // ---- offset MLP: network[BD-1]^T .. network[0]^T
        Act<P, NS_DR, false> dr;
        static_for<0, NS_DR>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            const float v0 = (2 * s < 3) ? g_off[2 * s < 3 ? 2 * s : 0] : 0.0f, v1 = (2 * s + 1 < 3) ? g_off[2 * s + 1 < 3 ? 2 * s + 1 : 0] : 0.0f;
            dr.template set<s, 0>(h ? v1 : v0);
        });
        Act<P, NB, false> ba, bb;
        constexpr auto NTB = std::integral_constant<int, PL::NT_BW>{};
        constexpr auto NTR = std::integral_constant<int, PL::NT_RW>{};
        static_assert(PL::NT_BW <= 2 && PL::NT_RW <= 2, "hv holds two tiles");
        fetch_acts(a.acts_b, A::BW, std::integral_constant<int, A::BD - 2>{}, NTB);
        dense_b<P, false, PL, PL::L_BEND(A::BD - 1), NS_DR>(st, bias_lane, dr, [&](auto tc, const f32x16& acc) {
            mask_store(a.acts_b, a.dz_b, A::BW, std::integral_constant<int, A::BD - 2>{}, tc, acc, ba); });
        static_for<0, A::BD - 2>([&](auto kc) {
            constexpr int k = decltype(kc)::value;             // 0 .. BD-3
            constexpr int i = A::BD - 2 - k;                   // network[i]^T: d z_i -> d h_{i-1}
            auto run = [&](auto& src, auto& dst) {
                fetch_acts(a.acts_b, A::BW, std::integral_constant<int, i - 1>{}, NTB);
                dense_b<P, false, PL, PL::L_BEND(i), NB>(st, bias_lane, src, [&](auto tc, const f32x16& acc) {
                    mask_store(a.acts_b, a.dz_b, A::BW, std::integral_constant<int, i - 1>{}, tc, acc, dst); });
            };
            if constexpr (k % 2 == 0) run(ba, bb); else run(bb, ba);
        });
        // network[0]^T, latent rows only: gradient wrt this sample's latent inputs
        auto take_lat = [&](auto tc, const f32x16& acc) {
            constexpr int t = decltype(tc)::value;
            if (ok) {
                const size_t row = so * A::LAT + 32 * t + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (32 * t + 8 * q + 4 * h + 3 < A::LAT) store4<P>(a.d_lat, row + 8 * q, acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            }
        };
        if constexpr ((A::BD - 2) % 2 == 0) dense_b<P, false, PL, PL::L_BEND(0), NB>(st, bias_lane, ba, take_lat);
        else dense_b<P, false, PL, PL::L_BEND(0), NB>(st, bias_lane, bb, take_lat);
        // ---- rigidity MLP: rigidity_network[RD-1]^T .. rigidity_network[1]^T
        Act<P, NS_DR, false> drr;
        static_for<0, NS_DR>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            drr.template set<s, 0>((s == 0 && h == 0) ? g_logit : 0.0f);
        });
        Act<P, NR, false> ra, rb;
        fetch_acts(a.acts_r, A::RW, std::integral_constant<int, A::RD - 2>{}, NTR);
        dense_b<P, false, PL, PL::L_RIG(A::RD - 1), NS_DR>(st, bias_lane, drr, [&](auto tc, const f32x16& acc) {
            mask_store(a.acts_r, a.dz_r, A::RW, std::integral_constant<int, A::RD - 2>{}, tc, acc, ra); });
        static_for<0, A::RD - 2>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int i = A::RD - 2 - k;                   // rigidity_network[i]^T, i >= 1
            auto run = [&](auto& src, auto& dst) {
                fetch_acts(a.acts_r, A::RW, std::integral_constant<int, i - 1>{}, NTR);
                dense_b<P, false, PL, PL::L_RIG(i), NR>(st, bias_lane, src, [&](auto tc, const f32x16& acc) {
                    mask_store(a.acts_r, a.dz_r, A::RW, std::integral_constant<int, i - 1>{}, tc, acc, dst); });
            };
            if constexpr (k % 2 == 0) run(ra, rb); else run(rb, ra);
        });
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Divergence regulariser (reference compute_divergence_loss / divergence_approx, run_nerf_helpers.py:22-116;
// train.py:244-287): per point  d = e^T J e,  J = d(masked offsets)/d(point).  The reference forms  J^T e  by a
// vector-Jacobian product with create_graph=True and back-propagates through that graph (double backward through both
// MLPs).  Here d is ONE forward-mode tangent through the MLPs (dense_b's tangent operand, as in the exact-view-direction
// kernels): with z_i = W_i h_{i-1} + b_i, h_i = relu(z_i),
//     tz_i = W_i th_{i-1},   th_i = [z_i > 0] tz_i,   th_0 = (e, 0),
//     m = (tanh(r) + 1)/2,   tm = (1 - tanh(r)^2)/2 * tr,          (0 and 0 below the cutoff knob)
//     d = s * e . (tm * off + m * toff)
// and its backward pass is two chains that share weights and relu masks (the derivative of the mask itself is zero, as
// in autograd's double backward of relu):
//     g_off = g s tm e,  g_toff = g s m e,  g_m = g s (e . toff),  g_tm = g s (e . off)
//     g_r = g_m (1 - t^2)/2 - g_tm tr t (1 - t^2),   g_tr = g_tm (1 - t^2)/2          (t = tanh(r))
//     dz_i = [z_i > 0] W_{i+1}^T dz_{i+1},   dtz_i = [z_i > 0] W_{i+1}^T dtz_{i+1}
//     dW_i = dz_i^T h_{i-1} + dtz_i^T th_{i-1},   db_i = sum dz_i,   d latent = (latent rows of W_0)^T dz_0
// Points are independent here (no ray structure): a block is 32 consecutive points, every point has its own latent row.
// ------------------------------------------------------------------------------------------------------------------
template <class A, class SA>
__global__ void __launch_bounds__(256, 2) bend_div_fwd(const BendDivArgs a) {
    using P = PolF32;
    using PL = Plan<P, A, true, false, false>;
    constexpr int SP = P::SP, WAVES = 4;
    constexpr int NS_BIN = PL::NS_BIN, NS_RIN = PL::NS_RIN;
    constexpr int NB = PL::NT_BW * SP, NR = PL::NT_RW * SP;
    using ST = WResident<P, PL::NFRAGS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bias_lds = (float*)(smem + ST::BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    ST st;
    st.init(a.wstream, smem, tid, WAVES * 64, lane);
    for (int i = tid; i < PL::NTILES * 32; i += WAVES * 64) bias_lds[i] = a.bias[i];
    __syncthreads();
    const BiasPtr bias_lane = bias_lane_ptr(bias_lds, h);

    const size_t M = (size_t)a.m;
    const long long nblocks = (a.m + 31) >> 5;
    for (long long blk = (long long)blockIdx.x * WAVES + wave; blk < nblocks; blk += (long long)gridDim.x * WAVES) {
        const long long sidx = blk * 32 + j;
        const bool ok = sidx < a.m;
        const size_t so = ok ? (size_t)sidx : M - 1;
        float p[3], ev[3];
        const float* lat;
        if (a.rays) {                   // ray mode (BendDivArgs): the sample's point and the ray's unit direction, computed here
            const size_t ray = so / (size_t)a.S;
            const int k = (int)(so - ray * (size_t)a.S);
            const float* rp = a.rays + ray * (size_t)a.ray_stride;
            float z;
            if (a.zr) z = a.zr[so];
            else {                      // train.py:847-852, as the network kernels
                const float t = lin01(k, a.S), near = rp[6], far = rp[7];
                if (a.lindisp) z = __fdiv_rn(1.0f, __fadd_rn(__fmul_rn(__fdiv_rn(1.0f, near), __fsub_rn(1.0f, t)), __fmul_rn(__fdiv_rn(1.0f, far), t)));
                else z = __fadd_rn(__fmul_rn(near, __fsub_rn(1.0f, t)), __fmul_rn(far, t));
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) { p[c] = __fadd_rn(rp[c], __fmul_rn(rp[3 + c], z)); ev[c] = rp[8 + c]; }       // train.py:871-873; 380-381
            lat = a.latents + ray * (size_t)a.lat_stride;
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) { p[c] = a.pts[so * 3 + c]; ev[c] = a.e[so * 3 + c]; }
            lat = a.latents + so * (size_t)a.lat_stride;
        }
        auto binval = [&](auto idxc) -> float {
            constexpr int idx = decltype(idxc)::value;
            if constexpr (idx < 3) return p[idx];
            else if constexpr (idx < 8) return 0.0f;
            else if constexpr (idx - 8 < A::LAT) return lat[idx - 8];
            else return 0.0f;
        };
        auto tanval = [&](auto idxc) -> float {            // tangent of the first layers' inputs: (e, 0, ...)
            constexpr int idx = decltype(idxc)::value;
            if constexpr (idx < 3) return ev[idx];
            else return 0.0f;
        };
        Act<P, NS_BIN, false> bin;
        Tan<P, NS_BIN> tbin;
        static_for<0, NS_BIN>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            const float v0 = binval(std::integral_constant<int, 2 * s>{}), v1 = binval(std::integral_constant<int, 2 * s + 1>{});
            bin.template set<s, 0>(h ? v1 : v0);
            const float t0 = tanval(std::integral_constant<int, 2 * s>{}), t1 = tanval(std::integral_constant<int, 2 * s + 1>{});
            tbin.template set<s, 0>(h ? t1 : t0);
        });
        // hidden layer `layer`: keep value and tangent (true feature order) and hand both on
        auto keep = [&](void* base, void* tbase, int width, auto lc, auto tc, const f32x16& acc, const f32x16& tacc, auto& out, auto& tout) {
            constexpr int layer = decltype(lc)::value, t = decltype(tc)::value;
            if (ok && base) {           // (ray mode: nothing is kept)
                const size_t row = ((size_t)layer * M + so) * width + 32 * t + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    store4<SA>(base, row + 8 * q, relu_bits(acc[4 * q]), relu_bits(acc[4 * q + 1]), relu_bits(acc[4 * q + 2]), relu_bits(acc[4 * q + 3]));
                    store4<SA>(tbase, row + 8 * q, acc[4 * q] > 0.0f ? tacc[4 * q] : 0.0f, acc[4 * q + 1] > 0.0f ? tacc[4 * q + 1] : 0.0f,
                              acc[4 * q + 2] > 0.0f ? tacc[4 * q + 2] : 0.0f, acc[4 * q + 3] > 0.0f ? tacc[4 * q + 3] : 0.0f);
                }
            }
            pack_act<P, t>(acc, out);
            pack_tan<P, t>(acc, tacc, tout);
        };
        Act<P, NB, false> ba, bb;
        Tan<P, NB> ta, tb;
        float off[3], toff[3];
        dense_b<P, false, PL, PL::L_BEND0, NS_BIN>(st, bias_lane, bin, [&](auto tc, const f32x16& acc, const f32x16& tacc) {
            keep(a.acts_b, a.tacts_b, A::BW, std::integral_constant<int, 0>{}, tc, acc, tacc, ba, ta); }, tbin);
        static_for<1, A::BD - 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i % 2 == 1)
                dense_b<P, false, PL, PL::L_BEND0 + i, NB>(st, bias_lane, ba, [&](auto tc, const f32x16& acc, const f32x16& tacc) {
                    keep(a.acts_b, a.tacts_b, A::BW, ic, tc, acc, tacc, bb, tb); }, ta);
            else
                dense_b<P, false, PL, PL::L_BEND0 + i, NB>(st, bias_lane, bb, [&](auto tc, const f32x16& acc, const f32x16& tacc) {
                    keep(a.acts_b, a.tacts_b, A::BW, ic, tc, acc, tacc, ba, ta); }, tb);
        });
        auto take_off = [&](auto, const f32x16& acc, const f32x16& tacc) {
            off[0] = acc[0]; off[1] = acc[1]; off[2] = acc[2];
            toff[0] = tacc[0]; toff[1] = tacc[1]; toff[2] = tacc[2];
        };
        if constexpr ((A::BD - 2) % 2 == 1) dense_b<P, false, PL, PL::L_BEND0 + A::BD - 1, NB>(st, bias_lane, bb, take_off, tb);
        else dense_b<P, false, PL, PL::L_BEND0 + A::BD - 1, NB>(st, bias_lane, ba, take_off, ta);
        // rigidity MLP (input = xyz only)
        Act<P, NS_RIN, false> rin;
        Tan<P, NS_RIN> trin;
        static_for<0, NS_RIN>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            const float v0 = (2 * s < 3) ? p[2 * s < 3 ? 2 * s : 0] : 0.0f, v1 = (2 * s + 1 < 3) ? p[2 * s + 1 < 3 ? 2 * s + 1 : 0] : 0.0f;
            rin.template set<s, 0>(h ? v1 : v0);
            const float t0 = (2 * s < 3) ? ev[2 * s < 3 ? 2 * s : 0] : 0.0f, t1 = (2 * s + 1 < 3) ? ev[2 * s + 1 < 3 ? 2 * s + 1 : 0] : 0.0f;
            trin.template set<s, 0>(h ? t1 : t0);
        });
        Act<P, NR, false> ra, rb;
        Tan<P, NR> tra, trb;
        float logit, tlogit;
        dense_b<P, false, PL, PL::L_RIG0, NS_RIN>(st, bias_lane, rin, [&](auto tc, const f32x16& acc, const f32x16& tacc) {
            keep(a.acts_r, a.tacts_r, A::RW, std::integral_constant<int, 0>{}, tc, acc, tacc, ra, tra); }, trin);
        static_for<1, A::RD - 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i % 2 == 1)
                dense_b<P, false, PL, PL::L_RIG0 + i, NR>(st, bias_lane, ra, [&](auto tc, const f32x16& acc, const f32x16& tacc) {
                    keep(a.acts_r, a.tacts_r, A::RW, ic, tc, acc, tacc, rb, trb); }, tra);
            else
                dense_b<P, false, PL, PL::L_RIG0 + i, NR>(st, bias_lane, rb, [&](auto tc, const f32x16& acc, const f32x16& tacc) {
                    keep(a.acts_r, a.tacts_r, A::RW, ic, tc, acc, tacc, ra, tra); }, trb);
        });
        auto take_logit = [&](auto, const f32x16& acc, const f32x16& tacc) { logit = acc[0]; tlogit = tacc[0]; };
        if constexpr ((A::RD - 2) % 2 == 1) dense_b<P, false, PL, PL::L_RIG0 + A::RD - 1, NR>(st, bias_lane, rb, take_logit, trb);
        else dense_b<P, false, PL, PL::L_RIG0 + A::RD - 1, NR>(st, bias_lane, ra, take_logit, tra);

        const float th = tanhf(logit);
        float mask = (th + 1.0f) / 2.0f;                                                   // rnh:559-561
        float tmask = 0.5f * (1.0f - th * th) * tlogit;
        if (a.knobs.has_cutoff && mask <= a.knobs.cutoff) { mask = 0.0f; tmask = 0.0f; }   // rnh:563-564: the assignment has no derivative
        float d = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) d += ev[c] * (tmask * off[c] + mask * toff[c]);       // e . d(mask * off)/dt   (rnh:567)
        if (a.knobs.has_scaling) d *= a.knobs.scaling;                                     // rnh:568-569
        if (ok && h == 0) {
            if (a.tvec) {               // the tangent itself: d(masked offsets)/dp . e  (J . e = e + this for the bent point)
                const float sc = a.knobs.has_scaling ? a.knobs.scaling : 1.0f;
                float* tv = a.tvec + so * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) tv[c] = sc * (tmask * off[c] + mask * toff[c]);
            }
            if (a.dirs_out) {           // NeRF.exact_nonrigid_viewdirs (rnh:367-378): J d = d + d(masked offsets)/dp . d, normalised, eps OUTSIDE the division
                const float sc = a.knobs.has_scaling ? a.knobs.scaling : 1.0f;
                float jd[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) jd[c] = __fadd_rn(ev[c], __fmul_rn(sc, __fadd_rn(__fmul_rn(tmask, off[c]), __fmul_rn(mask, toff[c]))));
                const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(jd[0], jd[0]), __fmul_rn(jd[1], jd[1])), __fmul_rn(jd[2], jd[2])));
#pragma unroll
                for (int c = 0; c < 3; ++c) a.dirs_out[so * 3 + c] = __fadd_rn(__fdiv_rn(jd[c], nrm), 0.000001f);
            }
            if (a.div) a.div[so] = d;
            if (a.bent4) {              // bend_fwd_train's own output (rnh:567-570): bent = p + scaling * mask * off, .w = the mask after the cutoff
                float bent[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float mo = __fmul_rn(mask, off[c]);
                    if (a.knobs.has_scaling) mo = __fmul_rn(mo, a.knobs.scaling);
                    bent[c] = __fadd_rn(p[c], mo);
                }
                *(f32x4*)(a.bent4 + so * 4) = f32x4{bent[0], bent[1], bent[2], mask};
            }
            if (a.off4) *(f32x4*)(a.off4 + so * 4) = f32x4{off[0], off[1], off[2], th};
            if (a.toff4) *(f32x4*)(a.toff4 + so * 4) = f32x4{toff[0], toff[1], toff[2], tlogit};
        }
    }
}

template <class A, class SA>
__global__ void __launch_bounds__(256, 2) bend_div_bwd(const BendDivArgs a) {
    using P = PolF32;
    using PL = PlanBB<P, A>;
    constexpr int SP = P::SP, WAVES = 4;
    constexpr int NS_DR = PL::NS_DR, NB = PL::NT_BW * SP, NR = PL::NT_RW * SP;
    using ST = WResident<P, PL::NFRAGS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bias_lds = (float*)(smem + ST::BYTES);          // zero: backward layers have no bias
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    ST st;
    st.init(a.wstream, smem, tid, WAVES * 64, lane);
    for (int i = tid; i < PL::NTILES * 32; i += WAVES * 64) bias_lds[i] = 0.0f;
    __syncthreads();
    const BiasPtr bias_lane = bias_lane_ptr(bias_lds, h);

    const size_t M = (size_t)a.m;
    const long long nblocks = (a.m + 31) >> 5;
    for (long long blk = (long long)blockIdx.x * WAVES + wave; blk < nblocks; blk += (long long)gridDim.x * WAVES) {
        const long long sidx = blk * 32 + j;
        const bool ok = sidx < a.m;
        const size_t so = ok ? (size_t)sidx : M - 1;
        const float ev[3] = {a.e[so * 3], a.e[so * 3 + 1], a.e[so * 3 + 2]};
        const f32x4 ot = *(const f32x4*)(a.off4 + so * 4);         // offsets xyz, tanh(logit)
        const f32x4 tt = *(const f32x4*)(a.toff4 + so * 4);        // their tangents; .w = tangent of the logit
        const float sc = a.knobs.has_scaling ? a.knobs.scaling : 1.0f;
        const float th = ot[3], tlogit = tt[3];
        const float s2 = 0.5f * (1.0f - th * th);
        const bool cut = a.knobs.has_cutoff && (th + 1.0f) / 2.0f <= a.knobs.cutoff;
        const float mask = cut ? 0.0f : (th + 1.0f) / 2.0f, tmask = cut ? 0.0f : s2 * tlogit;
        // cotangent of the masked tangent vector sc (tmask off + mask toff): the divergence e . (that) hands down g_div e; a caller
        // that consumes the vector itself (exact view directions, rnh:358-385) hands in its own
        float gv[3];
        if (a.g_tvec) {
#pragma unroll
            for (int c = 0; c < 3; ++c) gv[c] = ok ? a.g_tvec[so * 3 + c] * sc : 0.0f;
        } else {
            const float g = ok ? a.g_div[so] * sc : 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) gv[c] = g * ev[c];
        }
        float g_off[3], g_toff[3], g_m = 0.0f, g_tm = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g_off[c] = gv[c] * tmask;
            g_toff[c] = gv[c] * mask;
            g_m += gv[c] * tt[c];
            g_tm += gv[c] * ot[c];
        }
        if (a.r_g_bent4 || a.r_g_unmasked || a.r_g_mask) {
            // the render pass' cotangents for the same evaluation (bend_bwd's own first lines): bent = p + sc mask off (rnh:567-570)
            f32x4 gb = a.r_g_bent4 ? *(const f32x4*)(a.r_g_bent4 + so * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (a.r_g_bent4_b) gb += *(const f32x4*)(a.r_g_bent4_b + so * 4);
            float rm = a.r_g_mask ? a.r_g_mask[so] : 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float go = gb[c] * mask * sc + (a.r_g_unmasked ? a.r_g_unmasked[so * 3 + c] : 0.0f);
                g_off[c] += ok ? go : 0.0f;
                rm += gb[c] * ot[c] * sc;
            }
            g_m += ok ? rm : 0.0f;
        }
        const float g_logit = cut ? 0.0f : g_m * s2 - g_tm * tlogit * th * (1.0f - th * th);
        const float g_tlogit = cut ? 0.0f : g_tm * s2;
        if (ok && h == 0) {
            *(f32x4*)(a.dz_out4 + so * 4) = f32x4{g_off[0], g_off[1], g_off[2], g_logit};
            *(f32x4*)(a.dtz_out4 + so * 4) = f32x4{g_toff[0], g_toff[1], g_toff[2], g_tlogit};
        }
        // (d h, d th) of a hidden layer's tile -> (d z, d tz): both masked with the saved activation, stored, handed on
        f32x4 hv[2][4];                                    // the layer's saved activations, requested before its transposed layer (see bend_bwd)
        auto fetch_acts = [&](const void* acts, int width, auto lc, auto ntc) {
            constexpr int layer = decltype(lc)::value, nt = decltype(ntc)::value;
#pragma unroll
            for (int t = 0; t < nt; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) hv[t][q] = load4<SA>(acts, ((size_t)layer * M + so) * width + 32 * t + 4 * h + 8 * q);
        };
        auto mask_store = [&](const void*, void* dz, void* dtz, int width, auto lc, auto tc, const f32x16& acc, const f32x16& tacc,
                              auto& out, auto& tout) {
            constexpr int layer = decltype(lc)::value, t = decltype(tc)::value;
            const size_t row = ((size_t)layer * M + so) * width + 32 * t + 4 * h;
            f32x16 gv = acc, gt = tacc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    gv[4 * q + k] = (hv[t][q][k] > 0.0f) ? acc[4 * q + k] : 0.0f;
                    gt[4 * q + k] = (hv[t][q][k] > 0.0f) ? tacc[4 * q + k] : 0.0f;
                }
                if (ok) {
                    store4<SA>(dz, row + 8 * q, gv[4 * q], gv[4 * q + 1], gv[4 * q + 2], gv[4 * q + 3]);
                    store4<SA>(dtz, row + 8 * q, gt[4 * q], gt[4 * q + 1], gt[4 * q + 2], gt[4 * q + 3]);
                }
            }
            pack_lin<P, t>(gv, out);
            pack_lin<P, t>(gt, tout);
        };
        // ---- offset MLP: network[BD-1]^T .. network[0]^T
        Act<P, NS_DR, false> dr;
        Tan<P, NS_DR> tdr;
        static_for<0, NS_DR>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            const float v0 = (2 * s < 3) ? g_off[2 * s < 3 ? 2 * s : 0] : 0.0f, v1 = (2 * s + 1 < 3) ? g_off[2 * s + 1 < 3 ? 2 * s + 1 : 0] : 0.0f;
            dr.template set<s, 0>(h ? v1 : v0);
            const float t0 = (2 * s < 3) ? g_toff[2 * s < 3 ? 2 * s : 0] : 0.0f, t1 = (2 * s + 1 < 3) ? g_toff[2 * s + 1 < 3 ? 2 * s + 1 : 0] : 0.0f;
            tdr.template set<s, 0>(h ? t1 : t0);
        });
        Act<P, NB, false> ba, bb;
        Tan<P, NB> ta, tb;
        constexpr auto NTB = std::integral_constant<int, PL::NT_BW>{};
        constexpr auto NTR = std::integral_constant<int, PL::NT_RW>{};
        static_assert(PL::NT_BW <= 2 && PL::NT_RW <= 2, "hv holds two tiles");
        fetch_acts(a.acts_b, A::BW, std::integral_constant<int, A::BD - 2>{}, NTB);
        dense_b<P, false, PL, PL::L_BEND(A::BD - 1), NS_DR>(st, bias_lane, dr, [&](auto tc, const f32x16& acc, const f32x16& tacc) {
            mask_store(a.acts_b, a.dz_b, a.dtz_b, A::BW, std::integral_constant<int, A::BD - 2>{}, tc, acc, tacc, ba, ta); }, tdr);
        static_for<0, A::BD - 2>([&](auto kc) {
            constexpr int k = decltype(kc)::value;             // 0 .. BD-3
            constexpr int i = A::BD - 2 - k;                   // network[i]^T: (d z_i, d tz_i) -> (d h_{i-1}, d th_{i-1})
            auto run = [&](auto& src, auto& tsrc, auto& dst, auto& tdst) {
                fetch_acts(a.acts_b, A::BW, std::integral_constant<int, i - 1>{}, NTB);
                dense_b<P, false, PL, PL::L_BEND(i), NB>(st, bias_lane, src, [&](auto tc, const f32x16& acc, const f32x16& tacc) {
                    mask_store(a.acts_b, a.dz_b, a.dtz_b, A::BW, std::integral_constant<int, i - 1>{}, tc, acc, tacc, dst, tdst); }, tsrc);
            };
            if constexpr (k % 2 == 0) run(ba, ta, bb, tb); else run(bb, tb, ba, ta);
        });
        // network[0]^T, latent rows, value chain only (the tangent of the latent inputs is zero: they are not differentiated
        // with respect to position)
        auto take_lat = [&](auto tc, const f32x16& acc) {
            constexpr int t = decltype(tc)::value;
            if (ok) {
                const size_t row = so * A::LAT + 32 * t + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (32 * t + 8 * q + 4 * h + 3 < A::LAT) store4<P>(a.d_lat, row + 8 * q, acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            }
        };
        if constexpr ((A::BD - 2) % 2 == 0) dense_b<P, false, PL, PL::L_BEND(0), NB>(st, bias_lane, ba, take_lat);
        else dense_b<P, false, PL, PL::L_BEND(0), NB>(st, bias_lane, bb, take_lat);
        // ---- rigidity MLP: rigidity_network[RD-1]^T .. rigidity_network[1]^T
        Act<P, NS_DR, false> drr;
        Tan<P, NS_DR> tdrr;
        static_for<0, NS_DR>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            drr.template set<s, 0>((s == 0 && h == 0) ? g_logit : 0.0f);
            tdrr.template set<s, 0>((s == 0 && h == 0) ? g_tlogit : 0.0f);
        });
        Act<P, NR, false> ra, rb;
        Tan<P, NR> tra, trb;
        fetch_acts(a.acts_r, A::RW, std::integral_constant<int, A::RD - 2>{}, NTR);
        dense_b<P, false, PL, PL::L_RIG(A::RD - 1), NS_DR>(st, bias_lane, drr, [&](auto tc, const f32x16& acc, const f32x16& tacc) {
            mask_store(a.acts_r, a.dz_r, a.dtz_r, A::RW, std::integral_constant<int, A::RD - 2>{}, tc, acc, tacc, ra, tra); }, tdrr);
        static_for<0, A::RD - 2>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int i = A::RD - 2 - k;                   // rigidity_network[i]^T, i >= 1
            auto run = [&](auto& src, auto& tsrc, auto& dst, auto& tdst) {
                fetch_acts(a.acts_r, A::RW, std::integral_constant<int, i - 1>{}, NTR);
                dense_b<P, false, PL, PL::L_RIG(i), NR>(st, bias_lane, src, [&](auto tc, const f32x16& acc, const f32x16& tacc) {
                    mask_store(a.acts_r, a.dz_r, a.dtz_r, A::RW, std::integral_constant<int, i - 1>{}, tc, acc, tacc, dst, tdst); }, tsrc);
            };
            if constexpr (k % 2 == 0) run(ra, tra, rb, trb); else run(rb, trb, ra, tra);
        });
    }
}

// Weight and bias gradients of both MLPs in one launch: dW = dz^T x over the samples for a list of (dz, x) pairs of
// row-major fp32 arrays (at most 64 x 64 each), on v_mfma_f32_32x32x2_f32 with the SAMPLE as contraction index: the A
// operand of a k-step is dz[sample s + h][32 tr + i], the B operand x[sample s + h][32 tc + j] -- one coalesced 128-byte
// row segment per lane half, no transposes.  Every wave owns a contiguous range of samples and writes one fp32 partial
// (dW [64][64] and the row sums of dz = db [64]) per job; the caller adds the partials.
template <int U>        // k-steps per batch of loads (a template only so that the header can be included by several units)
__global__ void __launch_bounds__(256) bend_wgrad(const BendWgradArgs a) {
    const BendWgradJob jb = a.job[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, i = lane & 31;
    const int part = (int)blockIdx.x * 4 + wave;
    const long long pairs = (a.m + 1) / 2;
    const long long per = (pairs + a.nparts - 1) / a.nparts;
    const long long p0 = (long long)part * per, p1 = (p0 + per < pairs) ? p0 + per : pairs;
    const bool f1 = jb.f > 32, g1 = jb.g > 32;           // second row / column tile in use (wave-uniform)
    const bool fa0 = i < jb.f, fa1 = 32 + i < jb.f, gb0 = i < jb.g, gb1 = 32 + i < jb.g;
    f32x16 acc[2][2] = {{f32x16{}, f32x16{}}, {f32x16{}, f32x16{}}};
    float bsum[2] = {0.0f, 0.0f};
    for (int pass = 0; pass < (jb.dz2 ? 2 : 1); ++pass) {          // second product (dz2, x2): same shapes, same dW, not in db
        const void* dzp = pass ? jb.dz2 : jb.dz;
        const void* xp = pass ? jb.x2 : jb.x;
        // a batch = U k-steps (2 U samples): 4 U dword loads per lane, then up to 4 U MFMAs.  Two register sets: the loads of
        // batch n + 1 are in flight while the MFMAs of batch n run (a wave's range is short -- a few dozen batches -- and each
        // load phase used to cost a full memory latency: 145 us per launch at 65 536 samples)
        auto load = [&](long long p, float (&av)[U][2], float (&bv)[U][2]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long s = 2 * (p + u) + h;
                const bool ok = (p + u < p1) && s < a.m;
                const float* dr = (const float*)dzp + (size_t)s * jb.ldz + i;       // (fp32 kernel: fp32 arrays only)
                av[u][0] = (ok && fa0) ? dr[0] : 0.0f;
                av[u][1] = (ok && fa1) ? dr[32] : 0.0f;
                if (xp) {
                    const float* xr = (const float*)xp + (size_t)s * jb.ldx + i;
                    bv[u][0] = (ok && gb0) ? xr[0] : 0.0f;
                    bv[u][1] = (ok && gb1) ? xr[32] : 0.0f;
                } else {                            // column c of [point (3), latent code]: c = i (first tile), 32 + i (second)
                    const long long ray = ok ? s / a.S : 0;
                    const float* rp = a.rays + (size_t)ray * a.ray_stride;
                    const float* lp = a.latents + (size_t)ray * a.lat_stride;
                    float v0 = 0.0f, v1 = 0.0f;
                    if (ok && gb0) v0 = (i < 3) ? __fadd_rn(rp[i], __fmul_rn(rp[3 + i], a.z[s])) : lp[i - 3];
                    if (ok && gb1) v1 = lp[29 + i];
                    bv[u][0] = v0;
                    bv[u][1] = v1;
                }
            }
        };
        auto compute = [&](const float (&av)[U][2], const float (&bv)[U][2]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (pass == 0) {
                    bsum[0] += av[u][0];
                    bsum[1] += av[u][1];
                }
                acc[0][0] = PolF32::mfma(av[u][0], bv[u][0], acc[0][0]);
                if (g1) acc[0][1] = PolF32::mfma(av[u][0], bv[u][1], acc[0][1]);
                if (f1) {
                    acc[1][0] = PolF32::mfma(av[u][1], bv[u][0], acc[1][0]);
                    if (g1) acc[1][1] = PolF32::mfma(av[u][1], bv[u][1], acc[1][1]);
                }
            }
        };
        float av0[U][2], bv0[U][2], av1[U][2], bv1[U][2];
        long long p = p0;
        if (p < p1) load(p, av0, bv0);
        while (p < p1) {
            const long long n1 = p + U;
            if (n1 < p1) load(n1, av1, bv1);
            __builtin_amdgcn_sched_barrier(0);
            compute(av0, bv0);
            __builtin_amdgcn_sched_barrier(0);
            if (n1 >= p1) break;
            const long long n2 = n1 + U;
            if (n2 < p1) load(n2, av0, bv0);
            __builtin_amdgcn_sched_barrier(0);
            compute(av1, bv1);
            __builtin_amdgcn_sched_barrier(0);
            p = n2;
        }
    }
    float* out = a.out + ((size_t)part * a.njobs + blockIdx.y) * BEND_WGRAD_SLOT;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(32 * u + tile_row(r, h)) * 64 + 32 * v + i] = acc[u][v][r];
        const float rs = bsum[u] + __shfl_xor(bsum[u], 32);
        if (h == 0) out[64 * 64 + 32 * u + i] = rs;
    }
}

// bf16 mode: the same products on v_mfma_f32_32x32x16_bf16 -- the fp32 rows are loaded exactly as above (the loads are
// what the kernel is made of: 32 dwords per lane per 16 samples either way) and rounded to bf16 in registers, fp32
// accumulation.  The fp32 MFMA form issues 32 MFMAs of 64 cycles per 16 samples and 64 x 64 product and was bound by them
// (bend_wgrad 1.8 ms per launch at 16 384 rays for 0.9 ms of HBM time); here it is 4 MFMAs of 32 cycles.  Operand k of a
// lane = 8 consecutive samples of one feature: element e of lane (i, h) of chunk c is sample 16 c + 8 h + e, column i.
// The gradients entering these products come out of a bf16 trunk in this mode, so the rounding of the operands (2^-9
// relative per element, averaged over the samples) is below what they carry already; fp32 mode keeps the kernel above.
template <int UNUSED>
__global__ void __launch_bounds__(256, 2) bend_wgrad16(const BendWgradArgs a) {
    typedef PolBF16::frag frag;
    const BendWgradJob jb = a.job[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, i = lane & 31;
    const int part = (int)blockIdx.x * 4 + wave;
    const unsigned chunks = (unsigned)((a.m + 15) / 16);
    const unsigned per = (chunks + a.nparts - 1) / a.nparts;
    const unsigned c0 = (unsigned)part * per, c1 = (c0 + per < chunks) ? c0 + per : chunks;
    // Which features a lane holds.  fp32 array: lane i holds features i (tile 0) and 32 + i (tile 1), one dword load each.
    // bf16 array (the saved arrays): lane i holds features 2 i and 2 i + 1 -- ONE dword load per row gives both, tile 0 = the
    // even features, tile 1 = the odd ones (two-byte loads, one per tile, moved 128 bytes per instruction and made this
    // kernel 2.5 x slower than on fp32 arrays); the epilogue writes dW / db rows and columns back in feature order.
    const bool dz16 = jb.dz16 != 0, x16 = jb.x16 != 0;
    const int fd0 = dz16 ? 2 * i : i, fd1 = dz16 ? 2 * i + 1 : 32 + i;          // features of this lane's dz elements
    const int fx0 = x16 ? 2 * i : i, fx1 = x16 ? 2 * i + 1 : 32 + i;            // ... of its x elements
    const bool fa0 = fd0 < jb.f, fa1 = fd1 < jb.f, gb0 = fx0 < jb.g, gb1 = fx1 < jb.g;
    // Every array element is fetched by a raw buffer load: a 32-bit byte offset off a base held in scalar registers, which
    // returns 0 beyond the array's last byte, so rows past the end need no clamping or masking.  A lane without a feature
    // reads the row's first dword and is zeroed at conversion time; the second dword of a row is only asked for from an
    // fp32 array (one wave-uniform branch per eight loads).  One loop body serves both element types: nothing but the row
    // pitch and wave-uniform selects at conversion time depend on it.  (Two earlier forms: a run-time choice of the element
    // type around each load broke the loop into hundreds of basic blocks that each waited for their own load, 2.4 ms per
    // launch; the loop compiled once per type combination spilled the accumulators.)
    const unsigned rbd = (unsigned)jb.ldz * (dz16 ? 2u : 4u), rbx = (unsigned)jb.ldx * (x16 ? 2u : 4u);      // row pitch in bytes
    const unsigned nd = (unsigned)(((size_t)(a.m - 1) * jb.ldz + jb.f) * (dz16 ? 2 : 4));                    // bytes of the array
    const unsigned nx = (unsigned)(((size_t)(a.m - 1) * jb.ldx + jb.g) * (x16 ? 2 : 4));
    const unsigned od0 = fa0 ? 4u * i : 0u, od1 = fa1 ? 4u * i + 128u : 0u;
    const unsigned ox0 = gb0 ? 4u * i : 0u, ox1 = gb1 ? 4u * i + 128u : 0u;
    f32x16 acc[2][2] = {{f32x16{}, f32x16{}}, {f32x16{}, f32x16{}}};
    float bsum[2] = {0.0f, 0.0f};
    const unsigned last = (unsigned)(a.m - 1);
    auto run = [&](auto x_generated) {              // the main loop, compiled for x from an array and for x made here
        constexpr bool GEN = decltype(x_generated)::value;
        for (int pass = 0; pass < (jb.dz2 ? 2 : 1); ++pass) {          // second product (dz2, x2): same shapes and types, same dW, not in db
            const void* xp = pass ? jb.x2 : jb.x;
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pass ? jb.dz2 : jb.dz), 0, (int)nd, 0x00020000);
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(GEN ? jb.dz : xp), 0, GEN ? 0 : (int)nx, 0x00020000);
            // x made here: the rays, their latent codes and the depths, each behind its own descriptor
            const unsigned n_rays = (unsigned)((a.m + a.S - 1) / a.S);
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.rays), 0, GEN ? (int)(((size_t)(n_rays - 1) * a.ray_stride + 6) * 4) : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.latents), 0, GEN ? (int)(((size_t)(n_rays - 1) * a.lat_stride + jb.g - 3) * 4) : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rzz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.z), 0, GEN ? (int)((size_t)a.m * 4) : 0, 0x00020000);
            const unsigned gi3 = (i < 3) ? i : 0, gil = (i >= 3 && gb0) ? i - 3 : 0, gih = gb1 ? 29 + i : 0;
            auto load = [&](unsigned c, unsigned (&ra)[8][2], unsigned (&rb)[8][2]) {
                unsigned row0 = 16u * c + 8u * h;
                asm volatile("" : "+v"(row0));          // opaque: otherwise 32 offsets become loop induction variables, per register set, and spill
                const unsigned bd = row0 * rbd, bx = row0 * rbx;
                if constexpr (GEN) {
                    // column c of [point (3), latent code], as in bend_wgrad: o + z d for lanes 0..2, the ray's latent code
                    // beyond.  Buffer loads again (no predication, 0 beyond the arrays); asked for BEFORE dz so that the
                    // arithmetic below waits for these small, cache-resident loads only
                    float ro[8], rdv[8], rz[8], l0[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned s = row0 + e, ray = s / (unsigned)a.S;
                        const unsigned br = ray * (unsigned)a.ray_stride * 4u, bl = ray * (unsigned)a.lat_stride * 4u;
                        ro[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, br + 4u * gi3, 0, 0));
                        rdv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, br + 12u + 4u * gi3, 0, 0));
                        rz[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rzz, 4u * s, 0, 0));
                        l0[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, bl + 4u * gil, 0, 0));
                        rb[e][1] = __builtin_amdgcn_raw_buffer_load_b32(rl, bl + 4u * gih, 0, 0);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) ra[e][0] = __builtin_amdgcn_raw_buffer_load_b32(rd, bd + od0 + e * rbd, 0, 0);
                    if (!dz16) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ra[e][1] = __builtin_amdgcn_raw_buffer_load_b32(rd, bd + od1 + e * rbd, 0, 0);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) rb[e][0] = __builtin_bit_cast(unsigned, (i < 3) ? __fadd_rn(ro[e], __fmul_rn(rdv[e], rz[e])) : l0[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ra[e][0] = __builtin_amdgcn_raw_buffer_load_b32(rd, bd + od0 + e * rbd, 0, 0);
                    if (!dz16) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ra[e][1] = __builtin_amdgcn_raw_buffer_load_b32(rd, bd + od1 + e * rbd, 0, 0);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) rb[e][0] = __builtin_amdgcn_raw_buffer_load_b32(rx, bx + ox0 + e * rbx, 0, 0);
                    if (!x16) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) rb[e][1] = __builtin_amdgcn_raw_buffer_load_b32(rx, bx + ox1 + e * rbx, 0, 0);
                    }
                }
            };
            auto compute = [&](const unsigned (&ra)[8][2], const unsigned (&rb)[8][2]) {
                frag fa[2], fb[2];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned ua1 = dz16 ? ra[e][0] & 0xffff0000u : ra[e][1], ub1 = x16 ? rb[e][0] & 0xffff0000u : rb[e][1];
                    const float a0 = __builtin_bit_cast(float, fa0 ? (dz16 ? ra[e][0] << 16 : ra[e][0]) : 0u);
                    const float a1 = __builtin_bit_cast(float, fa1 ? ua1 : 0u);
                    const float b0 = __builtin_bit_cast(float, gb0 ? (x16 ? rb[e][0] << 16 : rb[e][0]) : 0u);
                    const float b1 = __builtin_bit_cast(float, gb1 ? ub1 : 0u);
                    if (pass == 0) {
                        bsum[0] += a0;
                        bsum[1] += a1;
                    }
                    fa[0][e] = (__bf16)a0; fa[1][e] = (__bf16)a1;
                    fb[0][e] = (__bf16)b0; fb[1][e] = (__bf16)b1;
                }
                // all four tiles, used or not (zero operands where a tile has no features): a wave-uniform skip of the unused
                // ones made the compiler keep a second copy of the accumulators and move 16 registers per tile per chunk
                acc[0][0] = PolBF16::mfma(fa[0], fb[0], acc[0][0]);
                acc[0][1] = PolBF16::mfma(fa[0], fb[1], acc[0][1]);
                acc[1][0] = PolBF16::mfma(fa[1], fb[0], acc[1][0]);
                acc[1][1] = PolBF16::mfma(fa[1], fb[1], acc[1][1]);
            };
            unsigned ra0[8][2], rb0[8][2], ra1[8][2], rb1[8][2];
            unsigned c = c0;
            if (c < c1) load(c, ra0, rb0);
            while (c < c1) {
                if (c + 1 < c1) load(c + 1, ra1, rb1);
                __builtin_amdgcn_sched_barrier(0);
                compute(ra0, rb0);
                __builtin_amdgcn_sched_barrier(0);
                if (c + 1 >= c1) break;
                if (c + 2 < c1) load(c + 2, ra0, rb0);
                __builtin_amdgcn_sched_barrier(0);
                compute(ra1, rb1);
                __builtin_amdgcn_sched_barrier(0);
                c += 2;
            }
        }
    };
    if (jb.x) run(std::false_type{}); else run(std::true_type{});
    // D tile (u, v): lane (h, i) holds rows tile_row(r, h), column i of the tile; tile rows / columns -> features as above
    float* out = a.out + ((size_t)part * a.njobs + blockIdx.y) * BEND_WGRAD_SLOT;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int col = x16 ? 2 * i + v : 32 * v + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = dz16 ? 2 * tile_row(r, h) + u : 32 * u + tile_row(r, h);
                out[row * 64 + col] = acc[u][v][r];
            }
        }
        const float rs = bsum[u] + __shfl_xor(bsum[u], 32);
        if (h == 0) out[64 * 64 + (dz16 ? 2 * i + u : 32 * u + i)] = rs;
    }
}

template <class A, bool BWD, class SA>
static hipError_t launch_bend_train(const BendTrainArgs& a, int num_cus, hipStream_t stream) {
    using P = PolF32;
    constexpr int NFRAGS = BWD ? PlanBB<P, A>::NFRAGS : Plan<P, A, true, false, false>::NFRAGS;
    constexpr int NTILES = BWD ? PlanBB<P, A>::NTILES : Plan<P, A, true, false, false>::NTILES;
    const size_t lds = (size_t)NFRAGS * P::FRAG_BYTES + (size_t)NTILES * 32 * sizeof(float);
    void (*kern)(const BendTrainArgs) = nullptr;
    if constexpr (BWD) kern = bend_bwd<A, SA>; else kern = bend_fwd_train<A, SA>;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const long long nblocks = (long long)a.n_rays * ((a.S + 31) / 32);
    if (nblocks >= (1ll << 31)) return hipErrorInvalidValue;
    const long long want = (nblocks + 3) / 4;
    if (want <= 0) return hipSuccess;
    // two workgroups per CU when their resident weights fit twice (forward 20 / 28 KiB, backward 65 / 97 KiB of 160)
    const long long resident = (long long)num_cus * ((2 * lds <= 160 * 1024) ? 2 : 1);
    const int grid = (int)(want < resident ? want : resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    return hipGetLastError();
}


template <class A, bool BWD, class SA>
static hipError_t launch_bend_div(const BendDivArgs& a, int num_cus, hipStream_t stream) {
    using P = PolF32;
    constexpr int NFRAGS = BWD ? PlanBB<P, A>::NFRAGS : Plan<P, A, true, false, false>::NFRAGS;
    constexpr int NTILES = BWD ? PlanBB<P, A>::NTILES : Plan<P, A, true, false, false>::NTILES;
    const size_t lds = (size_t)NFRAGS * P::FRAG_BYTES + (size_t)NTILES * 32 * sizeof(float);
    void (*kern)(const BendDivArgs) = nullptr;
    if constexpr (BWD) kern = bend_div_bwd<A, SA>; else kern = bend_div_fwd<A, SA>;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const long long want = ((a.m + 31) / 32 + 3) / 4;
    if (want <= 0) return hipSuccess;
    const long long resident = (long long)num_cus * ((2 * lds <= 160 * 1024) ? 2 : 1);
    const int grid = (int)(want < resident ? want : resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    return hipGetLastError();
}

}  // namespace nrn
