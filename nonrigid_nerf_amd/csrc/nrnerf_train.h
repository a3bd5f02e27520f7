// nrnerf_train.h -- training support for the canonical NeRF trunk (reference NeRF.forward run_nerf_helpers.py:272-306 under
// autograd; the training step is training_wrapper_class.forward + backward, train.py:152-287, 1594-1610).
//
// Forward and backward-data are built from the forward kernel's parts (weight ring, dense layer, in-register activation hand-off):
//
//   trunk_fwd_train   positional encoding + 8x256 trunk + head on ready-made points (the bent points: the deformation
//                     MLPs have kernels of their own, nrnerf_train_bend.h),
//                     like the inference kernel, but every hidden activation h_i = relu(W_i x_i + b_i) is also written
//                     to HBM, [layer][sample][256] in TRUE feature order: the D tile holds features 32t+8q+4h .. +3 of a
//                     sample in four consecutive accumulator registers, i.e. 16 (fp32) or 8 (bf16) contiguous bytes.
//   trunk_bwd         backward-data: the same dataflow with the layers reversed and the weights transposed (PlanB,
//                     nrnerf_plan.h): d h_{D-1} = W_out^T d raw, then for i = D-1 .. 1:  d z_i = d h_i * [h_i > 0]
//                     (stored for the weight gradients),  d x_i = W_i^T d z_i  (x_5 = [encoding, h_4]: its first two tiles
//                     are the encoding's gradient), finally d enc += W_0^T d z_0 and, through the derivative of the
//                     encoding, the gradient wrt the input point.
//   trunk_wgrad       bf16 mode: the weight gradients  dW_i = d z_i^T x_i  and bias gradients of a trunk in one launch over
//                     the two stored arrays;  trunk_wgrad_f32: the same on the fp32 mode's row-major arrays.
// fp32 mode (exact, the gradient-parity mode) and bf16 mode (bf16 operands incl. the stored activations and d z).
//
// Layout of the stored arrays.  fp32 mode: [layer][sample][256], true feature order (rows of features: trunk_wgrad_f32
// contracts over samples with one dword per lane and MFMA, so a lane reads along a row).  bf16 mode: [layer][block][feature][32 samples] -- one 256 x 32 tile per 32-sample block with the SAMPLES
// contiguous: register r of a D tile holds one feature for the 32 samples of the block in the 32 lanes of a wave half, so a
// store instruction writes two whole 64-byte rows; and it is the layout the weight-gradient kernel (trunk_wgrad, below)
// wants, whose contraction runs over samples: a lane's 8 consecutive k of an MFMA operand are 8 consecutive samples of
// one feature = one 16-byte load, no transpose anywhere.  Lanes of samples beyond the ray's end hold finite activations
// (those of the clamped sample) and zero gradients, so the padded columns contribute nothing.  The backward kernel does not
// read the activations at all in this mode: the forward kernel also writes which values passed the relu, 16 bits per lane
// and tile, one record per lane and layer ([layer][block][64 lanes][W/32 tiles] u16, 1/16 of the activations' bytes).
#pragma once
#include "nrnerf_net_impl.h"

namespace nrn {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// four consecutive features of one sample, stored / loaded in the array's element type
template <class P>
__device__ __forceinline__ void store4(void* base, size_t elem_index, float a, float b, float c, float d) {
    if constexpr (P::KH == 1) {
        *(f32x4*)((float*)base + elem_index) = f32x4{a, b, c, d};
    } else {
        const bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
        *(bf16x4*)((__bf16*)base + elem_index) = v;
    }
}
template <class P>
__device__ __forceinline__ f32x4 load4(const void* base, size_t elem_index) {
    if constexpr (P::KH == 1) {
        return *(const f32x4*)((const float*)base + elem_index);
    } else {
        const bf16x4 v = *(const bf16x4*)((const __bf16*)base + elem_index);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
}

// bf16 mode: the 16 values of a D tile this lane holds, written into the [feature][32 samples] tile of its block straight
// from the two B-operand fragments pack_tile has just produced (word k of fragment u = registers 8u + 2k, 8u + 2k + 1 as
// bf16): 16 two-byte stores, no second conversion, no temporaries.  `dst` points at (feature 32 t + 4 h, sample j).
template <int R>
__device__ __forceinline__ void store_half_bf16(__bf16* dst, unsigned word) {
    constexpr int off = ((R & 3) + 8 * (R >> 2)) * 32 * 2;              // bytes: feature row of register R, 64 bytes per row
    if constexpr (R & 1) asm volatile("global_store_short_d16_hi %0, %1, off offset:%2" ::"v"(dst), "v"(word), "n"(off) : "memory");
    else asm volatile("global_store_short %0, %1, off offset:%2" ::"v"(dst), "v"(word), "n"(off) : "memory");
}
template <class FRAG, int... R>
__device__ __forceinline__ void store_tile_bf16_impl(__bf16* dst, const FRAG& f0, const FRAG& f1, std::integer_sequence<int, R...>) {
    const u32x4 w0 = __builtin_bit_cast(u32x4, f0), w1 = __builtin_bit_cast(u32x4, f1);
    (store_half_bf16<R>(dst, (R < 8 ? w0 : w1)[(R & 7) >> 1]), ...);
}
template <class FRAG>
__device__ __forceinline__ void store_tile_bf16(__bf16* dst, const FRAG& f0, const FRAG& f1) {
    store_tile_bf16_impl(dst, f0, f1, std::make_integer_sequence<int, 16>{});
}

// ------------------------------------------------------------------------------------------
// forward with saved activations
// ------------------------------------------------------------------------------------------
// VIEWS: the view-dependent head (rnh:284-304) behind the trunk, as in the inference kernel -- alpha_linear, then
// relu(views_linears[0]([feature_linear(h), enc(direction)])) as ONE layer (feature_linear folded into its weights by the
// packer), then rgb_linear -- on the per-sample directions the caller hands in (TrunkArgs::dirs: the finite differences of the
// bent points, or the rays' own).  Saved for the backward pass: the colour branch's hidden activation hv (W / 2 values per
// sample, layout of `acts`) and, bf16 mode, its relu bits.
template <class P, class A, int WAVES, bool VIEWS = false>
__global__ void __launch_bounds__(WAVES * 64, (P::KH == 1) ? 1 : 2) trunk_fwd_train(const TrunkArgs a) {
    using PL = Plan<P, A, false, VIEWS>;
    using frag = typename P::frag;
    using PE = std::conditional_t<P::KH == 1, PolF32, PolF16>;
    using efrag = typename PE::frag;
    constexpr int KH = P::KH, SP = P::SP, NS_ENC = PL::NS_ENC, NT_W = PL::NT_W;
    static_assert(!A::TCB, "no training support for the time-conditioned baseline");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* bias_lds = (float*)(smem + RING * P::UNIT_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    for (int i = tid; i < PL::NTILES * 32; i += WAVES * 64) bias_lds[i] = a.bias[i];
    __syncthreads();
    const BiasPtr bias_lane = bias_lane_ptr(bias_lds, h);
    WRing<P, WAVES, PL::NUP> st;
    st.init(a.wstream, ring, wave, lane);

    const int S = a.S;
    const int bpr = (S + 31) >> 5;
    const long long nblocks = (long long)a.n_rays * bpr;
    const size_t M = (size_t)a.n_rays * S;
    for (long long tile0 = (long long)blockIdx.x * WAVES; tile0 < nblocks; tile0 += (long long)gridDim.x * WAVES) {
        const long long blk = tile0 + wave;
        const bool blk_ok = blk < nblocks;
        const long long b = blk_ok ? blk : nblocks - 1;
        const int ray = (int)(b / bpr);
        const int sidx = (int)(b % bpr) * 32 + j;
        const bool ok = blk_ok && sidx < S;
        const size_t so = (size_t)ray * S + (sidx < S ? sidx : S - 1);
        const f32x4 q4 = *(const f32x4*)(a.pts4 + so * 4);
        const float p[3] = {q4[0], q4[1], q4[2]};

        // positional encoding in B-operand order (as the inference kernel)
        constexpr int F0 = enc_F0(A::L);
        constexpr int NSLOT = PL::NS_ENC_XYZ * KH;
        float ev[NSLOT];
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) ev[q] = 0.0f;
        ev[0] = h ? p[2] : p[0];
        ev[1] = h ? 0.0f : p[1];
        const float fscale = h ? (float)(1 << F0) : 1.0f;
        const float prev_[3] = {p[0] * 0.15915494309189535f, p[1] * 0.15915494309189535f, p[2] * 0.15915494309189535f};
        static_for<0, F0>([&](auto fc) {
            constexpr int fl = decltype(fc)::value;
            static_for<0, 3>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                float sv, cv;
                enc_sincos<KH == 1>(p[c], prev_[c], fscale * (float)(1 << fl), &sv, &cv);
                ev[2 + 2 * (3 * fl + c)] = sv;
                ev[2 + 2 * (3 * fl + c) + 1] = cv;
            });
        });
        efrag enc[NS_ENC];
        static_for<0, NS_ENC>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            static_for<0, KH>([&](auto ec) { constexpr int e = decltype(ec)::value; PE::template set<e>(enc[s], ev[s * KH + e]); });
        });

        constexpr int NH = NT_W * SP;
        frag ha[NH], hb[NH];
        Empty none;
        unsigned mw = 0;                    // relu masks of the tile pair in flight
        // epilogue of hidden layer LAYER: relu, keep for the backward pass, hand to the next layer
        auto keep = [&](auto lc, auto tc, const f32x16& acc_in, auto& out) {
            constexpr int layer = decltype(lc)::value, t = decltype(tc)::value;
            f32x16 acc = acc_in;
            if constexpr (layer == 0 || layer == A::SKIP + 1) {
                if (a.ray_bias) {       // per-ray bias (time-conditioned baseline): features of this lane's accumulator rows
                    const float* rb = a.ray_bias + ((size_t)ray * 2 + (layer == 0 ? 0 : 1)) * A::W + 32 * t + 4 * h;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 bv = *(const f32x4*)(rb + 8 * q);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[4 * q + k] += bv[k];
                    }
                }
            }
            if constexpr (KH == 1) {
                if (ok) {
                    const size_t row = ((size_t)layer * M + so) * A::W + 32 * t + 4 * h;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        store4<P>(a.acts, row + 8 * q, relu_bits(acc[4 * q]), relu_bits(acc[4 * q + 1]), relu_bits(acc[4 * q + 2]),
                                  relu_bits(acc[4 * q + 3]));
                }
            }
            pack_tile<P, true, t>(acc, out);
            if constexpr (KH != 1) {
                if (blk_ok) store_tile_bf16((__bf16*)a.acts + (((size_t)layer * nblocks + b) * A::W + 32 * t + 4 * h) * 32 + j, out[t * SP], out[t * SP + 1]);
                // which of this lane's 16 values passed the relu: all the backward kernel needs of the activations.  The NT_W
                // tile masks of a layer form ONE 2 NT_W-byte record per lane and layer ([layer][block][lane][tile] u16), written a
                // tile pair (one dword) at a time: the backward kernel reads a layer's record with one load -- 64 dependent
                // two-byte loads per block cost it 0.9 ms per launch at 16 384 rays (probe: profiles/r03_train_store_probe.txt)
                unsigned m = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) m |= (acc[r] > 0.0f ? 1u : 0u) << r;
                if constexpr ((t & 1) == 0) {
                    mw = m;
                } else {                // tiles come in pairs (t - 1, t): one dword per pair
                    mw |= m << 16;
                    if (blk_ok) ((unsigned*)(a.mask + (((size_t)layer * nblocks + b) * 64 + lane) * NT_W))[t / 2] = mw;
                }
            }
        };
        dense<PE, P, PL, PL::L_TRUNK0, NS_ENC, 0>(st, bias_lane, enc, none, [&](auto tc, const f32x16& acc) {
            keep(std::integral_constant<int, 0>{}, tc, acc, ha); });
        static_for<1, A::D>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr bool skip = (i - 1 == A::SKIP);
            if constexpr (i % 2 == 1) {
                if constexpr (skip)
                    dense<PE, P, PL, PL::L_TRUNK0 + i, NS_ENC, NH>(st, bias_lane, enc, ha, [&](auto tc, const f32x16& acc) { keep(ic, tc, acc, hb); });
                else
                    dense<P, P, PL, PL::L_TRUNK0 + i, NH, 0>(st, bias_lane, ha, none, [&](auto tc, const f32x16& acc) { keep(ic, tc, acc, hb); });
            } else {
                if constexpr (skip)
                    dense<PE, P, PL, PL::L_TRUNK0 + i, NS_ENC, NH>(st, bias_lane, enc, hb, [&](auto tc, const f32x16& acc) { keep(ic, tc, acc, ha); });
                else
                    dense<P, P, PL, PL::L_TRUNK0 + i, NH, 0>(st, bias_lane, hb, none, [&](auto tc, const f32x16& acc) { keep(ic, tc, acc, ha); });
            }
        });
        float raw[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (!VIEWS) {
            auto take_raw = [&](auto, const f32x16& acc) { raw[0] = acc[0]; raw[1] = acc[1]; raw[2] = acc[2]; raw[3] = acc[3]; raw[4] = acc[4]; };
            if constexpr ((A::D - 1) % 2 == 1) dense<P, P, PL, PL::L_HEAD, NH, 0>(st, bias_lane, hb, none, take_raw);
            else dense<P, P, PL, PL::L_HEAD, NH, 0>(st, bias_lane, ha, none, take_raw);
        } else {
            // encoding of the sample's view direction in B-operand order (Embedder with LV frequencies, as the point's above)
            constexpr int NS_ENCV = PL::NS_ENCV, F0V = enc_F0(A::LV), NSLOTV = NS_ENCV * KH;
            const float* dp = a.dirs + so * 3;
            const float dirv[3] = {dp[0], dp[1], dp[2]};
            float evv[NSLOTV];
#pragma unroll
            for (int q = 0; q < NSLOTV; ++q) evv[q] = 0.0f;
            evv[0] = h ? dirv[2] : dirv[0];
            evv[1] = h ? 0.0f : dirv[1];
            const float vscale = h ? (float)(1 << F0V) : 1.0f;
            const float drev[3] = {dirv[0] * 0.15915494309189535f, dirv[1] * 0.15915494309189535f, dirv[2] * 0.15915494309189535f};
            static_for<0, F0V>([&](auto fc) {
                constexpr int fl = decltype(fc)::value;
                static_for<0, 3>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    float sv, cv;
                    enc_sincos<KH == 1>(dirv[c], drev[c], vscale * (float)(1 << fl), &sv, &cv);
                    evv[2 + 2 * (3 * fl + c)] = sv;
                    evv[2 + 2 * (3 * fl + c) + 1] = cv;
                });
            });
            efrag encv[NS_ENCV];
            static_for<0, NS_ENCV>([&](auto sc_) {
                constexpr int s = decltype(sc_)::value;
                static_for<0, KH>([&](auto ec) { constexpr int e = decltype(ec)::value; PE::template set<e>(encv[s], evv[s * KH + e]); });
            });
            constexpr int NTV = NT_W / 2, NV = NTV * SP;
            frag hv[NV];
            unsigned mv = 0;
            // epilogue of the colour branch's hidden layer: relu, keep (rows [M][W/2] fp32, or bf16 tiles [block][W/2][32] + relu bits
            // [block][lane][NTV] u16), hand to rgb_linear
            auto keep_v = [&](auto tc, const f32x16& acc, auto& out) {
                constexpr int t = decltype(tc)::value;
                if constexpr (KH == 1) {
                    if (ok) {
                        const size_t row = so * (A::W / 2) + 32 * t + 4 * h;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            store4<P>(a.hv, row + 8 * q, relu_bits(acc[4 * q]), relu_bits(acc[4 * q + 1]), relu_bits(acc[4 * q + 2]), relu_bits(acc[4 * q + 3]));
                    }
                }
                pack_tile<P, true, t>(acc, out);
                if constexpr (KH != 1) {
                    if (blk_ok) store_tile_bf16((__bf16*)a.hv + (((size_t)b) * (A::W / 2) + 32 * t + 4 * h) * 32 + j, out[t * SP], out[t * SP + 1]);
                    unsigned m = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) m |= (acc[r] > 0.0f ? 1u : 0u) << r;
                    if constexpr ((t & 1) == 0) {
                        mv = m;
                    } else {
                        mv |= m << 16;
                        if (blk_ok) ((unsigned*)(a.hv_mask + ((size_t)b * 64 + lane) * NTV))[t / 2] = mv;
                    }
                }
            };
            auto head = [&](auto& hx) {
                dense<P, P, PL, PL::L_ALPHA, NH, 0>(st, bias_lane, hx, none, [&](auto, const f32x16& acc) { raw[3] = acc[0]; });
                dense<PE, P, PL, PL::L_VIEWS, NS_ENCV, NH>(st, bias_lane, encv, hx, [&](auto tc, const f32x16& acc) { keep_v(tc, acc, hv); });
                dense<P, P, PL, PL::L_RGB, NV, 0>(st, bias_lane, hv, none, [&](auto, const f32x16& acc) {
                    raw[0] = acc[0]; raw[1] = acc[1]; raw[2] = acc[2]; });
            };
            if constexpr ((A::D - 1) % 2 == 1) head(hb); else head(ha);
        }
        if (ok && h == 0) {
            *(f32x4*)(a.raw4 + so * 4) = f32x4{raw[0], raw[1], raw[2], raw[3]};
            if (a.raw_out) {
                float* ro = a.raw_out + so * a.raw_ch;
                ro[0] = raw[0]; ro[1] = raw[1]; ro[2] = raw[2]; ro[3] = raw[3];
                if (a.raw_ch > 4) ro[4] = raw[4];
            }
        }
        static_for<PL::NUNITS, PL::NUP>([&](auto uc) { st.template advance<decltype(uc)::value>(); });
    }
    st.drain();
}

// ------------------------------------------------------------------------------------------
// backward-data
// ------------------------------------------------------------------------------------------
// VIEWS: the view-dependent head first -- rgb_linear^T on d raw's colour channels gives d hv, masked with hv's relu bits =
// d z_v (stored for the weight gradients); then ONE layer takes [d raw (sigma), d z_v] to [gradient of the direction encoding
// (one tile, encoding-slot order), d h_{D-1}]: alpha_linear^T and the transposed folded views layer share accumulators
// (PlanB<.., true>).  The direction encoding's gradient goes through the encoding's derivative to TrunkArgs::d_dirs.
template <class P, class A, int WAVES, bool VIEWS = false>
__global__ void __launch_bounds__(WAVES * 64, (P::KH == 1) ? 1 : 2) trunk_bwd(const TrunkArgs a) {
    using PL = PlanB<P, A, VIEWS>;
    using frag = typename P::frag;
    constexpr int KH = P::KH, SP = P::SP, NT_W = PL::NT_W, NT_E = PL::NT_E, NS_DR = PL::NS_DR;
    static_assert(NT_E == 2, "the encoding gradient is kept in two accumulator tiles");
    static_assert(!VIEWS || PL::NT_EV == 1, "the direction encoding's gradient is kept in one accumulator tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* bias_lds = (float*)(smem + RING * P::UNIT_BYTES);           // all zero: backward layers have no bias
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    for (int i = tid; i < PL::NTILES * 32; i += WAVES * 64) bias_lds[i] = 0.0f;
    __syncthreads();
    const BiasPtr bias_lane = bias_lane_ptr(bias_lds, h);
    WRing<P, WAVES, PL::NUP> st;
    st.init(a.wstream, ring, wave, lane);

    const int S = a.S;
    const int bpr = (S + 31) >> 5;
    const long long nblocks = (long long)a.n_rays * bpr;
    const size_t M = (size_t)a.n_rays * S;
    for (long long tile0 = (long long)blockIdx.x * WAVES; tile0 < nblocks; tile0 += (long long)gridDim.x * WAVES) {
        const long long blk = tile0 + wave;
        const bool blk_ok = blk < nblocks;
        const long long b = blk_ok ? blk : nblocks - 1;
        const int ray = (int)(b / bpr);
        const int sidx = (int)(b % bpr) * 32 + j;
        const bool ok = blk_ok && sidx < S;
        const size_t so = (size_t)ray * S + (sidx < S ? sidx : S - 1);

        // d raw as the B operand of the head^T layer: logical vector v[ch], element (s, h, e) = v[(2s + h) KH + e]
        const f32x4 g4 = *(const f32x4*)(a.d_raw4 + so * 4);
        frag dr[NS_DR];
        static_for<0, NS_DR>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            static_for<0, KH>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                constexpr int i0 = (2 * s) * KH + e, i1 = (2 * s + 1) * KH + e;
                const float v0 = (i0 < 4) ? g4[i0 < 4 ? i0 : 0] : 0.0f;
                const float v1 = (i1 < 4) ? g4[i1 < 4 ? i1 : 0] : 0.0f;
                P::template set<e>(dr[s], ok ? (h ? v1 : v0) : 0.0f);
            });
        });

        constexpr int NH = NT_W * SP;
        frag ha[NH], hb[NH];
        Empty none;
        // bf16 mode: the relu masks, one record of NT_W 16-bit tile masks per lane and layer (trunk_fwd_train), requested
        // right before the transposed layer whose epilogue applies it: the load's latency hides behind that layer's MFMAs
        // (in two halves of NT_W / 2 tiles -- the second one requested when the first has been used up, a tile pair's MFMAs
        //  before it is needed: four registers held across a layer spilled two at this kernel's 128-register budget)
        constexpr int MH = NT_W / 4;                                   // dwords per half record
        unsigned mk[MH];
        auto load_masks = [&](auto lc, auto hc) {
            constexpr int layer = decltype(lc)::value, half = decltype(hc)::value;
            if constexpr (KH != 1) {
                const unsigned* mp = (const unsigned*)(a.mask + (((size_t)layer * nblocks + b) * 64 + lane) * NT_W) + half * MH;
                if constexpr (MH == 2) {
                    const unsigned long long v = *(const unsigned long long*)mp;
                    mk[0] = (unsigned)v; mk[1] = (unsigned)(v >> 32);
                } else {
                    static_assert(MH == 1, "trunk widths 256 / 128");
                    mk[0] = *mp;
                }
            }
        };
        // epilogue producing d z_LAYER from tile t of d h_LAYER: mask with the saved activation, store, hand on
        auto mask_store = [&](auto lc, auto tc, const f32x16& acc, auto& out) {
            constexpr int layer = decltype(lc)::value, t = decltype(tc)::value;
            f32x16 g = acc;
            if constexpr (KH == 1) {
                const size_t row = ((size_t)layer * M + so) * A::W + 32 * t + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 hv = load4<P>(a.acts, row + 8 * q);
#pragma unroll
                    for (int k = 0; k < 4; ++k) g[4 * q + k] = (hv[k] > 0.0f) ? acc[4 * q + k] : 0.0f;
                    if (ok) store4<P>(a.d_pre, row + 8 * q, g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]);
                }
            } else {
                // this tile's 16 mask bits out of the layer's record (requested a layer ahead, see load_masks)
                const unsigned m = (mk[(t / 2) % MH] >> (16 * (t & 1))) & 0xffffu;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // bit r of the mask sign-extended to all ones / all zeros, ANDed in: two VALU per value and no condition
                    // registers (the select form spilled two registers into scratch at the 256-register budget).  The
                    // scalar copy matters: __builtin_bit_cast applied to a vector ELEMENT reads element 0 (hipcc 7.2).
                    const float v = acc[r];
                    g[r] = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & ((int)(m << (31 - r)) >> 31));
                }
            }
            pack_tile<P, false, t>(g, out);
            if constexpr (KH != 1) {
                if (blk_ok)
                    store_tile_bf16((__bf16*)a.d_pre + (((size_t)layer * nblocks + b) * A::W + 32 * t + 4 * h) * 32 + j, out[t * SP], out[t * SP + 1]);
                if constexpr (t == NT_W / 2 - 1) load_masks(lc, std::integral_constant<int, 1>{});      // second half of this layer's record
            }
        };
        // d h_{D-1} from tile t of head^T's output, masked and stored
        auto last_hidden = [&](auto tc, const f32x16& acc) { mask_store(std::integral_constant<int, A::D - 1>{}, tc, acc, ha); };
        if constexpr (!VIEWS) {
            // head^T: d h_{D-1}
            load_masks(std::integral_constant<int, A::D - 1>{}, std::integral_constant<int, 0>{});
            dense<P, P, PL, 0, NS_DR, 0>(st, bias_lane, dr, none, last_hidden);
        } else {
            constexpr int NTV = NT_W / 2, NV = NTV * SP;
            static_assert(KH == 1 || NTV == 4, "one 8-byte record of relu bits per lane");
            frag dv[NV];
            f32x16 dencv;                   // gradient of the direction encoding, slot q of this lane half in register q
            unsigned long long mvw = 0;
            if constexpr (KH != 1) mvw = *(const unsigned long long*)(a.hv_mask + ((size_t)b * 64 + lane) * NTV);
            // rgb_linear^T: d hv, masked with hv > 0 = d z_v, stored (layout of d_pre) and handed on
            dense<P, P, PL, 0, NS_DR, 0>(st, bias_lane, dr, none, [&](auto tc, const f32x16& acc) {
                constexpr int t = decltype(tc)::value;
                f32x16 g = acc;
                if constexpr (KH == 1) {
                    const size_t row = so * (A::W / 2) + 32 * t + 4 * h;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 hvv = load4<P>(a.hv, row + 8 * q);
#pragma unroll
                        for (int k = 0; k < 4; ++k) g[4 * q + k] = (hvv[k] > 0.0f) ? acc[4 * q + k] : 0.0f;
                        if (ok) store4<P>(a.d_pre_v, row + 8 * q, g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]);
                    }
                } else {
                    const unsigned m = (unsigned)(mvw >> (16 * t)) & 0xffffu;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[r];
                        g[r] = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & ((int)(m << (31 - r)) >> 31));
                    }
                }
                pack_tile<P, false, t>(g, dv);
                if constexpr (KH != 1) {
                    if (blk_ok) store_tile_bf16((__bf16*)a.d_pre_v + (((size_t)b) * (A::W / 2) + 32 * t + 4 * h) * 32 + j, dv[t * SP], dv[t * SP + 1]);
                }
            });
            // [alpha_linear; views o feature]^T: tile 0 = the direction encoding's gradient, tiles 1 .. NT_W = d h_{D-1}
            load_masks(std::integral_constant<int, A::D - 1>{}, std::integral_constant<int, 0>{});
            dense<P, P, PL, 1, NS_DR, NV>(st, bias_lane, dr, dv, [&](auto tc, const f32x16& acc) {
                constexpr int t = decltype(tc)::value;
                if constexpr (t == 0) dencv = acc;
                else last_hidden(std::integral_constant<int, t - 1>{}, acc);
            });
            // (right away: sixteen registers held to the end of the pass spilled two at this kernel's 256-register budget)
            if (a.d_dirs) {             // through the direction's encoding (LV frequencies), as for the point at the end of the pass
                const float* dq = a.dirs + so * 3;
                const float dv3[3] = {dq[0], dq[1], dq[2]};
                constexpr int F0V = enc_F0(A::LV);
                float dd[3] = {0.f, 0.f, 0.f};
                auto vslot = [&](auto qc) -> float { return dencv[decltype(qc)::value]; };
                dd[0] = h ? 0.0f : vslot(std::integral_constant<int, 0>{});
                dd[1] = h ? 0.0f : vslot(std::integral_constant<int, 1>{});
                dd[2] = h ? vslot(std::integral_constant<int, 0>{}) : 0.0f;
                const float vscale = h ? (float)(1 << F0V) : 1.0f;
                const float drev[3] = {dv3[0] * 0.15915494309189535f, dv3[1] * 0.15915494309189535f, dv3[2] * 0.15915494309189535f};
                static_for<0, F0V>([&](auto fc) {
                    constexpr int fl = decltype(fc)::value;
                    static_for<0, 3>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        if (h * F0V + fl < A::LV) {
                            const float scale = vscale * (float)(1 << fl);
                            float sv, cv;
                            enc_sincos<KH == 1>(dv3[c], drev[c], scale, &sv, &cv);
                            const float ds = vslot(std::integral_constant<int, 2 + 2 * (3 * fl + c)>{});
                            const float dc = vslot(std::integral_constant<int, 2 + 2 * (3 * fl + c) + 1>{});
                            dd[c] += scale * (cv * ds - sv * dc);
                        }
                    });
                });
#pragma unroll
                for (int c = 0; c < 3; ++c) dd[c] += __shfl_xor(dd[c], 32);
                if (ok && h == 0) { float* o = a.d_dirs + so * 3; o[0] = dd[0]; o[1] = dd[1]; o[2] = dd[2]; }
            }
        }
        f32x16 denc[NT_E];
        // layers D-1 .. 1 (transposed): input d z_i in the buffer the previous step filled, output d h_{i-1}
        static_for<0, A::D - 1>([&](auto kc) {
            constexpr int k = decltype(kc)::value;          // 0 .. D-2
            constexpr int i = A::D - 1 - k;                 // forward layer whose transpose runs now
            constexpr int LI = PL::layer_of(i);
            constexpr bool skip = (i - 1 == A::SKIP);
            auto run = [&](auto& src, auto& dst) {
                load_masks(std::integral_constant<int, i - 1>{}, std::integral_constant<int, 0>{});
                if constexpr (skip) {
                    dense<P, P, PL, LI, NH, 0>(st, bias_lane, src, none, [&](auto tc, const f32x16& acc) {
                        constexpr int t = decltype(tc)::value;
                        if constexpr (t < NT_E) denc[t] = acc;
                        else mask_store(std::integral_constant<int, i - 1>{}, std::integral_constant<int, t - NT_E>{}, acc, dst);
                    });
                } else {
                    dense<P, P, PL, LI, NH, 0>(st, bias_lane, src, none, [&](auto tc, const f32x16& acc) {
                        mask_store(std::integral_constant<int, i - 1>{}, tc, acc, dst); });
                }
            };
            if constexpr (k % 2 == 0) run(ha, hb); else run(hb, ha);
        });
        // layer 0^T: the rest of the encoding's gradient
        constexpr bool LAST_IN_B = ((A::D - 1) % 2 == 1);
        auto add_enc = [&](auto tc, const f32x16& acc) { denc[decltype(tc)::value] += acc; };
        if constexpr (LAST_IN_B) dense<P, P, PL, PL::layer_of(0), NH, 0>(st, bias_lane, hb, none, add_enc);
        else dense<P, P, PL, PL::layer_of(0), NH, 0>(st, bias_lane, ha, none, add_enc);

        // through the encoding (Embedder, run_nerf_helpers.py:120-150): slot q of this lane half holds d/d(value q);
        // d p_c = d id_c + sum_f 2^f (cos(2^f p_c) d sin - sin(2^f p_c) d cos)
        const f32x4 q4 = *(const f32x4*)(a.pts4 + so * 4);
        const float p[3] = {q4[0], q4[1], q4[2]};
        constexpr int F0 = enc_F0(A::L);
        float dp[3] = {0.f, 0.f, 0.f};
        auto dslot = [&](auto qc) -> float { constexpr int q = decltype(qc)::value; return denc[q / 16][q % 16]; };
        dp[0] = h ? 0.0f : dslot(std::integral_constant<int, 0>{});
        dp[1] = h ? 0.0f : dslot(std::integral_constant<int, 1>{});
        dp[2] = h ? dslot(std::integral_constant<int, 0>{}) : 0.0f;
        const float fscale = h ? (float)(1 << F0) : 1.0f;
        const float prev_[3] = {p[0] * 0.15915494309189535f, p[1] * 0.15915494309189535f, p[2] * 0.15915494309189535f};
        static_for<0, F0>([&](auto fc) {
            constexpr int fl = decltype(fc)::value;
            static_for<0, 3>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if (h * F0 + fl < A::L) {       // a frequency that exists (odd L: the upper half has one fewer)
                    const float scale = fscale * (float)(1 << fl);
                    float sv, cv;
                    enc_sincos<KH == 1>(p[c], prev_[c], scale, &sv, &cv);
                    const float ds = dslot(std::integral_constant<int, 2 + 2 * (3 * fl + c)>{});
                    const float dc = dslot(std::integral_constant<int, 2 + 2 * (3 * fl + c) + 1>{});
                    dp[c] += scale * (cv * ds - sv * dc);
                }
            });
        });
#pragma unroll
        for (int c = 0; c < 3; ++c) dp[c] += __shfl_xor(dp[c], 32);          // the two halves hold different frequencies
        if (ok && h == 0) *(f32x4*)(a.d_pts4 + so * 4) = f32x4{dp[0], dp[1], dp[2], 0.0f};
        static_for<PL::NUNITS, PL::NUP>([&](auto uc) { st.template advance<decltype(uc)::value>(); });
    }
    st.drain();
}

// ------------------------------------------------------------------------------------------
// weight gradients (bf16 mode)
// ------------------------------------------------------------------------------------------
// dW = dz^T x summed over the samples, for a list of (dz, x) pairs -- the seven hidden-to-hidden layers (x = the previous
// layer's activations) and the two layers that see the encoding (x = the encoding, 64 columns) -- straight from the
// [block][feature][32 samples] arrays the two kernels above fill: the contraction index of the MFMA is the sample, a lane's
// operand is 8 consecutive samples of one feature (one 16-byte global load), so there is no LDS, no transpose and no
// barrier.  Workgroup c of job j accumulates the job over blocks c, c + kch_j, ... in registers and writes one fp32 partial
// [W][xw]; the caller adds the partials.  The jobs have their own kch: a 64-column job costs 5/16 of a hidden-to-hidden one
// per block, so it gets 5/16 of the workgroups and all workgroups of the launch (<= one per CU) finish together.  4 waves (one per SIMD, up to 256 accumulator registers): wave
// (wr, wc) owns tile rows wr * NTR/2 .. and tile columns wc * NTC/2 ..; next block's 16 fragments are requested before the
// current block's MFMAs.  Measured 3.2 TB/s of unique HBM traffic at 16 384 rays; 8 waves with half the tiles each (more
// loads in flight, but every fragment requested by more waves) were 10 % slower.  The bias gradient comes along: the waves of column 0 add up the dz fragments they hold (eight
// conversions + adds per fragment on the VALU, one register per row tile).
// DZW: features of dz (rows of the product: W, or W / 2 for the colour branch's hidden layer); TCW: column tiles of x per wave,
// compile-time so that the MFMAs issue back to back
template <int DZW, int TCW>
__device__ __forceinline__ void trunk_wgrad_job(const WgradJob& jb, const long long nblocks, const long long pstride, const int sync_every, int c) {
    using P = PolBF16;
    typedef typename P::frag frag;
    constexpr int NTR = DZW / 32;                   // row tiles of dz^T (features of this layer)
    constexpr int TR = NTR / 2;                     // per wave
    static_assert(NTR % 2 == 0, "two wave rows");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int h = lane >> 5, i = lane & 31;
    const __bf16* dz = (const __bf16*)jb.dz;
    const __bf16* x = (const __bf16*)jb.x;
    f32x16 acc[TR][TCW];
    float bsum[TR];          // bias gradient: this lane's share of the row sums of dz (its 8 samples of feature i per fragment)
#pragma unroll
    for (int u = 0; u < TR; ++u) {
        bsum[u] = 0.0f;
#pragma unroll
        for (int v = 0; v < TCW; ++v) acc[u][v] = f32x16{};
    }
    auto load = [&](long long blk, frag (&fa)[TR][2], frag (&fb)[TCW][2]) {
#pragma unroll
        for (int u = 0; u < TR; ++u)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                fa[u][ks] = *(const frag*)(dz + ((size_t)blk * DZW + 32 * (wr * TR + u) + i) * 32 + ks * 16 + 8 * h);
#pragma unroll
        for (int v = 0; v < TCW; ++v)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                fb[v][ks] = *(const frag*)(x + ((size_t)blk * jb.xw + 32 * (wc * TCW + v) + i) * 32 + ks * 16 + 8 * h);
    };
    auto step = [&](const frag (&fa)[TR][2], const frag (&fb)[TCW][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int u = 0; u < TR; ++u) {
                if (wc == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[u] += (float)fa[u][ks][e];
                }
#pragma unroll
                for (int v = 0; v < TCW; ++v) acc[u][v] = P::mfma(fa[u][ks], fb[v][ks], acc[u][v]);
            }
        }
    };
    frag fa0[TR][2], fb0[TCW][2], fa1[TR][2], fb1[TCW][2];
    long long blk = c;
    if (blk < nblocks) load(blk, fa0, fb0);
    int since_sync = 0;
    while (blk < nblocks) {
        if (sync_every > 0 && ++since_sync >= sync_every) {      // all four waves have the same trip count: uniform
            since_sync = 0;
            __builtin_amdgcn_s_barrier();                            // (loads already requested stay in flight across it)
        }
        // (scheduling fences keep the two halves of the loop apart: request block n + 1, then the 32 MFMAs of block n.
        //  Free-running on purpose: a workgroup barrier every 2 / 4 / 8 blocks removes the redundant HBM reads -- the waves
        //  that share fragments stay in step -- but costs 15-45 %: tools/experiments/README.md)
        const long long n1 = blk + jb.kch;
        if (n1 < nblocks) load(n1, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        step(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (n1 >= nblocks) break;
        const long long n2 = n1 + jb.kch;
        if (n2 < nblocks) load(n2, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        step(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        blk = n2;
    }
    // D tile: lane (h, j) holds rows tile_row(r, h), column j
    float* dw = jb.dw + (size_t)c * pstride;
#pragma unroll
    for (int u = 0; u < TR; ++u) {
#pragma unroll
        for (int v = 0; v < TCW; ++v) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dw[(size_t)(32 * (wr * TR + u) + tile_row(r, h)) * jb.xw + 32 * (wc * TCW + v) + i] = acc[u][v][r];
        }
        const float rowsum = bsum[u] + __shfl_xor(bsum[u], 32);        // the two lane halves hold samples 8h .. 8h + 7 of each k-step
        if (wc == 0 && h == 0) jb.db[(size_t)c * pstride + 32 * (wr * TR + u) + i] = rowsum;
    }
}

// VIEWS: with the three jobs of the view-dependent head's colour branch (rows = W / 2).  A kernel of its own, its four job
// shapes as out-of-line functions: inlined side by side, hipcc 7.2 allocates 256 registers + scratch to all of them (the plain
// kernel's two shapes get 442 and no scratch).
template <int DZW, int TCW>
__device__ __attribute__((noinline)) void trunk_wgrad_job_call(const WgradJob jb, const long long nblocks, const long long pstride, const int sync_every, int c) {
    trunk_wgrad_job<DZW, TCW>(jb, nblocks, pstride, sync_every, c);
}
template <class A, bool VIEWS = false>
__global__ void __launch_bounds__(256, 1) trunk_wgrad(const WgradArgs a) {
    int j = 0;
    for (int k = 1; k < a.njobs; ++k)
        if ((int)blockIdx.x >= a.job[k].wg0) j = k;           // jobs are listed in grid order
    const WgradJob jb = a.job[j];
    const int c = (int)blockIdx.x - jb.wg0;
    if constexpr (!VIEWS) {
        if (jb.xw == A::W) trunk_wgrad_job<A::W, A::W / 64>(jb, a.nblocks, a.pstride, a.sync_every, c);          // hidden-to-hidden layer
        else trunk_wgrad_job<A::W, 1>(jb, a.nblocks, a.pstride, a.sync_every, c);                                 // 64 columns: encoding / head
    } else if (jb.rows == A::W) {
        if (jb.xw == A::W) trunk_wgrad_job_call<A::W, A::W / 64>(jb, a.nblocks, a.pstride, a.sync_every, c);
        else trunk_wgrad_job_call<A::W, 1>(jb, a.nblocks, a.pstride, a.sync_every, c);
    } else {                                                                     // view-dependent head: rows = W / 2
        if (jb.xw == A::W) trunk_wgrad_job_call<A::W / 2, A::W / 64>(jb, a.nblocks, a.pstride, a.sync_every, c); // d z_v^T h_{D-1}
        else trunk_wgrad_job_call<A::W / 2, 1>(jb, a.nblocks, a.pstride, a.sync_every, c);                        // d z_v^T enc(direction), hv^T d raw
    }
}

template <class A, bool VIEWS = false>
static hipError_t launch_trunk_wgrad(const WgradArgs& a, hipStream_t stream) {
    if (a.njobs <= 0 || a.nwg <= 0 || a.nblocks <= 0) return hipSuccess;
    for (int j = 0; j < a.njobs; ++j)
        if ((a.job[j].xw != 64 && a.job[j].xw != A::W) || (a.job[j].rows != A::W && !(VIEWS && a.job[j].rows == A::W / 2)) || a.job[j].kch < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL((trunk_wgrad<A, VIEWS>), dim3(a.nwg), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// weight gradients (fp32 mode)
// ------------------------------------------------------------------------------------------
// The same jobs on the fp32 mode's row-major arrays ([sample][feature]: what trunk_fwd_train / trunk_bwd write in that mode),
// v_mfma_f32_32x32x2_f32 with the SAMPLE as the contraction index: lane (i, k = lane >> 5) of an A operand holds dz[m0 + k][row],
// of a B operand x[m0 + k][col] -- one dword per lane and MFMA.  A lane therefore loads CR consecutive features of dz and CC of x
// per sample (16 / 8 / 4 bytes: a wave half reads 32 CR consecutive floats of one row) and feeds CR x CC MFMAs from them: MFMA
// (c, d) contracts rows {CR i + c} with columns {CC i + d} -- any assignment of features to the 32 rows of a tile is as good
// as any other, the epilogue writes each accumulator where it belongs.  Wave (wr, wc) owns rows wr 32 CR .. and columns wc 32 CC
// ..: 2 x 2 waves cover a [64 CR] x [64 CC] product (CR = W / 64; CC = W / 64 for a hidden-to-hidden layer, 1 for the 64-column
// ones).  16 MFMAs of 64 cycles per two samples and wave at W = 256 against 2 KB of loads per workgroup: matrix-pipe-bound (64
// flop per byte).  Workgroup c of a job takes a contiguous range of samples; the next eight samples' operands are requested
// before the current eight's MFMAs.  Same record layout of the partial sums as the bf16 kernel.
template <int N> struct fvec { typedef float type __attribute__((ext_vector_type(N))); };
template <> struct fvec<1> { typedef float type; };
template <int N>
__device__ __forceinline__ float fvec_get(const typename fvec<N>::type& v, int k) {
    if constexpr (N == 1) return v; else return v[k];
}
template <int DZW, int CR, int CC>       // DZW: features of dz (its row length); CR = DZW / 64
__device__ __forceinline__ void trunk_wgrad_f32_job(const WgradArgs& a, const WgradJob& jb, int c) {
    constexpr int W = DZW;
    typedef typename fvec<CR>::type vr;
    typedef typename fvec<CC>::type vc;
#ifndef NRN_WGF_G
#define NRN_WGF_G 4
#endif
    constexpr int G = NRN_WGF_G;                      // sample pairs per group (loads in flight: two groups)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int k = lane >> 5, i = lane & 31;
    const long long M = a.nblocks;                    // fp32 mode: samples
    long long per = (M + jb.kch - 1) / jb.kch;
    per = (per + 2 * G - 1) / (2 * G) * (2 * G);
    const long long m_begin = (long long)c * per;
    const long long m_end = (m_begin + per < M) ? m_begin + per : M;
    const float* dz = (const float*)jb.dz + wr * 32 * CR + CR * i;
    const float* x = (const float*)jb.x + wc * 32 * CC + CC * i;
    const int xw = jb.xw;
    f32x16 acc[CR][CC];
    float bsum[CR];
#pragma unroll
    for (int u = 0; u < CR; ++u) {
        bsum[u] = 0.0f;
#pragma unroll
        for (int v = 0; v < CC; ++v) acc[u][v] = f32x16{};
    }
    auto load = [&](long long m0, vr (&fa)[G], vc (&fb)[G]) {          // a whole group inside the range
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const size_t row = (size_t)(m0 + 2 * g + k);
            fa[g] = *(const vr*)(dz + row * W);
            fb[g] = *(const vc*)(x + row * xw);
        }
    };
    // the array's own end inside a group (a workgroup's range is a multiple of 2 G samples, so this is the last group of the last
    // workgroup only): the row of the last sample is read instead and dz zeroed -- x is finite there, the product and the row sum vanish
    auto load_tail = [&](long long m0, vr (&fa)[G], vc (&fb)[G]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const long long m = m0 + 2 * g + k;
            const bool in = m < M;
            const size_t row = (size_t)(in ? m : M - 1);
            const vr va = *(const vr*)(dz + row * W);
            fb[g] = *(const vc*)(x + row * xw);
            if constexpr (CR == 1) fa[g] = in ? va : 0.0f;
            else {
#pragma unroll
                for (int e = 0; e < CR; ++e) fa[g][e] = in ? va[e] : 0.0f;
            }
        }
    };
    auto step = [&](const vr (&fa)[G], const vc (&fb)[G]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int u = 0; u < CR; ++u) {
                if (wc == 0) bsum[u] += fvec_get<CR>(fa[g], u);
#pragma unroll
                for (int v = 0; v < CC; ++v) acc[u][v] = PolF32::mfma(fvec_get<CR>(fa[g], u), fvec_get<CC>(fb[g], v), acc[u][v]);
            }
        }
    };
    // (one loop exit: with the bf16 kernel's two-step loop -- an exit after each half -- hipcc 7.2 keeps a second copy of the 256
    //  accumulator registers and spills 500 of them)
    vr fa0[G], fa1[G];
    vc fb0[G], fb1[G];
    const long long m_full = m_begin + (m_end > m_begin ? (m_end - m_begin) / (2 * G) * (2 * G) : 0);
    long long m0 = m_begin;
    if (m0 < m_full) load(m0, fa0, fb0);
    // (vmcnt(0) once: otherwise the first group's loads are "pending" on one path into the loop and hipcc makes every iteration's
    //  first MFMAs wait for the loads issued right in front of them)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (; m0 < m_full; m0 += 4 * G) {
        const bool more1 = m0 + 2 * G < m_full, more2 = m0 + 4 * G < m_full;
        if (more1) load(m0 + 2 * G, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        step(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (more2) load(m0 + 4 * G, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (more1) step(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (m_full < m_end) {
        load_tail(m_full, fa0, fb0);
        step(fa0, fb0);
    }
    // D tile of MFMA (u, v): lane (k, i) register r = (row CR tile_row(r, k) + u, column CC i + v) of this wave's block
    float* dw = jb.dw + (size_t)c * a.pstride;
#pragma unroll
    for (int u = 0; u < CR; ++u) {
#pragma unroll
        for (int v = 0; v < CC; ++v) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dw[(size_t)(wr * 32 * CR + CR * tile_row(r, k) + u) * xw + wc * 32 * CC + CC * i + v] = acc[u][v][r];
        }
        const float rowsum = bsum[u] + __shfl_xor(bsum[u], 32);        // the two lane halves hold the two samples of each pair
        if (wc == 0 && k == 0) jb.db[(size_t)c * a.pstride + wr * 32 * CR + CR * i + u] = rowsum;
    }
}

template <class A, bool VIEWS = false>
__global__ void __launch_bounds__(256, 1) trunk_wgrad_f32(const WgradArgs a) {
    int j = 0;
    for (int q = 1; q < a.njobs; ++q)
        if ((int)blockIdx.x >= a.job[q].wg0) j = q;           // jobs are listed in grid order
    const WgradJob jb = a.job[j];
    const int c = (int)blockIdx.x - jb.wg0;
    if (jb.rows == A::W) {
        if (jb.xw == A::W) trunk_wgrad_f32_job<A::W, A::W / 64, A::W / 64>(a, jb, c);      // hidden-to-hidden layer
        else trunk_wgrad_f32_job<A::W, A::W / 64, 1>(a, jb, c);                             // 64 columns: encoding / head
    } else if constexpr (VIEWS) {                                                           // view-dependent head: rows = W / 2
        if (jb.xw == A::W) trunk_wgrad_f32_job<A::W / 2, A::W / 128, A::W / 64>(a, jb, c);
        else trunk_wgrad_f32_job<A::W / 2, A::W / 128, 1>(a, jb, c);
    }
}

template <class A, bool VIEWS = false>
static hipError_t launch_trunk_wgrad_f32(const WgradArgs& a, hipStream_t stream) {
    if (a.njobs <= 0 || a.nwg <= 0 || a.nblocks <= 0) return hipSuccess;
    for (int j = 0; j < a.njobs; ++j)
        if ((a.job[j].xw != 64 && a.job[j].xw != A::W) || (a.job[j].rows != A::W && !(VIEWS && a.job[j].rows == A::W / 2)) || a.job[j].kch < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL((trunk_wgrad_f32<A, VIEWS>), dim3(a.nwg), dim3(256), 0, stream, a);
    return hipGetLastError();
}

template <class P, class A, int WAVES, bool BWD, bool VIEWS = false>
static hipError_t launch_trunk_train(const TrunkArgs& a, int num_cus, hipStream_t stream) {
    constexpr int NTILES = BWD ? PlanB<P, A, VIEWS>::NTILES : Plan<P, A, false, VIEWS>::NTILES;
    const size_t lds = (size_t)RING * P::UNIT_BYTES + (size_t)NTILES * 32 * sizeof(float);
    void (*kern)(const TrunkArgs) = nullptr;
    if constexpr (BWD) kern = trunk_bwd<P, A, WAVES, VIEWS>; else kern = trunk_fwd_train<P, A, WAVES, VIEWS>;
    if (VIEWS && (!a.dirs || !a.hv || (BWD && !a.d_pre_v) || (P::KH != 1 && !a.hv_mask))) return hipErrorInvalidValue;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const long long nblocks = (long long)a.n_rays * ((a.S + 31) / 32);
    const long long ntiles = (nblocks + WAVES - 1) / WAVES;
    if (ntiles <= 0) return hipSuccess;
    const long long resident = (long long)num_cus * ((P::KH == 1) ? 1 : 8 / WAVES);
    const int grid = (int)(ntiles < resident ? ntiles : resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace nrn
