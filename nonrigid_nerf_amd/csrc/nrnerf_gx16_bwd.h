// nrnerf_gx16_bwd.h -- backward-data of a NON-COMPILED trunk (plain head) on the width-class kernel's dataflow (nrnerf_gx16.h): the layers
// in reverse with transposed weights (plan kinds GX_BHEAD / GX_BHID / GX_BSKIP / GX_BIN, nrnerf_gx16_plan.h), dense_x16 on the run-time
// LDS weight ring, activations' gradients handed from layer to layer in registers.  Reference: what autograd derives from NeRF.forward
// (run_nerf_helpers.py:272-306) for d raw -> d (pre-activations) and d (encoding); training of any --netdepth / --netwidth,
// train.py:1004-1010.  Replaces the backward-data mode of the run-time-parameterised kernel (nrnerf_generic.h, mode 2) where it applies
// (bf16, plain head, no latent input columns): 0.07 of the matrix pipe's peak there -- activations in LDS, every wave pulling its own
// weights -- 2.45 of the 4.5 ms of a 2048-ray step at width 192 (tools/generic_step_sequence.py).
//
// Per layer i = D - 1 .. 0 the epilogue of a tile pair masks the incoming d h_i with "h_i passed the relu" -- ONE byte per lane and pair,
// written by the forward kernel beside the activations (GxArgs::relu_bits: [layer][16-sample block][lane][WC / 32] bytes, bit e = element e
// of the lane's packed fragment) and fetched with one 8- or 16-byte load per block and layer --, stores d z_i as [sample][save_w] bf16 rows
// (what nrnerf_tn_products contracts over) and keeps it as the next layer's B operand.  The encoding's gradient (the four tiles in front of
// the hidden ones in the two layers that read the encoding) goes to memory in the reference's column order, fp32.
#pragma once
#include "nrnerf_gx16.h"
#include "nrnerf_gx16_bwd_api.h"

namespace nrn {

// two D tiles -> the B operand of the next k-step, masked by 8 bits (bit e = element e kept), no relu: 4 x v_cvt_pk + 4 x v_and
template <class P>
__device__ __forceinline__ typename P::frag x16_pack_masked(const f32x4& d0, const f32x4& d1, unsigned bits) {
    typedef typename P::frag2 F2;
    u32x4 w;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x2 t = (k < 2) ? f32x2{d0[2 * k], d0[2 * k + 1]} : f32x2{d1[2 * (k - 2)], d1[2 * (k - 2) + 1]};
        const unsigned q = __builtin_bit_cast(unsigned, __builtin_convertvector(t, F2));
        const unsigned m = (((bits >> (2 * k)) & 1u) ? 0x0000ffffu : 0u) | (((bits >> (2 * k + 1)) & 1u) ? 0xffff0000u : 0u);
        w[k] = q & m;
    }
    return __builtin_bit_cast(typename P::frag, w);
}

template <class P, int WC, int NB>
__global__ void __launch_bounds__(4 * 64, 1) gx16_bwd_kernel(const GxBwdArgs a) {
    constexpr int WAVES = 4;
    using frag = typename P::frag;
    using PBHEAD = PlanGX<WC, GX_BHEAD>;
    using PBHID = PlanGX<WC, GX_BHID>;
    using PBSKIP = PlanGX<WC, GX_BSKIP>;
    using PBIN = PlanGX<WC, GX_BIN>;
    constexpr int NS_H = WC / 32;
    constexpr int PF = (WC > 256) ? 4 : 8;
    constexpr int MW = (NS_H + 3) / 4;           // dwords of relu bits per lane, block and layer

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* bias_lds = (float*)(smem + RING * P::UNIT_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, n = lane & 15;
    for (int i = tid; i < a.n_bias_tiles * 16; i += WAVES * 64) bias_lds[i] = a.bias[i];
    __syncthreads();
    typedef const __attribute__((address_space(3))) f32x4* BP;
    const BP bias_lane0 = (BP)(bias_lds + 4 * g);
    WRingRT<P, WAVES> st;
    st.init(a.wstream, ring, wave, lane);

    const int S = a.S, D = a.depth, skip = a.skip;
    const int bpr = (S + 15) >> 4;
    const long long nblocks = (long long)a.n_rays * bpr;
    const long long per_wg = (long long)WAVES * NB;
    for (long long b0 = (long long)blockIdx.x * per_wg; b0 < nblocks; b0 += (long long)gridDim.x * per_wg) {
        unsigned so[NB];
        int blk[NB];                // (< 2^31 blocks: the launcher checks)
        bool ok[NB];
        frag draw[NB][1];
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            const long long blk_raw = b0 + (long long)wave * NB + b;
            const bool blk_ok = blk_raw < nblocks;
            blk[b] = (int)(blk_ok ? blk_raw : nblocks - 1);
            const int ray = (int)(blk[b] / bpr), bir = (int)(blk[b] % bpr);
            const int sidx = bir * 16 + n;
            ok[b] = blk_ok && sidx < S;
            so[b] = (unsigned)ray * (unsigned)S + (unsigned)(sidx < S ? sidx : S - 1);
            const f32x4 gr = *(const f32x4*)(a.d_raw4 + (size_t)so[b] * 4);
            // d raw as the first layer's operand: position 8 g + e = channel e of group 0; a sample beyond the ray's end contributes nothing
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = (g == 0 && e < 4 && ok[b]) ? gr[e < 4 ? e : 0] : 0.0f;
                if constexpr (std::is_same_v<P, PolBF16>) draw[b][0][e] = (__bf16)v;
                else draw[b][0][e] = (_Float16)v;
            }
        });
        frag ha[NB][NS_H], hb[NB][NS_H];
        frag none[NB][1];
        unsigned mb[NB][MW];        // relu bits of the layer whose d h the dense call in flight produces
        int lcur = 0;               // ... that layer's index
        auto fetch_bits = [&](int layer) {
            lcur = layer;
            static_for<0, NB>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                const unsigned* p = (const unsigned*)((const char*)a.relu_bits + (((size_t)layer * nblocks + (size_t)blk[b]) * 64 + lane) * (size_t)(4 * MW));
#pragma unroll
                for (int w = 0; w < MW; ++w) mb[b][w] = p[w];
            });
        };
        // hidden tile pair HP (0-based among the hidden tiles) of block K: mask, keep, store d z
        auto hidden = [&](auto& out, auto hpc, auto kc, const f32x4& d0, const f32x4& d1) {
            constexpr int hp = decltype(hpc)::value, k = decltype(kc)::value;
            const unsigned bits = (mb[k][hp >> 2] >> (8 * (hp & 3))) & 0xffu;
            const frag v = x16_pack_masked<P>(d0, d1, bits);
            out[k][hp] = v;
            const int col = 32 * hp + 4 * g;
            if (ok[k] && col < a.save_w) {
                const u32x4 w = __builtin_bit_cast(u32x4, v);
                typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
                unsigned short* row = (unsigned short*)a.d_pre + ((size_t)lcur * a.save_stride + (size_t)so[k] * a.save_w);
                *(u32x2_*)(row + col) = u32x2_{w[0], w[1]};
                if (col + 16 < a.save_w) *(u32x2_*)(row + col + 16) = u32x2_{w[2], w[3]};
            }
        };
        // encoding tile pair EP (positions 32 EP + ..) of block K -> d_enc rows, the reference's columns
        auto enc_out = [&](float* dst, auto epc, auto kc, const f32x4& d0, const f32x4& d1) {
            constexpr int ep = decltype(epc)::value, k = decltype(kc)::value;
            if (!ok[k]) return;
            float* row = dst + (size_t)so[k] * a.enc_w;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c0 = gx_enc_col_of_pos(a.L, 32 * ep + 4 * g + i), c1 = gx_enc_col_of_pos(a.L, 32 * ep + 16 + 4 * g + i);
                if (c0 >= 0) row[c0] = d0[i];
                if (c1 >= 0) row[c1] = d1[i];
            }
        };
        BP bl = bias_lane0;
        asm volatile("" : "+v"(bl));
        // ---- d h_{D-1} = output_linear^T d raw
        fetch_bits(D - 1);
        dense_x16<P, P, PBHEAD, 0, 1, 0, NB, PF>(st, bl, draw, none, [&](auto pc, auto kc, const f32x4& d0, const f32x4& d1) { hidden(ha, pc, kc, d0, d1); });
        st.template end_layer<PBHEAD>();
        bl += PBHEAD::NT * 4;
        // ---- layers D - 1 .. 1: d x_i = pts_linears[i]^T d z_i; two at a time (which array holds d z is then a compile-time fact)
        auto layer = [&](int i, auto& in, auto& out) __attribute__((always_inline)) {
            asm volatile("" : "+v"(bl));
            fetch_bits(i - 1);
            if (i - 1 == skip) {
                dense_x16<P, P, PBSKIP, 0, NS_H, 0, NB, PF>(st, bl, in, none, [&](auto pc, auto kc, const f32x4& d0, const f32x4& d1) {
                    constexpr int p = decltype(pc)::value;
                    if constexpr (p < 2) enc_out(a.d_enc1, pc, kc, d0, d1);
                    else hidden(out, std::integral_constant<int, p - 2>{}, kc, d0, d1);
                });
                st.template end_layer<PBSKIP>();
                bl += PBSKIP::NT * 4;
            } else {
                dense_x16<P, P, PBHID, 0, NS_H, 0, NB, PF>(st, bl, in, none, [&](auto pc, auto kc, const f32x4& d0, const f32x4& d1) { hidden(out, pc, kc, d0, d1); });
                st.template end_layer<PBHID>();
                bl += PBHID::NT * 4;
            }
        };
        int i = D - 1;
        for (; i - 1 >= 1; i -= 2) {
            layer(i, ha, hb);
            layer(i - 1, hb, ha);
        }
        // ---- the encoding's gradient through pts_linears[0]
        auto first = [&](auto& hx) __attribute__((always_inline)) {
            asm volatile("" : "+v"(bl));
            dense_x16<P, P, PBIN, 0, NS_H, 0, NB, PF>(st, bl, hx, none, [&](auto pc, auto kc, const f32x4& d0, const f32x4& d1) { enc_out(a.d_enc0, pc, kc, d0, d1); });
            static_for<PBIN::NUNITS, PBIN::NUP>([&](auto uc) { st.template advance<decltype(uc)::value>(); });
        };
        if (i >= 1) {
            layer(i, ha, hb);
            first(hb);
        } else {
            first(ha);
        }
        st.rewind(a.wstream);
    }
    st.drain();
}

template <class P, int WC>
static hipError_t launch_gx16_bwd_t(const GxBwdArgs& a, int num_cus, hipStream_t stream) {
    // (two blocks per wave from width class 256 up: with four, the two 128-register d z arrays + the epilogue's masks and store addresses
    //  spilled 27-38 registers whatever the prefetch depth)
    constexpr int WAVES = 4, NB = (WC >= 256) ? 2 : 4;
    if (!a.d_raw4 || !a.relu_bits || !a.d_pre || !a.d_enc0 || a.S < 1 || a.depth < 1 || a.L < 0 || a.L > GX_MAX_L) return hipErrorInvalidValue;
    if (a.skip >= 0 && a.skip <= a.depth - 2 && !a.d_enc1) return hipErrorInvalidValue;
    if (a.save_w % 4 != 0 || a.save_w < 4 || a.save_w > WC || a.enc_w != 3 + 6 * a.L) return hipErrorInvalidValue;
    if ((long long)a.n_rays * a.S >= (1ll << 32) || (long long)a.n_rays * ((a.S + 15) / 16) >= (1ll << 31)) return hipErrorInvalidValue;
    const size_t lds = (size_t)RING * P::UNIT_BYTES + (size_t)a.n_bias_tiles * 16 * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = gx16_bwd_kernel<P, WC, NB>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return hipErrorUnknown;
    const long long bpr = (a.S + 15) / 16;
    const long long want = ((long long)a.n_rays * bpr + WAVES * NB - 1) / (WAVES * NB);
    if (want <= 0) return hipSuccess;
    const int grid = (int)(want < num_cus ? want : num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace nrn
