// nrnerf_net_x16.h -- the trunk-only network kernel on v_mfma_f32_16x16x32_{bf16,f16}.
//
// Why a second tiling of the same layers: the trunk kernels of nrnerf_net_mb.h are power-bound, and under the
// socket's cap the 16x16x32 MFMA sustains 10-18 % more flops than the 32x32x16 one on the same operands
// (tools/probes/mfma_shape_power.hip, profiles/r04_mfma_shape_power.txt): per 32 x 32 x 256 of work the register file sees 512 accesses
// instead of 640 (the accumulator is a quarter the size).  Everything else is kept: the L2 -> LDS weight ring (WRing), one wave per
// SIMD with 64 samples, every weight fragment read from LDS once per wave and fed to four MFMAs (four 16-sample blocks; the mb
// kernel: two 32-sample blocks), activations handed from layer to layer in registers (PlanX16, nrnerf_plan.h: two consecutive D
// tiles of a lane ARE the next layer's B operand of one k-step).
//
// Scope: the pass the split-bender path spends its time in -- positional encoding of READY-MADE points (NetArgs::pts4: the bent
// points of the stand-alone bender kernel), the 8 x 256 trunk with its skip connection, the 4/5-channel head; bf16 or f16; no
// bender, no view-dependent head, no detail outputs (those calls keep the kernels of nrnerf_net_mb.h).
#pragma once
#include "nrnerf_net_impl.h"
#include "nrnerf_composite_ray.h"

namespace nrn {

template <class P> struct X16;
template <> struct X16<PolBF16> {
    static __device__ __forceinline__ f32x4 mfma(PolBF16::frag a, PolBF16::frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct X16<PolF16> {
    static __device__ __forceinline__ f32x4 mfma(PolF16::frag a, PolF16::frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// two D tiles (features 32 s + 4 g + i and 32 s + 16 + 4 g + i of this lane's sample) -> the B operand of k-step s, through relu:
// 4 x v_cvt_pk + 4 x v_pk_max_i16 (as pack16)
template <class P>
__device__ __forceinline__ typename P::frag x16_pack(const f32x4& d0, const f32x4& d1) {
    typedef typename P::frag2 F2;
    u32x4 w;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x2 t = (k < 2) ? f32x2{d0[2 * k], d0[2 * k + 1]} : f32x2{d1[2 * (k - 2)], d1[2 * (k - 2) + 1]};
        s16x2 q = __builtin_bit_cast(s16x2, __builtin_convertvector(t, F2));
        q = __builtin_elementwise_max(q, (s16x2)(short)0);
        w[k] = __builtin_bit_cast(unsigned, q);
    }
    return __builtin_bit_cast(typename P::frag, w);
}

#ifndef NRN_X16_NB
#define NRN_X16_NB 4          // 16-sample blocks per wave
#endif

// One dense layer.  Stream order (PlanX16 = place_fragments): tile pairs (2 p, 2 p + 1), their k-steps interleaved; an odd last tile
// (the head) alone.  Per fragment NB MFMAs (one per block).  Two accumulator sets: pair p runs in set p & 1 while the epilogue of
// pair p - 1 (pack -> out[b][p - 1], NB chunks of ~8 VALU) is issued among its first k-steps; the last pair's epilogue follows the
// layer.  The first MFMA of a chain takes the bias as its C operand (no copies).
template <class P0, class P1, class PL, int LI, int NS0, int NS1, int NB, class ST, class IN0, class IN1, class EPI>
__device__ __forceinline__ void dense_x16(ST& st, const __attribute__((address_space(3))) f32x4* bias_lane, const IN0 (&in0)[NB], const IN1 (&in1)[NB],
                                          EPI&& epi) {
    constexpr LayerSpec spec = PL::TB.layers[LI];
    static_assert(spec.ns == NS0 + NS1, "k-step count mismatch between kernel and plan");
    constexpr int NS = NS0 + NS1, NT = spec.nt, Q = NT * NS, PF = P1::PF;
    using SQ = SeqPos<NT, NS>;
    constexpr int G0 = PL::TB.tiles[spec.tile0].gbase;
    typename P1::frag a[PF];
    auto load = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int s = SQ::slab(q);
        if constexpr (s < NS0) a[q % PF] = __builtin_bit_cast(typename P1::frag, st.template frag<P0, G0 + q>());
        else a[q % PF] = st.template frag<P1, G0 + q>();
    };
    static_for<0, (PF < Q ? PF : Q)>([&](auto qc) { load(qc); });
    f32x4 acc[2][2][NB];          // [set][tile of the pair][block]
    f32x4 bias[2];
    static_for<0, Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int t = SQ::tile(q), s = SQ::slab(q);
        constexpr int p = t >> 1, u = t & 1, set = p & 1;
        if constexpr (s == 0) bias[u] = bias_lane[(spec.tile0 + t) * 4];         // [tile][16 rows]: this lane's rows 4 g .. 4 g + 3
        st.template ready<(Q - 1 - q < PF - 1) ? Q - 1 - q : PF - 1>(a[q % PF]);
        const typename P1::frag cur = a[q % PF];
        if constexpr (q + PF < Q) load(std::integral_constant<int, q + PF>{});
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            const f32x4 c = (s == 0) ? bias[u] : acc[set][u][b];
            if constexpr (s < NS0) acc[set][u][b] = X16<P0>::mfma(__builtin_bit_cast(typename P0::frag, cur), in0[b][s], c);
            else acc[set][u][b] = X16<P1>::mfma(cur, in1[b][s - NS0], c);
        });
        // epilogue of the previous pair: block k after the second tile's MFMAs of k-step k
        if constexpr (p > 0 && u == 1 && s < NB) {
            epi(std::integral_constant<int, p - 1>{}, std::integral_constant<int, s>{}, acc[set ^ 1][0][s], acc[set ^ 1][1][s]);
        }
        // a layer with fewer k-steps than blocks (the encoding layer: 2): the rest of that epilogue at its last k-step
        if constexpr (p > 0 && u == 1 && s == NS - 1 && NS < NB) {
            static_for<NS, NB>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                epi(std::integral_constant<int, p - 1>{}, kc, acc[set ^ 1][0][k], acc[set ^ 1][1][k]);
            });
        }
        if constexpr (q == Q - 1) {              // the last pair (or the lone head tile)
            static_for<0, NB>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (u == 1) epi(std::integral_constant<int, p>{}, kc, acc[set][0][k], acc[set][1][k]);
                else epi(std::integral_constant<int, p>{}, kc, acc[set][0][k], acc[set][0][k]);
            });
        }
    });
}

template <class P, class A, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 1) net_kernel_x16(const NetArgs a) {
    using PL = PlanX16<P, A>;
    using PE = PolF16;                                                // the encoding's operands are f16 in both modes
    using frag = typename P::frag;
    using efrag = typename PE::frag;
    constexpr int NB = NRN_X16_NB, NS_H = PL::NS_H, NS_E = PL::NS_E;
    static_assert(A::L == 10, "the encoding's slot layout below is spelt out for ten frequencies (x16_enc_col)");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* bias_lds = (float*)(smem + RING * P::UNIT_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, n = lane & 15;
    for (int i = tid; i < PL::NTILES * 16; i += WAVES * 64) bias_lds[i] = a.bias[i];
    __syncthreads();
    const __attribute__((address_space(3))) f32x4* bias_lane = (const __attribute__((address_space(3))) f32x4*)(bias_lds + 4 * g);
    asm volatile("" : "+v"(bias_lane));
    WRing<P, WAVES, PL::NUP> st;
    st.init(a.wstream, ring, wave, lane);

    const int S = a.S;
    const int bpr = (S + 15) >> 4;
    const long long nblocks = (long long)a.n_rays * bpr;
    const long long per_wg = (long long)WAVES * NB;
    for (long long b0 = (long long)blockIdx.x * per_wg; b0 < nblocks; b0 += (long long)gridDim.x * per_wg) {
        size_t so[NB];
        bool ok[NB];
        efrag enc[NB][NS_E];
        // ---- points and their positional encoding, in B-operand order (x16_enc_col): slots 2 i, 2 i + 1 of this lane's group =
        //      (sin, cos) of pair m = 4 i + g (frequency m / 3, coordinate m % 3); groups 2, 3: slots 14, 15 = x, y | z, 0
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            const long long blk_raw = b0 + (long long)wave * NB + b;
            const bool blk_ok = blk_raw < nblocks;
            const long long blk = blk_ok ? blk_raw : nblocks - 1;
            const int ray = (int)(blk / bpr);
            const int sidx = (int)(blk % bpr) * 16 + n;
            ok[b] = blk_ok && sidx < S;
            so[b] = (size_t)ray * S + (sidx < S ? sidx : S - 1);
            const f32x4 q4 = *(const f32x4*)(a.pts4 + so[b] * 4);
            const float prev[3] = {q4[0] * 0.15915494309189535f, q4[1] * 0.15915494309189535f, q4[2] * 0.15915494309189535f};
            float ev[16];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = 4 * i + g;
                const int f = m / 3, c = m - 3 * f;
                const float xr = c == 0 ? prev[0] : (c == 1 ? prev[1] : prev[2]);
                const float r = __builtin_amdgcn_fractf(xr * (float)(1 << f));          // (power-of-two scaling: exact)
                ev[2 * i] = __builtin_amdgcn_sinf(r);
                ev[2 * i + 1] = __builtin_amdgcn_cosf(r);
            }
            if (g >= 2) {                       // pair 28 + g does not exist: the identity columns
                ev[14] = (g == 2) ? q4[0] : q4[2];
                ev[15] = (g == 2) ? q4[1] : 0.0f;
            }
#pragma unroll
            for (int s = 0; s < NS_E; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) enc[b][s][e] = (_Float16)ev[8 * s + e];
        });

        frag ha[NB][NS_H], hb[NB][NS_H];
        frag none[NB][1];                   // (the second source of a layer that has one: never indexed)
        auto keep = [&](auto& out) {
            return [&](auto pc, auto kc, const f32x4& d0, const f32x4& d1) {
                out[decltype(kc)::value][decltype(pc)::value] = x16_pack<P>(d0, d1);
            };
        };
        dense_x16<PE, P, PL, 0, NS_E, 0, NB>(st, bias_lane, enc, none, keep(ha));
        static_for<1, A::D>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr bool skip = (i - 1 == A::SKIP);
            if constexpr (i % 2 == 1) {
                if constexpr (skip) dense_x16<PE, P, PL, i, NS_E, NS_H, NB>(st, bias_lane, enc, ha, keep(hb));
                else dense_x16<P, P, PL, i, NS_H, 0, NB>(st, bias_lane, ha, none, keep(hb));
            } else {
                if constexpr (skip) dense_x16<PE, P, PL, i, NS_E, NS_H, NB>(st, bias_lane, enc, hb, keep(ha));
                else dense_x16<P, P, PL, i, NS_H, 0, NB>(st, bias_lane, hb, none, keep(ha));
            }
        });
        // ---- head: one tile; group 0 holds channels 0..3 (rgb, sigma) of its sample, group 1 channel 4 in its first register
        f32x4 raw[NB];
        auto take = [&](auto, auto kc, const f32x4& d0, const f32x4&) { raw[decltype(kc)::value] = d0; };
        if constexpr ((A::D - 1) % 2 == 1) dense_x16<P, P, PL, PL::L_HEAD, NS_H, 0, NB>(st, bias_lane, hb, none, take);
        else dense_x16<P, P, PL, PL::L_HEAD, NS_H, 0, NB>(st, bias_lane, ha, none, take);
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            if (ok[b] && g == 0) {
                *(f32x4*)(a.raw4 + so[b] * 4) = raw[b];
                if (a.raw_out) {
                    float* ro = a.raw_out + so[b] * a.raw_ch;
                    ro[0] = raw[b][0]; ro[1] = raw[b][1]; ro[2] = raw[b][2]; ro[3] = raw[b][3];
                }
            }
            if (ok[b] && g == 1 && a.raw_out && a.raw_ch > 4) a.raw_out[so[b] * a.raw_ch + 4] = raw[b][0];
        });
        static_for<PL::NUNITS, PL::NUP>([&](auto uc) { st.template advance<decltype(uc)::value>(); });
    }
    st.drain();
}

template <class P, class A>
static hipError_t launch_net_x16_t(const NetArgs& a, int num_cus, hipStream_t stream) {
    constexpr int WAVES = 4;
    using PL = PlanX16<P, A>;
    if (!a.pts4 || !a.raw4 || a.fuse_on || a.S < 1) return hipErrorInvalidValue;
    const size_t lds = (size_t)RING * P::UNIT_BYTES + (size_t)PL::NTILES * 16 * sizeof(float);
    auto kern = net_kernel_x16<P, A, WAVES>;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const long long nblocks = (long long)a.n_rays * ((a.S + 15) / 16);
    const long long ntiles = (nblocks + WAVES * NRN_X16_NB - 1) / (WAVES * NRN_X16_NB);
    if (ntiles <= 0) return hipSuccess;
    const int grid = (int)(ntiles < num_cus ? ntiles : num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace nrn
