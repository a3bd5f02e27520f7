// nrnerf_net_mb_impl.h -- multi-block variant of the network kernel: every wave owns NB blocks of 32 samples that
// SHARE each weight fragment (one ds_read_b128 feeds NB MFMAs on independent accumulators).  4 waves per workgroup,
// one per SIMD, up to 512 registers each.  Per MFMA this halves (NB = 2) the LDS fragment reads, the LDS-DMA issues
// and the ring barriers of the 8-wave kernel in nrnerf_net_impl.h, at the price of having no second wave on the
// SIMD to cover a stall.  Same plan, same packed stream, same numerics; no view-dependent head (VIEWS stays on the
// single-block kernel).
#pragma once
#include "nrnerf_net_impl.h"

namespace nrn {

template <class T, int NB, int N>
struct Arr2 {     // [block][slab] register array with compile-time indices
    T v[NB][N > 0 ? N : 1];
};

// dense layer over NB blocks: acc[b] += A(t,s) * B_b(s); fragments prefetched PF deep; two accumulator sets
template <class P0, class P1, class PL, int LI, int NS0, int NS1, int NB, class ST, class IN0, class IN1, class EPI>
__device__ __forceinline__ void dense_mb(ST& st, const float* bias_lds, int h, const IN0& in0, const IN1& in1, EPI&& epi) {
    constexpr LayerSpec spec = PL::TB.layers[LI];
    static_assert(spec.ns == NS0 + NS1 && spec.split == 0, "slab count mismatch between kernel and plan");
    constexpr int NS = NS0 + NS1, Q = spec.nt * NS, PF = P1::PF;
    constexpr int G0 = PL::TB.tiles[spec.tile0].gbase;
    typename P1::frag a[PF];
    auto load = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int s = q % NS;
        if constexpr (s < NS0) a[q % PF] = __builtin_bit_cast(typename P1::frag, st.template frag<P0, G0 + q>());
        else a[q % PF] = st.template frag<P1, G0 + q>();
    };
    static_for<0, (PF < Q ? PF : Q)>([&](auto qc) { load(qc); });
    constexpr int DLY = (NS - 1 < NRN_EPI_DELAY) ? NS - 1 : NRN_EPI_DELAY;
    f32x16 accs[2][NB];
    {
        const f32x16 b0 = load_bias(bias_lds, spec.tile0, h);
        static_for<0, NB>([&](auto bc) { accs[0][decltype(bc)::value] = b0; });
        if constexpr (spec.nt > 1) {
            const f32x16 b1 = load_bias(bias_lds, spec.tile0 + 1, h);
            static_for<0, NB>([&](auto bc) { accs[1][decltype(bc)::value] = b1; });
        }
    }
    static_for<0, Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int t = q / NS, s = q % NS;
        const typename P1::frag cur = a[q % PF];
        if constexpr (q + PF < Q) load(std::integral_constant<int, q + PF>{});
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            if constexpr (s < NS0) accs[t & 1][b] = P0::mfma(__builtin_bit_cast(typename P0::frag, cur), in0.v[b][s], accs[t & 1][b]);
            else accs[t & 1][b] = P1::mfma(cur, in1.v[b][s - NS0], accs[t & 1][b]);
        });
        if constexpr (t > 0 && s == DLY) {
            static_for<0, NB>([&](auto bc) { epi(std::integral_constant<int, t - 1>{}, bc, accs[(t - 1) & 1][decltype(bc)::value]); });
            if constexpr (t + 1 < spec.nt) {
                const f32x16 bn = load_bias(bias_lds, spec.tile0 + t + 1, h);
                static_for<0, NB>([&](auto bc) { accs[(t + 1) & 1][decltype(bc)::value] = bn; });
            }
        }
        if constexpr (t == spec.nt - 1 && s == NS - 1)
            static_for<0, NB>([&](auto bc) { epi(std::integral_constant<int, t>{}, bc, accs[t & 1][decltype(bc)::value]); });
    });
}

// bender layer over NB blocks (3-term split product when SPLIT)
template <class PE, bool SPLIT, class PL, int LI, int NS, int NB, class ST, class ACT, class EPI>
__device__ __forceinline__ void dense_b_mb(ST& st, const float* bias_lds, int h, const ACT (&in)[NB], EPI&& epi) {
    constexpr LayerSpec spec = PL::TB.layers[LI];
    static_assert(spec.ns == NS && spec.split == (SPLIT ? 1 : 0), "bender layer mismatch between kernel and plan");
    static_for<0, spec.nt>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr TileInfo ti = PL::TB.tiles[spec.tile0 + t];
        const f32x16 bias = load_bias(bias_lds, spec.tile0 + t, h);
        f32x16 acc[NB], corr[NB];
        static_for<0, NB>([&](auto bc) { acc[decltype(bc)::value] = bias; corr[decltype(bc)::value] = f32x16{}; });
        static_for<0, NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (SPLIT) {
                const auto whi = st.template frag<PE, ti.gbase + 2 * s>();
                const auto wlo = st.template frag<PE, ti.gbase + 2 * s + 1>();
                static_for<0, NB>([&](auto bc) {
                    constexpr int b = decltype(bc)::value;
                    corr[b] = PE::mfma(wlo, in[b].hi[s], corr[b]);
                    corr[b] = PE::mfma(whi, in[b].lo[s], corr[b]);
                    acc[b] = PE::mfma(whi, in[b].hi[s], acc[b]);
                });
            } else {
                const auto w = st.template frag<PE, ti.gbase + s>();
                static_for<0, NB>([&](auto bc) { constexpr int b = decltype(bc)::value; acc[b] = PE::mfma(w, in[b].hi[s], acc[b]); });
            }
        });
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            if constexpr (SPLIT) acc[b] += corr[b] * (1.0f / PE::LO_SCALE);
            epi(tc, bc, acc[b]);
        });
    });
}

template <class P, class A, bool HAS_BEND, int WAVES, int NB>
__global__ void __launch_bounds__(WAVES * 64, 1) net_kernel_mb(const NetArgs a) {
    using PL = Plan<P, A, HAS_BEND, false>;
    using frag = typename P::frag;
    using PE = std::conditional_t<P::KH == 1, PolF32, PolF16>;
    using efrag = typename PE::frag;
    constexpr int KH = P::KH, SP = P::SP;
    constexpr int NS_ENC = PL::NS_ENC, NT_W = PL::NT_W, NH = NT_W * SP;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* bias_lds = (float*)(smem + RING * P::UNIT_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int j = lane & 31;

    for (int i = tid; i < PL::NTILES * 32; i += WAVES * 64) bias_lds[i] = a.bias[i];
    __syncthreads();
    WRing<P, WAVES, PL::NUP> st;
    st.init(a.wstream, ring, wave, lane);

    const int S = a.S;
    const int bpr = (S + 31) >> 5;
    const long long nblocks = (long long)a.n_rays * bpr;
    constexpr int TB = WAVES * NB;      // blocks per workgroup tile

    for (long long tile0 = (long long)blockIdx.x * TB; tile0 < nblocks; tile0 += (long long)gridDim.x * TB) {
        // ---- per-block state (all block indices are compile-time: everything stays in registers)
        int ray[NB], sc[NB];
        bool writer[NB];
        size_t so[NB];
        float p[NB][3], rig_mask[NB];
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            const long long blk = tile0 + (long long)wave * NB + b;
            const bool blk_ok = blk < nblocks;
            const long long bb = blk_ok ? blk : nblocks - 1;
            ray[b] = (int)(bb / bpr);
            const int sidx = (int)(bb % bpr) * 32 + j;
            sc[b] = sidx < S ? sidx : S - 1;
            writer[b] = blk_ok && sidx < S && h == 0;
            so[b] = (size_t)ray[b] * S + sc[b];
            const float* rp = a.rays + (size_t)ray[b] * a.ray_stride;
            float z;
            if (a.z) z = a.z[so[b]];
            else {
                const float t = lin01(sc[b], S);
                z = __fadd_rn(__fmul_rn(rp[6], __fsub_rn(1.0f, t)), __fmul_rn(rp[7], t));       // train.py:849
            }
            p[b][0] = __fadd_rn(rp[0], __fmul_rn(rp[3], z));                                     // train.py:871-873
            p[b][1] = __fadd_rn(rp[1], __fmul_rn(rp[4], z));
            p[b][2] = __fadd_rn(rp[2], __fmul_rn(rp[5], z));
            rig_mask[b] = 0.0f;
            if (writer[b] && a.ex.init_pts) {
                a.ex.init_pts[so[b] * 3 + 0] = p[b][0]; a.ex.init_pts[so[b] * 3 + 1] = p[b][1]; a.ex.init_pts[so[b] * 3 + 2] = p[b][2];
            }
        });

        if constexpr (HAS_BEND) {
            constexpr int NS_BIN = PL::NS_BIN, NS_RIN = PL::NS_RIN;
            constexpr int NBH = PL::NT_BW * SP, NR = PL::NT_RW * SP;
            constexpr bool SPLIT = P::SPLIT;
            Act<PE, NS_BIN, SPLIT> bin[NB];
            Act<PE, NS_RIN, SPLIT> rin[NB];
            static_for<0, NB>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                const float* lat = a.latents + (size_t)ray[b] * a.lat_stride;
                auto binval = [&](auto idxc) -> float {
                    constexpr int idx = decltype(idxc)::value;
                    if constexpr (idx < 3) return p[b][idx];
                    else if constexpr (idx < 8) return 0.0f;
                    else if constexpr (idx - 8 < A::LAT) return lat[idx - 8];
                    else return 0.0f;
                };
                auto rinval = [&](auto idxc) -> float {
                    constexpr int idx = decltype(idxc)::value;
                    if constexpr (idx < 3) return p[b][idx]; else return 0.0f;
                };
                static_for<0, NS_BIN>([&](auto sc_) {
                    constexpr int s = decltype(sc_)::value;
                    static_for<0, KH>([&](auto ec) {
                        constexpr int e = decltype(ec)::value;
                        const float v0 = binval(std::integral_constant<int, (2 * s) * KH + e>{});
                        const float v1 = binval(std::integral_constant<int, (2 * s + 1) * KH + e>{});
                        bin[b].template set<s, e>(h ? v1 : v0);
                    });
                });
                static_for<0, NS_RIN>([&](auto sc_) {
                    constexpr int s = decltype(sc_)::value;
                    static_for<0, KH>([&](auto ec) {
                        constexpr int e = decltype(ec)::value;
                        const float v0 = rinval(std::integral_constant<int, (2 * s) * KH + e>{});
                        const float v1 = rinval(std::integral_constant<int, (2 * s + 1) * KH + e>{});
                        rin[b].template set<s, e>(h ? v1 : v0);
                    });
                });
            });
            // ---- offset MLP (run_nerf_helpers.py:525-541)
            Act<PE, NBH, SPLIT> ba[NB], bb[NB];
            dense_b_mb<PE, SPLIT, PL, PL::L_BEND0, NS_BIN, NB>(st, bias_lds, h, bin, [&](auto tc, auto bc, const f32x16& acc) {
                pack_act<PE, decltype(tc)::value>(acc, ba[decltype(bc)::value]); });
            static_for<1, A::BD - 1>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i % 2 == 1)
                    dense_b_mb<PE, SPLIT, PL, PL::L_BEND0 + i, NBH, NB>(st, bias_lds, h, ba, [&](auto tc, auto bc, const f32x16& acc) {
                        pack_act<PE, decltype(tc)::value>(acc, bb[decltype(bc)::value]); });
                else
                    dense_b_mb<PE, SPLIT, PL, PL::L_BEND0 + i, NBH, NB>(st, bias_lds, h, bb, [&](auto tc, auto bc, const f32x16& acc) {
                        pack_act<PE, decltype(tc)::value>(acc, ba[decltype(bc)::value]); });
            });
            float off[NB][3], logit[NB];
            auto take_off = [&](auto, auto bc, const f32x16& acc) {
                constexpr int b = decltype(bc)::value; off[b][0] = acc[0]; off[b][1] = acc[1]; off[b][2] = acc[2]; };
            if constexpr ((A::BD - 2) % 2 == 1) dense_b_mb<PE, SPLIT, PL, PL::L_BEND0 + A::BD - 1, NBH, NB>(st, bias_lds, h, bb, take_off);
            else dense_b_mb<PE, SPLIT, PL, PL::L_BEND0 + A::BD - 1, NBH, NB>(st, bias_lds, h, ba, take_off);
            // ---- rigidity MLP (run_nerf_helpers.py:545-561)
            Act<PE, NR, SPLIT> ra[NB], rb[NB];
            dense_b_mb<PE, SPLIT, PL, PL::L_RIG0, NS_RIN, NB>(st, bias_lds, h, rin, [&](auto tc, auto bc, const f32x16& acc) {
                pack_act<PE, decltype(tc)::value>(acc, ra[decltype(bc)::value]); });
            static_for<1, A::RD - 1>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i % 2 == 1)
                    dense_b_mb<PE, SPLIT, PL, PL::L_RIG0 + i, NR, NB>(st, bias_lds, h, ra, [&](auto tc, auto bc, const f32x16& acc) {
                        pack_act<PE, decltype(tc)::value>(acc, rb[decltype(bc)::value]); });
                else
                    dense_b_mb<PE, SPLIT, PL, PL::L_RIG0 + i, NR, NB>(st, bias_lds, h, rb, [&](auto tc, auto bc, const f32x16& acc) {
                        pack_act<PE, decltype(tc)::value>(acc, ra[decltype(bc)::value]); });
            });
            auto take_logit = [&](auto, auto bc, const f32x16& acc) { logit[decltype(bc)::value] = acc[0]; };
            if constexpr ((A::RD - 2) % 2 == 1) dense_b_mb<PE, SPLIT, PL, PL::L_RIG0 + A::RD - 1, NR, NB>(st, bias_lds, h, rb, take_logit);
            else dense_b_mb<PE, SPLIT, PL, PL::L_RIG0 + A::RD - 1, NR, NB>(st, bias_lds, h, ra, take_logit);

            static_for<0, NB>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                float m = (tanhf(logit[b]) + 1.0f) / 2.0f;                                   // rnh:559-561
                if (a.knobs.has_cutoff && m <= a.knobs.cutoff) m = 0.0f;                     // rnh:563-564
                rig_mask[b] = m;
                float mo[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    mo[c] = __fmul_rn(m, off[b][c]);                                         // rnh:567
                    if (a.knobs.has_scaling) mo[c] = __fmul_rn(mo[c], a.knobs.scaling);      // rnh:568-569
                }
                if (writer[b]) {
                    const size_t o3 = so[b] * 3;
                    if (a.ex.unmasked) { a.ex.unmasked[o3] = off[b][0]; a.ex.unmasked[o3 + 1] = off[b][1]; a.ex.unmasked[o3 + 2] = off[b][2]; }
                    if (a.ex.masked) { a.ex.masked[o3] = mo[0]; a.ex.masked[o3 + 1] = mo[1]; a.ex.masked[o3 + 2] = mo[2]; }
                    if (a.ex.rigidity) a.ex.rigidity[so[b]] = m;
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) p[b][c] = __fadd_rn(p[b][c], mo[c]);             // rnh:570
            });
        }

        // ---- positional encoding of the (bent) points, directly in B-operand order
        Arr2<efrag, NB, NS_ENC> enc;
        Arr2<frag, NB, NH> ha, hb;
        Arr2<frag, NB, 0> none;
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            if (writer[b] && a.ex.in_pts) {
                a.ex.in_pts[so[b] * 3 + 0] = p[b][0]; a.ex.in_pts[so[b] * 3 + 1] = p[b][1]; a.ex.in_pts[so[b] * 3 + 2] = p[b][2];
            }
            constexpr int F0 = enc_F0(A::L);
            constexpr int NSLOT = NS_ENC * KH;
            float ev[NSLOT];
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) ev[q] = 0.0f;
            ev[0] = h ? p[b][2] : p[b][0];
            ev[1] = h ? 0.0f : p[b][1];
            const float fscale = h ? (float)(1 << F0) : 1.0f;
            static_for<0, F0>([&](auto fc) {
                constexpr int fl = decltype(fc)::value;
                static_for<0, 3>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    float sv, cv;
                    sincosf(p[b][c] * (fscale * (float)(1 << fl)), &sv, &cv);
                    ev[2 + 2 * (3 * fl + c)] = sv;
                    ev[2 + 2 * (3 * fl + c) + 1] = cv;
                });
            });
            static_for<0, NS_ENC>([&](auto sc_) {
                constexpr int s = decltype(sc_)::value;
                static_for<0, KH>([&](auto ec) {
                    constexpr int e = decltype(ec)::value;
                    PE::template set<e>(enc.v[b][s], ev[s * KH + e]);
                });
            });
        });

        // ---- trunk (run_nerf_helpers.py:272-282) and head (:306)
        dense_mb<PE, P, PL, PL::L_TRUNK0, NS_ENC, 0, NB>(st, bias_lds, h, enc, none, [&](auto tc, auto bc, const f32x16& acc) {
            pack_tile<P, true, decltype(tc)::value>(acc, ha.v[decltype(bc)::value]); });
        static_for<1, A::D>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr bool skip = (i - 1 == A::SKIP);
            if constexpr (i % 2 == 1) {
                if constexpr (skip)
                    dense_mb<PE, P, PL, PL::L_TRUNK0 + i, NS_ENC, NH, NB>(st, bias_lds, h, enc, ha, [&](auto tc, auto bc, const f32x16& acc) {
                        pack_tile<P, true, decltype(tc)::value>(acc, hb.v[decltype(bc)::value]); });
                else
                    dense_mb<P, P, PL, PL::L_TRUNK0 + i, NH, 0, NB>(st, bias_lds, h, ha, none, [&](auto tc, auto bc, const f32x16& acc) {
                        pack_tile<P, true, decltype(tc)::value>(acc, hb.v[decltype(bc)::value]); });
            } else {
                if constexpr (skip)
                    dense_mb<PE, P, PL, PL::L_TRUNK0 + i, NS_ENC, NH, NB>(st, bias_lds, h, enc, hb, [&](auto tc, auto bc, const f32x16& acc) {
                        pack_tile<P, true, decltype(tc)::value>(acc, ha.v[decltype(bc)::value]); });
                else
                    dense_mb<P, P, PL, PL::L_TRUNK0 + i, NH, 0, NB>(st, bias_lds, h, hb, none, [&](auto tc, auto bc, const f32x16& acc) {
                        pack_tile<P, true, decltype(tc)::value>(acc, ha.v[decltype(bc)::value]); });
            }
        });
        float raw[NB][5];
        auto take_raw = [&](auto, auto bc, const f32x16& acc) {
            constexpr int b = decltype(bc)::value;
            raw[b][0] = acc[0]; raw[b][1] = acc[1]; raw[b][2] = acc[2]; raw[b][3] = acc[3]; raw[b][4] = acc[4];
        };
        if constexpr ((A::D - 1) % 2 == 1) dense_mb<P, P, PL, PL::L_HEAD, NH, 0, NB>(st, bias_lds, h, hb, none, take_raw);
        else dense_mb<P, P, PL, PL::L_HEAD, NH, 0, NB>(st, bias_lds, h, ha, none, take_raw);

        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            if (HAS_BEND && a.knobs.detailed && a.knobs.has_removal && rig_mask[b] >= a.knobs.removal)
                raw[b][3] = raw[b][3] * 0.0f;                                                // rnh:308-311
            if (writer[b]) {
                *(f32x4*)(a.raw4 + so[b] * 4) = f32x4{raw[b][0], raw[b][1], raw[b][2], raw[b][3]};
                if (a.raw_out) {
                    float* ro = a.raw_out + so[b] * a.raw_ch;
                    ro[0] = raw[b][0]; ro[1] = raw[b][1]; ro[2] = raw[b][2]; ro[3] = raw[b][3];
                    if (a.raw_ch > 4) ro[4] = raw[b][4];
                }
            }
        });
        static_for<PL::NUNITS, PL::NUP>([&](auto uc) { st.template advance<decltype(uc)::value>(); });
    }
    st.drain();
}

template <class P, class A, bool HAS_BEND, int WAVES, int NB>
static hipError_t launch_one_mb(const NetArgs& a, int num_cus, hipStream_t stream) {
    using PL = Plan<P, A, HAS_BEND, false>;
    const size_t lds = (size_t)RING * P::UNIT_BYTES + (size_t)PL::NTILES * 32 * sizeof(float);
    auto kern = net_kernel_mb<P, A, HAS_BEND, WAVES, NB>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int bpr = (a.S + 31) / 32;
    const long long nblocks = (long long)a.n_rays * bpr;
    const long long ntiles = (nblocks + WAVES * NB - 1) / (WAVES * NB);
    if (ntiles <= 0) return hipSuccess;
    const int grid = (int)(ntiles < (long long)num_cus ? ntiles : (long long)num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace nrn
