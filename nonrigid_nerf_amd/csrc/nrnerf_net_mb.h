// nrnerf_net_mb.h -- the network kernel with TWO 32-sample blocks per wave (one wave per SIMD, 512 registers).
//
// Every weight fragment read from LDS feeds two MFMAs (one per block), which halves the LDS->VGPR traffic per flop of
// nrnerf_net_impl.h's kernel; the kernel is power-bound, so less data movement buys clock.  Needs
// `-mllvm -amdgpu-mfma-vgpr-form` (Makefile): above 256 registers hipcc otherwise parks the accumulators in AccVGPRs and
// pays one v_accvgpr_read per epilogue value.  A lone wave per SIMD hides none of its own issue, hence:
//   * dense_mb: four accumulator sets per block; the epilogue of tile pair p is issued in chunks among the MFMAs of pair
//     p+1, the last pair of a layer is finished inside the next layer, and the fragment prefetch queue runs across layers;
//   * dense_b2: the two blocks run the bender half a layer apart, so one block's hi/lo packing overlaps the other's MFMAs.
// The view-dependent head is supported with finite-difference or ray directions (the mailbox of nrnerf_net_impl.h,
// indexed by the block's position in the tile); the exact-direction (Jacobian) variants stay on nrnerf_net_impl.h.
// Same packed weight stream; same results bit for bit (with finite-difference directions: up to one-ulp flips of an
// f16-rounded direction encoding on 0.05 % of the samples).
#pragma once
#include "nrnerf_net_impl.h"
#include "nrnerf_composite_ray.h"

namespace nrn {

// MB blocks per wave (MB > 1: one wave per SIMD, 512 registers): every weight fragment read from LDS feeds MB MFMAs, one
// per 32-sample block, so LDS->VGPR reads and L2->LDS DMA per flop drop by MB.  With a single wave per SIMD nothing
// else hides this wave's non-MFMA work, so the epilogue of pair p (convert + relu of 2 tiles x MB blocks) is issued in
// 2*MB chunks spread over the MFMAs of pair p+1, which accumulate into the other half of four accumulator sets.
// State a wave carries from one dense_mb layer into the next: the four accumulator sets per block (the last tile pair
// of a layer stays un-converted in sets 2,3 and is finished inside the next layer) and the fragment prefetch queue
// (the weight stream is linear, so the queue simply runs on into the next layer's fragments).
template <class P, int MB>
struct MbState {
    f32x16 accs[MB][4];
    typename P::frag a[P::PF];
};
struct NoEpi {
    template <class B, class T> __device__ __forceinline__ void operator()(B, T, const f32x16&) const {}
};

// PREV_NT > 0: sets 2,3 hold tiles PREV_NT-2, PREV_NT-1 of the previous layer; `prev_epi` converts them, in chunks, while
//              this layer's first tile group runs (their output slabs are only read from slab NS-4 on).
// DEFER:       leave this layer's last pair in sets 2,3 for the next layer (needs NT % 4 == 0).
// CONT_IN / CONT_OUT: the prefetch queue already holds this layer's first PF fragments / keeps running past the end.
template <class P0, class P1, class PL, int LI, int NS0, int NS1, int MB, int PREV_NT, bool DEFER, bool CONT_IN, bool CONT_OUT,
          class ST, class IN0, class IN1, class EPI, class PEPI>
__device__ __forceinline__ void dense_mb(ST& st, BiasPtr bias_lane, MbState<P1, MB>& ms, const IN0 (&in0)[MB], const IN1 (&in1)[MB],
                                         EPI&& epi, PEPI&& prev_epi) {
    constexpr LayerSpec spec = PL::TB.layers[LI];
    static_assert(spec.ns == NS0 + NS1 && spec.split == 0, "slab count mismatch between kernel and plan");
    constexpr int NS = NS0 + NS1, NT = spec.nt, Q = NT * NS, PF = P1::PF;
    using SQ = SeqPos<NT, NS>;
    constexpr int G0 = PL::TB.tiles[spec.tile0].gbase;
    constexpr int NACC = 4;
    static_assert(!DEFER || (NT % 4 == 0 && NT >= 4), "a deferred pair must sit in accumulator sets 2,3");
    static_assert(PREV_NT == 0 || (PREV_NT % 4 == 0 && NS >= 12), "pending pair: sets 2,3, consumed before slab NS-4");
    static_assert(!CONT_OUT || (Q >= PF && Q % PF == 0), "the queue slot of fragment q is q % PF in both layers");
    static_assert(!CONT_OUT || PL::TB.tiles[PL::TB.layers[LI + 1].tile0].gbase == G0 + Q, "next layer must follow in the stream");
    auto& a = ms.a;
    auto& accs = ms.accs;
    auto load = [&](auto qc) {          // q may run past Q into the next layer (CONT_OUT): same stream, same fragment size
        constexpr int q = decltype(qc)::value;
        constexpr bool enc_slab = (q < Q) && (SQ::slab(q < Q ? q : 0) < NS0);
        if constexpr (enc_slab) a[q % PF] = __builtin_bit_cast(typename P1::frag, st.template frag<P0, G0 + q>());
        else a[q % PF] = st.template frag<P1, G0 + q>();
    };
    if constexpr (!CONT_IN) static_for<0, (PF < Q ? PF : Q)>([&](auto qc) { load(qc); });
    constexpr int NCH = 2 * MB;         // chunks of a pair's epilogue: tile (k & 1) of block (k >> 1)
    // sets 0,1 (tiles 0,1) start now; sets 2,3 once the pending pair has left them
    static_for<0, (NT < 2 ? NT : 2)>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const f32x16 bv = load_bias(bias_lane, spec.tile0 + t);
        static_for<0, MB>([&](auto bc) { accs[decltype(bc)::value][t] = bv; });
    });
    auto arm23 = [&]() {
        static_for<2, (NT < 4 ? NT : 4)>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const f32x16 bv = load_bias(bias_lane, spec.tile0 + t);
            static_for<0, MB>([&](auto bc) { accs[decltype(bc)::value][t] = bv; });
        });
    };
    if constexpr (PREV_NT == 0) arm23();
    constexpr int DLY = (NS - 1 < NRN_EPI_DELAY) ? NS - 1 : NRN_EPI_DELAY;
    constexpr int CH = (NS - 1 - DLY) / NCH > 0 ? (NS - 1 - DLY) / NCH : 0;      // slab steps between chunks (0: burst)
    constexpr int FIRST_T = (NT >= 2) ? 1 : 0;                                    // last tile of the first group
    static_for<0, Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int t = SQ::tile(q), s = SQ::slab(q);
        constexpr int inflight = CONT_OUT ? PF - 1 : ((Q - 1 - q < PF - 1) ? Q - 1 - q : PF - 1);
        st.template ready<inflight>(a[q % PF]);
        const typename P1::frag cur = a[q % PF];
        if constexpr (CONT_OUT || q + PF < Q) load(std::integral_constant<int, q + PF>{});
        static_for<0, MB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            if constexpr (s < NS0) accs[b][t % NACC] = P0::mfma(__builtin_bit_cast(typename P0::frag, cur), in0[b][s], accs[b][t % NACC]);
            else accs[b][t % NACC] = P1::mfma(cur, in1[b][s - NS0], accs[b][t % NACC]);
        });
        // the previous layer's last pair: chunk k at slab 1 + k of the first group, then sets 2,3 are free
        if constexpr (PREV_NT > 0 && t == FIRST_T) {
            static_for<0, NCH>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (s == 1 + k) {
                    constexpr int b = k >> 1, u = k & 1;
                    prev_epi(std::integral_constant<int, b>{}, std::integral_constant<int, PREV_NT - 2 + u>{}, accs[b][2 + u]);
                }
            });
            if constexpr (s == NCH) arm23();
        }
        // delayed, chunked epilogue of this layer's previous pair (while a paired tile runs: t odd)
        if constexpr (t >= 2 && (t & 1) == 1 && t < 2 * (NT / 2)) {
            constexpr int tp = (t & ~1) - 2;
            static_for<0, NCH>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int at = (CH > 0) ? DLY + k * CH : DLY;
                if constexpr (s == at) {
                    constexpr int tt = tp + (k & 1), b = k >> 1;
                    epi(std::integral_constant<int, b>{}, std::integral_constant<int, tt>{}, accs[b][tt % NACC]);
                }
            });
            constexpr int last_at = (CH > 0) ? DLY + (NCH - 1) * CH : DLY;
            if constexpr (s == last_at) {           // both tiles of pair tp are consumed: re-arm their sets with biases
                static_for<0, 2>([&](auto uc) {
                    constexpr int tn = tp + 4 + decltype(uc)::value;
                    if constexpr (tn < NT) {
                        const f32x16 bv = load_bias(bias_lane, spec.tile0 + tn);
                        static_for<0, MB>([&](auto bc) { accs[decltype(bc)::value][tn % NACC] = bv; });
                    }
                });
            }
        }
        if constexpr (q == Q - 1 && !DEFER) {     // drain: tiles whose epilogue has not run yet
            constexpr int NPAIRED = 2 * (NT / 2);
            constexpr int first = (NPAIRED >= 2) ? NPAIRED - 2 : 0;
            static_for<first, NT>([&](auto tc) {
                static_for<0, MB>([&](auto bc) {
                    epi(bc, tc, accs[decltype(bc)::value][decltype(tc)::value % NACC]);
                });
            });
        }
    });
}

// Two blocks per wave, skewed by half a layer (MB = 2 only): the hi/lo packing of the bender activations is VALU-heavy
// and a lone wave per SIMD has nobody to overlap it with but itself.  Block 0's MFMAs of layer L run while block 1's
// epilogue of layer L-1 is issued; block 1's MFMAs of layer L (same fragments, kept in registers) run while block 0's
// epilogue of layer L is issued; block 1's accumulators are handed to the next layer un-converted (BendPend).
struct BendPend {
    f32x16 acc[2], corr[2];
};
template <class PE, bool SPLIT>
__device__ __forceinline__ f32x16 bend_combine(const f32x16& acc, const f32x16& corr) {
    if constexpr (SPLIT) return acc + corr * (1.0f / PE::LO_SCALE);
    else return acc;
}
template <class PE, bool SPLIT, class PL, int LI, int NS, int PREV_W, class ST, class ACT, class EPI, class PEPI>
__device__ __forceinline__ void dense_b2(ST& st, BiasPtr bias_lane, BendPend& pend, const ACT (&in)[2], EPI&& epi, PEPI&& prev_epi) {
    constexpr LayerSpec spec = PL::TB.layers[LI];
    static_assert(spec.ns == NS && spec.split == (SPLIT ? 1 : 0), "bender layer mismatch between kernel and plan");
    constexpr int W = spec.nt;
    static_assert(W <= 2, "one tile group per layer");
    constexpr int FP = SPLIT ? 2 : 1;
    typename PE::frag fr[NS][W * FP];
    auto load = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        static_for<0, W>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr TileInfo ti = PL::TB.tiles[spec.tile0 + u];
            static_for<0, FP>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                fr[s][u * FP + f] = st.template frag<PE, ti.gbase + s * ti.gstride + f>();
            });
        });
    };
    auto step = [&](auto sc, auto bc, f32x16 (&acc)[W], f32x16 (&corr)[W]) {
        constexpr int s = decltype(sc)::value, b = decltype(bc)::value;
        if constexpr (SPLIT) {
            static_for<0, W>([&](auto uc) { constexpr int u = decltype(uc)::value; corr[u] = PE::mfma(fr[s][2 * u + 1], in[b].hi[s], corr[u]); });
            static_for<0, W>([&](auto uc) { constexpr int u = decltype(uc)::value; acc[u] = PE::mfma(fr[s][2 * u], in[b].hi[s], acc[u]); });
            static_for<0, W>([&](auto uc) { constexpr int u = decltype(uc)::value; corr[u] = PE::mfma(fr[s][2 * u], in[b].lo[s], corr[u]); });
        } else {
            static_for<0, W>([&](auto uc) { constexpr int u = decltype(uc)::value; acc[u] = PE::mfma(fr[s][u], in[b].hi[s], acc[u]); });
        }
    };
    load(std::integral_constant<int, 0>{});
    f32x16 bias[W];
    static_for<0, W>([&](auto uc) { bias[decltype(uc)::value] = load_bias(bias_lane, spec.tile0 + decltype(uc)::value); });
    // ---- block 0, with the previous layer's block-1 epilogue in its shadow
    f32x16 acc0[W], corr0[W];
    static_for<0, W>([&](auto uc) { acc0[decltype(uc)::value] = bias[decltype(uc)::value]; corr0[decltype(uc)::value] = f32x16{}; });
    static_for<0, NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s + 1 < NS) load(std::integral_constant<int, s + 1>{});
        static_for<0, W * FP>([&](auto fc) { st.template ready<(s + 1 < NS) ? W * FP : 0>(fr[s][decltype(fc)::value]); });
        step(sc, std::integral_constant<int, 0>{}, acc0, corr0);
        static_for<0, PREV_W>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if constexpr (s == (u < NS ? u : NS - 1))
                prev_epi(std::integral_constant<int, 1>{}, uc, bend_combine<PE, SPLIT>(pend.acc[u], pend.corr[u]));
        });
    });
    // ---- block 1 (its inputs were completed by prev_epi above), with block 0's epilogue in its shadow
    f32x16 acc1[W], corr1[W];
    static_for<0, W>([&](auto uc) { acc1[decltype(uc)::value] = bias[decltype(uc)::value]; corr1[decltype(uc)::value] = f32x16{}; });
    static_for<0, NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        step(sc, std::integral_constant<int, 1>{}, acc1, corr1);
        static_for<0, W>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if constexpr (s == (u < NS ? u : NS - 1))
                epi(std::integral_constant<int, 0>{}, uc, bend_combine<PE, SPLIT>(acc0[u], corr0[u]));
        });
    });
    static_for<0, W>([&](auto uc) { pend.acc[decltype(uc)::value] = acc1[decltype(uc)::value]; pend.corr[decltype(uc)::value] = corr1[decltype(uc)::value]; });
}
template <class PE, bool SPLIT, int PREV_W, class PEPI>
__device__ __forceinline__ void bend_drain(BendPend& pend, PEPI&& prev_epi) {
    static_for<0, PREV_W>([&](auto uc) {
        prev_epi(std::integral_constant<int, 1>{}, uc, bend_combine<PE, SPLIT>(pend.acc[decltype(uc)::value], pend.corr[decltype(uc)::value]));
    });
}

#define NRN_FORB(b) static_for<0, MB>([&](auto bc_) { constexpr int b = decltype(bc_)::value;
#define NRN_ENDB });
template <class P, class A, bool HAS_BEND, bool VIEWS, int WAVES, int MB>
__global__ void __launch_bounds__(WAVES * 64, 1) net_kernel_mb(const NetArgs a) {
    static_assert(MB == 2 && P::KH == 8, "two blocks per wave, 16-bit policies");
    using PL = Plan<P, A, HAS_BEND, VIEWS>;
    using frag = typename P::frag;                                                   // hidden activations
    using PE = std::conditional_t<P::KH == 1, PolF32, PolF16>;                      // encodings, bender (nrnerf_plan.h frag_is_f16)
    using efrag = typename PE::frag;
    constexpr int KH = P::KH, SP = P::SP;
    constexpr int NS_ENC = PL::NS_ENC;
    constexpr int NT_W = PL::NT_W;

    extern __shared__ __attribute__((aligned(16))) char smem[];     // one array: ring | bias (G17: 16-B aligned carve)
    char* ring = smem;
    float* bias_lds = (float*)(smem + RING * P::UNIT_BYTES);

    float* mailbox = bias_lds + PL::NTILES * 32;      // [2][WAVES * MB][4]: last bent point of each block (VIEWS only)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // provably wave-uniform (SGPR)
    const int h = lane >> 5;
    const int j = lane & 31;

    for (int i = tid; i < PL::NTILES * 32; i += WAVES * 64) bias_lds[i] = a.bias[i];
    __syncthreads();

    const BiasPtr bias_lane = bias_lane_ptr(bias_lds, h);

    WRing<P, WAVES, PL::NUP> st;
    st.init(a.wstream, ring, wave, lane);

    const int S = a.S;
    const int bpr = (S + 31) >> 5;                 // 32-sample blocks per ray
    const long long nblocks = (long long)a.n_rays * bpr;
    // Block -> workgroup assignment: tiles of WAVES * MB consecutive blocks, strided over the grid; wave w of a tile owns
    // blocks w and WAVES + w of it.
    long long blk_begin, blk_end, tile_stride;
    if constexpr (VIEWS) {      // contiguous whole rays per workgroup: a sample's direction needs its predecessor's bent point
        const long long rays_per_wg = (a.n_rays + gridDim.x - 1) / gridDim.x;
        blk_begin = (long long)blockIdx.x * rays_per_wg * bpr;
        blk_end = blk_begin + rays_per_wg * bpr;
        if (blk_end > nblocks) blk_end = nblocks;
        if (blk_begin > nblocks) blk_begin = nblocks;
        tile_stride = WAVES * MB;
    } else {
        blk_begin = (long long)blockIdx.x * (WAVES * MB);
        blk_end = nblocks;
        tile_stride = (long long)gridDim.x * (WAVES * MB);
    }

    // Fused compositing (variants without a fused bender, NetArgs::fuse_on): a WAVE owns whole rays.  Its blocks come in
    // groups of RW rays = RW * bpr blocks = TG tiles of the workgroup (RW = 1 when bpr is even, else MB), groups strided over
    // the grid; the raw outputs of a group's blocks are staged in the wave's own LDS area, and after the group's last tile
    // the wave composites its RW rays itself (composite_ray): no cross-wave exchange, no barrier, the raw array of the pass
    // never reaches HBM.  Per sample the arithmetic is that of the unfused mapping (bit-identical raw).
    bool fuse = false;
    int RW = 1, TG = 1, tg = 0;
    long long ngroups = 0, grp = blockIdx.x;
    f32x4* stage_w = nullptr;
    if constexpr (!HAS_BEND) {
        fuse = a.fuse_on != 0;
        RW = (bpr % MB) ? MB : 1;
        TG = RW * bpr / MB;
        ngroups = ((long long)a.n_rays + WAVES * RW - 1) / (WAVES * RW);
        stage_w = (f32x4*)(mailbox + 2 * WAVES * MB * 4) + (size_t)wave * RW * bpr * 32;
        // the compositing arguments live in LDS, not in SGPRs: ~40 scalar registers held across the whole tile would be
        // spilled into vector registers the MFMA section needs (measured: +60 VGPRs, scratch in the TCB variants)
        if (fuse) {
            int* dst = (int*)((f32x4*)(mailbox + 2 * WAVES * MB * 4) + (size_t)WAVES * RW * bpr * 32);
            const int* src = (const int*)&a.fuse;
            for (int i = tid; i < (int)(sizeof(CompositeArgs) / 4); i += WAVES * 64) dst[i] = src[i];
            __syncthreads();
        }
    }

#ifdef NRN_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    int iter = 0;
    float cpre[MB][8];          // fused compositing: direction and depths of this wave's rays, requested one tile ahead of their use
    for (long long tile0 = blk_begin; fuse ? (grp < ngroups) : (tile0 < blk_end); ++iter) {
        const unsigned long long t_pass = NRN_NOW();
        if constexpr (!HAS_BEND) {
            if (fuse && tg == TG - 1) {
                const CompositeArgs& fa = *(const CompositeArgs*)((f32x4*)(mailbox + 2 * WAVES * MB * 4) + (size_t)WAVES * RW * bpr * 32);
                static_for<0, MB>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    if (r < RW) {
                        const long long rr = (grp * WAVES + wave) * RW + r;
                        composite_prefetch(fa, (int)(rr < a.n_rays ? rr : a.n_rays - 1), lane, (S + 63) >> 6, cpre[r]);
                    }
                });
            }
        }
        int ray[MB], sidx[MB], sc[MB];
        bool ok[MB], writer[MB];
        size_t so[MB];              // flat sample index for per-sample outputs
        const float* rp[MB];
        float p[MB][3];
        NRN_FORB(b)
            bool blk_ok;
            int bir;                // block within its ray
            if (fuse) {             // block q of this wave's group: ray (grp * WAVES + wave) * RW + q / bpr
                const int q = tg * MB + b;
                const long long rr = (grp * WAVES + wave) * RW + q / bpr;
                blk_ok = rr < a.n_rays;
                ray[b] = (int)(blk_ok ? rr : a.n_rays - 1);
                bir = q % bpr;
            } else {
                const long long blk = tile0 + b * WAVES + wave;
                blk_ok = blk < blk_end;
                const long long bb_ = blk_ok ? blk : blk_end - 1;
                ray[b] = (int)(bb_ / bpr);
                bir = (int)(bb_ % bpr);
            }
            sidx[b] = bir * 32 + j;
            ok[b] = blk_ok && sidx[b] < S;
            sc[b] = sidx[b] < S ? sidx[b] : S - 1;
            rp[b] = a.rays + (size_t)ray[b] * a.ray_stride;
            const float ox = rp[b][0], oy = rp[b][1], oz = rp[b][2], dx = rp[b][3], dy = rp[b][4], dz = rp[b][5];
            float z;
            if (a.z) {
                z = a.z[(size_t)ray[b] * S + sc[b]];
            } else {
                const float near = rp[b][6], far = rp[b][7];
                const float t = lin01(sc[b], S);
                if (a.lindisp)                                                               // train.py:850-852
                    z = __fdiv_rn(1.0f, __fadd_rn(__fmul_rn(__fdiv_rn(1.0f, near), __fsub_rn(1.0f, t)),
                                                  __fmul_rn(__fdiv_rn(1.0f, far), t)));
                else
                    z = __fadd_rn(__fmul_rn(near, __fsub_rn(1.0f, t)), __fmul_rn(far, t));  // train.py:849
            }
            p[b][0] = __fadd_rn(ox, __fmul_rn(dx, z)); p[b][1] = __fadd_rn(oy, __fmul_rn(dy, z));
            p[b][2] = __fadd_rn(oz, __fmul_rn(dz, z));                                       // train.py:871-873
            so[b] = (size_t)ray[b] * S + sc[b];
            writer[b] = ok[b] && h == 0;
            if constexpr (!HAS_BEND) {
                if (a.pts4) {       // points bent by the stand-alone bender kernel (nrnerf_bend.h)
                    const f32x4 q = *(const f32x4*)(a.pts4 + so[b] * 4);
                    p[b][0] = q[0]; p[b][1] = q[1]; p[b][2] = q[2];
                }
            }
            if (writer[b] && a.ex.init_pts) {
                a.ex.init_pts[so[b] * 3 + 0] = p[b][0]; a.ex.init_pts[so[b] * 3 + 1] = p[b][1]; a.ex.init_pts[so[b] * 3 + 2] = p[b][2];
            }
        NRN_ENDB

        NRN_TACC(1, t_pass);
        const unsigned long long t_bend = NRN_NOW();
        float rig_mask[MB];
        NRN_FORB(b) rig_mask[b] = 0.0f; NRN_ENDB
        if constexpr (HAS_BEND) {
            constexpr int NS_BIN = PL::NS_BIN, NS_RIN = PL::NS_RIN;
            constexpr int NB = PL::NT_BW * SP, NR = PL::NT_RW * SP;
            constexpr bool SPLIT = P::SPLIT;
            Act<PE, NS_BIN, SPLIT> bin[MB];
            NRN_FORB(b)
                const float* lat = a.latents + (size_t)ray[b] * a.lat_stride;
                auto binval = [&](auto idxc) -> float {
                    constexpr int idx = decltype(idxc)::value;
                    if constexpr (idx < 3) return p[b][idx];
                    else if constexpr (idx < 8) return 0.0f;
                    else if constexpr (idx - 8 < A::LAT) return lat[idx - 8];
                    else return 0.0f;
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
            NRN_ENDB
            // ---- offset MLP (run_nerf_helpers.py:525-541) and rigidity MLP (:545-561); rigidity input = xyz only
            Act<PE, NB, SPLIT> ba[MB], bb[MB];
            Act<PE, NS_RIN, SPLIT> rin[MB];
            NRN_FORB(b)
                auto rinval = [&](auto idxc) -> float {
                    constexpr int idx = decltype(idxc)::value;
                    if constexpr (idx < 3) return p[b][idx];
                    else return 0.0f;
                };
                static_for<0, NS_RIN>([&](auto sc_) {
                    constexpr int s = decltype(sc_)::value;
                    static_for<0, KH>([&](auto ec) {
                        constexpr int e = decltype(ec)::value;
                        const float v0 = rinval(std::integral_constant<int, (2 * s) * KH + e>{});
                        const float v1 = rinval(std::integral_constant<int, (2 * s + 1) * KH + e>{});
                        rin[b].template set<s, e>(h ? v1 : v0);
                    });
                });
            NRN_ENDB
            Act<PE, NR, SPLIT> ra[MB], rb[MB];
            float off[MB][3], logit[MB];
            auto to_ba = [&](auto bc, auto tc, const f32x16& acc) { pack_act<PE, decltype(tc)::value>(acc, ba[decltype(bc)::value]); };
            auto to_bb = [&](auto bc, auto tc, const f32x16& acc) { pack_act<PE, decltype(tc)::value>(acc, bb[decltype(bc)::value]); };
            auto to_ra = [&](auto bc, auto tc, const f32x16& acc) { pack_act<PE, decltype(tc)::value>(acc, ra[decltype(bc)::value]); };
            auto to_rb = [&](auto bc, auto tc, const f32x16& acc) { pack_act<PE, decltype(tc)::value>(acc, rb[decltype(bc)::value]); };
            auto take_off = [&](auto bc, auto, const f32x16& acc) {
                constexpr int b = decltype(bc)::value;
                off[b][0] = acc[0]; off[b][1] = acc[1]; off[b][2] = acc[2];
            };
            auto take_logit = [&](auto bc, auto, const f32x16& acc) { logit[decltype(bc)::value] = acc[0]; };
            static_assert(PL::NT_BW <= 2 && PL::NT_RW <= 2, "dense_b2 handles one tile group per layer");
            {
                BendPend pend;
                constexpr int WB = PL::NT_BW, WR = PL::NT_RW;
                dense_b2<PE, SPLIT, PL, PL::L_BEND0, NS_BIN, 0>(st, bias_lane, pend, bin, to_ba, NoEpi{});
                static_for<1, A::BD - 1>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr (i % 2 == 1) dense_b2<PE, SPLIT, PL, PL::L_BEND0 + i, NB, WB>(st, bias_lane, pend, ba, to_bb, to_ba);
                    else dense_b2<PE, SPLIT, PL, PL::L_BEND0 + i, NB, WB>(st, bias_lane, pend, bb, to_ba, to_bb);
                });
                constexpr bool OFF_IN_B = ((A::BD - 2) % 2 == 1);       // buffer feeding the output layer
                if constexpr (OFF_IN_B) dense_b2<PE, SPLIT, PL, PL::L_BEND0 + A::BD - 1, NB, WB>(st, bias_lane, pend, bb, take_off, to_bb);
                else dense_b2<PE, SPLIT, PL, PL::L_BEND0 + A::BD - 1, NB, WB>(st, bias_lane, pend, ba, take_off, to_ba);
                dense_b2<PE, SPLIT, PL, PL::L_RIG0, NS_RIN, 1>(st, bias_lane, pend, rin, to_ra, take_off);
                static_for<1, A::RD - 1>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr (i % 2 == 1) dense_b2<PE, SPLIT, PL, PL::L_RIG0 + i, NR, WR>(st, bias_lane, pend, ra, to_rb, to_ra);
                    else dense_b2<PE, SPLIT, PL, PL::L_RIG0 + i, NR, WR>(st, bias_lane, pend, rb, to_ra, to_rb);
                });
                constexpr bool LOGIT_IN_B = ((A::RD - 2) % 2 == 1);
                if constexpr (LOGIT_IN_B) dense_b2<PE, SPLIT, PL, PL::L_RIG0 + A::RD - 1, NR, WR>(st, bias_lane, pend, rb, take_logit, to_rb);
                else dense_b2<PE, SPLIT, PL, PL::L_RIG0 + A::RD - 1, NR, WR>(st, bias_lane, pend, ra, take_logit, to_ra);
                bend_drain<PE, SPLIT, 1>(pend, take_logit);
            }

            NRN_FORB(b)
                rig_mask[b] = (tanhf(logit[b]) + 1.0f) / 2.0f;                                  // rnh:559-561
                if (a.knobs.has_cutoff && rig_mask[b] <= a.knobs.cutoff) rig_mask[b] = 0.0f;    // rnh:563-564
                float mo[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    mo[c] = __fmul_rn(rig_mask[b], off[b][c]);                                  // rnh:567
                    if (a.knobs.has_scaling) mo[c] = __fmul_rn(mo[c], a.knobs.scaling);         // rnh:568-569
                }
                if (writer[b]) {
                    if (a.ex.unmasked) { a.ex.unmasked[so[b] * 3 + 0] = off[b][0]; a.ex.unmasked[so[b] * 3 + 1] = off[b][1]; a.ex.unmasked[so[b] * 3 + 2] = off[b][2]; }
                    if (a.ex.masked) { a.ex.masked[so[b] * 3 + 0] = mo[0]; a.ex.masked[so[b] * 3 + 1] = mo[1]; a.ex.masked[so[b] * 3 + 2] = mo[2]; }
                    if (a.ex.rigidity) a.ex.rigidity[so[b]] = rig_mask[b];
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) p[b][c] = __fadd_rn(p[b][c], mo[c]);                // rnh:570
            NRN_ENDB
        }
        NRN_TACC(2, t_bend);
        const unsigned long long t_mid = NRN_NOW();
        NRN_FORB(b)
            if (writer[b] && a.ex.in_pts) {
                a.ex.in_pts[so[b] * 3 + 0] = p[b][0]; a.ex.in_pts[so[b] * 3 + 1] = p[b][1]; a.ex.in_pts[so[b] * 3 + 2] = p[b][2];
            }
            if (writer[b] && a.bent4) *(f32x4*)(a.bent4 + so[b] * 4) = f32x4{p[b][0], p[b][1], p[b][2], rig_mask[b]};
        NRN_ENDB

        // ---- view direction of the sample (VIEWS): finite difference of the bent points along the ray, or the ray's own
        //      unit direction without a bender (run_nerf_helpers.py:288-290, 339-351; train.py:73-76).  Block (b, wave) is
        //      block b * WAVES + wave of the tile, so its predecessor's mailbox slot is that index minus one.
        constexpr int NS_ENCV = PL::NS_ENCV;
        efrag encv[MB][VIEWS ? NS_ENCV : 1];
        if constexpr (VIEWS) {
            float dirv[MB][3];
            if constexpr (HAS_BEND) {
                NRN_FORB(b)
                    if (j == 31 && h == 0) {
                        float* mb = mailbox + ((iter & 1) * (WAVES * MB) + b * WAVES + wave) * 4;
                        mb[0] = p[b][0]; mb[1] = p[b][1]; mb[2] = p[b][2];
                    }
                NRN_ENDB
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                NRN_FORB(b)
                    float prev[3], next[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) { prev[c] = __shfl_up(p[b][c], 1); next[c] = __shfl_down(p[b][c], 1); }
                    const bool first_in_ray = (sidx[b] == 0);
                    if (j == 0 && !first_in_ray) {
                        constexpr int NV_ = WAVES * MB;
                        const int v = b * WAVES + wave;
                        const float* mb = (v > 0) ? mailbox + ((iter & 1) * NV_ + v - 1) * 4
                                                  : mailbox + (((iter + 1) & 1) * NV_ + NV_ - 1) * 4;
                        prev[0] = mb[0]; prev[1] = mb[1]; prev[2] = mb[2];
                    }
                    float dd[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) dd[c] = first_in_ray ? __fsub_rn(next[c], p[b][c]) : __fsub_rn(p[b][c], prev[c]);
                    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dd[0], dd[0]), __fmul_rn(dd[1], dd[1])), __fmul_rn(dd[2], dd[2])));
#pragma unroll
                    for (int c = 0; c < 3; ++c) dirv[b][c] = __fdiv_rn(dd[c], __fadd_rn(nrm, 0.000001f));
                NRN_ENDB
            } else if (a.pts4) {
                // split-bender path: the points are the bent points of nrnerf_bend.h, so the direction is their finite
                // difference along the ray exactly as in the fused kernel, the neighbour read from the same array
                NRN_FORB(b)
                    const bool first_in_ray = (sidx[b] == 0);
                    const f32x4 nb = *(const f32x4*)(a.pts4 + (first_in_ray ? so[b] + 1 : so[b] - 1) * 4);
                    float dd[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) dd[c] = first_in_ray ? __fsub_rn(nb[c], p[b][c]) : __fsub_rn(p[b][c], nb[c]);
                    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dd[0], dd[0]), __fmul_rn(dd[1], dd[1])), __fmul_rn(dd[2], dd[2])));
#pragma unroll
                    for (int c = 0; c < 3; ++c) dirv[b][c] = __fdiv_rn(dd[c], __fadd_rn(nrm, 0.000001f));
                NRN_ENDB
            } else {
                NRN_FORB(b) dirv[b][0] = rp[b][8]; dirv[b][1] = rp[b][9]; dirv[b][2] = rp[b][10]; NRN_ENDB
            }
            constexpr int F0V = enc_F0(A::LV);
            constexpr int NSLOTV = NS_ENCV * KH;
            NRN_FORB(b)
                float evv[NSLOTV];
#pragma unroll
                for (int q = 0; q < NSLOTV; ++q) evv[q] = 0.0f;
                evv[0] = h ? dirv[b][2] : dirv[b][0];
                evv[1] = h ? 0.0f : dirv[b][1];
                const float vscale = h ? (float)(1 << F0V) : 1.0f;
                const float drev[3] = {dirv[b][0] * 0.15915494309189535f, dirv[b][1] * 0.15915494309189535f, dirv[b][2] * 0.15915494309189535f};
                static_for<0, F0V>([&](auto fc) {
                    constexpr int fl = decltype(fc)::value;
                    static_for<0, 3>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        float sv, cv;
                        enc_sincos<KH == 1>(dirv[b][c], drev[c], vscale * (float)(1 << fl), &sv, &cv);
                        evv[2 + 2 * (3 * fl + c)] = sv;
                        evv[2 + 2 * (3 * fl + c) + 1] = cv;
                    });
                });
                static_for<0, NS_ENCV>([&](auto sc_) {
                    constexpr int s = decltype(sc_)::value;
                    static_for<0, KH>([&](auto ec) {
                        constexpr int e = decltype(ec)::value;
                        PE::template set<e>(encv[b][s], evv[s * KH + e]);
                    });
                });
            NRN_ENDB
        }

        // ---- positional encoding of the (bent) point, directly in B-operand order
        constexpr int F0 = enc_F0(A::L);
        constexpr int NSLOT = PL::NS_ENC_XYZ * KH;
        efrag enc[MB][NS_ENC];
        NRN_FORB(b)
            float ev[NSLOT];
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) ev[q] = 0.0f;
            ev[0] = h ? p[b][2] : p[b][0];
            ev[1] = h ? 0.0f : p[b][1];
            const float fscale = h ? (float)(1 << F0) : 1.0f;
            const float prev_[3] = {p[b][0] * 0.15915494309189535f, p[b][1] * 0.15915494309189535f, p[b][2] * 0.15915494309189535f};
            static_for<0, F0>([&](auto fc) {
                constexpr int fl = decltype(fc)::value;
                static_for<0, 3>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    float sv, cv;
                    enc_sincos<KH == 1>(p[b][c], prev_[c], fscale * (float)(1 << fl), &sv, &cv);   // power-of-two scaling is exact
                    ev[2 + 2 * (3 * fl + c)] = sv;
                    ev[2 + 2 * (3 * fl + c) + 1] = cv;
                });
            });
            static_for<0, PL::NS_ENC_XYZ>([&](auto sc_) {
                constexpr int s = decltype(sc_)::value;
                static_for<0, KH>([&](auto ec) {
                    constexpr int e = decltype(ec)::value;
                    PE::template set<e>(enc[b][s], ev[s * KH + e]);
                });
            });
            if constexpr (A::TCB) {      // time-conditioned baseline: the ray's latent code follows the encoding (rnh:273-274)
                const float* lat = a.latents + (size_t)ray[b] * a.lat_stride;
                static_for<PL::NS_ENC_XYZ, NS_ENC>([&](auto sc_) {
                    constexpr int s = decltype(sc_)::value;
                    static_for<0, KH>([&](auto ec) {
                        constexpr int e = decltype(ec)::value;
                        constexpr int i0 = (2 * (s - PL::NS_ENC_XYZ)) * KH + e, i1 = i0 + KH;
                        const float v0 = (i0 < A::LAT) ? lat[i0 < A::LAT ? i0 : 0] : 0.0f;
                        const float v1 = (i1 < A::LAT) ? lat[i1 < A::LAT ? i1 : 0] : 0.0f;
                        PE::template set<e>(enc[b][s], h ? v1 : v0);
                    });
                });
            }
        NRN_ENDB

        NRN_TACC(3, t_mid);
        const unsigned long long t_trunk = NRN_NOW();
        // ---- trunk (run_nerf_helpers.py:272-282) and head (:306)
        constexpr int NH = NT_W * SP;
        frag ha[MB][NH], hb[MB][NH];
        Empty none[MB];
        auto to_ha = [&](auto bc, auto tc, const f32x16& acc) { pack_tile<P, true, decltype(tc)::value>(acc, ha[decltype(bc)::value]); };
        auto to_hb = [&](auto bc, auto tc, const f32x16& acc) { pack_tile<P, true, decltype(tc)::value>(acc, hb[decltype(bc)::value]); };
        static_assert(NT_W % 4 == 0, "dense_mb keeps a deferred tile pair in accumulator sets 2,3");
        MbState<P, MB> ms;
        dense_mb<PE, P, PL, PL::L_TRUNK0, NS_ENC, 0, MB, 0, true, false, true>(st, bias_lane, ms, enc, none, to_ha, NoEpi{});
        static_for<1, A::D>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr bool skip = (i - 1 == A::SKIP);
            if constexpr (i % 2 == 1) {
                if constexpr (skip) dense_mb<PE, P, PL, PL::L_TRUNK0 + i, NS_ENC, NH, MB, NT_W, true, true, true>(st, bias_lane, ms, enc, ha, to_hb, to_ha);
                else dense_mb<P, P, PL, PL::L_TRUNK0 + i, NH, 0, MB, NT_W, true, true, true>(st, bias_lane, ms, ha, none, to_hb, to_ha);
            } else {
                if constexpr (skip) dense_mb<PE, P, PL, PL::L_TRUNK0 + i, NS_ENC, NH, MB, NT_W, true, true, true>(st, bias_lane, ms, enc, hb, to_ha, to_hb);
                else dense_mb<P, P, PL, PL::L_TRUNK0 + i, NH, 0, MB, NT_W, true, true, true>(st, bias_lane, ms, hb, none, to_ha, to_hb);
            }
        });
        float raw[MB][5];
        NRN_FORB(b) raw[b][0] = raw[b][1] = raw[b][2] = raw[b][3] = raw[b][4] = 0.f; NRN_ENDB
        constexpr bool LAST_IN_B = ((A::D - 1) % 2 == 1);       // buffer holding the trunk output
        if constexpr (!VIEWS) {
            auto take_raw = [&](auto bc, auto, const f32x16& acc) {
                constexpr int b = decltype(bc)::value;
                raw[b][0] = acc[0]; raw[b][1] = acc[1]; raw[b][2] = acc[2]; raw[b][3] = acc[3]; raw[b][4] = acc[4];
            };
            if constexpr (LAST_IN_B) dense_mb<P, P, PL, PL::L_HEAD, NH, 0, MB, NT_W, false, true, false>(st, bias_lane, ms, hb, none, take_raw, to_hb);
            else dense_mb<P, P, PL, PL::L_HEAD, NH, 0, MB, NT_W, false, true, false>(st, bias_lane, ms, ha, none, take_raw, to_ha);
        } else {
            // view-dependent head (run_nerf_helpers.py:284-304): alpha from the trunk output, then
            // relu(views_linear([feature_linear(trunk output), enc(dir)])) -- ONE layer on the trunk output, feature_linear folded
            // into its weights by the packer -- and rgb_linear; output = [rgb, alpha].  hx = trunk output.
            auto head = [&](auto& hx, auto&& to_hx) {
                auto take_alpha = [&](auto bc, auto, const f32x16& acc) { raw[decltype(bc)::value][3] = acc[0]; };
                // alpha finishes the trunk's deferred pair in its shadow and drains itself (one tile): hx is complete after it
                dense_mb<P, P, PL, PL::L_ALPHA, NH, 0, MB, NT_W, false, true, true>(st, bias_lane, ms, hx, none, take_alpha, to_hx);
                constexpr int NV = (NT_W / 2) * SP;
                frag hv[MB][NV];
                auto to_hv = [&](auto bc, auto tc, const f32x16& acc) { pack_tile<P, true, decltype(tc)::value>(acc, hv[decltype(bc)::value]); };
                dense_mb<PE, P, PL, PL::L_VIEWS, NS_ENCV, NH, MB, 0, false, true, true>(st, bias_lane, ms, encv, hx, to_hv, NoEpi{});
                auto take_rgb = [&](auto bc, auto, const f32x16& acc) {
                    constexpr int b = decltype(bc)::value;
                    raw[b][0] = acc[0]; raw[b][1] = acc[1]; raw[b][2] = acc[2];
                };
                dense_mb<P, P, PL, PL::L_RGB, NV, 0, MB, 0, false, true, false>(st, bias_lane, ms, hv, none, take_rgb, NoEpi{});
            };
            if constexpr (LAST_IN_B) head(hb, to_hb); else head(ha, to_ha);
        }

        NRN_TACC(4, t_trunk);
        const unsigned long long t_out = NRN_NOW();
        NRN_FORB(b)
            if (HAS_BEND && a.knobs.detailed && a.knobs.has_removal && rig_mask[b] >= a.knobs.removal)
                raw[b][3] = raw[b][3] * 0.0f;                                                  // rnh:308-311
            if (writer[b]) {
                if (!fuse) *(f32x4*)(a.raw4 + so[b] * 4) = f32x4{raw[b][0], raw[b][1], raw[b][2], raw[b][3]};
                if (a.raw_out) {
                    float* ro = a.raw_out + so[b] * a.raw_ch;
                    ro[0] = raw[b][0]; ro[1] = raw[b][1]; ro[2] = raw[b][2]; ro[3] = raw[b][3];
                    if (a.raw_ch > 4) ro[4] = raw[b][4];
                }
            }
            if constexpr (!HAS_BEND) {
                if (fuse && h == 0) stage_w[(tg * MB + b) * 32 + j] = f32x4{raw[b][0], raw[b][1], raw[b][2], raw[b][3]};
            }
        NRN_ENDB
        // padding units (keep the ring phase identical every pass and prime the next pass' first units)
        static_for<PL::NUNITS, PL::NUP>([&](auto uc) { st.template advance<decltype(uc)::value>(); });
        if constexpr (!HAS_BEND) {
            if (fuse) {
                if (++tg == TG) {        // the group's last tile: composite this wave's rays from its LDS stage (train.py:943-950)
                    // the surface reduction reads bent4 rows this wave's OTHER lanes have just written (only when this kernel
                    // writes that array itself: a model without bender): make the stores visible first
                    const CompositeArgs& fa = *(const CompositeArgs*)((f32x4*)(mailbox + 2 * WAVES * MB * 4) + (size_t)WAVES * RW * bpr * 32);
                    if (a.bent4 && fa.bent4) __threadfence();
                    static_for<0, MB>([&](auto rc) {
                        constexpr int r = decltype(rc)::value;
                        if (r < RW) {
                            const long long rr = (grp * WAVES + wave) * RW + r;
                            const bool ray_ok = rr < a.n_rays;
                            const int cray = (int)(ray_ok ? rr : a.n_rays - 1);
                            const f32x4* sw = stage_w + r * bpr * 32;
                            auto raw_at = [&](int ic) { return sw[ic]; };
                            switch ((S + 63) >> 6) {
                                case 1: { float cz[2], cw[1]; composite_ray<1>(fa, cray, ray_ok, lane, raw_at, cz, cw, cpre[r]); break; }
                                case 2: { float cz[3], cw[2]; composite_ray<2>(fa, cray, ray_ok, lane, raw_at, cz, cw, cpre[r]); break; }
                                case 3: { float cz[4], cw[3]; composite_ray<3>(fa, cray, ray_ok, lane, raw_at, cz, cw, cpre[r]); break; }
                                default: { float cz[5], cw[4]; composite_ray<4>(fa, cray, ray_ok, lane, raw_at, cz, cw, cpre[r]); break; }
                            }
                        }
                    });
                    tg = 0;
                    grp += gridDim.x;
                }
            } else {
                tile0 += tile_stride;
            }
        } else {
            tile0 += tile_stride;
        }
        NRN_TACC(5, t_out);
        NRN_TACC(0, t_pass);
#ifdef NRN_TIMING
        tacc[7] += 1;
#endif
    }
    st.drain();     // no LDS-DMA may be in flight when the workgroup's LDS is released
#ifdef NRN_TIMING
    if (blockIdx.x == 0 && lane == 0 && wave < 8) {
        tacc[6] = st.bar_cycles;
        for (int i = 0; i < 8; ++i) g_nrn_timing[wave][i] += tacc[i];
    }
#endif
}


template <class P, class A, bool HAS_BEND, bool VIEWS, int WAVES, int MB>
static hipError_t launch_one_mb(const NetArgs& a, int num_cus, hipStream_t stream) {
    using PL = Plan<P, A, HAS_BEND, VIEWS>;
    const int bpr = (a.S + 31) / 32;
    size_t lds = (size_t)RING * P::UNIT_BYTES + (size_t)PL::NTILES * 32 * sizeof(float) + 2 * WAVES * MB * 4 * sizeof(float);
    const int RW = (bpr % MB) ? MB : 1;      // fused compositing: rays per wave and group (see the kernel)
    if (a.fuse_on) {
        if (HAS_BEND || a.S > 256 || a.fuse.n_importance != 0 || a.fuse.S != a.S) return hipErrorInvalidValue;
        lds += (size_t)WAVES * RW * bpr * 32 * 16 + 256;       // the waves' raw stages + the compositing arguments
    }
    auto kern = net_kernel_mb<P, A, HAS_BEND, VIEWS, WAVES, MB>;
    // function attributes are per device: one flag per ordinal (idempotent; racing threads set the same value)
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        // the largest request any launch of this variant can make: ring + bias + mailbox + the fused stages at bpr = 7 (RW = MB)
        const size_t lds_max = (size_t)RING * P::UNIT_BYTES + (size_t)PL::NTILES * 32 * sizeof(float) + 2 * WAVES * MB * 4 * sizeof(float) +
                               (HAS_BEND ? 0 : (size_t)WAVES * MB * 7 * 32 * 16 + 256);
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const long long nblocks = (long long)a.n_rays * bpr;
    long long want = (nblocks + WAVES * MB - 1) / (WAVES * MB);
    if (want <= 0) return hipSuccess;
    if (a.fuse_on) {    // groups of WAVES * RW whole rays
        want = ((long long)a.n_rays + WAVES * RW - 1) / (WAVES * RW);
    } else if (VIEWS) {    // contiguous whole-ray ranges: no more workgroups than ray groups that fill a tile
        const long long rays_per_tile = (WAVES * MB + bpr - 1) / bpr;
        want = ((long long)a.n_rays + rays_per_tile - 1) / rays_per_tile;
    }
    const int grid = (int)(want < num_cus ? (want > 0 ? want : 1) : num_cus);       // persistent: one 4-wave workgroup per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace nrn
