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
#ifndef NRN_X16_WAVES
#define NRN_X16_WAVES 4       // waves per workgroup (4: one per SIMD, up to 512 registers; 8: two per SIMD, 256)
#endif
// the same two knobs for the 128-wide trunk (ArchNarrow): a quarter of the flops per sample, so the latency-bound ends of an iteration
// (points, encoding, outputs, compositing) weigh four times as much and a second wave per SIMD pays (measured, profiles/r05_*)
#ifndef NRN_X16_NB_NARROW
#define NRN_X16_NB_NARROW 4
#endif
#ifndef NRN_X16_WAVES_NARROW
#define NRN_X16_WAVES_NARROW 8
#endif
#ifndef NRN_X16_PF_NARROW
#define NRN_X16_PF_NARROW 2       // weight fragments requested ahead in the narrow trunk: 2 fits 256 registers without spilling (two waves per
                                  // SIMD cover the LDS latency); measured fine pass 8 (39 spilled): 6.80 ms, 4 (9 spilled): 6.69, 2: 6.75 (profiles/r05_w128_x16_ab.txt)
#endif
#ifndef NRN_X16_PF
#define NRN_X16_PF 8          // weight fragments requested from LDS ahead of their MFMAs (4: 24.1 ms per fine pass, 6: 23.7, 8: 23.7)
#endif

// One dense layer.  Stream order (PlanX16 = place_fragments): tile pairs (2 p, 2 p + 1), their k-steps interleaved; an odd last tile
// (the head) alone.  Per fragment NB MFMAs (one per block).  Two accumulator sets: pair p runs in set p & 1 while the epilogue of
// pair p - 1 (pack -> out[b][p - 1], NB chunks of ~8 VALU) is issued among its first k-steps; the last pair's epilogue follows the
// layer.  The first MFMA of a chain takes the bias as its C operand (no copies).
template <class A> struct X16Cfg {
    static constexpr int NB = (A::W <= 128) ? NRN_X16_NB_NARROW : NRN_X16_NB, WAVES = (A::W <= 128) ? NRN_X16_WAVES_NARROW : NRN_X16_WAVES;
    static constexpr int PF = (A::W <= 128) ? NRN_X16_PF_NARROW : NRN_X16_PF;
};
// (PF_: fragments requested ahead; the stand-alone bender, nrnerf_bend_x16.h, runs several waves per SIMD and asks for fewer)
template <class P0, class P1, class PL, int LI, int NS0, int NS1, int NB, int PF_ = NRN_X16_PF, class ST, class IN0, class IN1, class EPI>
__device__ __forceinline__ void dense_x16(ST& st, const __attribute__((address_space(3))) f32x4* bias_lane, const IN0 (&in0)[NB], const IN1 (&in1)[NB],
                                          EPI&& epi) {
    constexpr LayerSpec spec = PL::TB.layers[LI];
    static_assert(spec.ns == NS0 + NS1, "k-step count mismatch between kernel and plan");
    constexpr int NS = NS0 + NS1, NT = spec.nt, Q = NT * NS, PF = PF_;
    using SQ = SeqPos<NT, NS>;
    constexpr int G0 = PL::TB.tiles[spec.tile0].gbase;
    typename P1::frag a[PF];
    f32x4 bias[NT];               // (one name per tile: a bias is requested PF k-steps ahead, across tile pairs in the short layers)
    const unsigned bias_addr = (unsigned)(size_t)bias_lane;
    auto load = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int s = SQ::slab(q);
        // A tile's bias ([tile][16 rows]: this lane's rows 4 g .. 4 g + 3) travels in the same queue, right ahead of the tile's
        // first fragment, and is covered by that fragment's counted wait (LDS operations retire in order).  Read with a plain
        // load, hipcc -- which cannot see the fragment reads inside the asm statements -- waits for it with lgkmcnt(0) at every
        // tile pair and drains the prefetch queue there (87 times per pass of the network).
        if constexpr (s == 0) {
            u32x4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(bias_addr), "n"((spec.tile0 + SQ::tile(q)) * 64));
            bias[SQ::tile(q)] = __builtin_bit_cast(f32x4, v);
        }
        if constexpr (s < NS0) a[q % PF] = __builtin_bit_cast(typename P1::frag, st.template frag<P0, G0 + q>());
        else a[q % PF] = st.template frag<P1, G0 + q>();
    };
    // fragment q has landed -- and with it everything older in the queue (N: LDS operations younger than fragment q, at most)
    auto ready = [&](auto qc, auto& f) {
        constexpr int q = decltype(qc)::value;
        constexpr int F = (Q - 1 - q < PF - 1) ? Q - 1 - q : PF - 1;          // younger fragments ...
        constexpr int N = F + [] { int nb = 0; for (int j = q + 1; j <= q + F; ++j) nb += SQ::slab(j) == 0; return nb; }();   // ... and biases
        static_assert(ST::ASM_FRAGS && N >= 0 && N <= 15, "counted waits need the 16-byte asm fragment reads");
        u32x4 v = __builtin_bit_cast(u32x4, f);
        if constexpr (SQ::slab(q) == 0) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(v), "+v"(bias[SQ::tile(q)]) : "n"(N));
        else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
        f = __builtin_bit_cast(typename P1::frag, v);
    };
    static_for<0, (PF < Q ? PF : Q)>([&](auto qc) { load(qc); });
    f32x4 acc[2][2][NB];          // [set][tile of the pair][block]
    // an accumulator as the epilogue sees it: through a volatile asm, which stays behind the counted LDS wait of the k-step it is
    // written after -- left alone, hipcc hoists the conversions up to the accumulator's last MFMA and waits out its latency there
    auto pinned = [](f32x4 v) { asm volatile("" : "+v"(v)); return v; };
    static_for<0, Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int t = SQ::tile(q), s = SQ::slab(q);
        constexpr int p = t >> 1, u = t & 1, set = p & 1;
        ready(qc, a[q % PF]);
        const typename P1::frag cur = a[q % PF];
        if constexpr (q + PF < Q) load(std::integral_constant<int, q + PF>{});
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            const f32x4 c = (s == 0) ? bias[t] : acc[set][u][b];
            if constexpr (s < NS0) acc[set][u][b] = X16<P0>::mfma(__builtin_bit_cast(typename P0::frag, cur), in0[b][s], c);
            else acc[set][u][b] = X16<P1>::mfma(cur, in1[b][s - NS0], c);
        });
        // epilogue of the previous pair: block k after the second tile's MFMAs of k-step k (or after the lone last tile's, when the
        // layer has an odd tile count > 1: the views layer's alpha tile)
        constexpr bool host = (u == 1) || (NT % 2 == 1 && t == NT - 1);
        if constexpr (p > 0 && host && s < NB) {
            epi(std::integral_constant<int, p - 1>{}, std::integral_constant<int, s>{}, pinned(acc[set ^ 1][0][s]), pinned(acc[set ^ 1][1][s]));
        }
        // a layer with fewer k-steps than blocks (the encoding layer: 2): the rest of that epilogue at its last k-step
        if constexpr (p > 0 && host && s == NS - 1 && NS < NB) {
            static_for<NS, NB>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                epi(std::integral_constant<int, p - 1>{}, kc, pinned(acc[set ^ 1][0][k]), pinned(acc[set ^ 1][1][k]));
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

// EPL: 0 = the raw outputs go to memory (NetArgs::raw4); 1..4 = the compositing is fused in (NetArgs::fuse_on) for passes of up to
// 64 EPL samples per ray, a lane owning EPL of them (composite_ray<EPL>).  One kernel per case rather than a switch inside one: the loop
// body is straight-line code several times the instruction cache (every jump to code that is not next in line costs ~2000 cycles,
// NRN_TIMING), the flag and the case fold at compile time, and the fine pass measured 0.5-1 % faster (tools/experiments/README.md).
// VIEWS: the view-dependent head (rnh:284-304) behind the trunk -- a sample's direction = the finite difference of the points of
// NetArgs::pts4 along its ray (rnh:339-351; the neighbour read from the same array, as the 32x32x16 trunk-only kernel does), its
// encoding one more B operand; [views_linears[0] o feature_linear | alpha_linear] and rgb_linear as two more dense_x16 calls.
// SAMPLE (EPL > 0): the pass is the COARSE one of a hierarchical render -- behind composite_ray the wave also draws its ray's importance
// depths, z_std and the merge (sample_merge_ray, nrnerf_composite_ray.h: composite_kernel<EPL, true>'s own code) from a private LDS area,
// i.e. K1 of nrnerf_render's sequence becomes this kernel's epilogue and the coarse raw array never reaches HBM (train.py:889-920).
template <class P, class A, int WAVES, int EPL, bool VIEWS = false, bool SAMPLE = false>
__global__ void __launch_bounds__(WAVES * 64, 1) net_kernel_x16(const NetArgs a) {
    using PL = PlanX16<P, A, VIEWS>;
    using PE = PolF16;                                                // the encoding's operands are f16 in both modes
    using frag = typename P::frag;
    using efrag = typename PE::frag;
    constexpr int NB = X16Cfg<A>::NB, NS_H = PL::NS_H, NS_E = PL::NS_E, PFK = X16Cfg<A>::PF;
    static_assert(WAVES == X16Cfg<A>::WAVES, "launched with the architecture's own workgroup size");
    static_assert(A::L == 10, "the encoding's slot layout below is spelt out for ten frequencies (x16_enc_col)");

#ifdef NRN_TIMING
    const unsigned long long rt_kernel_start = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* bias_lds = (float*)(smem + RING * P::UNIT_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, n = lane & 15;
    for (int i = tid; i < PL::NTILES * 16; i += WAVES * 64) bias_lds[i] = a.bias[i];
    if (tid < 32) bias_lds[PL::NTILES * 16 + tid] = 0.0f;             // (the tile of zeros and the mailbox)
    __syncthreads();
    const __attribute__((address_space(3))) f32x4* bias_lane = (const __attribute__((address_space(3))) f32x4*)(bias_lds + 4 * g);
    asm volatile("" : "+v"(bias_lane));
    WRing<P, WAVES, PL::NUP> st;
    st.init(a.wstream, ring, wave, lane);

    const int S = a.S;
    const int bpr = (S + 15) >> 4;
    const long long nblocks = (long long)a.n_rays * bpr;
    const long long per_wg = (long long)WAVES * NB;
    // Fused compositing (NetArgs::fuse_on; as nrnerf_net_mb.h): a WAVE owns whole rays.  Its blocks come in groups of RW rays = RW * bpr
    // blocks = TG iterations of NB blocks (RW: the fewest rays whose blocks fill whole iterations), groups strided over the grid; the raw
    // outputs of a group are staged in the wave's own LDS area and after the group's last iteration the wave composites its RW rays
    // itself (composite_ray): no exchange between waves, no barrier, the pass' raw array never reaches HBM.
    constexpr bool fuse = EPL > 0;
    const int RW = (bpr % NB == 0) ? 1 : ((2 * bpr) % NB == 0 ? 2 : NB);
    const int TG = RW * bpr / NB;
    const long long ngroups = ((long long)a.n_rays + WAVES * RW - 1) / (WAVES * RW);
    // (behind the bias table: 64 bytes of zeros, then the mailbox through which wave 0 hands the workgroup's next group index to the
    //  other waves -- apart from the table, so that nothing that reads a tile past its end can ever see an index)
    int* const mailbox = (int*)(bias_lds + PL::NTILES * 16 + 16);
    f32x4* const stage0 = (f32x4*)(bias_lds + PL::NTILES * 16 + 32);
    f32x4* const stage_w = stage0 + (size_t)wave * RW * bpr * 16;
    const CompositeArgs& fa = *(const CompositeArgs*)(stage0 + (size_t)WAVES * RW * bpr * 16);      // (in LDS: see nrnerf_net_mb.h)
    static_assert(sizeof(CompositeArgs) <= 256, "the compositing arguments' LDS slot");
    // (SAMPLE) this wave's cdf / bins [64 EPL each] and merge list [S + I <= 256 (+ 4)]
    constexpr int SCR = SAMPLE ? (2 * 64 * EPL + 260) : 0;
    float* const scr_w = (float*)((char*)(stage0 + (size_t)WAVES * RW * bpr * 16) + 256) + (size_t)wave * SCR;
    if constexpr (fuse) {
        int* dst = (int*)(stage0 + (size_t)WAVES * RW * bpr * 16);
        const int* src = (const int*)&a.fuse;
        for (int i = tid; i < (int)(sizeof(CompositeArgs) / 4); i += WAVES * 64) dst[i] = src[i];
        __syncthreads();
    }
    // Which groups (fused compositing: of WAVES * RW rays = TG iterations; else: one iteration's WAVES * NB blocks) a workgroup takes: group
    // blockIdx.x and every gridDim.x-th one after it, or -- NetArgs::work_counter -- the next index of a device counter whenever it starts a
    // new group.  The index is needed at the advance point of the group's LAST iteration (the next iteration's points are requested
    // there); wave 0 asks for it two iterations earlier (an atomic behind the iteration's last LDS-DMA requests, where the point loads
    // sit: the vector-memory queue retires in order) and publishes it one iteration earlier through the LDS mailbox (slot = index & 1:
    // rewritten two groups later at the earliest); the ring barriers of the iteration in between order the write before the reads.
    const bool dyn = a.work_counter != nullptr;
    const int TGd = fuse ? TG : 1;                                  // iterations per group of the counter's numbering
    const int tg_pub = ((-2 % TGd) + TGd) % TGd, tg_iss = ((-3 % TGd) + TGd) % TGd;
    unsigned pend = 0;                                              // (wave 0, lane 0) the index asked for at the previous issue point
    int kseq = 0;                                                   // this workgroup's groups so far
    long long first_id = blockIdx.x;
    if (dyn) {
        if (tid == 0) {
            mailbox[0] = (int)atomicAdd(a.work_counter, 1u);
            if (TGd == 1) { mailbox[1] = (int)atomicAdd(a.work_counter, 1u); pend = atomicAdd(a.work_counter, 1u); }
            else if (TGd == 2) pend = atomicAdd(a.work_counter, 1u);
        }
        __syncthreads();
        first_id = __builtin_amdgcn_readfirstlane(mailbox[0]);
        __syncthreads();                                            // (slot 0 is written again at the publish point of index 2)
    }
    int tg = 0;
    long long grp = first_id;
    // (opaque: re-derived inside the loop, gridDim.x is a scalar load from the dispatch packet -- ~4000 cycles per iteration
    //  measured with NRN_TIMING on the path that advances every iteration)
    unsigned gdim = gridDim.x;
    asm volatile("" : "+s"(gdim));
    float cpre[NB][8];          // direction and depths of this wave's rays, requested one iteration ahead of their use
    // where block b of the iteration (tg_, grp_ | b0_) lies: its sample's row in the [N, S] arrays, and whether the lane has a sample
    // (rows as 32-bit numbers: n_rays <= 2^20 per launch and S <= 1024)
    auto locate = [&](int tg_, long long grp_, long long b0_, int b, unsigned& so_, bool& ok_, unsigned& nb_, bool& first_) {
        bool blk_ok;
        int ray, bir;
        if constexpr (fuse) {                   // block q of this wave's group: ray (grp * WAVES + wave) * RW + q / bpr
            const int q = tg_ * NB + b;
            const long long rr = (grp_ * WAVES + wave) * RW + q / bpr;
            blk_ok = rr < a.n_rays;
            ray = (int)(blk_ok ? rr : a.n_rays - 1);
            bir = q % bpr;
        } else {
            const long long blk_raw = b0_ + (long long)wave * NB + b;
            blk_ok = blk_raw < nblocks;
            const long long blk = blk_ok ? blk_raw : nblocks - 1;
            ray = (int)(blk / bpr);
            bir = (int)(blk % bpr);
        }
        const int sidx = bir * 16 + n;
        ok_ = blk_ok && sidx < S;
        const int sc = sidx < S ? sidx : S - 1;
        so_ = (unsigned)ray * (unsigned)S + (unsigned)sc;
        first_ = sc == 0;                                          // (VIEWS) sample 0 takes sample 1's direction, rnh:346-348
        nb_ = (unsigned)ray * (unsigned)S + (unsigned)(sc == 0 ? (S > 1 ? 1 : 0) : sc - 1);
    };
    // The points of an iteration are requested during the PREVIOUS one -- right after its last LDS-DMA requests, before its output
    // stores and its compositing: a lone wave per SIMD has nothing else to put between the request and the use, and the loads'
    // latency was 4 000 of an iteration's 96 000 cycles (NRN_TIMING).  Not earlier: the vector-memory queue retires in order, and the
    // ring's counted vmcnt waits would wait for them.
    unsigned so[NB];
    bool ok[NB];
    f32x4 q4n[NB];
    f32x4 q4p[VIEWS ? NB : 1];          // (VIEWS) the neighbouring sample's point
    bool first[NB];
    static_for<0, NB>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        unsigned nb;
        locate(0, grp, first_id * per_wg, b, so[b], ok[b], nb, first[b]);
        q4n[b] = *(const f32x4*)(a.pts4 + (size_t)so[b] * 4);
        if constexpr (VIEWS) q4p[b] = *(const f32x4*)(a.pts4 + (size_t)nb * 4);
    });
#ifdef NRN_TIMING
    // slots: 0 iteration, 1 points + encoding, 2 layers + head, 3 outputs + ring tail, 4 compositing, 5 iteration in 100 MHz ticks,
    //        6 ring waits + barriers (inside 2), 7 iterations
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (long long b0 = first_id * per_wg; fuse ? (grp < ngroups) : (b0 < nblocks); ) {
#ifdef NRN_TIMING
        const unsigned long long t_it = NRN_NOW(), r_it = __builtin_amdgcn_s_memrealtime();
#endif
        if (fuse && tg == TG - 1) {
            static_for<0, NB>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                if (r < RW) {
                    const long long rr = (grp * WAVES + wave) * RW + r;
                    composite_prefetch(fa, (int)(rr < a.n_rays ? rr : a.n_rays - 1), lane, EPL, cpre[r]);
                }
            });
        }
        efrag enc[NB][NS_E];
        // ---- the points' positional encoding, in B-operand order (x16_enc_col): slots 2 i, 2 i + 1 of this lane's group =
        //      (sin, cos) of pair m = 4 i + g (frequency m / 3, coordinate m % 3); groups 2, 3: slots 14, 15 = x, y | z, 0
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            const f32x4 q4 = q4n[b];
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

        // ---- (VIEWS) the samples' directions and their encoding, one k-step in B-operand order (x16_dir_col): slots 2 i, 2 i + 1 (i < 3)
        //      = (sin, cos) of pair m = 4 i + g; slots 6, 7 of groups 0, 1 = the identity columns x, y | z, 0
        efrag encv[NB][1];
        if constexpr (VIEWS) {
            static_assert(A::LV == 4, "the direction encoding's slot layout is spelt out for four frequencies (x16_dir_col)");
            static_for<0, NB>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                const f32x4 q4 = q4n[b], nb4 = q4p[b];
                float dd[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) dd[c] = first[b] ? __fsub_rn(nb4[c], q4[c]) : __fsub_rn(q4[c], nb4[c]);
                const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dd[0], dd[0]), __fmul_rn(dd[1], dd[1])), __fmul_rn(dd[2], dd[2])));
                float dir[3], drev[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { dir[c] = __fdiv_rn(dd[c], __fadd_rn(nrm, 0.000001f)); drev[c] = dir[c] * 0.15915494309189535f; }
                float ev[8];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int m = 4 * i + g;
                    const int f = m / 3, c = m - 3 * f;
                    const float xr = c == 0 ? drev[0] : (c == 1 ? drev[1] : drev[2]);
                    const float r = __builtin_amdgcn_fractf(xr * (float)(1 << f));
                    ev[2 * i] = __builtin_amdgcn_sinf(r);
                    ev[2 * i + 1] = __builtin_amdgcn_cosf(r);
                }
                ev[6] = (g == 0) ? dir[0] : (g == 1 ? dir[2] : 0.0f);
                ev[7] = (g == 0) ? dir[1] : 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) encv[b][0][e] = (_Float16)ev[e];
            });
        }
#ifdef NRN_TIMING
        NRN_TACC(1, t_it);
        const unsigned long long t_net = NRN_NOW();
#endif
        frag ha[NB][NS_H], hb[NB][NS_H];
        frag none[NB][1];                   // (the second source of a layer that has one: never indexed)
        auto keep = [&](auto& out) {
            return [&](auto pc, auto kc, const f32x4& d0, const f32x4& d1) {
                out[decltype(kc)::value][decltype(pc)::value] = x16_pack<P>(d0, d1);
            };
        };
        dense_x16<PE, P, PL, 0, NS_E, 0, NB, PFK>(st, bias_lane, enc, none, keep(ha));
        static_for<1, A::D>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr bool skip = (i - 1 == A::SKIP);
            if constexpr (i % 2 == 1) {
                if constexpr (skip) dense_x16<PE, P, PL, i, NS_E, NS_H, NB, PFK>(st, bias_lane, enc, ha, keep(hb));
                else dense_x16<P, P, PL, i, NS_H, 0, NB, PFK>(st, bias_lane, ha, none, keep(hb));
            } else {
                if constexpr (skip) dense_x16<PE, P, PL, i, NS_E, NS_H, NB, PFK>(st, bias_lane, enc, hb, keep(ha));
                else dense_x16<P, P, PL, i, NS_H, 0, NB, PFK>(st, bias_lane, hb, none, keep(ha));
            }
        });
        // ---- head: one tile; group 0 holds channels 0..3 (rgb, sigma) of its sample, group 1 channel 4 in its first register
        f32x4 raw[NB];
        auto take = [&](auto, auto kc, const f32x4& d0, const f32x4&) { raw[decltype(kc)::value] = d0; };
        if constexpr (!VIEWS) {
            if constexpr ((A::D - 1) % 2 == 1) dense_x16<P, P, PL, PL::L_HEAD, NS_H, 0, NB, PFK>(st, bias_lane, hb, none, take);
            else dense_x16<P, P, PL, PL::L_HEAD, NS_H, 0, NB, PFK>(st, bias_lane, ha, none, take);
        } else {
            // view-dependent head: hv = relu(views o feature ([enc(dir), h])) in tile pairs 0 .. NT_V / 2 - 1, sigma = the lone last tile's
            // row 0 (group 0's first register, no relu); then rgb = rgb_linear(hv): raw = [rgb, sigma]
            constexpr int NS_V = PL::NS_V;
            frag hv[NB][NS_V];
            float sigma[NB];
            auto views_epi = [&](auto pc, auto kc, const f32x4& d0, const f32x4& d1) {
                constexpr int p = decltype(pc)::value, k = decltype(kc)::value;
                if constexpr (p < NS_V) hv[k][p] = x16_pack<P>(d0, d1);
                else sigma[k] = d0[0];
            };
            if constexpr ((A::D - 1) % 2 == 1) dense_x16<PE, P, PL, PL::L_VIEWS, 1, NS_H, NB, PFK>(st, bias_lane, encv, hb, views_epi);
            else dense_x16<PE, P, PL, PL::L_VIEWS, 1, NS_H, NB, PFK>(st, bias_lane, encv, ha, views_epi);
            dense_x16<P, P, PL, PL::L_HEAD, NS_V, 0, NB, PFK>(st, bias_lane, hv, none, take);
            static_for<0, NB>([&](auto bc) { raw[decltype(bc)::value][3] = sigma[decltype(bc)::value]; });
        }
#ifdef NRN_TIMING
        NRN_TACC(2, t_net);
        const unsigned long long t_out = NRN_NOW();
#endif
        // the stream's padding units: the ring runs on into the next iteration's first units
        static_for<PL::NUNITS, PL::NUP>([&](auto uc) { st.template advance<decltype(uc)::value>(); });
        // the next iteration's points, requested behind the ring's last LDS-DMA requests and ahead of everything that is left to do here
        int tg_n = tg;
        long long grp_n = grp, b0_n = b0;
        const int tgd = fuse ? tg : 0;
        long long next_id = 0;
        if (dyn) {
            if (tgd + 1 == TGd) next_id = __builtin_amdgcn_readfirstlane(mailbox[(kseq + 1) & 1]);      // published an iteration ago
            if (wave == 0 && lane == 0) {
                if (tgd == tg_pub) mailbox[(kseq + (TGd == 1 ? 2 : 1)) & 1] = (int)pend;               // (asked for an iteration ago)
                if (tgd == tg_iss) pend = atomicAdd(a.work_counter, 1u);
            }
            if (tgd + 1 == TGd) ++kseq;
        }
        if constexpr (fuse) {
            if (tg + 1 == TG) { tg_n = 0; grp_n = dyn ? next_id : grp + gdim; } else tg_n = tg + 1;
        } else {
            b0_n = dyn ? next_id * per_wg : b0 + (long long)gdim * per_wg;
        }
        unsigned so_n[NB];
        bool ok_n[NB];
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            unsigned nb;
            locate(tg_n, grp_n, b0_n, b, so_n[b], ok_n[b], nb, first[b]);
            q4n[b] = *(const f32x4*)(a.pts4 + (size_t)so_n[b] * 4);
            if constexpr (VIEWS) q4p[b] = *(const f32x4*)(a.pts4 + (size_t)nb * 4);
        });
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            if (ok[b] && g == 0) {
                if (!fuse) *(f32x4*)(a.raw4 + (size_t)so[b] * 4) = raw[b];
                if (a.raw_out) {
                    float* ro = a.raw_out + (size_t)so[b] * a.raw_ch;
                    ro[0] = raw[b][0]; ro[1] = raw[b][1]; ro[2] = raw[b][2]; ro[3] = raw[b][3];
                }
            }
            if (ok[b] && g == 1 && a.raw_out && a.raw_ch > 4) a.raw_out[(size_t)so[b] * a.raw_ch + 4] = raw[b][0];
            if (fuse && g == 0) stage_w[(tg * NB + b) * 16 + n] = raw[b];
        });
#ifdef NRN_TIMING
        NRN_TACC(3, t_out);
        const unsigned long long t_comp = NRN_NOW();
#endif
        // (hinted: the iteration that does not composite falls through to the back edge)
        if constexpr (fuse) {
            if (__builtin_expect(tg + 1 == TG, 0)) {    // the group's last iteration: composite this wave's rays from its LDS stage (train.py:943-950)
                static_for<0, NB>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    if (r < RW) {
                        const long long rr = (grp * WAVES + wave) * RW + r;
                        const bool ray_ok = rr < a.n_rays;
                        const int cray = (int)(ray_ok ? rr : a.n_rays - 1);
                        const f32x4* sw = stage_w + r * bpr * 16;
                        auto raw_at = [&](int ic) { return sw[ic]; };
                        float cz[EPL + 1], cw[EPL];
                        composite_ray<EPL, true>(fa, cray, ray_ok, lane, raw_at, cz, cw, cpre[r]);
                        if constexpr (SAMPLE) sample_merge_ray<EPL, true>(fa, cray, ray_ok, lane, cz, cw, scr_w, scr_w + 64 * EPL, scr_w + 128 * EPL);
                    }
                });
            }
        }
        tg = tg_n; grp = grp_n; b0 = b0_n;
        static_for<0, NB>([&](auto bc) { so[decltype(bc)::value] = so_n[decltype(bc)::value]; ok[decltype(bc)::value] = ok_n[decltype(bc)::value]; });
#ifdef NRN_TIMING
        NRN_TACC(4, t_comp);
        NRN_TACC(0, t_it);
        tacc[5] += __builtin_amdgcn_s_memrealtime() - r_it;
        tacc[7] += 1;
#endif
    }
    st.drain();
#ifdef NRN_TIMING
    if (blockIdx.x == 0 && lane == 0 && wave < 8) {
        tacc[6] = st.bar_cycles;
        for (int i = 0; i < 8; ++i) g_nrn_timing[wave][i] += tacc[i];
    }
    // (WAVES == 4: rows 4..7 are free) when the waves 0 of sixteen workgroups spread over the grid left the loop, in 100 MHz ticks: the
    // spread is the share of the launch that runs with idle CUs
    // (workgroups 0..7: one per XCD; 8..15 their neighbours on the same XCDs)
    if (WAVES == 4 && lane == 0 && wave == 0 && blockIdx.x < 16) {
        const int slot = (int)blockIdx.x;
        g_nrn_timing[4 + slot / 8][slot % 8] = __builtin_amdgcn_s_memrealtime();
        if (blockIdx.x == 0) g_nrn_timing[6][0] = rt_kernel_start;
    }
#endif
}

template <class P, class A, int EPL, bool VIEWS = false, bool SAMPLE = false>
static hipError_t launch_net_x16_t(const NetArgs& a, int num_cus, hipStream_t stream) {
    constexpr int WAVES = X16Cfg<A>::WAVES;
    using PL = PlanX16<P, A, VIEWS>;
    constexpr int NB = X16Cfg<A>::NB;
    if (!a.pts4 || (!a.raw4 && !a.fuse_on) || a.S < 1) return hipErrorInvalidValue;
    if ((a.fuse_on != 0) != (EPL > 0) || (EPL > 0 && (a.S + 63) / 64 != EPL)) return hipErrorInvalidValue;      // (the dispatcher's job)
    const int bpr = (a.S + 15) / 16;
    const int RW = (bpr % NB == 0) ? 1 : ((2 * bpr) % NB == 0 ? 2 : NB);
    size_t lds = (size_t)RING * P::UNIT_BYTES + (size_t)PL::NTILES * 16 * sizeof(float) + 128;     // ring | bias table | a tile of zeros | mailbox
    if (a.fuse_on) {
        if (a.S > 256 || (a.fuse.n_importance != 0) != SAMPLE || a.fuse.S != a.S) return hipErrorInvalidValue;
        lds += (size_t)WAVES * RW * bpr * 16 * 16 + 256;            // the waves' raw stages + the compositing arguments
        if (SAMPLE) {
            if (a.S + a.fuse.n_importance > 256 || !a.fuse.z_out) return hipErrorInvalidValue;
            lds += (size_t)WAVES * (2 * 64 * EPL + 260) * sizeof(float);
        }
    }
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = net_kernel_x16<P, A, WAVES, EPL, VIEWS, SAMPLE>;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        // the largest request any launch can make: ring + bias table + the fused stages at bpr = 15 (RW = NB)
        const size_t lds_max = (size_t)RING * P::UNIT_BYTES + (size_t)PL::NTILES * 16 * sizeof(float) + 128 + (size_t)WAVES * NB * 15 * 256 + 256 +
                               (SAMPLE ? (size_t)WAVES * (2 * 64 * EPL + 260) * sizeof(float) : 0);
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_max < 160 * 1024 ? lds_max : 160 * 1024));
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    long long want;
    if (a.fuse_on) want = ((long long)a.n_rays + WAVES * RW - 1) / (WAVES * RW);          // groups of WAVES * RW whole rays
    else want = ((long long)a.n_rays * bpr + WAVES * NB - 1) / (WAVES * NB);
    if (want <= 0) return hipSuccess;
    const int grid = (int)(want < num_cus ? want : num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace nrn
