#pragma once
// nrnerf_net_impl.h -- the fused per-sample network kernel for gfx950 (MI355X).
//
// One launch evaluates, for every sample of every ray of a pass:
//   point generation           (reference train.py:847-873 / 921-923)
//   ray_bending offset MLP + rigidity MLP + masking      (run_nerf_helpers.py:507-577)
//   positional encoding of the bent point                (run_nerf_helpers.py:120-150, 582-584)
//   8x256 trunk with skip concatenation + output head    (run_nerf_helpers.py:272-306)
// and writes raw[N,S,4] for the composite kernel.  Nothing else touches HBM: the 95-float
// per-sample network input the reference materialises (train.py:88-90) never exists.
//
// Execution model:
//   * a wave owns a block of 32 consecutive samples of one ray; lane = sample (lane & 31), the
//     two lane halves split the k (feature) index as the MFMA B operand requires;
//   * every layer is D^T = W * H^T on v_mfma_f32_32x32x16_{bf16,f16} (or 32x32x2_f32): weights are
//     the A operand, activations the B operand, and the D tile is already (a permutation of) the
//     next layer's B operand -- activations stay in VGPRs from the encoding to the head;
//   * weights are a linear stream of pre-permuted A fragments in HBM/L2 (nrnerf_plan.h); the
//     workgroup stages it through an LDS ring unit by unit, all waves consume each unit;
//   * workgroups are persistent: grid = #CUs, each loops over tiles of WAVES blocks.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "nrnerf_kernels.h"
#include "nrnerf_plan.h"
#include "nrnerf_composite_ray.h"

#ifndef NRN_PF16
#define NRN_PF16 4
#endif
#ifndef NRN_EPI_DELAY
#define NRN_EPI_DELAY 4      // MFMAs of the next tile issued before the previous tile's epilogue
#endif

namespace nrn {

// Opt-in phase timing (make TUNE=-DNRN_TIMING SUFFIX=_timing; tools/timing_probe.py): the waves of workgroup 0
// accumulate s_memtime deltas per phase and around every ring barrier.  Costs ~10 % and is never in the shipped build.
#ifdef NRN_TIMING
static __device__ unsigned long long g_nrn_timing[8][8];
#define NRN_NOW() __builtin_amdgcn_s_memtime()
#define NRN_TACC(slot, t0) tacc[slot] += NRN_NOW() - (t0)
#else
#define NRN_NOW() 0ull
#define NRN_TACC(slot, t0) ((void)0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// compile-time loop, in order, instantiation depth log2(N - I) (the f32 layers have > 1000 steps)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N - I == 1) {
        f(std::integral_constant<int, I>{});
    } else if constexpr (N - I > 1) {
        constexpr int M = I + (N - I) / 2;
        static_for<I, M>(f);
        static_for<M, N>(f);
    }
}

// ------------------------------------------------------------------------------------------
// precision policies
// ------------------------------------------------------------------------------------------
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// accumulator registers 8U..8U+7 -> one 16-bit B slab, optionally through relu: 4 x v_cvt_pk_{bf16,f16}_f32 and
// 4 x v_pk_max_i16 (relu on the packed pairs: negative 16-bit floats are negative int16; the sign survives
// rounding).  Written pair-wise because hipcc only selects the packed conversion for 2-vectors.
template <class F8, class F2, int U, bool RELU>
__device__ __forceinline__ F8 pack16(const f32x16& c) {
    u32x4 w;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x2 t = {c[8 * U + 2 * k], c[8 * U + 2 * k + 1]};
        const F2 p = __builtin_convertvector(t, F2);
        s16x2 q = __builtin_bit_cast(s16x2, p);
        if (RELU) q = __builtin_elementwise_max(q, (s16x2)(short)0);
        w[k] = __builtin_bit_cast(unsigned, q);
    }
    return __builtin_bit_cast(F8, w);
}

// relu(x) for a finite-or-inf float as ONE integer max: negative floats are negative as signed ints.
// (fmaxf on an MFMA result costs two v_max_f32: hipcc inserts a canonicalising max first.)
__device__ __forceinline__ float relu_bits(float x) {
    return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0));
}
struct PolBF16 : Shape<8, false> {       // bf16 trunk, single-product f16 bender (nrnerf_plan.h Shape::SPLIT)
    typedef __bf16 frag __attribute__((ext_vector_type(8)));
    static constexpr int PF = NRN_PF16;      // A-fragment software prefetch depth (4 VGPRs each)
    typedef __bf16 frag2 __attribute__((ext_vector_type(2)));
    template <int U, bool RELU>
    static __device__ __forceinline__ frag from_acc(const f32x16& c) { return pack16<frag, frag2, U, RELU>(c); }
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    template <int E>
    static __device__ __forceinline__ void set(frag& f, float v) { f[E] = (__bf16)v; }
    static __device__ __forceinline__ float round(float v) { return (float)(__bf16)v; }
};
struct PolF16 : Shape<8, true> {         // f16 trunk, fp32-equivalent split-product bender
    typedef _Float16 frag __attribute__((ext_vector_type(8)));
    static constexpr int PF = NRN_PF16;
    typedef _Float16 frag2 __attribute__((ext_vector_type(2)));
    template <int U, bool RELU>
    static __device__ __forceinline__ frag from_acc(const f32x16& c) { return pack16<frag, frag2, U, RELU>(c); }
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    template <int E>
    static __device__ __forceinline__ void set(frag& f, float v) { f[E] = (_Float16)v; }
    static __device__ __forceinline__ float round(float v) { return (float)(_Float16)v; }
};
struct PolF32 : Shape<1> {
    typedef float frag;
    static constexpr int PF = 8;
    template <int U, bool RELU>
    static __device__ __forceinline__ frag from_acc(const f32x16& c) { return RELU ? relu_bits(c[U]) : c[U]; }
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    template <int E>
    static __device__ __forceinline__ void set(frag& f, float v) { f = v; }
    static __device__ __forceinline__ float round(float v) { return v; }
};

// ------------------------------------------------------------------------------------------
// weight stream: HBM/L2 -> LDS ring by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip).
//
// The stream is a sequence of 16 KiB units; unit U lives in ring slot U % RING.  All waves of the
// workgroup consume every unit, each wave DMAs PW 1-KiB pieces of it.  advance<U>() is called (at a
// compile-time-known point) right before the first fragment of unit U is read.  With LAG = NRN_RING_LAG:
//     s_waitcnt vmcnt((RING-LAG-1)*PW) [lgkmcnt(0) if LAG == 1]   my pieces of unit U have landed
//     s_barrier                                 => everyone's pieces of U landed
//     issue DMA of unit U+RING-LAG into the slot of unit U-LAG
// Why the recycled slot is free.  LAG = 1: every wave waited lgkmcnt(0) before the barrier, so its reads of
// U-1 have retired (this also drains the fragment prefetch queue, a stall).  LAG = 2: a wave arriving at
// barrier U has issued the MFMAs of every fragment of unit U-2 (the prefetch runs at most PF < 16 fragments
// ahead), hence waited for those reads; no drain needed, at the price of one more ring slot per unit of lead.
// vmcnt retires in order, so other VMEM traffic in flight (the per-block ray loads / raw stores) can only
// make the counted wait stricter, never weaker.
// ------------------------------------------------------------------------------------------
#ifndef NRN_RING_LAG
#define NRN_RING_LAG 2
#endif
template <int N>
__device__ __forceinline__ void wait_ring() {
    if constexpr (NRN_RING_LAG == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <class P, int WAVES, int NUP>
struct WRing {
    static constexpr int UNIT = P::UNIT_BYTES;
    // (Tried and rejected on MI355X: letting only 2 or 4 "loader" waves issue the DMA so their SIMD partners keep the
    //  matrix pipe busy -- 2-4 % slower, and the wave-uniform branches alone cost 7 %.)
    static constexpr int PW = UNIT / 1024 / WAVES;          // DMA instructions per wave per unit
    static constexpr int LAG = NRN_RING_LAG;                // recycle the slot of unit U - LAG at advance<U>
    static_assert(LAG >= 1 && RING - LAG >= 2, "need at least one unit of DMA lead");
    static_assert(UNIT % (1024 * WAVES) == 0, "unit must split evenly over the waves");
    static_assert(NUP % RING == 0 && NUP >= RING, "unit count must be a padded multiple of the ring depth");
    const char* ubase;     // stream + this wave's piece offset: wave-uniform, lives in SGPRs
    unsigned lane16;       // lane * 16: the only per-lane part of a DMA source address (saddr + voffset form)
    char* ring;            // LDS ring base
    int wave_off;          // this wave's piece offset inside a unit (wave-uniform)
    int lane_off;          // lane * (FRAG_BYTES / 64)
    unsigned lane_addr;    // LDS byte address of this lane's part of fragment 0 of slot 0
#ifdef NRN_TIMING
    unsigned long long bar_cycles = 0, bar_count = 0;
#endif

    __device__ __forceinline__ void init(const void* stream, char* lds, int wave, int lane) {
        wave_off = wave * PW * 1024;
        ubase = (const char*)stream + wave_off;
        lane16 = (unsigned)lane * 16u;
        ring = lds;
        lane_off = lane * (P::FRAG_BYTES / 64);
        lane_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (unsigned)lane_off;
        static_for<0, RING - LAG>([&](auto uc) { issue<decltype(uc)::value>(); });
    }
    template <int V>
    __device__ __forceinline__ void issue() {
        // One address per unit: the offset is a compile-time constant, but hiding it from the optimiser stops LICM from
        // hoisting one 64-bit per-lane address per unit out of the persistent loop (spilled VGPR pairs, and every scratch
        // reload drains the DMA queue with a vmcnt(0)).  The pieces of a wave are 1 KiB apart in the stream and in the
        // slot, which is what the instruction's immediate offset (added to the global and to the LDS address) expresses.
        unsigned off = (unsigned)(V * UNIT);
        asm volatile("" : "+s"(off));
        const char* src = (ubase + off) + lane16;
        char* dst = ring + (V % RING) * UNIT + wave_off;
        static_for<0, PW>([&](auto ic) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, decltype(ic)::value * 1024, 0);
        });
    }
    template <int U>
    __device__ __forceinline__ void advance() {
#ifdef NRN_TIMING
        const unsigned long long tb0 = NRN_NOW();
#endif
        wait_ring<(RING - LAG - 1) * PW>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef NRN_TIMING
        bar_cycles += NRN_NOW() - tb0; ++bar_count;
#endif
        issue<(U + RING - LAG) % NUP>();
    }
    // fragment GF (index in the whole stream) for this lane; advances the ring when GF opens a new unit.
    //
    // 16-byte fragments are read with an explicit `ds_read_b128 vdst, vaddr offset:imm` and consumed after an explicit
    // counted `s_waitcnt lgkmcnt(N)` (ready<N>): left to itself hipcc re-sinks the prefetched reads next to their MFMA
    // and drains the queue with lgkmcnt(0) -- which schedule it picks flips with unrelated changes of register pressure.
    // LDS operations retire in order, so at the consumer "at most N younger LDS operations outstanding" implies this
    // read is done; LDS traffic the compiler adds on its own (bias table, mailbox) only makes the wait stricter, and its
    // own waits (computed without knowing about these reads) can only be stricter than needed.  No scalar memory load
    // may be in flight across a counted wait (SMEM returns out of order): the kernel reads its arguments before the
    // persistent loop, and tools/check_isa.py asserts there is no s_load inside the loop.
    static constexpr bool ASM_FRAGS = (P::FRAG_BYTES == 1024);
    template <class PX, int GF>
    __device__ __forceinline__ typename PX::frag frag() {
        static_assert(PX::FRAG_BYTES == P::FRAG_BYTES, "mixed policies must share the fragment size");
        constexpr int UF = P::UNIT_FRAGS;
        if constexpr (GF % UF == 0) advance<GF / UF>();
        constexpr int OFF = ((GF / UF) % RING) * UNIT + (GF % UF) * P::FRAG_BYTES;
        if constexpr (ASM_FRAGS) {
            static_assert(OFF + 16 <= 65536, "ring must stay within the immediate ds_read offset");
            u32x4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lane_addr), "n"(OFF));
            return __builtin_bit_cast(typename PX::frag, v);
        } else {
            const char* p = ring + OFF + lane_off;
            return *(const typename PX::frag*)p;
        }
    }
    // the fragment read N LDS operations before the most recent one has landed
    template <int N, class F>
    __device__ __forceinline__ void ready(F& f) {
        if constexpr (ASM_FRAGS) {
            static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
            u32x4 v = __builtin_bit_cast(u32x4, f);
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
            f = __builtin_bit_cast(F, v);
        }
    }
    __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// ------------------------------------------------------------------------------------------
// one dense layer: for every output tile t, acc = bias; acc += A(t,s) * B(s) over the input slabs;
// `epi(t, acc)` consumes the 32x32 fp32 tile.  B slabs come from in0 (first NS0, policy P0) then
// in1 (NS1, policy P1): the skip layer mixes f16 encoding slabs with bf16 hidden slabs in one
// fp32 accumulator.
// ------------------------------------------------------------------------------------------
// `bias_lane` is the per-lane LDS address (table + 64 * (lane >> 5)), made opaque once in the kernel prologue
// (BiasPtr): the table sits above the 64 KiB ring, so with a visible base hipcc materialises one VGPR per tile
// address and hoists all ~70 of them out of the persistent loop (64 VGPRs + 12 scratch slots, each reload draining
// the DMA queue with a vmcnt(0)); from an opaque base every tile is an immediate ds_read offset.
typedef const __attribute__((address_space(3))) f32x4* BiasPtr;
__device__ __forceinline__ BiasPtr bias_lane_ptr(const float* bias_lds, int h) {
    BiasPtr p = (BiasPtr)(bias_lds + h * 16);
    asm volatile("" : "+v"(p));
    return p;
}
__device__ __forceinline__ f32x16 load_bias(BiasPtr bias_lane, int tile) {
    BiasPtr bp = bias_lane + tile * 8;
    const f32x4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
    return f32x16{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3],
                  b2[0], b2[1], b2[2], b2[3], b3[0], b3[1], b3[2], b3[3]};
}

// Position q of a layer's fragment sequence (stream order, nrnerf_plan.h) -> (tile, slab).  Tiles are paired with
// interleaved slabs; an odd last tile follows alone.
template <int NT, int NS>
struct SeqPos {
    static constexpr int NPAIR = NT / 2;
    static constexpr int tile(int q) { return q < NPAIR * 2 * NS ? 2 * (q / (2 * NS)) + (q & 1) : NT - 1; }
    static constexpr int slab(int q) { return q < NPAIR * 2 * NS ? (q % (2 * NS)) / 2 : q - NPAIR * 2 * NS; }
    static constexpr bool last_of_tile(int q) { return slab(q) == NS - 1; }
};

// Priority toggling (branch-free wave balancing).  The two waves sharing a SIMD run the same code; the older one wins
// every arbitration, finishes its unit early and then idles at the ring barrier while its partner runs alone (phase
// timing: waves 0-3 waited 31 % of the time, waves 4-7 9 %).  Each wave therefore raises its priority for NRN_PRIO_HALF
// MFMAs and lowers it for the next NRN_PRIO_HALF: whichever wave is ahead soon sits in a low-priority stretch while
// the laggard is in a high-priority one, so the pair advances in lock-step.  0 disables.
#ifndef NRN_PRIO_HALF
#define NRN_PRIO_HALF 0
#endif
template <int Q>
__device__ __forceinline__ void prio_tick() {
    if constexpr (NRN_PRIO_HALF > 0) {
        if constexpr (Q % (2 * NRN_PRIO_HALF) == 0) __builtin_amdgcn_s_setprio(1);
        else if constexpr (Q % (2 * NRN_PRIO_HALF) == NRN_PRIO_HALF) __builtin_amdgcn_s_setprio(0);
    }
}

#ifndef NRN_NACC
#define NRN_NACC 2      // accumulator sets per wave (4: the previous pair's epilogue overlaps the next pair's MFMAs, but the
                        // 32 extra registers spill inside the trunk at 256 VGPRs)
#endif

// Fragments are prefetched PF steps ahead across the whole layer (one flat sequence), so a ds_read_b128 is in
// flight for ~PF MFMAs before its consumer.  Two tiles run as interleaved accumulator chains (see nrnerf_plan.h);
// with NACC = 4 the chains of pair p+1 start in fresh accumulator sets while pair p's epilogue (convert, relu) is
// issued DLY steps later, so the VALU work overlaps the matrix pipe.
template <class P0, class P1, class PL, int LI, int NS0, int NS1, class ST, class IN0, class IN1, class EPI>
__device__ __forceinline__ void dense(ST& st, BiasPtr bias_lane, const IN0& in0, const IN1& in1, EPI&& epi) {
    constexpr LayerSpec spec = PL::TB.layers[LI];
    static_assert(spec.ns == NS0 + NS1 && spec.split == 0, "slab count mismatch between kernel and plan");
    constexpr int NS = NS0 + NS1, NT = spec.nt, Q = NT * NS, PF = P1::PF;
    using SQ = SeqPos<NT, NS>;
    constexpr int G0 = PL::TB.tiles[spec.tile0].gbase;
    constexpr int NACC = (NT >= 4) ? NRN_NACC : (NT >= 2 ? 2 : 1);
    typename P1::frag a[PF];
    auto load = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int s = SQ::slab(q);
        if constexpr (s < NS0) a[q % PF] = __builtin_bit_cast(typename P1::frag, st.template frag<P0, G0 + q>());
        else a[q % PF] = st.template frag<P1, G0 + q>();
    };
    static_for<0, (PF < Q ? PF : Q)>([&](auto qc) { load(qc); });
    constexpr int DLY = (NS - 1 < NRN_EPI_DELAY) ? NS - 1 : NRN_EPI_DELAY;
    f32x16 accs[NACC];
    static_for<0, (NACC < NT ? NACC : NT)>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (NACC == 4 || t < 2) accs[t % NACC] = load_bias(bias_lane, spec.tile0 + t);
    });
    static_for<0, Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int t = SQ::tile(q), s = SQ::slab(q);
        prio_tick<q>();
        st.template ready<(Q - 1 - q < PF - 1) ? Q - 1 - q : PF - 1>(a[q % PF]);     // reads q+1 .. q+PF-1 may be in flight
        const typename P1::frag cur = a[q % PF];
        if constexpr (q + PF < Q) load(std::integral_constant<int, q + PF>{});
        if constexpr (s < NS0) accs[t % NACC] = P0::mfma(__builtin_bit_cast(typename P0::frag, cur), in0[s], accs[t % NACC]);
        else accs[t % NACC] = P1::mfma(cur, in1[s - NS0], accs[t % NACC]);
        if constexpr (NACC == 4) {
            // delayed epilogue of the previous pair, then pre-load the biases of the pair after this one
            if constexpr (t >= 2 && (t & 1) == 1 && s == DLY) {
                constexpr int tp = (t & ~1) - 2;
                epi(std::integral_constant<int, tp>{}, accs[tp % NACC]);
                epi(std::integral_constant<int, tp + 1>{}, accs[(tp + 1) % NACC]);
                if constexpr (tp + 4 < NT) accs[(tp + 4) % NACC] = load_bias(bias_lane, spec.tile0 + tp + 4);
                if constexpr (tp + 5 < NT) accs[(tp + 5) % NACC] = load_bias(bias_lane, spec.tile0 + tp + 5);
            }
            if constexpr (q == Q - 1) {     // drain: the last pair (or the odd tile, plus the pair before it)
                constexpr int first = (NT & 1) ? (NT >= 3 ? NT - 3 : NT - 1) : NT - 2;
                static_for<first, NT>([&](auto tc) { epi(tc, accs[decltype(tc)::value % NACC]); });
            }
        } else {
            if constexpr (SQ::last_of_tile(q) && (((t & 1) == 1) || t == NT - 1)) {
                // pair finished: epilogue of its tile(s), then the next pair's biases
                if constexpr ((t & 1) == 1) epi(std::integral_constant<int, t - 1>{}, accs[(t - 1) % NACC]);
                epi(std::integral_constant<int, t>{}, accs[t % NACC]);
                if constexpr (t + 1 < NT) accs[(t + 1) % NACC] = load_bias(bias_lane, spec.tile0 + t + 1);
                if constexpr (t + 2 < NT) accs[(t + 2) % NACC] = load_bias(bias_lane, spec.tile0 + t + 2);
            }
        }
    });
}

// Activations of the bender / rigidity MLPs: value = hi (+ lo in the split 16-bit modes).
template <class PE, int N, bool SPLIT>
struct Act {
    typename PE::frag hi[N];
    typename PE::frag lo[SPLIT ? N : 1];
    template <int S, int E>
    __device__ __forceinline__ void set(float v) {
        PE::template set<E>(hi[S], v);
        if constexpr (SPLIT) PE::template set<E>(lo[S], (v - PE::round(v)) * PE::LO_SCALE);
    }
    // slab S := relu(accumulator registers 8U..8U+7), 16-bit policies.  Per pair of values: 2 x v_max_i32 (relu),
    // v_cvt_pk_f16_f32 (hi), v_pk_mul_f32 (x * 2^11) and 2 x v_fma_mix{lo,hi}_f16 computing f16(x * 2^11 - hi * 2^11)
    // straight from the packed hi halves -- 6 VALU instead of the 10-12 of the element-wise form (convert back,
    // subtract, scale, convert).  Same value: x - hi is exact in fp32.
    template <int S, int U>
    __device__ __forceinline__ void set_slab_relu(const f32x16& c) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        u32x4 wh, wl;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x2 t = {relu_bits(c[8 * U + 2 * k]), relu_bits(c[8 * U + 2 * k + 1])};
            const h2 hh = __builtin_convertvector(t, h2);
            wh[k] = __builtin_bit_cast(unsigned, hh);
            if constexpr (SPLIT) {
                // hipcc does not select the mixed-precision fma here (it converts hi back with v_cvt_f32_f16 + sdwa)
                f32x2 ts;
                const f32x2 sc = {PE::LO_SCALE, -PE::LO_SCALE};
                asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(ts) : "v"(t), "s"(sc));
                unsigned w;      // mixlo leaves the upper half alone; mixhi then defines it
                asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]"
                    : "=v"(w) : "v"(wh[k]), "s"(sc[1]), "v"(ts[0]));
                asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                    : "+v"(w) : "v"(wh[k]), "s"(sc[1]), "v"(ts[1]));
                wl[k] = w;
            }
        }
        hi[S] = __builtin_bit_cast(typename PE::frag, wh);
        if constexpr (SPLIT) lo[S] = __builtin_bit_cast(typename PE::frag, wl);
    }
};

// dense layer of the bender: 3-term split product  Whi*xhi + 2^-11 (Whi*xlo + Wlo*xhi)  when SPLIT.  The two tiles of
// a pair advance together, so no accumulator is written by two consecutive MFMAs.
// Optional tangent operand (exact non-rigid view directions, run_nerf_helpers.py:358-385): `tin` holds d(layer input)/dt
// for ONE direction t (forward-mode differentiation; only J . d is needed, not J).  The tangent of W x is W (dx/dt): one
// more MFMA per (tile, slab) with the hi weight fragment (directions tolerate the f16 rounding of the weights), into its
// own accumulator; `epi` then receives (tile, value, tangent) and applies the relu mask of the value to the tangent.
struct NoTan {};
template <class PE, int N>
struct Tan {
    typename PE::frag t[N];
    template <int S, int E>
    __device__ __forceinline__ void set(float v) { PE::template set<E>(t[S], v); }
};
template <class PE, bool SPLIT, class PL, int LI, int NS, class ST, class ACT, class EPI, class TIN = NoTan>
__device__ __forceinline__ void dense_b(ST& st, BiasPtr bias_lane, const ACT& in, EPI&& epi, const TIN& tin = TIN{}) {
    constexpr bool TANG = !std::is_same_v<TIN, NoTan>;
    constexpr LayerSpec spec = PL::TB.layers[LI];
    static_assert(spec.ns == NS && spec.split == (SPLIT ? 1 : 0), "bender layer mismatch between kernel and plan");
    constexpr int NT = spec.nt;
    constexpr int FP = SPLIT ? 2 : 1;                     // fragments per (tile, slab): hi [, lo]
    static_for<0, (NT + 1) / 2>([&](auto pc) {
        constexpr int t0 = 2 * decltype(pc)::value;
        constexpr int W = (t0 + 1 < NT) ? 2 : 1;          // tiles in this group
        // the group's fragments are consecutive in the stream (slab-major, then tile, then hi/lo): step s is read
        // into w[s & 1] while the MFMAs of step s - 1 run (the layers are too short to hide an LDS round trip per step)
        typename PE::frag w[2][W * FP];
        auto load = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            static_for<0, W>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                constexpr TileInfo ti = PL::TB.tiles[spec.tile0 + t0 + u];
                static_for<0, FP>([&](auto fc) {
                    constexpr int f = decltype(fc)::value;
                    w[s & 1][u * FP + f] = st.template frag<PE, ti.gbase + s * ti.gstride + f>();
                });
            });
        };
        load(std::integral_constant<int, 0>{});
        f32x16 acc[W], corr[W], tacc[TANG ? W : 1];
        static_for<0, W>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            acc[u] = load_bias(bias_lane, spec.tile0 + t0 + u);
            corr[u] = f32x16{};
            if constexpr (TANG) tacc[u] = f32x16{};
        });
        static_for<0, NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            // start step s + 1, then wait for step s (the W * FP younger reads may stay in flight)
            if constexpr (s + 1 < NS) load(std::integral_constant<int, s + 1>{});
            static_for<0, W * FP>([&](auto fc) {
                st.template ready<(s + 1 < NS) ? W * FP : 0>(w[s & 1][decltype(fc)::value]);
            });
            if constexpr (SPLIT) {
                static_for<0, W>([&](auto uc) { constexpr int u = decltype(uc)::value; corr[u] = PE::mfma(w[s & 1][2 * u + 1], in.hi[s], corr[u]); });
                static_for<0, W>([&](auto uc) { constexpr int u = decltype(uc)::value; acc[u] = PE::mfma(w[s & 1][2 * u], in.hi[s], acc[u]); });
                static_for<0, W>([&](auto uc) { constexpr int u = decltype(uc)::value; corr[u] = PE::mfma(w[s & 1][2 * u], in.lo[s], corr[u]); });
            } else {
                static_for<0, W>([&](auto uc) { constexpr int u = decltype(uc)::value; acc[u] = PE::mfma(w[s & 1][u], in.hi[s], acc[u]); });
            }
            if constexpr (TANG)
                static_for<0, W>([&](auto uc) { constexpr int u = decltype(uc)::value; tacc[u] = PE::mfma(w[s & 1][FP * u], tin.t[s], tacc[u]); });
        });
        static_for<0, W>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if constexpr (SPLIT) acc[u] += corr[u] * (1.0f / PE::LO_SCALE);
            if constexpr (TANG) epi(std::integral_constant<int, t0 + u>{}, acc[u], tacc[u]);
            else epi(std::integral_constant<int, t0 + u>{}, acc[u]);
        });
    });
}

// relu + convert a D tile into the SP B-operand slabs it provides to the next layer
template <class P, bool RELU, int T, class OUT>
__device__ __forceinline__ void pack_tile(const f32x16& acc, OUT& out) {
    static_for<0, P::SP>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        out[T * P::SP + u] = P::template from_acc<u, RELU>(acc);
    });
}
// tangent through the relu: d relu(v)/dt = (v > 0) dv/dt
template <class PE, int T, class TAN>
__device__ __forceinline__ void pack_tan(const f32x16& acc, const f32x16& tacc, TAN& out) {
    static_for<0, PE::SP>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        static_for<0, PE::KH>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            out.template set<T * PE::SP + u, e>(acc[u * PE::KH + e] > 0.0f ? tacc[u * PE::KH + e] : 0.0f);
        });
    });
}
template <class PE, int T, class ACT>
__device__ __forceinline__ void pack_act(const f32x16& acc, ACT& out) {
    static_for<0, PE::SP>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        if constexpr (PE::KH == 8) {
            out.template set_slab_relu<T * PE::SP + u, u>(acc);
        } else {
            static_for<0, PE::KH>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                out.template set<T * PE::SP + u, e>(relu_bits(acc[u * PE::KH + e]));
            });
        }
    });
}

// sin/cos of x * 2^k for the positional encoding.  fp32 mode: libm-accurate sincosf of the exact product (the
// reference evaluates torch.sin(x * 2^k) in fp32).  16-bit modes: the result is rounded to f16 (eps 4.9e-4), so the
// hardware transcendental is used: rev = x / 2pi once, scaled by the exact power of two, reduced with v_fract, then
// v_sin_f32 / v_cos_f32 (arguments in revolutions).  Error <= ~1e-4 at the highest frequency, ~6 instructions per
// pair instead of ~100.
template <bool EXACT>
__device__ __forceinline__ void enc_sincos(float x, float x_rev, float scale, float* sv, float* cv) {
    if constexpr (EXACT) {
        sincosf(x * scale, sv, cv);
    } else {
        const float r = __builtin_amdgcn_fractf(x_rev * scale);
        *sv = __builtin_amdgcn_sinf(r);
        *cv = __builtin_amdgcn_cosf(r);
    }
}

// torch.linspace(0, 1, n)[i] in fp32 (ATen RangeFactories: symmetric two-sided evaluation)
__device__ __forceinline__ float lin01(int i, int n) {
    if (n <= 1) return 0.0f;
    const float step = __fdiv_rn(1.0f, (float)(n - 1));
    return (i < n / 2) ? __fmul_rn(step, (float)i) : __fsub_rn(1.0f, __fmul_rn(step, (float)(n - 1 - i)));
}

struct Empty {
    template <class T> __device__ __forceinline__ float operator[](T) const { return 0.f; }
};

// EXACT (only with VIEWS && HAS_BEND): view directions = normalised J . d, J = d(bent point)/d(point), d = the ray's unit
// direction (exact_nonrigid_viewdirs, run_nerf_helpers.py:358-385), by forward-mode differentiation through the bender
// and rigidity MLPs inside this kernel; otherwise the finite-difference directions of the default configuration.
template <class P, class A, bool HAS_BEND, bool VIEWS, int WAVES, bool EXACT = false>
__global__ void __launch_bounds__(WAVES * 64, (P::KH == 1) ? 1 : 2) net_kernel(const NetArgs a) {
    static_assert(!EXACT || (VIEWS && HAS_BEND), "exact directions differentiate the bender");
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
    float* mailbox = bias_lds + PL::NTILES * 32;      // [2][WAVES][4]: last bent point of each block (VIEWS only)

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
    // Block -> workgroup assignment.  Default: tiles of WAVES consecutive blocks, strided over the grid.  VIEWS: every
    // workgroup owns a contiguous range of whole rays, because a sample's direction needs the bent point of the sample
    // before it (run_nerf_helpers.py:339-351) and that neighbour must be produced by the same workgroup.
    long long blk_begin, blk_end, tile_stride;
    if constexpr (VIEWS) {
        const long long rays_per_wg = (a.n_rays + gridDim.x - 1) / gridDim.x;
        blk_begin = (long long)blockIdx.x * rays_per_wg * bpr;
        blk_end = blk_begin + rays_per_wg * bpr;
        if (blk_end > nblocks) blk_end = nblocks;
        if (blk_begin > nblocks) blk_begin = nblocks;
        tile_stride = WAVES;
    } else {
        blk_begin = (long long)blockIdx.x * WAVES;
        blk_end = nblocks;
        tile_stride = (long long)gridDim.x * WAVES;
    }

    // Fused compositing (variants without a fused bender, NetArgs::fuse_on): a WAVE owns whole rays -- ray grp * WAVES + wave
    // of group grp, one block per tile, bpr tiles per group, groups strided over the grid; the raw outputs are staged in the
    // wave's own LDS area and composited by the wave itself after the ray's last block (composite_ray).  See nrnerf_net_mb.h.
    bool fuse = false;
    int tg = 0;
    long long ngroups = 0, grp = blockIdx.x;
    f32x4* stage_w = nullptr;
    if constexpr (!HAS_BEND) {
        fuse = a.fuse_on != 0;
        ngroups = ((long long)a.n_rays + WAVES - 1) / WAVES;
        stage_w = (f32x4*)(mailbox + 2 * WAVES * 4) + (size_t)wave * bpr * 32;
        if (fuse) {          // the compositing arguments live in LDS, not in SGPRs held across the whole tile (nrnerf_net_mb.h)
            int* dst = (int*)((f32x4*)(mailbox + 2 * WAVES * 4) + (size_t)WAVES * bpr * 32);
            const int* src = (const int*)&a.fuse;
            for (int i = tid; i < (int)(sizeof(CompositeArgs) / 4); i += WAVES * 64) dst[i] = src[i];
            __syncthreads();
        }
    }

#ifdef NRN_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    int iter = 0;
    float cpre[8];              // fused compositing: direction and depths of this wave's ray, requested one tile ahead of their use
    for (long long tile0 = blk_begin; fuse ? (grp < ngroups) : (tile0 < blk_end); ++iter) {
        const unsigned long long t_pass = NRN_NOW();
        if constexpr (!HAS_BEND) {
            if (fuse && tg == bpr - 1) {
                const CompositeArgs& fa = *(const CompositeArgs*)((f32x4*)(mailbox + 2 * WAVES * 4) + (size_t)WAVES * bpr * 32);
                const long long rr = grp * WAVES + wave;
                composite_prefetch(fa, (int)(rr < a.n_rays ? rr : a.n_rays - 1), lane, (S + 63) >> 6, cpre);
            }
        }
        bool blk_ok;
        int ray, bir;               // bir: block within its ray
        if (fuse) {
            const long long rr = grp * WAVES + wave;
            blk_ok = rr < a.n_rays;
            ray = (int)(blk_ok ? rr : a.n_rays - 1);
            bir = tg;
        } else {
            const long long blk = tile0 + wave;
            blk_ok = blk < blk_end;
            const long long b = blk_ok ? blk : blk_end - 1;
            ray = (int)(b / bpr);
            bir = (int)(b % bpr);
        }
        const int sidx = bir * 32 + j;
        const bool ok = blk_ok && sidx < S;
        const int sc = sidx < S ? sidx : S - 1;

        const float* rp = a.rays + (size_t)ray * a.ray_stride;
        const float ox = rp[0], oy = rp[1], oz = rp[2], dx = rp[3], dy = rp[4], dz = rp[5];
        float z;
        if (a.z) {
            z = a.z[(size_t)ray * S + sc];
        } else {
            const float near = rp[6], far = rp[7];
            const float t = lin01(sc, S);
            if (a.lindisp)                                                               // train.py:850-852
                z = __fdiv_rn(1.0f, __fadd_rn(__fmul_rn(__fdiv_rn(1.0f, near), __fsub_rn(1.0f, t)),
                                              __fmul_rn(__fdiv_rn(1.0f, far), t)));
            else
                z = __fadd_rn(__fmul_rn(near, __fsub_rn(1.0f, t)), __fmul_rn(far, t));  // train.py:849
        }
        float p[3] = {__fadd_rn(ox, __fmul_rn(dx, z)), __fadd_rn(oy, __fmul_rn(dy, z)),
                      __fadd_rn(oz, __fmul_rn(dz, z))};                                  // train.py:871-873
        const size_t so = (size_t)ray * S + sc;     // flat sample index for per-sample outputs
        const bool writer = ok && h == 0;
        if constexpr (!HAS_BEND) {
            if (a.pts4) {       // points bent by the stand-alone bender kernel (nrnerf_bend.h)
                const f32x4 q = *(const f32x4*)(a.pts4 + so * 4);
                p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
            }
        }

        if (writer && a.ex.init_pts) {
            a.ex.init_pts[so * 3 + 0] = p[0]; a.ex.init_pts[so * 3 + 1] = p[1]; a.ex.init_pts[so * 3 + 2] = p[2];
        }

        NRN_TACC(1, t_pass);
        const unsigned long long t_bend = NRN_NOW();
        float rig_mask = 0.0f;
        float jd[3] = {0.f, 0.f, 0.f};             // EXACT: J . d
        if constexpr (HAS_BEND) {
            constexpr int NS_BIN = PL::NS_BIN, NS_RIN = PL::NS_RIN;
            constexpr int NB = PL::NT_BW * SP, NR = PL::NT_RW * SP;
            constexpr bool SPLIT = P::SPLIT;
            const float* lat = a.latents + (size_t)ray * a.lat_stride;
            auto binval = [&](auto idxc) -> float {
                constexpr int idx = decltype(idxc)::value;
                if constexpr (idx < 3) return p[idx];
                else if constexpr (idx < 8) return 0.0f;
                else if constexpr (idx - 8 < A::LAT) return lat[idx - 8];
                else return 0.0f;
            };
            Act<PE, NS_BIN, SPLIT> bin;
            static_for<0, NS_BIN>([&](auto sc_) {
                constexpr int s = decltype(sc_)::value;
                static_for<0, KH>([&](auto ec) {
                    constexpr int e = decltype(ec)::value;
                    const float v0 = binval(std::integral_constant<int, (2 * s) * KH + e>{});
                    const float v1 = binval(std::integral_constant<int, (2 * s + 1) * KH + e>{});
                    bin.template set<s, e>(h ? v1 : v0);
                });
            });
            // ---- offset MLP (run_nerf_helpers.py:525-541)
            Act<PE, NB, SPLIT> ba, bb;
            float off[3], doff[3] = {0.f, 0.f, 0.f}, dlogit = 0.0f;
            float dvec[3] = {0.f, 0.f, 0.f};
            if constexpr (EXACT) { dvec[0] = rp[8]; dvec[1] = rp[9]; dvec[2] = rp[10]; }      // unbent unit direction (train.py:380, 397)
            auto tangent_in = [&](auto& tin, auto nslab) {      // d(input)/dt: t = d in the xyz slots, 0 elsewhere
                static_for<0, decltype(nslab)::value>([&](auto sc_) {
                    constexpr int s = decltype(sc_)::value;
                    static_for<0, KH>([&](auto ec) {
                        constexpr int e = decltype(ec)::value;
                        constexpr int i0 = (2 * s) * KH + e, i1 = (2 * s + 1) * KH + e;
                        const float v0 = (i0 < 3) ? dvec[i0 < 3 ? i0 : 0] : 0.0f;
                        const float v1 = (i1 < 3) ? dvec[i1 < 3 ? i1 : 0] : 0.0f;
                        tin.template set<s, e>(h ? v1 : v0);
                    });
                });
            };
            if constexpr (EXACT) {
                Tan<PE, NS_BIN> tin;
                tangent_in(tin, std::integral_constant<int, NS_BIN>{});
                Tan<PE, NB> ta, tb;
                dense_b<PE, SPLIT, PL, PL::L_BEND0, NS_BIN>(st, bias_lane, bin, [&](auto tc, const f32x16& acc, const f32x16& tacc) {
                    pack_act<PE, decltype(tc)::value>(acc, ba);
                    pack_tan<PE, decltype(tc)::value>(acc, tacc, ta);
                }, tin);
                static_for<1, A::BD - 1>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr (i % 2 == 1) {
                        dense_b<PE, SPLIT, PL, PL::L_BEND0 + i, NB>(st, bias_lane, ba, [&](auto tc, const f32x16& acc, const f32x16& tacc) {
                            pack_act<PE, decltype(tc)::value>(acc, bb);
                            pack_tan<PE, decltype(tc)::value>(acc, tacc, tb);
                        }, ta);
                    } else {
                        dense_b<PE, SPLIT, PL, PL::L_BEND0 + i, NB>(st, bias_lane, bb, [&](auto tc, const f32x16& acc, const f32x16& tacc) {
                            pack_act<PE, decltype(tc)::value>(acc, ba);
                            pack_tan<PE, decltype(tc)::value>(acc, tacc, ta);
                        }, tb);
                    }
                });
                auto take_off_t = [&](auto, const f32x16& acc, const f32x16& tacc) {
                    off[0] = acc[0]; off[1] = acc[1]; off[2] = acc[2];
                    doff[0] = tacc[0]; doff[1] = tacc[1]; doff[2] = tacc[2];
                };
                if constexpr ((A::BD - 2) % 2 == 1) dense_b<PE, SPLIT, PL, PL::L_BEND0 + A::BD - 1, NB>(st, bias_lane, bb, take_off_t, tb);
                else dense_b<PE, SPLIT, PL, PL::L_BEND0 + A::BD - 1, NB>(st, bias_lane, ba, take_off_t, ta);
            } else {
            dense_b<PE, SPLIT, PL, PL::L_BEND0, NS_BIN>(st, bias_lane, bin, [&](auto tc, const f32x16& acc) {
                pack_act<PE, decltype(tc)::value>(acc, ba);
            });
            static_for<1, A::BD - 1>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i % 2 == 1) {
                    dense_b<PE, SPLIT, PL, PL::L_BEND0 + i, NB>(st, bias_lane, ba, [&](auto tc, const f32x16& acc) {
                        pack_act<PE, decltype(tc)::value>(acc, bb);
                    });
                } else {
                    dense_b<PE, SPLIT, PL, PL::L_BEND0 + i, NB>(st, bias_lane, bb, [&](auto tc, const f32x16& acc) {
                        pack_act<PE, decltype(tc)::value>(acc, ba);
                    });
                }
            });
            auto take_off = [&](auto, const f32x16& acc) { off[0] = acc[0]; off[1] = acc[1]; off[2] = acc[2]; };
            if constexpr ((A::BD - 2) % 2 == 1) dense_b<PE, SPLIT, PL, PL::L_BEND0 + A::BD - 1, NB>(st, bias_lane, bb, take_off);
            else dense_b<PE, SPLIT, PL, PL::L_BEND0 + A::BD - 1, NB>(st, bias_lane, ba, take_off);
            }
            // ---- rigidity MLP (run_nerf_helpers.py:545-561); input = xyz only
            Act<PE, NS_RIN, SPLIT> rin;
            auto rinval = [&](auto idxc) -> float {
                constexpr int idx = decltype(idxc)::value;
                if constexpr (idx < 3) return p[idx];
                else return 0.0f;
            };
            static_for<0, NS_RIN>([&](auto sc_) {
                constexpr int s = decltype(sc_)::value;
                static_for<0, KH>([&](auto ec) {
                    constexpr int e = decltype(ec)::value;
                    const float v0 = rinval(std::integral_constant<int, (2 * s) * KH + e>{});
                    const float v1 = rinval(std::integral_constant<int, (2 * s + 1) * KH + e>{});
                    rin.template set<s, e>(h ? v1 : v0);
                });
            });
            Act<PE, NR, SPLIT> ra, rb;
            float logit;
            if constexpr (EXACT) {
                Tan<PE, NS_RIN> trin;
                tangent_in(trin, std::integral_constant<int, NS_RIN>{});
                Tan<PE, NR> tra, trb;
                dense_b<PE, SPLIT, PL, PL::L_RIG0, NS_RIN>(st, bias_lane, rin, [&](auto tc, const f32x16& acc, const f32x16& tacc) {
                    pack_act<PE, decltype(tc)::value>(acc, ra);
                    pack_tan<PE, decltype(tc)::value>(acc, tacc, tra);
                }, trin);
                static_for<1, A::RD - 1>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr (i % 2 == 1) {
                        dense_b<PE, SPLIT, PL, PL::L_RIG0 + i, NR>(st, bias_lane, ra, [&](auto tc, const f32x16& acc, const f32x16& tacc) {
                            pack_act<PE, decltype(tc)::value>(acc, rb);
                            pack_tan<PE, decltype(tc)::value>(acc, tacc, trb);
                        }, tra);
                    } else {
                        dense_b<PE, SPLIT, PL, PL::L_RIG0 + i, NR>(st, bias_lane, rb, [&](auto tc, const f32x16& acc, const f32x16& tacc) {
                            pack_act<PE, decltype(tc)::value>(acc, ra);
                            pack_tan<PE, decltype(tc)::value>(acc, tacc, tra);
                        }, trb);
                    }
                });
                auto take_logit_t = [&](auto, const f32x16& acc, const f32x16& tacc) { logit = acc[0]; dlogit = tacc[0]; };
                if constexpr ((A::RD - 2) % 2 == 1) dense_b<PE, SPLIT, PL, PL::L_RIG0 + A::RD - 1, NR>(st, bias_lane, rb, take_logit_t, trb);
                else dense_b<PE, SPLIT, PL, PL::L_RIG0 + A::RD - 1, NR>(st, bias_lane, ra, take_logit_t, tra);
            } else {
            dense_b<PE, SPLIT, PL, PL::L_RIG0, NS_RIN>(st, bias_lane, rin, [&](auto tc, const f32x16& acc) {
                pack_act<PE, decltype(tc)::value>(acc, ra);
            });
            static_for<1, A::RD - 1>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i % 2 == 1) {
                    dense_b<PE, SPLIT, PL, PL::L_RIG0 + i, NR>(st, bias_lane, ra, [&](auto tc, const f32x16& acc) {
                        pack_act<PE, decltype(tc)::value>(acc, rb);
                    });
                } else {
                    dense_b<PE, SPLIT, PL, PL::L_RIG0 + i, NR>(st, bias_lane, rb, [&](auto tc, const f32x16& acc) {
                        pack_act<PE, decltype(tc)::value>(acc, ra);
                    });
                }
            });
            auto take_logit = [&](auto, const f32x16& acc) { logit = acc[0]; };
            if constexpr ((A::RD - 2) % 2 == 1) dense_b<PE, SPLIT, PL, PL::L_RIG0 + A::RD - 1, NR>(st, bias_lane, rb, take_logit);
            else dense_b<PE, SPLIT, PL, PL::L_RIG0 + A::RD - 1, NR>(st, bias_lane, ra, take_logit);
            }

            const float th = tanhf(logit);
            rig_mask = (th + 1.0f) / 2.0f;                                            // rnh:559-561
            float dmask = EXACT ? (1.0f - th * th) * 0.5f * dlogit : 0.0f;           // d mask / dt
            if (a.knobs.has_cutoff && rig_mask <= a.knobs.cutoff) { rig_mask = 0.0f; dmask = 0.0f; }   // rnh:563-564 (assignment: no gradient)
            if constexpr (EXACT) {
                // bent = p + s * mask * off  =>  J . d = d + s * (dmask * off + mask * doff)        (rnh:567-570)
                const float sc_ = a.knobs.has_scaling ? a.knobs.scaling : 1.0f;
#pragma unroll
                for (int c = 0; c < 3; ++c) jd[c] = dvec[c] + sc_ * (dmask * off[c] + rig_mask * doff[c]);
            }
            float mo[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                mo[c] = __fmul_rn(rig_mask, off[c]);                                  // rnh:567
                if (a.knobs.has_scaling) mo[c] = __fmul_rn(mo[c], a.knobs.scaling);   // rnh:568-569
            }
            if (writer) {
                if (a.ex.unmasked) { a.ex.unmasked[so * 3 + 0] = off[0]; a.ex.unmasked[so * 3 + 1] = off[1]; a.ex.unmasked[so * 3 + 2] = off[2]; }
                if (a.ex.masked) { a.ex.masked[so * 3 + 0] = mo[0]; a.ex.masked[so * 3 + 1] = mo[1]; a.ex.masked[so * 3 + 2] = mo[2]; }
                if (a.ex.rigidity) a.ex.rigidity[so] = rig_mask;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = __fadd_rn(p[c], mo[c]);                // rnh:570
        }
        NRN_TACC(2, t_bend);
        const unsigned long long t_mid = NRN_NOW();
        if (writer && a.ex.in_pts) {
            a.ex.in_pts[so * 3 + 0] = p[0]; a.ex.in_pts[so * 3 + 1] = p[1]; a.ex.in_pts[so * 3 + 2] = p[2];
        }
        if (writer && a.bent4) *(f32x4*)(a.bent4 + so * 4) = f32x4{p[0], p[1], p[2], rig_mask};

        // ---- view direction of the sample (VIEWS): finite difference of the bent points along the ray, or the ray's own
        //      unit direction without a bender (run_nerf_helpers.py:288-290, 339-351; train.py:73-76)
        constexpr int NS_ENCV = PL::NS_ENCV;
        efrag encv[VIEWS ? NS_ENCV : 1];
        if constexpr (VIEWS) {
            float dirv[3];
            if constexpr (EXACT) {
                const float nrm = sqrtf(jd[0] * jd[0] + jd[1] * jd[1] + jd[2] * jd[2]);
#pragma unroll
                for (int c = 0; c < 3; ++c) dirv[c] = __fadd_rn(__fdiv_rn(jd[c], nrm), 0.000001f);   // rnh:374-378: eps outside the division
            } else if constexpr (HAS_BEND) {
                if (j == 31 && h == 0) {
                    float* mb = mailbox + ((iter & 1) * WAVES + wave) * 4;
                    mb[0] = p[0]; mb[1] = p[1]; mb[2] = p[2];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float prev[3], next[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { prev[c] = __shfl_up(p[c], 1); next[c] = __shfl_down(p[c], 1); }
                const bool first_in_ray = (sidx == 0);
                // the neighbour of lane 0 lives in the previous block: previous wave of this tile, or the last wave of the
                // previous tile of this workgroup (double-buffered by tile parity).  Read after the next ring barrier.
                __builtin_amdgcn_s_barrier();
                if (j == 0 && !first_in_ray) {
                    const float* mb = (wave > 0) ? mailbox + ((iter & 1) * WAVES + wave - 1) * 4
                                                 : mailbox + (((iter + 1) & 1) * WAVES + WAVES - 1) * 4;
                    prev[0] = mb[0]; prev[1] = mb[1]; prev[2] = mb[2];
                }
                float dd[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) dd[c] = first_in_ray ? __fsub_rn(next[c], p[c]) : __fsub_rn(p[c], prev[c]);
                const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dd[0], dd[0]), __fmul_rn(dd[1], dd[1])), __fmul_rn(dd[2], dd[2])));
#pragma unroll
                for (int c = 0; c < 3; ++c) dirv[c] = __fdiv_rn(dd[c], __fadd_rn(nrm, 0.000001f));
            } else if (a.pts4) {
                // split-bender path: the points are the bent points of nrnerf_bend.h, so the direction is their finite
                // difference along the ray exactly as in the fused kernel, the neighbour read from the same array
                const f32x4 nb = *(const f32x4*)(a.pts4 + (sidx == 0 ? so + 1 : so - 1) * 4);
                float dd[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) dd[c] = (sidx == 0) ? __fsub_rn(nb[c], p[c]) : __fsub_rn(p[c], nb[c]);
                const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dd[0], dd[0]), __fmul_rn(dd[1], dd[1])), __fmul_rn(dd[2], dd[2])));
#pragma unroll
                for (int c = 0; c < 3; ++c) dirv[c] = __fdiv_rn(dd[c], __fadd_rn(nrm, 0.000001f));
            } else {
                dirv[0] = rp[8]; dirv[1] = rp[9]; dirv[2] = rp[10];
            }
            constexpr int F0V = enc_F0(A::LV);
            constexpr int NSLOTV = NS_ENCV * KH;
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
            static_for<0, NS_ENCV>([&](auto sc_) {
                constexpr int s = decltype(sc_)::value;
                static_for<0, KH>([&](auto ec) {
                    constexpr int e = decltype(ec)::value;
                    PE::template set<e>(encv[s], evv[s * KH + e]);
                });
            });
        }

        // ---- positional encoding of the (bent) point, directly in B-operand order
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
                enc_sincos<KH == 1>(p[c], prev_[c], fscale * (float)(1 << fl), &sv, &cv);   // power-of-two scaling is exact
                ev[2 + 2 * (3 * fl + c)] = sv;
                ev[2 + 2 * (3 * fl + c) + 1] = cv;
            });
        });
        efrag enc[NS_ENC];
        static_for<0, PL::NS_ENC_XYZ>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            static_for<0, KH>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                PE::template set<e>(enc[s], ev[s * KH + e]);
            });
        });
        if constexpr (A::TCB) {      // time-conditioned baseline: the ray's latent code follows the encoding (rnh:273-274)
            const float* lat = a.latents + (size_t)ray * a.lat_stride;
            static_for<PL::NS_ENC_XYZ, NS_ENC>([&](auto sc_) {
                constexpr int s = decltype(sc_)::value;
                static_for<0, KH>([&](auto ec) {
                    constexpr int e = decltype(ec)::value;
                    constexpr int i0 = (2 * (s - PL::NS_ENC_XYZ)) * KH + e, i1 = i0 + KH;
                    const float v0 = (i0 < A::LAT) ? lat[i0 < A::LAT ? i0 : 0] : 0.0f;
                    const float v1 = (i1 < A::LAT) ? lat[i1 < A::LAT ? i1 : 0] : 0.0f;
                    PE::template set<e>(enc[s], h ? v1 : v0);
                });
            });
        }

        NRN_TACC(3, t_mid);
        const unsigned long long t_trunk = NRN_NOW();
        // ---- trunk (run_nerf_helpers.py:272-282) and head (:306)
        constexpr int NH = NT_W * SP;
        frag ha[NH], hb[NH];
        Empty none;
        dense<PE, P, PL, PL::L_TRUNK0, NS_ENC, 0>(st, bias_lane, enc, none, [&](auto tc, const f32x16& acc) {
            pack_tile<P, true, decltype(tc)::value>(acc, ha);
        });
        static_for<1, A::D>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr bool skip = (i - 1 == A::SKIP);
            if constexpr (i % 2 == 1) {
                if constexpr (skip)
                    dense<PE, P, PL, PL::L_TRUNK0 + i, NS_ENC, NH>(st, bias_lane, enc, ha, [&](auto tc, const f32x16& acc) {
                        pack_tile<P, true, decltype(tc)::value>(acc, hb); });
                else
                    dense<P, P, PL, PL::L_TRUNK0 + i, NH, 0>(st, bias_lane, ha, none, [&](auto tc, const f32x16& acc) {
                        pack_tile<P, true, decltype(tc)::value>(acc, hb); });
            } else {
                if constexpr (skip)
                    dense<PE, P, PL, PL::L_TRUNK0 + i, NS_ENC, NH>(st, bias_lane, enc, hb, [&](auto tc, const f32x16& acc) {
                        pack_tile<P, true, decltype(tc)::value>(acc, ha); });
                else
                    dense<P, P, PL, PL::L_TRUNK0 + i, NH, 0>(st, bias_lane, hb, none, [&](auto tc, const f32x16& acc) {
                        pack_tile<P, true, decltype(tc)::value>(acc, ha); });
            }
        });
        float raw[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        constexpr bool LAST_IN_B = ((A::D - 1) % 2 == 1);       // buffer holding the trunk output
        if constexpr (!VIEWS) {
            auto take_raw = [&](auto, const f32x16& acc) {
                raw[0] = acc[0]; raw[1] = acc[1]; raw[2] = acc[2]; raw[3] = acc[3]; raw[4] = acc[4];
            };
            if constexpr (LAST_IN_B) dense<P, P, PL, PL::L_HEAD, NH, 0>(st, bias_lane, hb, none, take_raw);
            else dense<P, P, PL, PL::L_HEAD, NH, 0>(st, bias_lane, ha, none, take_raw);
        } else {
            // view-dependent head (run_nerf_helpers.py:284-304): alpha from the trunk output, then
            // relu(views_linear([feature_linear(trunk output), enc(dir)])) -- one layer, feature_linear folded into its weights by
            // the packer -- and rgb_linear; output = [rgb, alpha]
            auto head = [&](auto& hx) {
                dense<P, P, PL, PL::L_ALPHA, NH, 0>(st, bias_lane, hx, none, [&](auto, const f32x16& acc) { raw[3] = acc[0]; });
                constexpr int NV = (NT_W / 2) * SP;
                frag hv[NV];
                dense<PE, P, PL, PL::L_VIEWS, NS_ENCV, NH>(st, bias_lane, encv, hx, [&](auto tc, const f32x16& acc) {
                    pack_tile<P, true, decltype(tc)::value>(acc, hv); });
                dense<P, P, PL, PL::L_RGB, NV, 0>(st, bias_lane, hv, none, [&](auto, const f32x16& acc) {
                    raw[0] = acc[0]; raw[1] = acc[1]; raw[2] = acc[2]; });
            };
            if constexpr (LAST_IN_B) head(hb); else head(ha);
        }

        NRN_TACC(4, t_trunk);
        const unsigned long long t_out = NRN_NOW();
        if (HAS_BEND && a.knobs.detailed && a.knobs.has_removal && rig_mask >= a.knobs.removal)
            raw[3] = raw[3] * 0.0f;                                                  // rnh:308-311
        if (writer) {
            if (!fuse) *(f32x4*)(a.raw4 + so * 4) = f32x4{raw[0], raw[1], raw[2], raw[3]};
            if (a.raw_out) {
                float* ro = a.raw_out + so * a.raw_ch;
                ro[0] = raw[0]; ro[1] = raw[1]; ro[2] = raw[2]; ro[3] = raw[3];
                if (a.raw_ch > 4) ro[4] = raw[4];
            }
        }
        if constexpr (!HAS_BEND) {
            if (fuse && h == 0) stage_w[tg * 32 + j] = f32x4{raw[0], raw[1], raw[2], raw[3]};
        }
        // padding units (keep the ring phase identical every pass and prime the next pass' first units)
        static_for<PL::NUNITS, PL::NUP>([&](auto uc) { st.template advance<decltype(uc)::value>(); });
        if constexpr (!HAS_BEND) {
            if (fuse) {
                if (++tg == bpr) {       // the ray's last block: composite it from this wave's LDS stage (train.py:943-950)
                    // the surface reduction reads bent4 rows this wave's OTHER lanes have just written (only when this kernel
                    // writes that array itself: a model without bender): make the stores visible first
                    const CompositeArgs& fa = *(const CompositeArgs*)((f32x4*)(mailbox + 2 * WAVES * 4) + (size_t)WAVES * bpr * 32);
                    if (a.bent4 && fa.bent4) __threadfence();
                    auto raw_at = [&](int ic) { return stage_w[ic]; };
                    switch ((S + 63) >> 6) {
                        case 1: { float cz[2], cw[1]; composite_ray<1>(fa, ray, blk_ok, lane, raw_at, cz, cw, cpre); break; }
                        case 2: { float cz[3], cw[2]; composite_ray<2>(fa, ray, blk_ok, lane, raw_at, cz, cw, cpre); break; }
                        case 3: { float cz[4], cw[3]; composite_ray<3>(fa, ray, blk_ok, lane, raw_at, cz, cw, cpre); break; }
                        default: { float cz[5], cw[4]; composite_ray<4>(fa, ray, blk_ok, lane, raw_at, cz, cw, cpre); break; }
                    }
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

// ------------------------------------------------------------------------------------------
// launch (one explicit instantiation per translation unit, see nrnerf_net_inst.hip)
// ------------------------------------------------------------------------------------------
template <class P, class A, bool HAS_BEND, bool VIEWS, int WAVES, bool EXACT = false>
static hipError_t launch_one(const NetArgs& a, int num_cus, hipStream_t stream) {
    using PL = Plan<P, A, HAS_BEND, VIEWS>;
    const int bpr_l = (a.S + 31) / 32;
    size_t lds = (size_t)RING * P::UNIT_BYTES + (size_t)PL::NTILES * 32 * sizeof(float) + 2 * WAVES * 4 * sizeof(float);
    if (a.fuse_on) {     // fused compositing: the waves' raw stages (one ray each)
        if (HAS_BEND || a.S > 256 || a.fuse.n_importance != 0 || a.fuse.S != a.S) return hipErrorInvalidValue;
        lds += (size_t)WAVES * bpr_l * 32 * 16 + 256;
    }
    auto kern = net_kernel<P, A, HAS_BEND, VIEWS, WAVES, EXACT>;
    // function attributes are per device: one flag per ordinal (idempotent; racing threads set the same value)
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        const size_t lds_max = (size_t)RING * P::UNIT_BYTES + (size_t)PL::NTILES * 32 * sizeof(float) + 2 * WAVES * 4 * sizeof(float) +
                               (HAS_BEND ? 0 : (size_t)WAVES * 8 * 32 * 16 + 256);      // + the fused stages at 256 samples per ray, the compositing arguments
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int bpr = (a.S + 31) / 32;
    const long long nblocks = (long long)a.n_rays * bpr;
    const long long ntiles = (nblocks + WAVES - 1) / WAVES;
    if (ntiles <= 0) return hipSuccess;
    // persistent grid: 16-bit builds keep 8 waves per CU resident (one 8-wave or two 4-wave workgroups, each with
    // its own LDS ring); the fp32 build one 4-wave workgroup (512 registers per wave)
    const long long resident = (long long)num_cus * ((P::KH == 1) ? 1 : 8 / WAVES);
    long long want = ntiles;
    if (a.fuse_on) {        // groups of WAVES whole rays
        want = ((long long)a.n_rays + WAVES - 1) / WAVES;
    } else if (VIEWS) {    // contiguous whole-ray ranges: no more workgroups than ray groups that fill a tile
        const long long rays_per_tile = (WAVES + bpr - 1) / bpr;
        want = ((long long)a.n_rays + rays_per_tile - 1) / rays_per_tile;
    }
    const int grid = (int)(want < resident ? (want > 0 ? want : 1) : resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace nrn
