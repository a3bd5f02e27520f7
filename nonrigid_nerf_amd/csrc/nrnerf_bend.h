// nrnerf_bend.h -- the ray bender as a kernel of its own (reference ray_bending.forward, run_nerf_helpers.py:507-577).
//
// Why it exists.  The bender is shared by the coarse and the fine network (run_nerf_helpers.py:213-215) and the S coarse
// depths are a subset of the S + I merged depths of the fine pass (train.py:920), so the fine pass only has to bend the
// I importance samples: the bent points of the coarse samples are carried over from the coarse launch.  In the fused
// kernel (nrnerf_net_impl.h / nrnerf_net_mb.h) the bender is also the worst-utilised phase -- 3 % of the algorithmic
// flops in 17-19 % of a pass, VALU-bound by the hi/lo packing of its fp32-equivalent split product -- so taking it out
// of the fine pass removes a third of that phase and lets the rest run in a kernel shaped for it:
//   * the whole bender + rigidity weight stream (78 fragments = 78 KiB in the 16-bit modes) is RESIDENT in LDS for the
//     lifetime of the workgroup: no ring, no DMA and no barrier inside the loop, waves are fully independent;
//   * eight waves per CU, two per SIMD, so one wave's packing VALU overlaps the other's MFMAs (sixteen / four in the
//     "bf16" mode, whose single-product bender needs half the LDS and registers).
// Arithmetic (MFMA order, split product, tanh, masking) is the fused kernels' own -- dense_b / Act / pack_act of
// nrnerf_net_impl.h -- so a bent point computed here equals the fused kernel's bit for bit.
#pragma once
#include "nrnerf_net_impl.h"

namespace nrn {

// weight fragments resident in LDS; same read interface as WRing (frag / ready), nothing to advance
template <class P, int NFRAGS>
struct WResident {
    static constexpr bool ASM_FRAGS = (P::FRAG_BYTES == 1024);
    // fp32 fragments (one dword per lane): explicit ds_read_b32 + counted waits as well -- with plain loads hipcc hoists
    // dozens of the loop-invariant fragment reads out of the persistent loop and spills them (80-150 registers of scratch in
    // the fp32 bender kernels)
    static constexpr bool ASM_FRAGS32 = (P::FRAG_BYTES == 256);
    static constexpr int BYTES = NFRAGS * P::FRAG_BYTES;
    char* base;
    int lane_off;
    unsigned lane_addr, lane_addr_hi;      // LDS byte address of this lane's part of fragment 0 / of the fragment at 64 KiB

    __device__ __forceinline__ void init(const void* stream, char* lds, int tid, int nthreads, int lane) {
        const u32x4* src = (const u32x4*)stream;
        u32x4* dst = (u32x4*)lds;
        for (int i = tid; i < BYTES / 16; i += nthreads) dst[i] = src[i];
        base = lds;
        lane_off = lane * (P::FRAG_BYTES / 64);
        lane_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (unsigned)lane_off;
        lane_addr_hi = lane_addr + 65536u;
        asm volatile("" : "+v"(lane_addr), "+v"(lane_addr_hi));      // one VGPR each, every fragment an immediate offset
    }
    template <class PX, int GF>
    __device__ __forceinline__ typename PX::frag frag() {
        static_assert(PX::FRAG_BYTES == P::FRAG_BYTES && GF < NFRAGS, "fragment outside the resident stream");
        constexpr int OFF = GF * P::FRAG_BYTES;
        if constexpr (ASM_FRAGS) {
            u32x4 v;
            if constexpr (OFF + 16 <= 65536) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lane_addr), "n"(OFF));
            else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lane_addr_hi), "n"(OFF - 65536));
            return __builtin_bit_cast(typename PX::frag, v);
        } else if constexpr (ASM_FRAGS32) {
            unsigned v;
            if constexpr (OFF + 4 <= 65536) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(lane_addr), "n"(OFF));
            else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(lane_addr_hi), "n"(OFF - 65536));
            return __builtin_bit_cast(typename PX::frag, v);
        } else {
            return *(const typename PX::frag*)(base + OFF + lane_off);
        }
    }
    template <int N, class F>
    __device__ __forceinline__ void ready(F& f) {
        if constexpr (ASM_FRAGS) {
            static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
            u32x4 v = __builtin_bit_cast(u32x4, f);
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
            f = __builtin_bit_cast(F, v);
        } else if constexpr (ASM_FRAGS32) {
            static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
            unsigned v = __builtin_bit_cast(unsigned, f);
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
            f = __builtin_bit_cast(F, v);
        }
    }
};

template <class P, class A, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, (P::KH == 1) ? 4 : 2) bend_kernel(const BendArgs a) {
    static_assert(P::KH == 1 ? WAVES == 4 : WAVES == 8, "fp32 mode: workgroups of four waves, four of them per CU; 16-bit modes: eight waves");
    using PL = Plan<P, A, true, false, false>;                  // bender + rigidity layers only
    using PE = std::conditional_t<P::KH == 1, PolF32, PolF16>;  // as in the fused kernels (nrnerf_plan.h frag_is_f16)
    constexpr int KH = P::KH, SP = P::SP;
    constexpr int NS_BIN = PL::NS_BIN, NS_RIN = PL::NS_RIN;
    constexpr int NB = PL::NT_BW * SP, NR = PL::NT_RW * SP;
    constexpr bool SPLIT = P::SPLIT;
    using ST = WResident<P, PL::NFRAGS>;

    extern __shared__ __attribute__((aligned(16))) char smem[];     // resident weights | bias table
    float* bias_lds = (float*)(smem + ST::BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int j = lane & 31;

    ST st;
    st.init(a.wstream, smem, tid, WAVES * 64, lane);
    for (int i = tid; i < PL::NTILES * 32; i += WAVES * 64) bias_lds[i] = a.bias[i];
    __syncthreads();
    const BiasPtr bias_lane = bias_lane_ptr(bias_lds, h);

    const int n = a.n_per_ray;
    const int bpr = (n + 31) >> 5;                 // 32-sample blocks per ray
    const int nblocks = a.n_rays * bpr;            // n_rays <= 2^20 per launch, bpr <= 8
    // Inputs of a block.  A block is 32 samples of ONE ray, so the ray record and its latent code are wave-uniform: read
    // through the constant address space they become scalar loads (38 SGPRs instead of 38 per-lane VMEM loads); the
    // per-sample depth and output row are the only vector loads.  The next block's inputs are requested before the
    // current block's MLPs run: with two waves per SIMD nothing else would hide that latency.
    typedef const __attribute__((address_space(4))) float* cfloat_p;
    struct In {
        float o[3], d[3], lat[A::LAT];
        float z;
        int row, ray;
        bool ok;
    };
    auto load_inputs = [&](int blk, In& in) {
        const int ray = __builtin_amdgcn_readfirstlane(blk / bpr);
        const int k = (blk - ray * bpr) * 32 + j;
        in.ok = k < n;
        const int kc = in.ok ? k : n - 1;
        in.ray = ray;
        cfloat_p rp = (cfloat_p)(a.rays + (size_t)ray * a.ray_stride);
        cfloat_p lp = (cfloat_p)(a.latents + (size_t)ray * a.lat_stride);
#pragma unroll
        for (int c = 0; c < 3; ++c) { in.o[c] = rp[c]; in.d[c] = rp[3 + c]; }
#pragma unroll
        for (int c = 0; c < A::LAT; ++c) in.lat[c] = lp[c];
        if (a.z) {
            in.z = a.z[(size_t)ray * n + kc];
        } else {                                   // coarse depths (train.py:847-852), as in the fused kernels
            const float near = rp[6], far = rp[7];
            const float t = lin01(kc, n);
            if (a.lindisp)
                in.z = __fdiv_rn(1.0f, __fadd_rn(__fmul_rn(__fdiv_rn(1.0f, near), __fsub_rn(1.0f, t)), __fmul_rn(__fdiv_rn(1.0f, far), t)));
            else
                in.z = __fadd_rn(__fmul_rn(near, __fsub_rn(1.0f, t)), __fmul_rn(far, t));
        }
        in.row = a.rank ? (int)a.rank[(size_t)ray * n + kc] : kc;
    };
    const int blk0 = (int)blockIdx.x * WAVES + wave, blk_step = (int)gridDim.x * WAVES;
    In cur;
    if (blk0 < nblocks) load_inputs(blk0, cur);
    // no barrier below: every wave strides over the blocks on its own
    for (int blk = blk0; blk < nblocks; blk += blk_step) {
        const bool ok = cur.ok;
        const int out_ray = cur.ray, out_row = cur.row;
        float p[3] = {__fadd_rn(cur.o[0], __fmul_rn(cur.d[0], cur.z)), __fadd_rn(cur.o[1], __fmul_rn(cur.d[1], cur.z)),
                      __fadd_rn(cur.o[2], __fmul_rn(cur.d[2], cur.z))};                  // train.py:921-923
        const float* lat = cur.lat;
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
        // `cur` is consumed (point and first-layer operand built): request the next block's inputs into the same registers
        if (blk + blk_step < nblocks) load_inputs(blk + blk_step, cur);
        // ---- offset MLP (run_nerf_helpers.py:525-541)
        Act<PE, NB, SPLIT> ba, bb;
        float off[3];
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

        float rig_mask = (tanhf(logit) + 1.0f) / 2.0f;                                       // rnh:559-561
        if (a.knobs.has_cutoff && rig_mask <= a.knobs.cutoff) rig_mask = 0.0f;               // rnh:563-564
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float mo = __fmul_rn(rig_mask, off[c]);                                          // rnh:567
            if (a.knobs.has_scaling) mo = __fmul_rn(mo, a.knobs.scaling);                    // rnh:568-569
            p[c] = __fadd_rn(p[c], mo);                                                      // rnh:570
        }
        if (ok && h == 0)
            *(f32x4*)(a.bent4 + ((size_t)out_ray * a.out_stride + out_row) * 4) = f32x4{p[0], p[1], p[2], rig_mask};
    }
}

template <class P, class A, int WAVES>
static hipError_t launch_bend_one(const BendArgs& a, int num_cus, hipStream_t stream) {
    using PL = Plan<P, A, true, false, false>;
    const size_t lds = (size_t)PL::NFRAGS * P::FRAG_BYTES + (size_t)PL::NTILES * 32 * sizeof(float);
    auto kern = bend_kernel<P, A, WAVES>;
    static bool attr_set[64] = {};       // function attributes are per device (idempotent; racing threads set the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const long long nblocks = (long long)a.n_rays * ((a.n_per_ray + 31) / 32);
    if (nblocks >= (1ll << 31)) return hipErrorInvalidValue;
    const long long want = (nblocks + WAVES - 1) / WAVES;
    if (want <= 0) return hipSuccess;
    // persistent: one workgroup per CU; two for the single-product 16-bit variant (39 KiB of resident weights and < 128
    // VGPRs per lane: sixteen waves per CU fit, four per SIMD to hide the VALU-heavy packing behind each other's MFMAs)
    // fp32: 20 KiB of resident weights and ~100 VGPRs per lane (explicit ds_read_b32 fragment reads): four workgroups per CU
    const long long per_cu = (P::KH == 1) ? 4 : ((!P::SPLIT) ? 2 : 1);
    const long long resident = per_cu * num_cus;
    const int grid = (int)(want < resident ? want : resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace nrn
