// nrnerf_bend_x16.h -- the stand-alone ray bender (reference ray_bending.forward, run_nerf_helpers.py:507-577) on
// v_mfma_f32_16x16x32_f16: the single-product bender of "bf16" mode in the dataflow of the 16x16x32 trunk kernel.
//
// Why a second bender kernel.  The 32x32x16 bender of nrnerf_bend.h evaluates ONE 32-sample block per wave: a serial chain of ~10
// small layers with two accumulator chains each, 39 MFMAs and ~650 other instructions per block.  Measured (round 5): 2 800 cycles
// per block and SIMD with four waves per SIMD, where its MFMAs alone are 1 250 -- the kernel waits on its own dependency chains, not
// on any pipe.  Here a wave owns NB blocks of 16 samples that share every weight fragment (NB independent chains per tile), the
// activations go from layer to layer in registers (two consecutive D tiles of a lane ARE the next B operand, as in nrnerf_net_x16.h:
// epilogue = 4 v_cvt_pk + 4 v_pk_max per block and tile pair instead of the 32x32x16 kernel's pack_act), and the layers are
// dense_x16 itself.  Weights resident in LDS (39 / 55 KiB), no ring, no barrier in the loop, waves fully independent.
// Arithmetic: the same f16 x f16 products accumulated in fp32 in another order (k-slices of 32 instead of 16), so a bent point is
// another fp32 rounding of the same sum: compared to the 32x32x16 bender in tests/test_gpu_parity.py (bender-alone test).
#pragma once
#include "nrnerf_bend.h"
#include "nrnerf_bend_x16_plan.h"
#include "nrnerf_net_x16.h"

namespace nrn {

#ifndef NRN_BX16_NB
#define NRN_BX16_NB 2          // 16-sample blocks per wave and iteration
#endif
#ifndef NRN_BX16_WAVES
#define NRN_BX16_WAVES 8       // waves per workgroup
#endif
#ifndef NRN_BX16_OCC
#define NRN_BX16_OCC 2         // workgroups per CU the register budget is sized for (8 waves x 2 = four waves per SIMD)
#endif
#ifndef NRN_BX16_PF
#define NRN_BX16_PF 4
#endif
#ifndef NRN_BX16_CHUNK
#define NRN_BX16_CHUNK 2       // groups of NB blocks a wave takes from its counter at a time
#endif

// PERRAY: a latent code per ray (lat_stride != 0); false: one code for the whole launch (a frame render), read once per wave
template <class A, int WAVES, int NB, bool PERRAY>
__global__ void __launch_bounds__(WAVES * 64, NRN_BX16_OCC) bend_kernel_x16(const BendArgs a) {
    using P = PolF16;
    using PL = PlanX16Bend<A>;
    using frag = typename P::frag;
    using ST = WResident<P, PL::NFRAGS>;
    constexpr int NS_B = PL::NS_B, NS_R = PL::NS_R, PF = NRN_BX16_PF;

#ifdef NRN_TIMING
    const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];     // resident weights | bias table [tile][16 rows]
    float* bias_lds = (float*)(smem + ST::BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, n = lane & 15;
    ST st;
    st.init(a.wstream, smem, tid, WAVES * 64, lane);
    for (int i = tid; i < PL::NTILES * 16; i += WAVES * 64) bias_lds[i] = a.bias[i];
    __syncthreads();
    const __attribute__((address_space(3))) f32x4* bias_lane = (const __attribute__((address_space(3))) f32x4*)(bias_lds + 4 * g);
    asm volatile("" : "+v"(bias_lane));

    const int npr = a.n_per_ray;
    const int bpr = (npr + 15) >> 4;               // 16-sample blocks per ray
    const int nblocks = a.n_rays * bpr;            // (< 2^31: the launcher checks)
    // 16-byte loads of the ray record (rows of 8 floats) and of the latent code when the caller's arrays allow it (wave-uniform)
    const bool ray_vec = (a.ray_stride % 4 == 0) && (((size_t)a.rays & 15) == 0);
    const bool lat_vec = (a.lat_stride % 4 == 0) && (((size_t)a.latents & 15) == 0);
    const int blk0 = ((int)blockIdx.x * WAVES + wave) * NB, blk_step = (int)gridDim.x * WAVES * NB;
    // Inputs of an iteration: requested one iteration ahead (right after the current iteration's operands are built), so that the loads'
    // latency runs under the MLPs.  One latent code for every ray (lat_stride == 0: a frame render, train.py:464-466) is read once.
    struct In { float o[3], d[3], z, near, far; int idx, kc; float l8[PERRAY ? 8 : 1]; };
    float lconst[8];
    if constexpr (!PERRAY) {
        const float* lp = a.latents + 8 * g;
#pragma unroll
        for (int e = 0; e < 8; ++e) lconst[e] = lp[e];
    }
    auto load_inputs = [&](int blk, In (&in)[NB]) {
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            const int bi = blk + b;
            const bool blk_ok = bi < nblocks;
            const int bcl = blk_ok ? bi : nblocks - 1;
            const int ray = bcl / bpr;
            const int k = (bcl - ray * bpr) * 16 + n;
            const bool ok = blk_ok && k < npr;
            const int kc = k < npr ? k : npr - 1;
            in[b].kc = kc;
            const float* rp = a.rays + (size_t)ray * a.ray_stride;
            if (ray_vec) {
                const f32x4 r0 = *(const f32x4*)rp, r1 = *(const f32x4*)(rp + 4);
                in[b].o[0] = r0[0]; in[b].o[1] = r0[1]; in[b].o[2] = r0[2]; in[b].d[0] = r0[3]; in[b].d[1] = r1[0]; in[b].d[2] = r1[1];
                in[b].near = r1[2]; in[b].far = r1[3];
            } else {
                in[b].o[0] = rp[0]; in[b].o[1] = rp[1]; in[b].o[2] = rp[2]; in[b].d[0] = rp[3]; in[b].d[1] = rp[4]; in[b].d[2] = rp[5];
                in[b].near = rp[6]; in[b].far = rp[7];
            }
            in[b].z = a.z ? a.z[(size_t)ray * npr + kc] : 0.0f;
            const int row = a.rank ? (int)a.rank[(size_t)ray * npr + kc] : kc;
            in[b].idx = ok ? ray * a.out_stride + row : -1;           // (n_rays * out_stride < 2^31: the launcher checks)
            if constexpr (PERRAY) {
                const float* lp = a.latents + (size_t)ray * a.lat_stride + 8 * g;
                if (lat_vec) {
                    const f32x4 l0 = *(const f32x4*)lp, l1 = *(const f32x4*)(lp + 4);
                    in[b].l8[0] = l0[0]; in[b].l8[1] = l0[1]; in[b].l8[2] = l0[2]; in[b].l8[3] = l0[3];
                    in[b].l8[4] = l1[0]; in[b].l8[5] = l1[1]; in[b].l8[6] = l1[2]; in[b].l8[7] = l1[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) in[b].l8[e] = lp[e];
                }
            }
        });
    };
    // Which groups of NB blocks a wave evaluates: group blk0 / NB, then every (grid x WAVES)-th one -- or, with BendArgs::work_counter, chunks
    // of NRN_BX16_CHUNK consecutive groups taken from the counter as the wave gets to them.  Either way the NEXT group is known one
    // iteration ahead (its inputs are requested then): the grab of a new chunk is issued before the MLPs and read after them.
    // The counters: ONE per pair of workgroups that the dispatcher puts on the same CU (workgroups k and k + K of a grid of 2 K), 64 bytes
    // apart, each over its own 1 / K of the groups -- the imbalance to even out is inside a CU (the older workgroup's waves win the issue
    // arbitration; so do the older waves of a workgroup), and ONE counter for the whole launch serialises: 300 000 same-address atomics
    // per frame were 1.4 ms (measured, profiles/r06_bender_dynamic_ab.txt)
    const int ngroups = (nblocks + NB - 1) / NB;
    const bool dynamic = a.work_counter != nullptr;
    const int K = ((int)gridDim.x + NRN_BX16_OCC - 1) / NRN_BX16_OCC, kc = (int)blockIdx.x % K;
    const int lo = (int)((long long)ngroups * kc / K), hi = dynamic ? (int)((long long)ngroups * (kc + 1) / K) : ngroups;
    unsigned* const counter = a.work_counter + 16 * kc;
    auto grab = [&]() -> int {                      // the next chunk of this counter's range (lane 0's value: read with readfirstlane)
        unsigned c = 0;
        if (lane == 0) c = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (int)c;
    };
    auto chunk_at = [&](int c, int& first, int& end) {      // (beyond the range: an empty chunk at `hi`)
        const long long f = (long long)lo + (long long)c * NRN_BX16_CHUNK;
        first = f < hi ? (int)f : hi;
        end = f + NRN_BX16_CHUNK < hi ? (int)(f + NRN_BX16_CHUNK) : hi;
    };
    int grp, grp_end, nxt, nxt_end;
    if (dynamic) {
        chunk_at(__builtin_amdgcn_readfirstlane(grab()), grp, grp_end);
        if (grp + 1 < grp_end) { nxt = grp + 1; nxt_end = grp_end; }
        else chunk_at(__builtin_amdgcn_readfirstlane(grab()), nxt, nxt_end);
    } else {
        grp = blk0 / NB; grp_end = ngroups; nxt = grp + blk_step / NB; nxt_end = ngroups;
    }
    In cur[NB];
    if (grp < hi) load_inputs(grp * NB, cur);
#ifdef NRN_TIMING
    // (tools/timing_probe_bender.py) slots: 0 iteration, 1 operands, 2 offset MLP, 3 rigidity MLP, 4 tail + stores, 5 iteration in 100 MHz ticks, 7 iterations
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // no barrier below: every wave works through its own groups of NB blocks
    for (; grp < hi; ) {
        const int blk = grp * NB;
#ifdef NRN_TIMING
        const unsigned long long t_it = NRN_NOW(), r_it = __builtin_amdgcn_s_memrealtime();
#endif
        float p[NB][3];
        int out_idx[NB];            // row of this lane's sample in bent4 (-1: nothing to write)
        frag bin[NB][2];            // first-layer operand: k-step 0 = xyz (group 0, elements 0..2), k-step 1 = the latent code
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            float z = cur[b].z;
            if (!a.z) {                                // coarse depths (train.py:847-852), as in the fused kernels
                const float t = lin01(cur[b].kc, npr), near = cur[b].near, far = cur[b].far;
                if (a.lindisp)
                    z = __fdiv_rn(1.0f, __fadd_rn(__fmul_rn(__fdiv_rn(1.0f, near), __fsub_rn(1.0f, t)), __fmul_rn(__fdiv_rn(1.0f, far), t)));
                else
                    z = __fadd_rn(__fmul_rn(near, __fsub_rn(1.0f, t)), __fmul_rn(far, t));
            }
            out_idx[b] = cur[b].idx;
#pragma unroll
            for (int c = 0; c < 3; ++c) p[b][c] = __fadd_rn(cur[b].o[c], __fmul_rn(cur[b].d[c], z));       // train.py:921-923
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bin[b][0][e] = (_Float16)((g == 0 && e < 3) ? p[b][e < 3 ? e : 0] : 0.0f);
                if constexpr (PERRAY) bin[b][1][e] = (_Float16)cur[b].l8[e];
                else bin[b][1][e] = (_Float16)lconst[e];
            }
        });
        // `cur` is consumed (points and first-layer operands built): request the next iteration's inputs into the same registers
        if (nxt < hi) load_inputs(nxt * NB, cur);
        // the group after the next: the next one's successor in its chunk (or stride), or the first of a new chunk -- asked for now
        const bool new_chunk = dynamic && nxt + 1 >= nxt_end;
        int grabbed = 0;
        if (new_chunk) grabbed = grab();
#ifdef NRN_TIMING
        NRN_TACC(1, t_it);
        const unsigned long long t_off = NRN_NOW();
#endif

        frag none[NB][1];
        auto keep = [&](auto& out) {
            return [&](auto pc, auto kc, const f32x4& d0, const f32x4& d1) {
                out[decltype(kc)::value][decltype(pc)::value] = x16_pack<P>(d0, d1);
            };
        };
        // ---- offset MLP (run_nerf_helpers.py:525-541)
        frag ha[NB][NS_B], hb[NB][NS_B];
        dense_x16<P, P, PL, PL::L_BEND0, 2, 0, NB, PF>(st, bias_lane, bin, none, keep(ha));
        static_for<1, A::BD - 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i % 2 == 1) dense_x16<P, P, PL, PL::L_BEND0 + i, NS_B, 0, NB, PF>(st, bias_lane, ha, none, keep(hb));
            else dense_x16<P, P, PL, PL::L_BEND0 + i, NS_B, 0, NB, PF>(st, bias_lane, hb, none, keep(ha));
        });
        float off[NB][3];
        auto take_off = [&](auto, auto kc, const f32x4& d0, const f32x4&) {
            constexpr int k = decltype(kc)::value;
            off[k][0] = d0[0]; off[k][1] = d0[1]; off[k][2] = d0[2];
        };
        if constexpr ((A::BD - 2) % 2 == 1) dense_x16<P, P, PL, PL::L_BEND0 + A::BD - 1, NS_B, 0, NB, PF>(st, bias_lane, hb, none, take_off);
        else dense_x16<P, P, PL, PL::L_BEND0 + A::BD - 1, NS_B, 0, NB, PF>(st, bias_lane, ha, none, take_off);
#ifdef NRN_TIMING
        asm volatile("" : "+v"(off[0][0]));
        NRN_TACC(2, t_off);
        const unsigned long long t_rig = NRN_NOW();
#endif
        // ---- rigidity MLP (run_nerf_helpers.py:545-561); input = xyz only: the first-layer operand's xyz k-step again
        frag ra[NB][NS_R], rb[NB][NS_R];
        dense_x16<P, P, PL, PL::L_RIG0, 1, 0, NB, PF>(st, bias_lane, bin, none, keep(ra));
        static_for<1, A::RD - 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i % 2 == 1) dense_x16<P, P, PL, PL::L_RIG0 + i, NS_R, 0, NB, PF>(st, bias_lane, ra, none, keep(rb));
            else dense_x16<P, P, PL, PL::L_RIG0 + i, NS_R, 0, NB, PF>(st, bias_lane, rb, none, keep(ra));
        });
        float logit[NB];
        auto take_logit = [&](auto, auto kc, const f32x4& d0, const f32x4&) { logit[decltype(kc)::value] = d0[0]; };
        if constexpr ((A::RD - 2) % 2 == 1) dense_x16<P, P, PL, PL::L_RIG0 + A::RD - 1, NS_R, 0, NB, PF>(st, bias_lane, rb, none, take_logit);
        else dense_x16<P, P, PL, PL::L_RIG0 + A::RD - 1, NS_R, 0, NB, PF>(st, bias_lane, ra, none, take_logit);

#ifdef NRN_TIMING
        asm volatile("" : "+v"(logit[0]));
        NRN_TACC(3, t_rig);
        const unsigned long long t_tail = NRN_NOW();
#endif
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            float rig_mask = (tanhf(logit[b]) + 1.0f) / 2.0f;                                    // rnh:559-561
            if (a.knobs.has_cutoff && rig_mask <= a.knobs.cutoff) rig_mask = 0.0f;               // rnh:563-564
            float q[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float mo = __fmul_rn(rig_mask, off[b][c]);                                       // rnh:567
                if (a.knobs.has_scaling) mo = __fmul_rn(mo, a.knobs.scaling);                    // rnh:568-569
                q[c] = __fadd_rn(p[b][c], mo);                                                   // rnh:570
            }
            if (out_idx[b] >= 0 && g == 0) *(f32x4*)(a.bent4 + (size_t)out_idx[b] * 4) = f32x4{q[0], q[1], q[2], rig_mask};
        });
        grp = nxt; grp_end = nxt_end;
        if (new_chunk) chunk_at(__builtin_amdgcn_readfirstlane(grabbed), nxt, nxt_end);
        else if (dynamic) { nxt = nxt + 1; }
        else { nxt = nxt + blk_step / NB; }
#ifdef NRN_TIMING
        NRN_TACC(4, t_tail);
        NRN_TACC(0, t_it);
        tacc[5] += __builtin_amdgcn_s_memrealtime() - r_it;
        tacc[7] += 1;
#endif
    }
#ifdef NRN_TIMING
    if (blockIdx.x == 0 && lane == 0 && wave < 6)
        for (int k = 0; k < 8; ++k) g_nrn_timing[wave][k] += tacc[k];
    // rows 6 / 7: (start, end) in 100 MHz ticks of wave 0 of the FIRST / LAST workgroup, every 64th workgroup's start in slots 2..7 of row 7
    if (lane == 0 && wave == 0) {
        const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
        if (blockIdx.x == 0) { g_nrn_timing[6][0] = rt_start; g_nrn_timing[6][1] = rt_end; }
        if (blockIdx.x == gridDim.x - 1) { g_nrn_timing[7][0] = rt_start; g_nrn_timing[7][1] = rt_end; }
        if (blockIdx.x == 255) { g_nrn_timing[6][2] = rt_start; g_nrn_timing[6][3] = rt_end; }
        if (blockIdx.x == 256) { g_nrn_timing[6][4] = rt_start; g_nrn_timing[6][5] = rt_end; }
        if (blockIdx.x == 384) { g_nrn_timing[6][6] = rt_start; g_nrn_timing[6][7] = rt_end; }
        if (blockIdx.x == 128) { g_nrn_timing[7][2] = rt_start; g_nrn_timing[7][3] = rt_end; }
    }
#endif
}

template <class A>
static hipError_t launch_bend_x16_t(const BendArgs& a, int num_cus, hipStream_t stream) {
    constexpr int WAVES = NRN_BX16_WAVES, NB = NRN_BX16_NB;
    using PL = PlanX16Bend<A>;
    const size_t lds = (size_t)PL::NFRAGS * PolF16::FRAG_BYTES + (size_t)PL::NTILES * 16 * sizeof(float);
    const bool per_ray = a.lat_stride != 0;
    auto kern = per_ray ? bend_kernel_x16<A, WAVES, NB, true> : bend_kernel_x16<A, WAVES, NB, false>;
    static bool attr_set[64][2] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (dev < 0 || dev >= 64 || !attr_set[dev][per_ray]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev][per_ray] = true;
    }
    const long long nblocks = (long long)a.n_rays * ((a.n_per_ray + 15) / 16);
    if (nblocks >= (1ll << 31) || (long long)a.n_rays * a.out_stride >= (1ll << 31)) return hipErrorInvalidValue;
    const long long want = (nblocks + WAVES * NB - 1) / (WAVES * NB);
    if (want <= 0) return hipSuccess;
    const long long resident = (long long)NRN_BX16_OCC * num_cus;
    const int grid = (int)(want < resident ? want : resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace nrn
