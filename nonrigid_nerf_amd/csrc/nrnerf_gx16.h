// nrnerf_gx16.h -- the width-class trunk kernel on v_mfma_f32_16x16x32 for architectures outside the compiled set (see
// nrnerf_gx16_plan.h for why and for the layer kinds).  Trunk-only, like net_kernel_x16's raw-to-memory case: positional encoding of
// READY-MADE points (the bender pass' output), pts_linears with any depth / one skip index, output_linear; bf16 or f16.
// Everything else of such a model (the bender, compositing, sampling) runs on the kernels it already had.
#pragma once
#include "nrnerf_gx16_plan.h"
#include "nrnerf_net_x16.h"
#include "nrnerf_x16_api.h"

namespace nrn {

// the LDS weight ring with a RUN-TIME source pointer: unit V of the current layer comes from lbase + V * UNIT; slots are compile-time
// because every layer streams a whole number of ring periods (PlanGX::NUP).  Otherwise WRing (nrnerf_net_impl.h).
template <class P, int WAVES>
struct WRingRT {
    static constexpr int UNIT = P::UNIT_BYTES;
    static constexpr int PW = UNIT / 1024 / WAVES;
    static constexpr int LAG = NRN_RING_LAG;
    static constexpr bool ASM_FRAGS = (P::FRAG_BYTES == 1024);
    static_assert(ASM_FRAGS && LAG >= 1 && RING - LAG >= 2 && UNIT % (1024 * WAVES) == 0, "16-bit fragments, at least one unit of DMA lead");
    const char* lbase;     // first unit of the current layer + this wave's piece offset (wave-uniform)
    unsigned lane16;
    char* ring;
    int wave_off;
    unsigned lane_addr;

    __device__ __forceinline__ void init(const void* stream, char* lds, int wave, int lane) {
        wave_off = wave * PW * 1024;
        lbase = (const char*)stream + wave_off;
        lane16 = (unsigned)lane * 16u;
        ring = lds;
        lane_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (unsigned)lane * 16u;
        static_for<0, RING - LAG>([&](auto uc) { issue<decltype(uc)::value>(); });
    }
    template <int V>
    __device__ __forceinline__ void issue() {
        unsigned off = (unsigned)(V * UNIT);
        asm volatile("" : "+s"(off));
        const char* src = (lbase + off) + lane16;
        char* dst = ring + (V % RING) * UNIT + wave_off;
        static_for<0, PW>([&](auto ic) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, decltype(ic)::value * 1024, 0);
        });
    }
    template <int U>
    __device__ __forceinline__ void advance() {
        wait_ring<(RING - LAG - 1) * PW>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue<U + RING - LAG>();                       // (beyond the layer's last unit: the next layer's first units, contiguous in the stream)
    }
    template <class PX, int GF>
    __device__ __forceinline__ typename PX::frag frag() {
        constexpr int UF = P::UNIT_FRAGS;
        if constexpr (GF % UF == 0) advance<GF / UF>();
        constexpr int OFF = ((GF / UF) % RING) * UNIT + (GF % UF) * P::FRAG_BYTES;
        static_assert(OFF + 16 <= 65536, "ring must stay within the immediate ds_read offset");
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lane_addr), "n"(OFF));
        return __builtin_bit_cast(typename PX::frag, v);
    }
    // the layer's padding units, then on to the next layer's block
    template <class PLK>
    __device__ __forceinline__ void end_layer() {
        static_for<PLK::NUNITS, PLK::NUP>([&](auto uc) { advance<decltype(uc)::value>(); });
        lbase += (size_t)PLK::NUP * UNIT;
    }
    __device__ __forceinline__ void rewind(const void* stream) { lbase = (const char*)stream + wave_off; }
    __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// VIEWS: the view-dependent head behind the trunk (directions = finite differences of the points' rows, as net_kernel_x16<VIEWS>)
// FUSE: the pass is a FINAL one and its compositing (raw2outputs, train.py:943-950) runs as the kernel's epilogue, as in net_kernel_x16 --
// a wave owns whole rays (groups of RW rays = TG iterations of NB blocks, strided over the grid), keeps their raw outputs in its own LDS
// stage and composites them after the group's last iteration (composite_ray: the composite kernel's own code, same bits); the pass' raw
// array never reaches HBM and the composite launch goes.  Passes of up to 256 samples (a lane owns 1..4 of them: one code block per
// case behind a switch -- this kernel is not the one whose last per cent is counted).
// SAVE (training forward of a non-compiled trunk, nrnerf_generic_trunk_forward; plain head): every hidden activation h_i = relu(W_i x_i + b_i)
// is also written to GxArgs::save as [layer][sample][save_w] rows in the model's 16-bit type -- what nrnerf_generic_trunk_backward masks with
// and nrnerf_tn_products contracts over -- straight from the registers: a lane holds features 32 p + 4 g .. + 3 and 32 p + 16 + 4 g .. + 3 of
// its sample, so the four lanes of a sample write 64 contiguous bytes per tile pair (two 8-byte stores each).
template <class P, int WC, int NB, bool VIEWS, bool FUSE = false, bool SAVE = false>
__global__ void __launch_bounds__(4 * 64, 1) gx16_kernel(const GxArgs a) {
    static_assert(!SAVE || (!VIEWS && !FUSE), "saved activations: the plain head, raw outputs to memory");
    constexpr int WAVES = 4;
    using PE = PolF16;                                                // the encoding's operands are f16 in both modes
    using frag = typename P::frag;
    using efrag = typename PE::frag;
    using PIN = PlanGX<WC, GX_IN>;
    using PHID = PlanGX<WC, GX_HID>;
    using PSKIP = PlanGX<WC, GX_SKIP>;
    using PHEAD = PlanGX<WC, GX_HEAD>;
    using PVIEWS = PlanGX<WC, GX_VIEWS>;
    using PRGB = PlanGX<WC, GX_RGB>;
    constexpr int NS_V = WC / 64;
    constexpr int NS_H = WC / 32, NS_E = GX_NS_E, NT = WC / 16;
    constexpr int PF = (WC > 256) ? 4 : 8;

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

    const int S = a.S, D = a.depth, skip = a.skip, L = a.L;
    const int bpr = (S + 15) >> 4;
    const long long nblocks = (long long)a.n_rays * bpr;
    const long long per_wg = (long long)WAVES * NB;
    // (FUSE) ray groups, as net_kernel_x16: RW = the fewest rays whose blocks fill whole iterations of NB blocks
    const int RW = (bpr % NB == 0) ? 1 : ((2 * bpr) % NB == 0 ? 2 : NB);
    const int TG = RW * bpr / NB;
    const long long ngroups = ((long long)a.n_rays + WAVES * RW - 1) / (WAVES * RW);
    f32x4* const stage0 = (f32x4*)(bias_lds + a.n_bias_tiles * 16);
    f32x4* const stage_w = stage0 + (size_t)wave * RW * bpr * 16;
    const CompositeArgs& fa = *(const CompositeArgs*)(stage0 + (size_t)WAVES * RW * bpr * 16);      // (in LDS: a kernel argument read per use costs scalar loads)
    if constexpr (FUSE) {
        static_assert(sizeof(CompositeArgs) <= 256, "the compositing arguments' LDS slot");
        int* dst = (int*)(stage0 + (size_t)WAVES * RW * bpr * 16);
        const int* src = (const int*)&a.fuse;
        for (int i = tid; i < (int)(sizeof(CompositeArgs) / 4); i += WAVES * 64) dst[i] = src[i];
        __syncthreads();
    }
    int tg = 0;
    long long grp = blockIdx.x;
    for (long long b0 = (long long)blockIdx.x * per_wg; FUSE ? (grp < ngroups) : (b0 < nblocks); b0 += (long long)gridDim.x * per_wg) {
        unsigned so[NB];
        bool ok[NB];
        long long blkid[NB];        // (SAVE) the block's index among all 16-sample blocks, -1: none
        efrag enc[NB][NS_E];
        efrag encv[NB][1];
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            bool blk_ok;
            int ray, bir;
            if constexpr (FUSE) {                   // block q of this wave's group: ray (grp * WAVES + wave) * RW + q / bpr
                const int q = tg * NB + b;
                const long long rr = (grp * WAVES + wave) * RW + q / bpr;
                blk_ok = rr < a.n_rays;
                ray = (int)(blk_ok ? rr : a.n_rays - 1);
                bir = q % bpr;
            } else {
                const long long blk_raw = b0 + (long long)wave * NB + b;
                blk_ok = blk_raw < nblocks;
                const long long blk = blk_ok ? blk_raw : nblocks - 1;
                ray = (int)(blk / bpr); bir = (int)(blk % bpr);
            }
            blkid[b] = blk_ok ? (long long)ray * bpr + bir : -1;
            const int sidx = bir * 16 + n;
            ok[b] = blk_ok && sidx < S;
            so[b] = (unsigned)ray * (unsigned)S + (unsigned)(sidx < S ? sidx : S - 1);
            const f32x4 q4 = *(const f32x4*)(a.pts4 + (size_t)so[b] * 4);
            if constexpr (VIEWS) {
                // the sample's direction: finite difference of the points along the ray, sample 0 takes sample 1's (rnh:339-351)
                const int sc = sidx < S ? sidx : S - 1;
                const bool first = sc == 0;
                const unsigned nbr = (unsigned)ray * (unsigned)S + (unsigned)(first ? (S > 1 ? 1 : 0) : sc - 1);
                const f32x4 nb4 = *(const f32x4*)(a.pts4 + (size_t)nbr * 4);
                float dd[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) dd[c] = first ? __fsub_rn(nb4[c], q4[c]) : __fsub_rn(q4[c], nb4[c]);
                const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dd[0], dd[0]), __fmul_rn(dd[1], dd[1])), __fmul_rn(dd[2], dd[2])));
                float dir[3], drev[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { dir[c] = __fdiv_rn(dd[c], __fadd_rn(nrm, 0.000001f)); drev[c] = dir[c] * 0.15915494309189535f; }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int p = 8 * g + 2 * i;
                    const int m = (p - 4) >> 1;
                    const int f = m / 3, c = m - 3 * f;
                    const float xr = c == 0 ? drev[0] : (c == 1 ? drev[1] : drev[2]);
                    const float r = __builtin_amdgcn_fractf(xr * __builtin_amdgcn_ldexpf(1.0f, f));
                    float sv = __builtin_amdgcn_sinf(r), cv = __builtin_amdgcn_cosf(r);
                    if (p < 4) { sv = (p == 0) ? dir[0] : dir[2]; cv = (p == 0) ? dir[1] : 0.0f; }
                    else if (m >= 3 * a.LV) { sv = 0.0f; cv = 0.0f; }
                    encv[b][0][2 * i] = (_Float16)sv;
                    encv[b][0][2 * i + 1] = (_Float16)cv;
                }
            }
            const float prev[3] = {q4[0] * 0.15915494309189535f, q4[1] * 0.15915494309189535f, q4[2] * 0.15915494309189535f};
#pragma unroll
            for (int s = 0; s < NS_E; ++s) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {          // the lane's 4 (sin, cos) pairs of this k-step: positions 32 s + 8 g + 2 i, + 1
                    const int p = 32 * s + 8 * g + 2 * i;
                    const int m = (p - 4) >> 1;        // (p = 0, 2 of lane group 0, k-step 0: the identity columns instead)
                    const int f = m / 3, c = m - 3 * f;
                    const float xr = c == 0 ? prev[0] : (c == 1 ? prev[1] : prev[2]);
                    const float r = __builtin_amdgcn_fractf(xr * __builtin_amdgcn_ldexpf(1.0f, f));
                    float sv = __builtin_amdgcn_sinf(r), cv = __builtin_amdgcn_cosf(r);
                    if (p < 4) { sv = (p == 0) ? q4[0] : q4[2]; cv = (p == 0) ? q4[1] : 0.0f; }
                    else if (m >= 3 * L) { sv = 0.0f; cv = 0.0f; }
                    enc[b][s][2 * i] = (_Float16)sv;
                    enc[b][s][2 * i + 1] = (_Float16)cv;
                }
            }
        });

        frag ha[NB][NS_H], hb[NB][NS_H];
        frag none[NB][1];
        int lsave = 0;              // (SAVE) index of the layer whose activations the epilogue in flight writes
        constexpr int MW = (NS_H + 3) / 4;
        unsigned mbits[NB][MW];     // (SAVE) which of the lane's values passed the relu: one byte per tile pair (GxArgs::relu_bits)
        auto flush_bits = [&]() {
            if constexpr (SAVE) {
                if (a.relu_bits) {
                    static_for<0, NB>([&](auto bc) {
                        constexpr int b = decltype(bc)::value;
                        if (blkid[b] >= 0) {
                            unsigned* p = (unsigned*)((char*)a.relu_bits + (((size_t)lsave * nblocks + blkid[b]) * 64 + lane) * (size_t)(4 * MW));
#pragma unroll
                            for (int w = 0; w < MW; ++w) p[w] = mbits[b][w];
                        }
                    });
                }
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int w = 0; w < MW; ++w) mbits[b][w] = 0u;
            }
        };
        if constexpr (SAVE) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int w = 0; w < MW; ++w) mbits[b][w] = 0u;
        }
        auto keep = [&](auto& out) {
            return [&](auto pc, auto kc, const f32x4& d0, const f32x4& d1) {
                constexpr int k = decltype(kc)::value, p = decltype(pc)::value;
                const frag v = x16_pack<P>(d0, d1);
                out[k][p] = v;
                if constexpr (SAVE) {
                    const int col = 32 * p + 4 * g;
                    if (ok[k] && col < a.save_w) {          // (save_w % 4 == 0: whole 4-feature pieces)
                        const u32x4 w = __builtin_bit_cast(u32x4, v);
                        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
                        unsigned short* row = (unsigned short*)a.save + ((size_t)lsave * a.save_stride + (size_t)so[k] * a.save_w);
                        *(u32x2_*)(row + col) = u32x2_{w[0], w[1]};
                        if (col + 16 < a.save_w) *(u32x2_*)(row + col + 16) = u32x2_{w[2], w[3]};
                    }
                    const u32x4 w = __builtin_bit_cast(u32x4, v);
                    unsigned byte = 0u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) byte |= ((w[q] & 0xffffu) ? 1u : 0u) << (2 * q) | ((w[q] >> 16) ? 1u : 0u) << (2 * q + 1);
                    mbits[k][p >> 2] |= byte << (8 * (p & 3));
                }
            };
        };
        BP bl = bias_lane0;
        asm volatile("" : "+v"(bl));
        dense_x16<PE, P, PIN, 0, NS_E, 0, NB, PF>(st, bl, enc, none, keep(ha));
        flush_bits();
        st.template end_layer<PIN>();
        bl += NT * 4;                                   // (f32x4 units: 16 floats per tile)
        // the layers two at a time (ha -> hb -> ha: which array holds the activations is then a compile-time fact in every code block;
        // a run-time flag for it cost 280 spilled registers at width class 256), an odd one last; the head reads whichever is current
        f32x4 raw[NB];
        auto take = [&](auto, auto kc, const f32x4& d0, const f32x4&) { raw[decltype(kc)::value] = d0; };
        auto layer = [&](int l, auto& in, auto& out) __attribute__((always_inline)) {
            lsave = l;
            asm volatile("" : "+v"(bl));
            if (l - 1 == skip) { dense_x16<PE, P, PSKIP, 0, NS_E, NS_H, NB, PF>(st, bl, enc, in, keep(out)); flush_bits(); st.template end_layer<PSKIP>(); }
            else { dense_x16<P, P, PHID, 0, NS_H, 0, NB, PF>(st, bl, in, none, keep(out)); flush_bits(); st.template end_layer<PHID>(); }
            bl += NT * 4;
        };
        int l = 1;
        for (; l + 1 < D; l += 2) {
            layer(l, ha, hb);
            layer(l + 1, hb, ha);
        }
        auto head = [&](auto& hx) __attribute__((always_inline)) {
            asm volatile("" : "+v"(bl));
            if constexpr (!VIEWS) {
                dense_x16<P, P, PHEAD, 0, NS_H, 0, NB, PF>(st, bl, hx, none, take);
                static_for<PHEAD::NUNITS, PHEAD::NUP>([&](auto uc) { st.template advance<decltype(uc)::value>(); });
            } else {
                // hv = relu(views o feature ([enc(dir), h])) in tile pairs, sigma = the lone last tile's row 0; then rgb = rgb_linear(hv)
                frag hv[NB][NS_V];
                float sigma[NB];
                auto views_epi = [&](auto pc, auto kc, const f32x4& d0, const f32x4& d1) {
                    constexpr int p = decltype(pc)::value, k = decltype(kc)::value;
                    if constexpr (p < NS_V) hv[k][p] = x16_pack<P>(d0, d1);
                    else sigma[k] = d0[0];
                };
                dense_x16<PE, P, PVIEWS, 0, 1, NS_H, NB, PF>(st, bl, encv, hx, views_epi);
                st.template end_layer<PVIEWS>();
                bl += PVIEWS::NT * 4;
                asm volatile("" : "+v"(bl));
                dense_x16<P, P, PRGB, 0, NS_V, 0, NB, PF>(st, bl, hv, none, take);
                static_for<PRGB::NUNITS, PRGB::NUP>([&](auto uc) { st.template advance<decltype(uc)::value>(); });
                static_for<0, NB>([&](auto bc) { raw[decltype(bc)::value][3] = sigma[decltype(bc)::value]; });
            }
        };
        if (l < D) {
            layer(l, ha, hb);
            head(hb);
        } else {
            head(ha);
        }
        // (the last layer's padding units ran on into the copy of the first layer's first units behind it) back to the stream's start
        st.rewind(a.wstream);

        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            if (ok[b] && g == 0) {
                if constexpr (!FUSE) *(f32x4*)(a.raw4 + (size_t)so[b] * 4) = raw[b];
                if (a.raw_out) {
                    float* ro = a.raw_out + (size_t)so[b] * a.raw_ch;
                    ro[0] = raw[b][0]; ro[1] = raw[b][1]; ro[2] = raw[b][2]; ro[3] = raw[b][3];
                }
            }
            if (ok[b] && g == 1 && a.raw_out && a.raw_ch > 4) a.raw_out[(size_t)so[b] * a.raw_ch + 4] = raw[b][0];
            if (FUSE && g == 0) stage_w[(tg * NB + b) * 16 + n] = raw[b];
        });
        if constexpr (FUSE) {
            if (tg + 1 == TG) {       // the group's last iteration: composite this wave's rays from its LDS stage (train.py:943-950)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (int r = 0; r < RW; ++r) {
                    const long long rr = (grp * WAVES + wave) * RW + r;
                    const bool ray_ok = rr < a.n_rays;
                    const int cray = (int)(ray_ok ? rr : a.n_rays - 1);
                    const f32x4* sw = stage_w + r * bpr * 16;
                    auto raw_at = [&](int ic) { return sw[ic]; };
                    switch ((S + 63) >> 6) {
                        case 1: { float cz[2], cw[1]; composite_ray<1, false>(fa, cray, ray_ok, lane, raw_at, cz, cw); break; }
                        case 2: { float cz[3], cw[2]; composite_ray<2, false>(fa, cray, ray_ok, lane, raw_at, cz, cw); break; }
                        case 3: { float cz[4], cw[3]; composite_ray<3, false>(fa, cray, ray_ok, lane, raw_at, cz, cw); break; }
                        default: { float cz[5], cw[4]; composite_ray<4, false>(fa, cray, ray_ok, lane, raw_at, cz, cw); break; }
                    }
                }
                tg = 0; grp += gridDim.x;
            } else {
                tg += 1;
            }
        }
    }
    st.drain();
}

template <class P, int WC, bool VIEWS, bool FUSE, bool SAVE = false>
static hipError_t launch_gx16_tf(const GxArgs& a, int num_cus, hipStream_t stream) {
    constexpr int WAVES = 4, NB = (WC > 256) ? 2 : 4;
    if (!a.pts4 || (!a.raw4 && !FUSE) || a.S < 1 || a.depth < 1 || a.L < 0 || a.L > GX_MAX_L) return hipErrorInvalidValue;
    size_t lds = (size_t)RING * P::UNIT_BYTES + (size_t)a.n_bias_tiles * 16 * sizeof(float);
    if (FUSE) {
        if (a.S > 256 || a.fuse.n_importance != 0 || a.fuse.S != a.S) return hipErrorInvalidValue;
        const int bpr_ = (a.S + 15) / 16;
        const int RW = (bpr_ % NB == 0) ? 1 : ((2 * bpr_) % NB == 0 ? 2 : NB);
        lds += (size_t)WAVES * RW * bpr_ * 16 * 16 + 256;            // the waves' raw stages + the compositing arguments
    }
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (VIEWS && (a.LV < 0 || a.LV > GX_MAX_LV)) return hipErrorInvalidValue;
    // sample rows are 32-bit numbers in the kernel (so[], the neighbour's row): refuse what would wrap (launch_bend_x16_t does the same)
    if ((long long)a.n_rays * a.S >= (1ll << 32) || (long long)a.n_rays * ((a.S + 15) / 16) >= (1ll << 31)) return hipErrorInvalidValue;
    auto kern = gx16_kernel<P, WC, NB, VIEWS, FUSE, SAVE>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return hipErrorUnknown;
    const long long bpr = (a.S + 15) / 16;
    long long want = ((long long)a.n_rays * bpr + WAVES * NB - 1) / (WAVES * NB);
    if (FUSE) {
        const int RW = (bpr % NB == 0) ? 1 : ((2 * bpr) % NB == 0 ? 2 : NB);
        want = ((long long)a.n_rays + WAVES * RW - 1) / (WAVES * RW);          // groups of WAVES * RW whole rays
    }
    if (want <= 0) return hipSuccess;
    const int grid = (int)(want < num_cus ? want : num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}
template <class P, int WC, bool VIEWS>
static hipError_t launch_gx16_t(const GxArgs& a, int num_cus, hipStream_t stream) {
    if (a.save) {
        if constexpr (!VIEWS && std::is_same_v<P, PolBF16>) {
            if (a.fuse_on || a.save_w % 4 != 0 || a.save_w < 4 || a.save_w > WC || a.save_stride < (long long)a.n_rays * a.S * a.save_w) return hipErrorInvalidValue;
            return launch_gx16_tf<P, WC, false, false, true>(a, num_cus, stream);
        } else {
            return hipErrorInvalidValue;          // (training runs in bf16; the view-dependent head keeps the run-time-parameterised kernel)
        }
    }
    return a.fuse_on ? launch_gx16_tf<P, WC, VIEWS, true>(a, num_cus, stream) : launch_gx16_tf<P, WC, VIEWS, false>(a, num_cus, stream);
}
// rays of one fused-compositing group of the width-class kernel (the API layer's "enough rays to fuse" threshold)
static inline long long gx16_rays_per_group_of(int wc, int S) {
    const int NB = (wc > 256) ? 2 : 4, bpr = (S + 15) / 16;
    const int RW = (bpr % NB == 0) ? 1 : ((2 * bpr) % NB == 0 ? 2 : NB);
    return 4ll * RW;
}

}  // namespace nrn
