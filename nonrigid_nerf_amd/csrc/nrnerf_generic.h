// nrnerf_generic.h -- the network kernels for ANY architecture the reference can build.
//
// The reference instantiates NeRF(D = --netdepth, W = --netwidth) / (--netdepth_fine, --netwidth_fine) with
// --multires / --multires_views frequencies and a ray bender fed by --ray_bending_latent_size codes (train.py:1004-1010,
// 1060, 1133-1139, 564-630).  The kernels of nrnerf_net_impl.h / nrnerf_net_mb.h are compiled per architecture: their
// activations never leave registers and their weight stream has compile-time addresses, which is what makes them fast and
// what ties them to a handful of shapes.  This file is the other end of the trade: ONE kernel per precision whose layer
// list, widths and encodings are run-time data, so every shape outside the compiled set still renders natively (slower
// -- activations take a round trip through LDS per layer and the weights come from L2 without a ring -- but on the matrix
// pipe, with the same arithmetic types, behind the same C ABI).
//
// Dataflow.  A workgroup (4 waves) owns a tile of NS = 32 * NSB consecutive samples of the flattened [ray, sample] index.
// Three activation buffers live in LDS, one row per sample:
//     E  the network input vector (positional encoding of the point [, latent code]; bender: [point, latent code])
//     V  the second input (direction encoding; bender: the bare point, input of the rigidity network)
//     H  the hidden activations (in place: a layer's outputs replace its inputs between two barriers)
// A layer is computed transposed like everywhere in this library, D^T = W . X^T: the weights are the MFMA A operand -- packed
// on the host in fragment order (tile, k-slab), one coalesced 16-byte (16-bit) / 4-byte (fp32) load per lane and fragment,
// straight from L2 -- and the activations the B operand, read from LDS (lane = sample, 8 / 1 consecutive k per lane).  A wave
// owns output tiles wave, wave + 4, ... for all NSB sample blocks (each weight fragment feeds NSB MFMAs); its accumulators
// hold the complete layer output, so one buffer suffices.  A layer reads up to two sources (the skip layer: [E | H], the
// view-dependent layer: [H | V]), in the reference's column order: the packer needs no permutation.
// Heads (<= 8 output rows) land in a small fp32 buffer O; the epilogue turns it into raw logits or the bent point.
//
// Precision: "f32" exact fp32 (v_mfma_f32_32x32x2_f32); "bf16" / "f16": H in that type, E and V always f16 (bounded by
// construction, like nrnerf_plan.h::frag_is_f16); the ray bender ALWAYS runs the fp32 instantiation (its offsets feed a
// 2^(L-1)-frequency encoding; 3 % of the flops).
#pragma once
#include <hip/hip_runtime.h>

#include "nrnerf_kernels.h"
#include "nrnerf_net_impl.h"      // precision policies, lin01, static_for

namespace nrn {

// A "slab" of the generic kernel is one 16-byte fetch per lane of each operand: 16-bit types 8 k per lane = ONE 32x32x16 MFMA; fp32
// 4 k per lane = FOUR 32x32x2 MFMAs (lane half h holds k = 8 s + 4 h + j, j = 0..3: MFMA j contracts the pair {8 s + j, 8 s + 4 + j}
// -- any pairing is a valid order of the sum as long as the packed weights use the same one).  One dword per lane and MFMA, as first
// built, left the fp32 instantiation waiting for L2 on every 64-cycle MFMA: 0.7 x the eager PyTorch path on a wide network.
template <class P>
struct GenTypes {
    using PE = std::conditional_t<P::KH == 1, PolF32, PolF16>;      // element type of E and V
    using elem = std::conditional_t<P::KH == 1, float, unsigned short>;
    using gfrag = std::conditional_t<P::KH == 1, f32x4, typename P::frag>;      // 16 bytes per lane either way
    static constexpr int KHG = (P::KH == 1) ? 4 : 8;               // k per lane and slab
    static constexpr int KSG = 2 * KHG;                            // k per slab
    static constexpr int PAD = (P::KH == 1) ? 4 : 8;               // row padding in elements: rows stay 16-byte aligned, 128-bit reads spread over the banks
};
constexpr int GEN_FRAG_BYTES = 1024;                               // 64 lanes x 16 bytes, every precision

__device__ __forceinline__ unsigned short gen_to_f16(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }
__device__ __forceinline__ unsigned short gen_to_bf16(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
template <class P> __device__ __forceinline__ typename GenTypes<P>::elem gen_cvt(float v) {
    if constexpr (P::KH == 1) return v;
    else if constexpr (std::is_same_v<P, PolBF16>) return gen_to_bf16(v);
    else return gen_to_f16(v);
}

// B operand of slab s of `buf` for sample row `n`: 8 (16-bit) / 4 (fp32) consecutive k from column s * KSG + h * KHG, one 128-bit LDS read
template <class PX>
__device__ __forceinline__ typename GenTypes<PX>::gfrag gen_bfrag(const void* buf, int stride, int n, int s, int h) {
    if constexpr (PX::KH == 1) {
        return *(const f32x4*)((const float*)buf + (size_t)n * stride + 8 * s + 4 * h);
    } else {
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        const u32x4_ v = *(const u32x4_*)((const unsigned short*)buf + (size_t)n * stride + 16 * s + 8 * h);
        return __builtin_bit_cast(typename PX::frag, v);
    }
}
// acc += A(slab) . B(slab): one MFMA (16-bit) / four (fp32)
template <class PX>
__device__ __forceinline__ f32x16 gen_mfma(const typename GenTypes<PX>::gfrag& a, const typename GenTypes<PX>::gfrag& b, f32x16 acc) {
    if constexpr (PX::KH == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = PX::mfma(a[j], b[j], acc);
        return acc;
    } else {
        return PX::mfma(a, b, acc);
    }
}

// MAXT: output tiles per wave and layer the instantiation has accumulators for (2: widths <= 256, half the registers, two
// workgroups per CU; 4: widths <= 512).  TRAIN: the training entry points' instantiations (nrnerf_generic_trunk_forward / _backward:
// saved activations, relu masks, the backward-data mode, per-sample directions, outputs straight to memory) -- compiled out of the
// rendering instantiations, whose register allocation they would otherwise cost (35 .. 95 spilled registers, measured on the ISA)
template <class P, int NSB, int MAXT, bool TRAIN>
__global__ void __launch_bounds__(GEN_WAVES * 64, (NSB * MAXT <= 4) ? 2 : 1) gen_kernel(const GenArgs a) {
    using PE = typename GenTypes<P>::PE;
    using elem = typename GenTypes<P>::elem;
    constexpr int PAD = GenTypes<P>::PAD;
    constexpr int NS = 32 * NSB, KH = P::KH, FB = GEN_FRAG_BYTES;
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    const int se = a.ke + PAD, sv = a.kv + PAD, sh = a.kh + PAD;            // row strides in elements
    elem* E = (elem*)gsm;
    elem* V = E + (size_t)NS * se;
    elem* H = V + (size_t)NS * sv;
    float* O = (float*)(H + (size_t)NS * sh);                               // [NS][8]
    float* Pt = O + NS * 8;                                                 // [NS][8]: point xyz, (bender) unit direction / spare
    float* Bt = Pt + NS * 8;                                                // the program's whole bias table (a.bias_in_lds)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, j = lane & 31;
    // biases from LDS, not from L2 once per layer and tile: nothing overlaps that latency with one or two workgroups per CU
    if (a.bias_in_lds) {
        for (int i = tid; i < a.n_bias_tiles * 32; i += GEN_WAVES * 64) Bt[i] = a.bias[i];
        __syncthreads();
    }
    const float* bias_tab = a.bias_in_lds ? Bt : a.bias;
    const int S = a.S;
    const long long M = (long long)a.n_rays * S;
    const long long ntile = (M + NS - 1) / NS;

    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long long m0 = tile * NS;
        if (TRAIN && a.mode == 2) {
            // ---- backward-data (training): H = the rows of d raw (columns >= draw_ch zero) in the model's type; no points, no E / V
            for (int idx = tid; idx < NS * 16; idx += GEN_WAVES * 64) {
                const int n = idx >> 4, c = idx & 15;
                const float v = (c < a.draw_ch && m0 + n < M) ? a.draw[(size_t)(m0 + n) * a.draw_ch + c] : 0.0f;
                H[(size_t)n * sh + a.draw_col + c] = gen_cvt<P>(v);
            }
        }
        // ---- points (and view directions) of the tile's samples
        if ((!TRAIN || a.mode != 2) && tid < NS) {
            const long long m = (m0 + tid < M) ? m0 + tid : M - 1;
            const int ray = (int)(m / S), si = (int)(m % S);
            const float* rp = a.rays + (size_t)ray * a.ray_stride;
            float p[3];
            if (a.mode == 1 && a.pts4) {
                const f32x4 q = *(const f32x4*)(a.pts4 + (size_t)m * 4);
                p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
            } else {
                float z;
                if (a.z) z = a.z[(size_t)ray * S + si];
                else {
                    const float near = rp[6], far = rp[7], t = lin01(si, S);
                    if (a.lindisp) z = __fdiv_rn(1.0f, __fadd_rn(__fmul_rn(__fdiv_rn(1.0f, near), __fsub_rn(1.0f, t)), __fmul_rn(__fdiv_rn(1.0f, far), t)));
                    else z = __fadd_rn(__fmul_rn(near, __fsub_rn(1.0f, t)), __fmul_rn(far, t));            // train.py:849-852
                }
                for (int c = 0; c < 3; ++c) p[c] = __fadd_rn(rp[c], __fmul_rn(rp[3 + c], z));               // train.py:871-873
            }
            float d[3] = {0.f, 0.f, 0.f};
            if (a.mode == 1 && a.LV >= 0) {
                if (TRAIN && a.dirs) {           // training: the caller's direction of this sample
                    const float* dp = a.dirs + (size_t)m * 3;
                    d[0] = dp[0]; d[1] = dp[1]; d[2] = dp[2];
                } else if (a.dirs_from_pts) {           // rnh:339-351: backward difference of the bent points, sample 0 copies sample 1
                    const bool first = (si == 0);
                    const f32x4 nb = *(const f32x4*)(a.pts4 + (size_t)(first ? m + 1 : m - 1) * 4);
                    float dd[3];
                    for (int c = 0; c < 3; ++c) dd[c] = first ? __fsub_rn(nb[c], p[c]) : __fsub_rn(p[c], nb[c]);
                    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dd[0], dd[0]), __fmul_rn(dd[1], dd[1])), __fmul_rn(dd[2], dd[2])));
                    for (int c = 0; c < 3; ++c) d[c] = __fdiv_rn(dd[c], __fadd_rn(nrm, 0.000001f));
                } else {
                    d[0] = rp[8]; d[1] = rp[9]; d[2] = rp[10];                                               // train.py:73-76
                }
            }
            float* pt = Pt + tid * 8;
            pt[0] = p[0]; pt[1] = p[1]; pt[2] = p[2]; pt[3] = 0.f; pt[4] = d[0]; pt[5] = d[1]; pt[6] = d[2]; pt[7] = 0.f;
            if (a.mode == 0 && m0 + tid < M && a.ex.init_pts) {
                a.ex.init_pts[(size_t)m * 3] = p[0]; a.ex.init_pts[(size_t)m * 3 + 1] = p[1]; a.ex.init_pts[(size_t)m * 3 + 2] = p[2];
            }
        }
        __syncthreads();
        // ---- E and V rows.  network: E = Embedder(point) [, latent], V = Embedder(direction) (rnh:120-150: [x, sin(2^0 x),
        //      cos(2^0 x), sin(2^1 x), ...]); bender: E = [point, latent] (rnh:525), V = [point] (rnh:546)
        const int enc_w = (a.mode == 1) ? 3 + 6 * a.L : 3;
        for (int idx = tid; (!TRAIN || a.mode != 2) && idx < NS * a.ke; idx += GEN_WAVES * 64) {
            const int n = idx / a.ke, c = idx - n * a.ke;
            const float* pt = Pt + n * 8;
            float v = 0.f;
            if (c < 3) v = pt[c];
            else if (c < enc_w) {
                const int q = c - 3, f = q / 6, r = q - 6 * f;
                const float x = pt[r % 3] * (float)(1 << f);                       // power-of-two scaling: exact
                v = (r < 3) ? sinf(x) : cosf(x);
            } else if (c < enc_w + a.lat) {
                const long long m = (m0 + n < M) ? m0 + n : M - 1;
                v = a.latents[(size_t)(m / S) * a.lat_stride + (c - enc_w)];
            }
            E[(size_t)n * se + c] = gen_cvt<PE>(v);
        }
        for (int idx = tid; (!TRAIN || a.mode != 2) && idx < NS * a.kv; idx += GEN_WAVES * 64) {
            const int n = idx / a.kv, c = idx - n * a.kv;
            const float* pt = Pt + n * 8 + (a.mode == 1 ? 4 : 0);
            float v = 0.f;
            if (c < 3) v = pt[c];
            else if (a.mode == 1 && c < 3 + 6 * a.LV) {
                const int q = c - 3, f = q / 6, r = q - 6 * f;
                const float x = pt[r % 3] * (float)(1 << f);
                v = (r < 3) ? sinf(x) : cosf(x);
            }
            V[(size_t)n * sv + c] = gen_cvt<PE>(v);
        }
        __syncthreads();

        // ---- layers.  The first weight fragments of layer l + 1 are requested before layer l's outputs are written back (two
        //      barriers away from their use): per layer only the k-slab pipeline's steady state is left exposed.
        typedef typename GenTypes<P>::gfrag wfrag;         // (fragments against E / V are f16, against H the model's type: same size)
        wfrag nxt[MAXT];
        auto first_frags = [&](int li) {
            const GenLayer& ly = a.layer[li];
            const char* wb = (const char*)a.wstream + (size_t)ly.w_frag * FB + (size_t)lane * (FB / 64);
            const int ns = ly.ns0 + ly.ns1;
#pragma unroll
            for (int i = 0; i < MAXT; ++i)
                if (wave + i * GEN_WAVES < ly.nt) nxt[i] = *(const wfrag*)(wb + (size_t)(wave + i * GEN_WAVES) * ns * FB);
        };
        first_frags(0);
        for (int li = 0; li < a.n_layers; ++li) {
            const GenLayer& ly = a.layer[li];
            const int ns = ly.ns0 + ly.ns1;
            f32x16 acc[MAXT][NSB];
            int ntw = 0;                                   // this wave's tiles: wave, wave + 4, ...
#pragma unroll
            for (int i = 0; i < MAXT; ++i) {
                const int t = wave + i * GEN_WAVES;
                if (t < ly.nt) {
                    ntw = i + 1;
                    const f32x4* bp = (const f32x4*)(bias_tab + ((size_t)(ly.bias_tile + t) * 32 + h * 16));
                    const f32x4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
                    const f32x16 bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3], b2[0], b2[1], b2[2], b2[3], b3[0], b3[1], b3[2], b3[3]};
#pragma unroll
                    for (int sb = 0; sb < NSB; ++sb) acc[i][sb] = bv;
                }
            }
            const char* wbase = (const char*)a.wstream + (size_t)ly.w_frag * FB + (size_t)lane * (FB / 64);
            auto src_ptr = [&](int b) -> const void* { return b == GB_E ? (const void*)E : (b == GB_V ? (const void*)V : (const void*)H); };
            auto src_stride = [&](int b) { return b == GB_E ? se : (b == GB_V ? sv : sh); };
            auto run = [&](auto pxc, const void* buf, int stride, int s_begin, int s_count, bool have_first) {
                using PX = typename decltype(pxc)::type;
                // weight fragments of slab s + 1 are requested (from L2) before the MFMAs of slab s
                using GF = typename GenTypes<PX>::gfrag;
                GF af[MAXT], an[MAXT];
                auto fetch = [&](GF (&dst)[MAXT], int s) {
#pragma unroll
                    for (int i = 0; i < MAXT; ++i)
                        if (i < ntw) dst[i] = *(const GF*)(wbase + ((size_t)(wave + i * GEN_WAVES) * ns + s_begin + s) * FB);
                };
                if (have_first) {
#pragma unroll
                    for (int i = 0; i < MAXT; ++i) af[i] = __builtin_bit_cast(GF, nxt[i]);
                } else if (s_count > 0) {
                    fetch(af, 0);
                }
                for (int s = 0; s < s_count; ++s) {
                    if (s + 1 < s_count) fetch(an, s + 1);
                    GF bf[NSB];
#pragma unroll
                    for (int sb = 0; sb < NSB; ++sb) bf[sb] = gen_bfrag<PX>(buf, stride, sb * 32 + j, s, h);
#pragma unroll
                    for (int i = 0; i < MAXT; ++i) {
                        if (i < ntw) {
#pragma unroll
                            for (int sb = 0; sb < NSB; ++sb) acc[i][sb] = gen_mfma<PX>(af[i], bf[sb], acc[i][sb]);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < MAXT; ++i) af[i] = an[i];
                }
            };
            struct TagE { using type = PE; };
            struct TagH { using type = P; };
            if (ly.src0 == GB_H) run(TagH{}, H + (TRAIN ? ly.boff0 : 0), sh, 0, ly.ns0, true); else run(TagE{}, src_ptr(ly.src0), src_stride(ly.src0), 0, ly.ns0, true);
            if (ly.ns1 > 0) {
                if (ly.src1 == GB_H) run(TagH{}, H + (TRAIN ? ly.boff1 : 0), sh, ly.ns0, ly.ns1, false); else run(TagE{}, src_ptr(ly.src1), src_stride(ly.src1), ly.ns0, ly.ns1, false);
            }
            if (li + 1 < a.n_layers) first_frags(li + 1);
            __syncthreads();                               // every wave has read the layer's inputs: H may be overwritten
#pragma unroll
            for (int i = 0; i < MAXT; ++i) {
                if (i < ntw) {
                    const int t = wave + i * GEN_WAVES;
#pragma unroll
                    for (int sb = 0; sb < NSB; ++sb) {
                        const int n = sb * 32 + j;
                        if (TRAIN && ly.dst >= GB_OUT0) { // training, backward-data: an encoding's gradient, straight to memory
                            float* go = a.gout[ly.dst - GB_OUT0];
                            const int gw = ly.dst == GB_OUT2 ? a.gout_w2 : a.gout_w;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int row = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                                if (row < ly.o_rows && m0 + n < M) go[(size_t)(m0 + n) * gw + row] = acc[i][sb][r];
                            }
                        } else if (ly.dst == GB_H) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {          // registers 4q .. 4q + 3 = rows 32 t + 8 q + 4 h + (0..3) (nrnerf_plan.h::tile_row)
                                float v[4];
#pragma unroll
                                for (int c = 0; c < 4; ++c) v[c] = ly.relu ? fmaxf(acc[i][sb][4 * q + c], 0.0f) : acc[i][sb][4 * q + c];
                                if (TRAIN && a.mask && ly.mask_idx >= 0) {  // training, backward-data: d pre = d h where the forward activation passed the relu
                                    const int col = 32 * t + 8 * q + 4 * h;
                                    const bool live = m0 + n < M && col + 3 < a.save_w;
                                    const elem* mp = (const elem*)a.mask + (size_t)ly.mask_idx * a.save_stride + (size_t)(live ? m0 + n : 0) * a.save_w + (live ? col : 0);
#pragma unroll
                                    for (int c = 0; c < 4; ++c) {
                                        bool pos;
                                        if constexpr (KH == 1) pos = mp[c] > 0.0f;
                                        else pos = (mp[c] & 0x7fff) != 0 && !(mp[c] & 0x8000);       // a positive 16-bit float (activations are >= 0)
                                        v[c] = (live && pos) ? v[c] : 0.0f;
                                    }
                                }
                                elem* dstp = H + (size_t)n * sh + 32 * t + 8 * q + 4 * h;
                                if constexpr (KH == 1) {                 // four floats = one 16-byte LDS store (rows are 16-byte aligned)
                                    *(f32x4*)dstp = f32x4{v[0], v[1], v[2], v[3]};
                                } else {                           // four 16-bit values = one 8-byte LDS store (rows are 8-byte aligned)
                                    typedef unsigned short u16x4_ __attribute__((ext_vector_type(4)));
                                    *(u16x4_*)dstp = u16x4_{gen_cvt<P>(v[0]), gen_cvt<P>(v[1]), gen_cvt<P>(v[2]), gen_cvt<P>(v[3])};
                                }
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int row = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                                if (row < ly.o_rows) O[n * 8 + ly.o_col + row] = ly.relu ? fmaxf(acc[i][sb][r], 0.0f) : acc[i][sb][r];
                            }
                        }
                    }
                }
            }
            __syncthreads();
            if (TRAIN && a.save && ly.save_idx >= 0) {  // training: this layer's H rows (forward: activations; backward: d pre) to memory, 4 elements per lane
                const int vpr = a.save_w >> 2;           // (save_w % 4 == 0: the launcher checks)
                elem* sp = (elem*)a.save + (size_t)ly.save_idx * a.save_stride;
                for (int idx = tid; idx < NS * vpr; idx += GEN_WAVES * 64) {
                    const int n = idx / vpr, c = (idx - n * vpr) * 4;
                    if (m0 + n < M) {
                        if constexpr (KH == 1) *(f32x4*)(sp + (size_t)(m0 + n) * a.save_w + c) = *(const f32x4*)(H + (size_t)n * sh + c);
                        else {
                            typedef unsigned short u16x4s_ __attribute__((ext_vector_type(4)));
                            *(u16x4s_*)(sp + (size_t)(m0 + n) * a.save_w + c) = *(const u16x4s_*)(H + (size_t)n * sh + c);
                        }
                    }
                }
                // (the next layer's write-back into H is two barriers away: these reads are done by then)
            }
        }

        // ---- epilogue
        if ((!TRAIN || a.mode != 2) && tid < NS && m0 + tid < M) {
            const long long m = m0 + tid;
            const float* o = O + tid * 8;
            if (a.mode == 1) {
                float sigma = o[3];
                if (a.knobs.detailed && a.knobs.has_removal && a.pts4 && a.bent4 && a.bent4[(size_t)m * 4 + 3] >= a.knobs.removal) sigma = sigma * 0.0f;   // rnh:308-311
                if (!a.pts4) {       // a model without ray bender: the points of this pass are its own (detail outputs, surface reduction)
                    const float* pt = Pt + tid * 8;
                    if (a.bent4) *(f32x4*)(a.bent4 + (size_t)m * 4) = f32x4{pt[0], pt[1], pt[2], 0.0f};
                    for (int c = 0; c < 3; ++c) {
                        if (a.ex.init_pts) a.ex.init_pts[(size_t)m * 3 + c] = pt[c];
                        if (a.ex.in_pts) a.ex.in_pts[(size_t)m * 3 + c] = pt[c];
                    }
                }
                if (a.raw4) *(f32x4*)(a.raw4 + (size_t)m * 4) = f32x4{o[0], o[1], o[2], sigma};
                if (a.raw_out) {
                    float* ro = a.raw_out + (size_t)m * a.raw_ch;
                    ro[0] = o[0]; ro[1] = o[1]; ro[2] = o[2]; ro[3] = sigma;
                    if (a.raw_ch > 4) ro[4] = o[4];
                }
            } else {
                const float* pt = Pt + tid * 8;
                float mask = (tanhf(o[3]) + 1.0f) / 2.0f;                                      // rnh:559-561
                if (a.knobs.has_cutoff && mask <= a.knobs.cutoff) mask = 0.0f;                 // rnh:563-564
                float mo[3], bent[3];
                for (int c = 0; c < 3; ++c) {
                    mo[c] = __fmul_rn(mask, o[c]);                                             // rnh:567
                    if (a.knobs.has_scaling) mo[c] = __fmul_rn(mo[c], a.knobs.scaling);        // rnh:568-569
                    bent[c] = __fadd_rn(pt[c], mo[c]);                                         // rnh:570
                }
                *(f32x4*)(a.bent4 + (size_t)m * 4) = f32x4{bent[0], bent[1], bent[2], mask};
                for (int c = 0; c < 3; ++c) {
                    if (a.ex.unmasked) a.ex.unmasked[(size_t)m * 3 + c] = o[c];
                    if (a.ex.masked) a.ex.masked[(size_t)m * 3 + c] = mo[c];
                    if (a.ex.in_pts) a.ex.in_pts[(size_t)m * 3 + c] = bent[c];
                }
                if (a.ex.rigidity) a.ex.rigidity[m] = mask;
            }
        }
        __syncthreads();
    }
}

template <class P, int NSB, int MAXT, bool TRAIN>
static hipError_t launch_gen_t(const GenArgs& a_in, int num_cus, hipStream_t stream) {
    constexpr int PAD = GenTypes<P>::PAD, NS = 32 * NSB;
    const size_t es = sizeof(typename GenTypes<P>::elem);
    GenArgs a = a_in;
    const size_t act = (size_t)NS * ((a.ke + PAD) + (a.kv + PAD) + (a.kh + PAD)) * es + (size_t)NS * 16 * sizeof(float);
    a.bias_in_lds = (a.n_bias_tiles > 0 && a.n_bias_tiles <= 320) ? 1 : 0;            // <= 40 KB of biases
    const size_t lds = act + (a.bias_in_lds ? (size_t)a.n_bias_tiles * 128 : 0);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = gen_kernel<P, NSB, MAXT, TRAIN>;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const long long M = (long long)a.n_rays * a.S;
    const long long ntile = (M + NS - 1) / NS;
    if (ntile <= 0) return hipSuccess;
    // workgroups per CU: by registers (launch bounds: two when NSB * MAXT <= 4) and by LDS
    const long long by_regs = (NSB * MAXT <= 4) ? 2 : 1, by_lds = (long long)(160 * 1024) / (long long)(lds + 1024);
    const long long resident = (long long)num_cus * (by_regs < by_lds ? by_regs : (by_lds < 1 ? 1 : by_lds));
    const int grid = (int)(ntile < resident ? ntile : resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(GEN_WAVES * 64), lds, stream, a);
    return hipGetLastError();
}
template <class P, int NSB, bool TRAIN>
static hipError_t launch_gen(const GenArgs& a, int num_cus, hipStream_t stream) {
    if (a.ke % 16 || a.kv % 16 || a.kh % 16 || a.ke > GEN_MAX_E || a.kv > GEN_MAX_V || a.kh > GEN_MAX_W || a.n_layers < 1 ||
        a.n_layers > GEN_MAX_LAYERS) return hipErrorInvalidValue;
    if (!TRAIN && (a.mode == 2 || a.save || a.mask || a.dirs)) return hipErrorInvalidValue;      // (the training instantiations' business)
    if ((a.save || a.mask) && (a.save_w % 4 != 0 || a.save_w < 4 || a.save_w > a.kh)) return hipErrorInvalidValue;
    if (a.mode == 2 && (!a.draw || a.draw_ch < 1 || a.draw_ch > 16 || a.draw_col % 16 != 0 || a.draw_col < 0 || a.draw_col + 16 > a.kh)) return hipErrorInvalidValue;
    int widest = 0;
    for (int l = 0; l < a.n_layers; ++l) widest = a.layer[l].nt > widest ? a.layer[l].nt : widest;
    if (widest > GEN_WAVES * GEN_MAXT) return hipErrorInvalidValue;
    return (widest <= GEN_WAVES * 2) ? launch_gen_t<P, NSB, 2, TRAIN>(a, num_cus, stream) : launch_gen_t<P, NSB, 4, TRAIN>(a, num_cus, stream);
}

}  // namespace nrn
