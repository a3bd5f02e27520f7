// nrnerf_gen_train.hip -- what the training of a NON-COMPILED architecture needs besides the forward / backward-data programs of
// nrnerf_generic.h (reference: NeRF.forward under autograd, run_nerf_helpers.py:240-314, any --netdepth / --netwidth, train.py:1004-1010):
//
//   tn_products       every weight and bias gradient of a trunk,  dW_i = d z_i^T x_i  and  db_i = sum_m d z_i[m],  over the two saved
//                     ROW-MAJOR arrays ([sample][feature], bf16 or fp32), as ONE launch over a list of jobs + one reduction of the
//                     partial sums.  Replaces the chunked library GEMMs (torch.bmm, 2.9 of a 6.5 ms step at width 192).
//   encoding_rows     Embedder.embed (rnh:120-150) of 3-vectors as rows -- the first operand of the products of the layers that read the
//                     encoding -- and its transpose-Jacobian product (the gradient wrt the points / directions from the encodings').
//                     Replaces a Python posenc under autograd (~45 elementwise launches over [M, 6 L + 3] tensors per pass).
//
// tn_products, bf16: the contraction runs over SAMPLES, but a row-major array has the FEATURES contiguous, so an MFMA operand (8
// consecutive k of one row / column per lane) is a strided gather.  gfx950's LDS transpose read does it in hardware: the tile is staged
// in LDS exactly as it lies in memory ([64 samples][128 features], 16-byte pieces), and ds_read_b64_tr_b16 hands lane i of a 16-lane group
// feature c0 + i of four consecutive samples (tools/probes/tr_b16_probe.hip: lane i supplies the address of piece (row i / 4, columns
// 4 (i % 4) ..) of a [4][16] block and receives column i).  Two of them make one 16x16x32 operand; the k index is a summation index, so
// lane group g takes samples {4 g .. 4 g + 3} and {16 + 4 g ..} of the k-step for BOTH operands -- then the 32 lanes of a half wave read
// sixteen consecutive 32-byte rows of a 16-column sub-tile per instruction: 512 contiguous bytes, conflict-free.
// fp32 (the gradient-parity mode): v_mfma_f32_16x16x4_f32, a lane's single k element read with ds_read_b32 from the same kind of tile.
#include <hip/hip_runtime.h>

#include "nrnerf_gen_train.h"

#ifdef TN_DBG_PLAINREAD
#define TN_TR "ds_read_b64"
#else
#define TN_TR "ds_read_b64_tr_b16"
#endif

namespace nrn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------------------------------------------
// tn_products
// ---------------------------------------------------------------------------------------------------------------------------------
// Panel = what ONE workgroup accumulates: up to 256 output rows x 256 output columns (2 x 4 waves of 128 x 64 = 8 x 4 tiles of 16 x 16,
// 128 accumulator registers per lane -- with 2 x 2 waves of 128 x 128 hipcc kept the 256 accumulator registers out of the AGPRs and
// spilled 840 bytes per lane; one workgroup of eight waves per CU).  Why that large: the contraction index is the SAMPLE, so a product reads
// both arrays once per panel pair -- at width <= 256 each saved array is read exactly once per layer (128 flop per byte, the operation's
// own intensity); a 128 x 128 panel read them twice to four times and ran at the L2's bandwidth (round 6, first version: 6.8 GB of
// traffic for 2.4 GB of arrays at width 192).
constexpr int TN_P = 256;
constexpr int TN_THREADS = 512;
template <bool F32> struct TnTile;
// bf16: the LDS image is cut into 16-column sub-tiles, [column tile][64 samples][16 columns] (a row = 32 bytes): a transpose read of a wave
// then covers 512 CONTIGUOUS bytes (16 samples x 16 columns) -- the one access pattern of ds_read_b64_tr_b16 that is conflict-free
// (cdna_hip_programming.md, T10).  With the tile as plain rows ([64][256], padded) the same reads took ~400 cycles each: the kernel spent
// 1.2 of its 1.45 ms on them (tools/probes/tn_probe.hip, -DTN_DBG_NOLOAD / -DTN_DBG_NOMFMA / -DTN_DBG_NOFRAG).
template <> struct TnTile<false> { static constexpr int KS = 64, SUB_BYTES = 64 * 32 + 64, TILE_BYTES = (TN_P / 16) * SUB_BYTES; };
template <> struct TnTile<true> { static constexpr int KS = 32, ROW_BYTES = TN_P * 4 + 64, TILE_BYTES = KS * ROW_BYTES; };    // 1088: rows 16 banks apart (ds_read_b32: 32 banks)

// fetch rows [m0, m0 + KS) x columns [c0, c0 + 256) of a row-major array as 16-byte pieces, eight per thread.  The arrays' rows and row
// pitch are 16-byte aligned and every row is padded to whole pieces in memory (tn_operand_ok: the API layer refuses anything else, the
// Python side pads) -- so all eight loads are issued unconditionally from clamped addresses and masked when they are staged: no branch,
// no wait between them.  (Written with a branch per piece and an element-wise edge path inline, hipcc serialised the fetch behind
// vmcnt(0) waits: 1.3 TB/s.)
template <bool F32>
__host__ __device__ inline bool tn_operand_ok(const void* base, int ld, int width) {
    constexpr int ES = F32 ? 4 : 2, EPP = 16 / ES;
    return ((((size_t)base | (size_t)((long long)ld * ES)) & 15) == 0) && ((width + EPP - 1) / EPP * EPP <= ld);
}
template <bool F32>
__device__ __forceinline__ void tn_fetch(const void* base, int ld, int width, long long m0, long long m_end, int c0, int tid, u32x4 (&regs)[4]) {
    constexpr int ES = F32 ? 4 : 2, EPP = 16 / ES, PPR = TN_P * ES / 16, NP = TnTile<F32>::KS * PPR / TN_THREADS;      // pieces per row; pieces per thread
    static_assert(NP == 4, "four 16-byte pieces per thread and tile");
    const int last_col = (width + EPP - 1) / EPP * EPP - EPP;          // first column of the row's last whole piece
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int piece = q * TN_THREADS + tid;
        const int r = piece / PPR, pc = piece % PPR;
        const long long m = m0 + r;
        const int col = c0 + pc * EPP;
        const long long mc = m < m_end ? m : m_end - 1;
        const int cc = col < last_col ? col : last_col;
        regs[q] = *(const u32x4*)((const char*)base + ((size_t)mc * ld + cc) * ES);      // (masked when it is staged: tn_stage)
    }
}

// write the fetched pieces into the LDS tile, masked here (what lies beyond the array's rows / columns becomes zero) --
// not where they are loaded, so that nothing between the eight loads needs their data
template <bool F32>
__device__ __forceinline__ void tn_stage(char* tile, int tid, int width, long long m0, long long m_end, int c0, const u32x4 (&regs)[4]) {
    constexpr int ES = F32 ? 4 : 2, EPP = 16 / ES, PPR = TN_P * ES / 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int piece = q * TN_THREADS + tid;
        const int r = piece / PPR, pc = piece % PPR;
        u32x4 v = regs[q];
        const int keep = (m0 + r < m_end) ? width - (c0 + pc * EPP) : 0;                 // valid elements of this piece (<= 0: none, >= EPP: all)
        if constexpr (F32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (e < keep) ? v[e] : 0u;
        } else {
#pragma unroll
            for (int d = 0; d < 4; ++d) v[d] &= ((2 * d < keep) ? 0x0000ffffu : 0u) | ((2 * d + 1 < keep) ? 0xffff0000u : 0u);
        }
        if constexpr (F32) *(u32x4*)(tile + r * TnTile<true>::ROW_BYTES + pc * 16) = v;
        else *(u32x4*)(tile + (pc >> 1) * TnTile<false>::SUB_BYTES + r * 32 + (pc & 1) * 16) = v;
    }
}

template <bool F32>
__global__ void __launch_bounds__(TN_THREADS, 1) tn_products_kernel(const TnKernelArgs a) {
    constexpr int KS = TnTile<F32>::KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // two LDS buffers of (A tile, B tile): tile i is multiplied out of buffer i & 1 while tile i + 1 is written into the other one and
    // tile i + 2 travels from memory into registers -- one barrier per tile, and a load has a whole tile's arithmetic to land in
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;            // rows wr * 128 .., columns wc * 64 ..
    const int g = lane >> 4, n = lane & 15;
    // (neighbouring workgroups = the panels of ONE chunk of samples: they read the same rows of the two arrays, from L2)
    const int sj = blockIdx.x % a.n_sub, chunk = blockIdx.x / a.n_sub;
    const TnSubJob J = a.sub[sj];
    const long long per = (a.n_rows + a.kch - 1) / a.kch;
    const long long perk = (per + KS - 1) / KS * KS;
    const long long m_begin = chunk * perk, m_end = (m_begin + perk < a.n_rows) ? m_begin + perk : a.n_rows;
    const int rows_here = J.wo - J.o0 < TN_P ? J.wo - J.o0 : TN_P, cols_here = J.wi - J.k0 < TN_P ? J.wi - J.k0 : TN_P;
    // this wave's 16 x 16 tiles: rows wr * 128 + 16 t (t < nt), columns wc * 64 + 16 u (u < nu)
    const int nt_ = (rows_here - wr * 128 + 15) / 16, nu_ = (cols_here - wc * 64 + 15) / 16;
    const int nt = nt_ < 0 ? 0 : (nt_ > 8 ? 8 : nt_), nu = nu_ < 0 ? 0 : (nu_ > 4 ? 4 : nu_);
    const bool bias_wave = J.bias_off >= 0 && J.k0 == 0 && wc == 0 && nt > 0;

    f32x4 acc[8][4];
    float accb[8];             // (bias waves) this lane's share of the column sums of A: rows 16 t + n, its k slots
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        accb[t] = 0.0f;
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    constexpr int TB = TnTile<F32>::TILE_BYTES;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    u32x4 ra[4], rb[4];
    if (m_begin < m_end) {                        // tile 0 -> buffer 0; tile 1 -> registers
        tn_fetch<F32>(J.a, J.lda, J.wo, m_begin, m_end, J.o0, tid, ra);
        tn_fetch<F32>(J.b, J.ldb, J.wi, m_begin, m_end, J.k0, tid, rb);
        tn_stage<F32>(smem, tid, J.wo, m_begin, m_end, J.o0, ra);
        tn_stage<F32>(smem + TB, tid, J.wi, m_begin, m_end, J.k0, rb);
        if (m_begin + KS < m_end) {
            tn_fetch<F32>(J.a, J.lda, J.wo, m_begin + KS, m_end, J.o0, tid, ra);
            tn_fetch<F32>(J.b, J.ldb, J.wi, m_begin + KS, m_end, J.k0, tid, rb);
        }
    }
    __syncthreads();
    int buf = 0;
    for (long long m0 = m_begin; m0 < m_end; m0 += KS, buf ^= 1) {
        char* const tileA = smem + buf * 2 * TB;
        char* const tileB = tileA + TB;
        const unsigned ldsA = lds0 + (unsigned)(buf * 2 * TB), ldsB = ldsA + (unsigned)TB;
        if (m0 + KS < m_end) {                    // the next tile: registers -> the other buffer (last read one barrier ago)
            tn_stage<F32>(smem + (buf ^ 1) * 2 * TB, tid, J.wo, m0 + KS, m_end, J.o0, ra);
            tn_stage<F32>(smem + (buf ^ 1) * 2 * TB + TB, tid, J.wi, m0 + KS, m_end, J.k0, rb);
        }
#ifndef TN_DBG_NOLOAD
        if (m0 + 2 * KS < m_end) {                // the tile after it travels while this one is multiplied
            tn_fetch<F32>(J.a, J.lda, J.wo, m0 + 2 * KS, m_end, J.o0, tid, ra);
            tn_fetch<F32>(J.b, J.ldb, J.wi, m0 + 2 * KS, m_end, J.k0, tid, rb);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);        // (the loads stay ahead of the arithmetic: hipcc sinks them towards their use otherwise)
#ifndef TN_DBG_NOFRAG
        if ((nt > 0 && nu > 0) || bias_wave) {
        if constexpr (!F32) {
#pragma unroll
            for (int kk = 0; kk < KS / 32; ++kk) {
                // lane i of group g: piece (row 32 kk + 4 g + i / 4 [+ 16], columns 4 (i % 4) .. of the sub-tile) -> feature c0 + i, 4 samples
                constexpr int SB = TnTile<false>::SUB_BYTES;
                const unsigned rowoff = (unsigned)((32 * kk + 4 * g + (n >> 2)) * 32 + 8 * (n & 3));
                // (no guard per tile: the tile's columns beyond the job's width are staged as zeros, so a wave whose panel is only partly
                //  covered multiplies zeros there -- the time of a step is the fully covered wave's anyway.  Guards per MFMA made hipcc
                //  carry the 256 accumulator registers through a phi copy at every branch: 13 000 cycles per k-step.)
                bf16x8 fa[8], fb[4];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    u32x2 lo, hi;
                    const unsigned ad = ldsA + rowoff + (unsigned)((wr * 8 + t) * SB);
                    asm volatile(TN_TR " %0, %1" : "=v"(lo) : "v"(ad));
                    asm volatile(TN_TR " %0, %1 offset:512" : "=v"(hi) : "v"(ad));
                    fa[t] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    u32x2 lo, hi;
                    const unsigned ad = ldsB + rowoff + (unsigned)((wc * 4 + u) * SB);
                    asm volatile(TN_TR " %0, %1" : "=v"(lo) : "v"(ad));
                    asm volatile(TN_TR " %0, %1 offset:512" : "=v"(hi) : "v"(ad));
                    fb[u] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int t = 0; t < 8; ++t) asm volatile("" : "+v"(fa[t]));
#pragma unroll
                for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(fb[u]));
                // (the lower half of the wave's rows only when the panel reaches it: ONE wave-uniform branch around 16 MFMAs)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t], fb[u], acc[t][u], 0, 0, 0);
                if (nt > 4) {
#pragma unroll
                    for (int t = 4; t < 8; ++t)
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t], fb[u], acc[t][u], 0, 0, 0);
                }
                if (bias_wave) {                  // column sums of A, on the vector ALU (8 registers instead of 8 accumulator tiles)
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        float sum = 0.0f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) sum += (float)fa[t][e];
                        accb[t] += sum;
                    }
                }
            }
        } else {
#pragma unroll 2
            for (int kk = 0; kk < KS / 4; ++kk) {
                constexpr int RB = TnTile<true>::ROW_BYTES;
                const char* ra_ = tileA + (4 * kk + g) * RB + (wr * 128 + n) * 4;
                const char* rb_ = tileB + (4 * kk + g) * RB + (wc * 64 + n) * 4;
                float fa[8], fb[4];
#pragma unroll
                for (int t = 0; t < 8; ++t) fa[t] = *(const float*)(ra_ + 64 * t);
#pragma unroll
                for (int u = 0; u < 4; ++u) fb[u] = *(const float*)(rb_ + 64 * u);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[t], fb[u], acc[t][u], 0, 0, 0);
                if (nt > 4) {
#pragma unroll
                    for (int t = 4; t < 8; ++t)
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[t], fb[u], acc[t][u], 0, 0, 0);
                }
                if (bias_wave) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) accb[t] += fa[t];
                }
            }
        }
        }
#endif
        __syncthreads();                          // this tile's fragment reads are done, the next tile's image is complete
    }
    // this workgroup's partial sums, at their FINAL positions of record `chunk` (D tile: lane (g, n) holds rows 4 g .. 4 g + 3, column n)
    float* rec = a.partials + (size_t)chunk * a.total;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (t >= nt) continue;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u >= nu) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = wr * 128 + 16 * t + 4 * g + e, col = wc * 64 + 16 * u + n;
                if (row < rows_here && col < cols_here) rec[J.out_off + (long long)(J.o0 + row) * J.ldo + (J.k0 + col)] = acc[t][u][e];
            }
        }
    }
    if (bias_wave) {              // (wave-uniform) the four lane groups hold the four quarters of every k-step: add them up
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float v = accb[t];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int row = wr * 128 + 16 * t + n;
            if (g == 0 && t < nt && row < rows_here) rec[J.bias_off + J.o0 + row] = v;
        }
    }
}

// out[i] = sum over the records, in order (deterministic); 4 loads in flight
__global__ void __launch_bounds__(256) tn_reduce_kernel(const float* parts, long long total, int kch, float* out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float s = 0.0f;
    int c = 0;
    for (; c + 3 < kch; c += 4) {
        const float v0 = parts[(size_t)c * total + i], v1 = parts[(size_t)(c + 1) * total + i], v2 = parts[(size_t)(c + 2) * total + i], v3 = parts[(size_t)(c + 3) * total + i];
        s = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(s, v0), v1), v2), v3);
    }
    for (; c < kch; ++c) s = __fadd_rn(s, parts[(size_t)c * total + i]);
    out[i] = s;
}

hipError_t launch_tn_clear(float* partials, long long total, int kch, hipStream_t stream) {
    // records start from zero: positions no job covers (a 5th head channel, ...) must sum to zero
    return hipMemsetAsync(partials, 0, (size_t)kch * total * sizeof(float), stream);
}
hipError_t launch_tn_products(const TnKernelArgs& a, bool f32, hipStream_t stream) {
    if (a.n_sub <= 0 || a.n_sub > TN_MAX_SUBJOBS || a.kch < 1 || a.total <= 0 || a.n_rows <= 0 || !a.partials) return hipErrorInvalidValue;
    for (int j = 0; j < a.n_sub; ++j) {          // 16-byte aligned rows, padded to whole 16-byte pieces (the kernel loads nothing narrower)
        const TnSubJob& J = a.sub[j];
        const bool ok = f32 ? (tn_operand_ok<true>(J.a, J.lda, J.wo) && tn_operand_ok<true>(J.b, J.ldb, J.wi))
                            : (tn_operand_ok<false>(J.a, J.lda, J.wo) && tn_operand_ok<false>(J.b, J.ldb, J.wi));
        if (!ok) return hipErrorInvalidValue;
    }
    const size_t lds = 4 * (size_t)(f32 ? TnTile<true>::TILE_BYTES : TnTile<false>::TILE_BYTES);
    const dim3 grid((unsigned)(a.n_sub * a.kch));
    if (f32) {
        if (hipFuncSetAttribute((const void*)tn_products_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return hipErrorUnknown;
        hipLaunchKernelGGL(tn_products_kernel<true>, grid, dim3(TN_THREADS), lds, stream, a);
    } else {
        if (hipFuncSetAttribute((const void*)tn_products_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return hipErrorUnknown;
        hipLaunchKernelGGL(tn_products_kernel<false>, grid, dim3(TN_THREADS), lds, stream, a);
    }
    return hipGetLastError();
}
hipError_t launch_tn_reduce(const float* partials, long long total, int kch, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, partials, total, kch, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// encoding rows
// ---------------------------------------------------------------------------------------------------------------------------------
// forward: one thread per (row, frequency slot): slot q < L writes the six columns of frequency q, slot L the identity columns, the
// appended code columns and the zero padding up to enc_cols
template <bool B16>
__global__ void __launch_bounds__(256) encoding_fwd_kernel(const EncodingArgs a) {
    const int Q = a.L + 1;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= a.n_rows * Q) return;
    const long long row = t / Q;
    const int q = (int)(t % Q);
    const float* sp = a.src + (size_t)row * a.src_stride;
    const float p[3] = {sp[0], sp[1], sp[2]};
    auto put = [&](int col, float v) {
        if constexpr (B16) ((__bf16*)a.enc)[(size_t)row * a.enc_cols + col] = (__bf16)v;
        else ((float*)a.enc)[(size_t)row * a.enc_cols + col] = v;
    };
    if (q < a.L) {
        const float sc = (float)(1 << q);                   // freq_bands = 2 ** linspace(0, L - 1, L): exact powers of two (rnh:139-141)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sn, cs;
            sincosf(__fmul_rn(p[c], sc), &sn, &cs);
            put(3 + 6 * q + c, sn);
            put(3 + 6 * q + 3 + c, cs);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) put(c, p[c]);
        int col = 3 + 6 * a.L;
        if (a.codes) {
            const float* cp = a.codes + (size_t)(row / a.rows_per_code) * a.n_lat;
            for (int k = 0; k < a.n_lat; ++k) put(col + k, cp[k]);
            col += a.n_lat;
        }
        for (; col < a.enc_cols; ++col) put(col, 0.0f);
    }
}
// backward: one thread per row: d src[c] = g[c] + sum_k 2^k (cos(2^k p_c) g[3 + 6 k + c] - sin(2^k p_c) g[3 + 6 k + 3 + c]),  g = d_enc0 (+ d_enc1)
__global__ void __launch_bounds__(256) encoding_bwd_kernel(const EncodingArgs a) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.n_rows) return;
    const float* sp = a.src + (size_t)row * a.src_stride;
    const float* g0 = a.d_enc0 + (size_t)row * a.d_enc_stride;
    const float* g1 = a.d_enc1 ? a.d_enc1 + (size_t)row * a.d_enc_stride : nullptr;
    auto gat = [&](int col) { return g1 ? __fadd_rn(g0[col], g1[col]) : g0[col]; };
    float d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = gat(c);
    for (int k = 0; k < a.L; ++k) {
        const float sc = (float)(1 << k);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sn, cs;
            sincosf(__fmul_rn(sp[c], sc), &sn, &cs);
            d[c] += sc * (cs * gat(3 + 6 * k + c) - sn * gat(3 + 6 * k + 3 + c));
        }
    }
    float* o = a.d_src + (size_t)row * a.d_src_stride;
    o[0] = d[0]; o[1] = d[1]; o[2] = d[2];
    for (int c = 3; c < a.d_src_stride; ++c) o[c] = 0.0f;
}

hipError_t launch_encoding_rows(const EncodingArgs& a, bool backward, hipStream_t stream) {
    if (a.n_rows <= 0) return hipSuccess;
    if (!a.src || a.src_stride < 3 || a.L < 0 || a.L > 16) return hipErrorInvalidValue;
    if (!backward) {
        if (!a.enc || a.enc_cols < 3 + 6 * a.L + (a.codes ? a.n_lat : 0) || (a.codes && (a.n_lat < 1 || a.rows_per_code < 1))) return hipErrorInvalidValue;
        const long long total = a.n_rows * (a.L + 1);
        const dim3 grid((unsigned)((total + 255) / 256));
        if (a.enc_bf16) hipLaunchKernelGGL(encoding_fwd_kernel<true>, grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(encoding_fwd_kernel<false>, grid, dim3(256), 0, stream, a);
    } else {
        if (!a.d_enc0 || !a.d_src || a.d_enc_stride < 3 + 6 * a.L || a.d_src_stride < 3) return hipErrorInvalidValue;
        hipLaunchKernelGGL(encoding_bwd_kernel, dim3((unsigned)((a.n_rows + 255) / 256)), dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

}  // namespace nrn
