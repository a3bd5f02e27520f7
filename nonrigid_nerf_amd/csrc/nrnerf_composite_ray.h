// nrnerf_composite_ray.h -- alpha compositing of ONE ray by ONE wavefront (reference raw2outputs, train.py:724-789) and the
// per-ray surface reduction (free_viewpoint_rendering.py:621-658), as a device function shared by
//   * composite_kernel (nrnerf_composite.hip): one wave per ray, raw [N,S,4] read from HBM, followed there by sample_pdf and
//     the merge when the pass is the coarse one of a hierarchical render;
//   * the network kernels' fused epilogue (nrnerf_net_mb.h / nrnerf_net_impl.h, variants without a fused bender): a wave owns
//     whole rays, keeps their raw outputs in LDS and composites them itself -- the final pass' raw array never reaches HBM
//     ("compositing fused into the ray loop", BASELINE.json north_star; train.py:943-959 calls raw2outputs inline).
// The SAME code in both places, so the two routes give the same bits (asserted: tests/test_gpu_parity.py).
//
// Lane l owns the EPL consecutive samples l*EPL .. l*EPL+EPL-1: the exclusive transmittance product is a short in-lane serial
// scan followed by one 64-lane prefix scan.
#pragma once
#include <hip/hip_runtime.h>

#include "nrnerf_kernels.h"

// NRN_FUSE_PREFETCH (build-time, experiments): 0 = the fused epilogue loads its inputs itself, 1 = only the ray direction is
// requested ahead, 2 = direction and depths (default)
#ifndef NRN_FUSE_PREFETCH
#define NRN_FUSE_PREFETCH 2
#endif

namespace nrn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float c_lin01(int i, int n) {     // torch.linspace(0,1,n)[i], fp32
    if (n <= 1) return 0.0f;
    const float step = __fdiv_rn(1.0f, (float)(n - 1));
    return (i < n / 2) ? __fmul_rn(step, (float)i) : __fsub_rn(1.0f, __fmul_rn(step, (float)(n - 1 - i)));
}

// The arguments' pointers are device-memory pointers.  Saying so matters where the argument block was copied to LDS (the fused
// epilogues): read back from there a pointer is generic, its accesses become FLAT instructions -- which count in both memory
// counters -- and hipcc's waits around them degrade to vmcnt(0), i.e. to waiting out the weight ring's LDS-DMA and earlier stores.
template <class T>
__device__ __forceinline__ __attribute__((address_space(1))) T* gmem(T* p) { return (__attribute__((address_space(1))) T*)p; }

// a value the optimiser cannot look into: whatever produced it is rounded to fp32 here, nothing is fused across it
__device__ __forceinline__ float rounded(float x) { asm("" : "+v"(x)); return x; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// inclusive prefix scans over the 64 lanes
__device__ __forceinline__ float wave_scan_add(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(v, o); if (lane >= o) v += t; }
    return v;
}
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(v, o); if (lane >= o) v *= t; }
    return v;
}

// what composite_ray reads from HBM before it can start, requested early: out[0..2] = direction, out[3 + k] = depth of sample
// lane * epl + k (k < epl <= 4; only when the pass has explicit depths)
__device__ __forceinline__ void composite_prefetch(const CompositeArgs& a, const int ray, const int lane, const int epl, float (&out)[8]) {
    const __attribute__((address_space(1))) float* rp = gmem(a.rays) + (size_t)ray * a.ray_stride;
    if (NRN_FUSE_PREFETCH == 0) return;
    out[0] = rp[3]; out[1] = rp[4]; out[2] = rp[5];
    if (a.z && NRN_FUSE_PREFETCH >= 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = lane * epl + k, ic = i < a.S ? i : a.S - 1;
            if (k < epl) out[3 + k] = gmem(a.z)[(size_t)ray * a.S + ic];
        }
    }
}

// `raw_at(i)` returns the f32x4 (r, g, b, sigma logits) of sample i of this ray (i < S).  On return z[k] / w[k] hold the
// depth / visibility weight of sample lane*EPL + k (z[EPL]: the next lane's first depth), which sample_pdf goes on to use.
// `pre` (optional): values the caller fetched ahead of time -- pre[0..2] = the ray's direction, pre[3 + k] = the depth of
// sample lane*EPL + k when a.z is given (composite_prefetch below) -- so that a fused epilogue does not wait for HBM.
template <int EPL, bool FENCED = false, class RAWF>
__device__ __forceinline__ void composite_ray(const CompositeArgs& a, const int ray, const bool ray_ok, const int lane,
                                              RAWF&& raw_at, float (&z)[EPL + 1], float (&w)[EPL], const float* pre = nullptr) {
    // FENCED (the 16x16x32 kernel's epilogue, where a lone wave per SIMD runs this code with nothing to overlap it): scheduling fences
    // between the phases.  Same instructions, same bits; measured with NRN_TIMING, together with ONE instantiation in the kernel
    // instead of four behind a switch: 6 600 -> 6 300 cycles per iteration (tools/experiments/README.md, round 4).
    auto phase_fence = [] { if constexpr (FENCED) __builtin_amdgcn_sched_barrier(0); };
    // No multiply-add fusion the source does not spell out (__fmaf_rn).  hipcc's `__fmul_rn` / `__fadd_rn` are plain * and +, this
    // build's -ffp-contract=fast fuses them wherever a product feeds a sum IN ONE BASIC BLOCK (pragmas are not honoured), and
    // which blocks an instantiation ends up with depends on everything around it: with the ray direction prefetched a tile ahead
    // the 16-bit network kernels evaluated |d| as fma(dz, dz, dx*dx + dy*dy) where the composite kernel did not -- an ulp of
    // `dnorm`, hence of every alpha, on 1 % of the rays (tools/experiments/debug_fused_composite.py).  `rounded()` hides a product
    // from the optimiser, so it is rounded before it is added in every instantiation.
    const int S = a.S;
    const __attribute__((address_space(1))) float* rp = gmem(a.rays) + (size_t)ray * a.ray_stride;
    if (NRN_FUSE_PREFETCH == 0) pre = nullptr;
    const float dx = pre ? pre[0] : rp[3], dy = pre ? pre[1] : rp[4], dz = pre ? pre[2] : rp[5];
    float near = 0.0f, far = 0.0f;
    if (!a.z) { near = rp[6]; far = rp[7]; }
    const float dnorm = sqrtf(__fadd_rn(__fadd_rn(rounded(dx * dx), rounded(dy * dy)), rounded(dz * dz)));      // train.py:748

    // ---- load this lane's samples
    float sig[EPL], col[EPL][3];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int i = lane * EPL + k;
        const int ic = i < S ? i : S - 1;
        if (a.z) z[k] = (pre && NRN_FUSE_PREFETCH >= 2) ? pre[3 + k] : gmem(a.z)[(size_t)ray * S + ic];
        else {
            const float t = c_lin01(ic, S);
            if (a.lindisp)                                                               // train.py:850-852
                z[k] = __fdiv_rn(1.0f, __fadd_rn(rounded(__fdiv_rn(1.0f, near) * __fsub_rn(1.0f, t)),
                                                 rounded(__fdiv_rn(1.0f, far) * t)));
            else
                z[k] = __fadd_rn(rounded(near * __fsub_rn(1.0f, t)), rounded(far * t));     // train.py:849
        }
        const f32x4 r = raw_at(ic);
        col[k][0] = r[0]; col[k][1] = r[1]; col[k][2] = r[2]; sig[k] = r[3];
        if (a.noise) sig[k] = __fadd_rn(sig[k], gmem(a.noise)[(size_t)ray * S + ic]);                // train.py:761
    }
    z[EPL] = __shfl_down(z[0], 1);     // first depth of the next lane
    phase_fence();

    // ---- alpha, transmittance, weights (train.py:740-775)
    float alpha[EPL];
    float run = 1.0f;                  // product of (1 - alpha + 1e-10) over this lane's samples so far
    float texcl[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int i = lane * EPL + k;
        float dist = (i == S - 1) ? 1e10f : __fsub_rn(z[k + 1], z[k]);      // :743-746
        dist = __fmul_rn(dist, dnorm);                                      // :748
        const float s = fmaxf(sig[k], 0.0f);
        alpha[k] = (i < S) ? __fsub_rn(1.0f, expf(-__fmul_rn(s, dist))) : 0.0f;   // :741
        texcl[k] = run;
        run = __fmul_rn(run, (i < S) ? __fadd_rn(__fsub_rn(1.0f, alpha[k]), 1e-10f) : 1.0f);
    }
    phase_fence();
    const float incl = wave_scan_mul(run, lane);
    float before = __shfl_up(incl, 1);
    if (lane == 0) before = 1.0f;
    float sr = 0.f, sg = 0.f, sb = 0.f, sdepth = 0.f, sacc = 0.f;
    phase_fence();
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int i = lane * EPL + k;
        w[k] = (i < S) ? __fmul_rn(alpha[k], __fmul_rn(before, texcl[k])) : 0.0f;
        const float r = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-col[k][0])));    // sigmoid, :750
        const float g = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-col[k][1])));
        const float b = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-col[k][2])));
        sr = __fmaf_rn(w[k], r, sr); sg = __fmaf_rn(w[k], g, sg); sb = __fmaf_rn(w[k], b, sb);
        sdepth = __fmaf_rn(w[k], z[k], sdepth); sacc = __fadd_rn(sacc, w[k]);
        if (ray_ok && i < S) {
            if (a.vis) gmem(a.vis)[(size_t)ray * S + i] = w[k];
            if (a.alpha) gmem(a.alpha)[(size_t)ray * S + i] = alpha[k];
            if (a.z_user) gmem(a.z_user)[(size_t)ray * S + i] = z[k];
        }
    }
    phase_fence();
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sdepth = wave_sum(sdepth); sacc = wave_sum(sacc);
    if (ray_ok && lane == 0) {
        if (a.white_bkgd) {                                                                                   // :786-787
            const float bg = __fsub_rn(1.0f, sacc);
            sr = __fadd_rn(sr, bg); sg = __fadd_rn(sg, bg); sb = __fadd_rn(sb, bg);
        }
        gmem(a.rgb)[(size_t)ray * 3 + 0] = sr; gmem(a.rgb)[(size_t)ray * 3 + 1] = sg; gmem(a.rgb)[(size_t)ray * 3 + 2] = sb;   // :776
        gmem(a.acc)[ray] = sacc;                                                                                    // :779
        const float q = __fdiv_rn(sdepth, sacc);                                  // 0/0 = NaN when acc == 0 ...
        gmem(a.disp)[ray] = __fdiv_rn(1.0f, (q != q) ? q : fmaxf(1e-10f, q));           // ... which torch.max propagates (:781-784)
    }

    phase_fence();
    // ---- surface reduction: index of the sample whose accumulated visibility is closest to 0.5 (first one on ties),
    //      and the bent point / rigidity there (free_viewpoint_rendering.py:621-648 does this on the host from the
    //      full per-sample tensors: ~15 KB/ray of D2H traffic instead of 20 B/ray)
    if (a.bent4) {
        // cumsum in strictly sequential order (carry handed from lane to lane): zero-weight plateaus then give
        // bit-identical prefixes, hence exact ties that resolve to the first index, as with torch.cumsum on the host
        float carry = 0.f, base = 0.f;
        for (int l = 0; l < 64; ++l) {
            float e = carry;
#pragma unroll
            for (int k = 0; k < EPL; ++k) e = __fadd_rn(e, w[k]);
            if (lane == l) base = carry;
            carry = __shfl(e, l);
        }
        // NaN weights (a diverged checkpoint, an overflowed f16 sigma) compare false everywhere: start from this lane's
        // first valid sample so the index is always in range, like the host argmin the reference uses (fvr:626-628)
        float best = 3.0e38f, run_c = base;
        int bidx = (lane * EPL < S) ? lane * EPL : S - 1;
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            const int i = lane * EPL + k;
            run_c = __fadd_rn(run_c, w[k]);
            const float dist = fabsf(__fsub_rn(run_c, 0.5f));
            if (i < S && dist < best) { best = dist; bidx = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(bidx, o);
            if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
        }
        if (ray_ok && lane == 0) {
            bidx = bidx < 0 ? 0 : (bidx > S - 1 ? S - 1 : bidx);
            const f32x4 b = *(const __attribute__((address_space(1))) f32x4*)(gmem(a.bent4) + ((size_t)ray * S + bidx) * 4);
            if (a.surf_pts) { gmem(a.surf_pts)[(size_t)ray * 3] = b[0]; gmem(a.surf_pts)[(size_t)ray * 3 + 1] = b[1]; gmem(a.surf_pts)[(size_t)ray * 3 + 2] = b[2]; }
            if (a.surf_rig) gmem(a.surf_rig)[ray] = b[3];
            if (a.med_idx) gmem(a.med_idx)[ray] = bidx;
        }
    }
    phase_fence();

}

// ---- the COARSE pass of a hierarchical render, after composite_ray: inverse-CDF sampling of the importance depths (sample_pdf,
// run_nerf_helpers.py:651-698), z_std (train.py:959) and the sort-merge of coarse and importance depths (train.py:920) of ONE ray by ONE
// wavefront.  z / w: what composite_ray returned (lane l: samples l * EPL ..).  s_cdf / s_bins [64 * EPL], s_z [S + I]: LDS areas of THIS
// wave.  Shared by composite_kernel<EPL, true> (one wave per ray) and the coarse epilogue of net_kernel_x16 (a wave owns whole rays and
// goes on to the next one): the same code, hence the same bits.  WAVE_LOCAL: the areas are private to the wave, so a compiler-level fence
// orders the phases (LDS operations of one wave execute in order); false: a workgroup barrier, as composite_kernel always had it.
template <bool WAVE_LOCAL>
__device__ __forceinline__ void ray_phase_sync() {
    if constexpr (WAVE_LOCAL) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

template <int EPL, bool WAVE_LOCAL>
__device__ __forceinline__ void sample_merge_ray(const CompositeArgs& a, const int ray, const bool ray_ok, const int lane,
                                                 const float (&z)[EPL + 1], const float (&w)[EPL], float* s_cdf, float* s_bins, float* s_z) {
    const int S = a.S, I = a.n_importance;
    const int nb = S - 1;          // bins = mid-points; cdf has nb entries (run_nerf_helpers.py:657-659)
    // ---- pdf / cdf over weights[1:-1] + 1e-5 (rnh:654-659)
    float v[EPL], vs = 0.f;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int i = lane * EPL + k;
        v[k] = (i >= 1 && i <= S - 2) ? __fadd_rn(w[k], 1e-5f) : 0.0f;
        vs += v[k];
    }
    const float total = wave_sum(vs);
    float lrun = 0.f, lpre[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) { lrun += __fdiv_rn(v[k], total); lpre[k] = lrun; }      // pdf = w / sum (:655)
    const float lincl = wave_scan_add(lrun, lane);
    const float lbase = lincl - lrun;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int i = lane * EPL + k;
        if (i < nb) {
            s_cdf[i] = (i == 0) ? 0.0f : lbase + lpre[k];                            // cumsum (:656)
            s_bins[i] = __fmul_rn(0.5f, __fadd_rn(z[k + 1], z[k]));                  // train.py:910
        }
        if (i < S) s_z[i] = z[k];
    }
    ray_phase_sync<WAVE_LOCAL>();
    // ---- inverse CDF at u = linspace(0,1,I) (det=True) or at the caller's uniforms (det=False), rnh:663-696
    float zsum = 0.f;
    for (int k = lane; k < I; k += 64) {
        const float u = a.u ? gmem(a.u)[(size_t)ray * I + k] : c_lin01(k, I);             // rnh:663-665
        int lo = 0, hi = nb;       // lower_bound: first idx with cdf[idx] >= u  (searchsorted right=False)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_cdf[mid] < u) lo = mid + 1; else hi = mid; }
        const int below = lo - 1 > 0 ? lo - 1 : 0;                                         // :683
        const int above = lo < nb - 1 ? lo : nb - 1;                                       // :684
        const float c0 = s_cdf[below], c1 = s_cdf[above];
        const float b0 = s_bins[below], b1 = s_bins[above];
        float denom = __fsub_rn(c1, c0);                                                   // :693
        if (denom < 1e-5f) denom = 1.0f;                                                   // :694
        const float t = __fdiv_rn(__fsub_rn(u, c0), denom);                                // :695
        const float zs = __fadd_rn(b0, __fmul_rn(t, __fsub_rn(b1, b0)));                   // :696
        s_z[S + k] = zs;
        zsum += zs;
    }
    ray_phase_sync<WAVE_LOCAL>();
    const int n = S + I;
    // ---- z_std (population std of the importance samples), train.py:959
    const float mean = wave_sum(zsum) / (float)I;
    float var = 0.f;
    for (int k = lane; k < I; k += 64) { const float d = s_z[S + k] - mean; var += d * d; }
    var = wave_sum(var) / (float)I;
    if (ray_ok && lane == 0 && a.z_std) gmem(a.z_std)[ray] = sqrtf(var);
    // ---- merge (train.py:920): the values torch.sort would return, via the stable rank of every depth.
    // Coarse depths are strictly increasing.  If the importance samples are non-decreasing too (the inverse CDF is
    // monotone; rounding can break it by an ulp), ranks follow from one binary search into the other list:
    //   rank(coarse i) = i + #{samples <  z_i}          (stable: ties keep the concatenation order, coarse first)
    //   rank(sample k) = k + #{coarse  <= s_k}
    // Otherwise fall back to counting against all n elements (correct for any input order).
    // split-bender path: coarse sample i keeps its bent point, moved to its row among the merged depths; importance
    // sample k is listed (depth, row) for the stand-alone bender kernel
    auto split_out = [&](int idx, int rank, float depth) {
        if (!a.rank_new) return;
        if (idx < S) {
            if (a.split_bent_in)      // (training asks for the new samples' list only: nrnerf_composite_args.z_new / rank_new)
                *(__attribute__((address_space(1))) f32x4*)(gmem(a.split_bent_out) + ((size_t)ray * n + rank) * 4) =
                    *(const __attribute__((address_space(1))) f32x4*)(gmem(a.split_bent_in) + ((size_t)ray * S + idx) * 4);
        } else {
            gmem(a.rank_new)[(size_t)ray * I + (idx - S)] = (uint8_t)rank;
            gmem(a.z_new)[(size_t)ray * I + (idx - S)] = depth;
        }
    };
    bool mono = true;
    for (int k = lane; k < I; k += 64)
        if (k > 0 && s_z[S + k] < s_z[S + k - 1]) mono = false;
    mono = __all(mono);
    if (mono) {
        for (int idx = lane; idx < n; idx += 64) {
            const float mine = s_z[idx];
            int lo, hi, rank;
            if (idx < S) {      // count samples strictly below
                lo = 0; hi = I;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_z[S + mid] < mine) lo = mid + 1; else hi = mid; }
                rank = idx + lo;
            } else {            // count coarse depths <= mine
                lo = 0; hi = S;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_z[mid] <= mine) lo = mid + 1; else hi = mid; }
                rank = (idx - S) + lo;
            }
            if (ray_ok) { gmem(a.z_out)[(size_t)ray * n + rank] = mine; split_out(idx, rank, mine); }
        }
    } else {
        for (int idx = lane; idx < n; idx += 64) {
            const float mine = s_z[idx];
            int rank = 0;
            for (int jj = 0; jj < n; ++jj) {
                const float o = s_z[jj];
                rank += (o < mine || (o == mine && jj < idx)) ? 1 : 0;
            }
            if (ray_ok) { gmem(a.z_out)[(size_t)ray * n + rank] = mine; split_out(idx, rank, mine); }
        }
    }
}

}  // namespace nrn
