// Micro-benchmark: MFMA issue efficiency of candidate inner-loop shapes of net_kernel on MI355X.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// Every variant streams A fragments from LDS with ds_read_b128 (1 KiB per wave per fragment, lane-linear = conflict
// free), keeps B operands (activations) in registers, runs 16 MFMAs (K = 256) per 32-feature output tile, 8 tiles per
// "layer", and packs relu(acc) to bf16 as the next layer's B operand -- the steady state of the trunk, minus the ring.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N - I == 1) f(std::integral_constant<int, I>{});
    else if constexpr (N - I > 1) { constexpr int M = I + (N - I) / 2; static_for<I, M>(f); static_for<M, N>(f); }
}
template <int U> __device__ __forceinline__ bf16x8 pack(const f32x16& c) {
    u32x4 w;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x2 t = {c[8 * U + 2 * k], c[8 * U + 2 * k + 1]};
        s16x2 q = __builtin_bit_cast(s16x2, __builtin_convertvector(t, bf16x2));
        q = __builtin_elementwise_max(q, (s16x2)(short)0);
        w[k] = __builtin_bit_cast(unsigned, q);
    }
    return __builtin_bit_cast(bf16x8, w);
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// VARIANT 0: 8 waves, 1 block/wave, one chain per tile, fragment read next to its MFMA (compiler schedules)
// VARIANT 1: 8 waves, 1 block/wave, PF-deep prefetch, two accumulator sets, delayed epilogue (the shipped kernel)
// VARIANT 2: 8 waves, 1 block/wave, two interleaved chains (tiles t, t+1), PF-deep prefetch
// VARIANT 3: 8 waves, MFMA + epilogue only, A operand from registers (no LDS): ceiling of the epilogue-carrying loop
// VARIANT 4: 4 waves (1 per SIMD, 512 regs), 2 blocks/wave sharing every A fragment, PF-deep prefetch
template <int VARIANT, int WAVES, int NB, int PF>
__global__ void __launch_bounds__(WAVES * 64, WAVES / 4) probe(const bf16x8* __restrict__ g, float* out, int layers) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 64 KiB: 64 fragments
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += WAVES * 64) ((bf16x8*)smem)[i] = g[i];
    __syncthreads();
    bf16x8 ha[NB][16], hb[NB][16];
    static_for<0, NB>([&](auto bc) { static_for<0, 16>([&](auto sc) {
        ha[decltype(bc)::value][decltype(sc)::value] = g[lane + (decltype(sc)::value + 16 * decltype(bc)::value) * 64]; }); });
    const char* base = smem + lane * 16;
    auto frag = [&](auto fc) { return *(const bf16x8*)(base + (decltype(fc)::value & 63) * 1024); };
    for (int l = 0; l < layers; ++l) {
        if constexpr (VARIANT == 0 || VARIANT == 3) {
            static_for<0, 8>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                f32x16 acc = {};
                static_for<0, 16>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    bf16x8 a;
                    if constexpr (VARIANT == 3) a = ha[0][(s + 5) & 15]; else a = frag(std::integral_constant<int, t * 16 + s>{});
                    acc = MFMA(a, ha[0][s], acc);
                });
                hb[0][2 * t] = pack<0>(acc); hb[0][2 * t + 1] = pack<1>(acc);
            });
        } else if constexpr (VARIANT == 1) {
            bf16x8 a[PF];
            static_for<0, PF>([&](auto ic) { a[decltype(ic)::value] = frag(ic); });
            f32x16 accs[2] = {f32x16{}, f32x16{}};
            static_for<0, 128>([&](auto qc) {
                constexpr int q = decltype(qc)::value, t = q >> 4, s = q & 15;
                const bf16x8 cur = a[q % PF];
                if constexpr (q + PF < 128) a[q % PF] = frag(std::integral_constant<int, q + PF>{});
                accs[t & 1] = MFMA(cur, ha[0][s], accs[t & 1]);
                if constexpr (t > 0 && s == 4) { hb[0][2 * (t - 1)] = pack<0>(accs[(t - 1) & 1]); hb[0][2 * (t - 1) + 1] = pack<1>(accs[(t - 1) & 1]); accs[(t - 1) & 1] = f32x16{}; }
                if constexpr (q == 127) { hb[0][14] = pack<0>(accs[1]); hb[0][15] = pack<1>(accs[1]); }
            });
        } else if constexpr (VARIANT == 2) {
            bf16x8 a[PF];
            static_for<0, PF>([&](auto ic) { a[decltype(ic)::value] = frag(ic); });
            static_for<0, 4>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                f32x16 acc0 = {}, acc1 = {};
                static_for<0, 32>([&](auto kc) {
                    constexpr int k = decltype(kc)::value, q = p * 32 + k, s = k >> 1;
                    const bf16x8 cur = a[q % PF];
                    if constexpr (q + PF < 128) a[q % PF] = frag(std::integral_constant<int, q + PF>{});
                    if constexpr (k & 1) acc1 = MFMA(cur, ha[0][s], acc1); else acc0 = MFMA(cur, ha[0][s], acc0);
                });
                hb[0][4 * p] = pack<0>(acc0); hb[0][4 * p + 1] = pack<1>(acc0);
                hb[0][4 * p + 2] = pack<0>(acc1); hb[0][4 * p + 3] = pack<1>(acc1);
            });
        } else if constexpr (VARIANT == 4) {
            bf16x8 a[PF];
            static_for<0, PF>([&](auto ic) { a[decltype(ic)::value] = frag(ic); });
            f32x16 accs[2][NB];
            static_for<0, NB>([&](auto bc) { accs[0][decltype(bc)::value] = f32x16{}; accs[1][decltype(bc)::value] = f32x16{}; });
            static_for<0, 128>([&](auto qc) {
                constexpr int q = decltype(qc)::value, t = q >> 4, s = q & 15;
                const bf16x8 cur = a[q % PF];
                if constexpr (q + PF < 128) a[q % PF] = frag(std::integral_constant<int, q + PF>{});
                static_for<0, NB>([&](auto bc) { constexpr int b = decltype(bc)::value; accs[t & 1][b] = MFMA(cur, ha[b][s], accs[t & 1][b]); });
                if constexpr (t > 0 && s == 4) static_for<0, NB>([&](auto bc) { constexpr int b = decltype(bc)::value;
                    hb[b][2 * (t - 1)] = pack<0>(accs[(t - 1) & 1][b]); hb[b][2 * (t - 1) + 1] = pack<1>(accs[(t - 1) & 1][b]); accs[(t - 1) & 1][b] = f32x16{}; });
                if constexpr (q == 127) static_for<0, NB>([&](auto bc) { constexpr int b = decltype(bc)::value;
                    hb[b][14] = pack<0>(accs[1][b]); hb[b][15] = pack<1>(accs[1][b]); });
            });
        }
        static_for<0, NB>([&](auto bc) { static_for<0, 16>([&](auto sc) {
            ha[decltype(bc)::value][decltype(sc)::value] = hb[decltype(bc)::value][decltype(sc)::value]; }); });
    }
    float r = 0;
    static_for<0, NB>([&](auto bc) { static_for<0, 16>([&](auto sc) { r += (float)ha[decltype(bc)::value][decltype(sc)::value][0]; }); });
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = r;
}

template <int V, int WAVES, int NB, int PF> void run(const char* name, const bf16x8* g, float* out) {
    const int layers = 4000;
    auto k = probe<V, WAVES, NB, PF>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(256), dim3(WAVES * 64), 65536, 0, g, out, layers);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    const double flops = 256.0 * WAVES * NB * layers * 128 * 32768.0;
    printf("%-62s %8.2f ms %8.1f TFLOP/s (%.1f %% of 2500)  err=%s\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0,
           hipGetErrorString(hipGetLastError()));
}
int main() {
    std::vector<unsigned short> h(4096 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + ((i * 2654435761u) >> 20 & 0x1ff)) ^ ((i * 40503u >> 7 & 1) ? 0x8000 : 0);
    bf16x8* g; float* out;
    hipMalloc(&g, h.size() * 2); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(g, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<3, 8, 1, 4>("V3 8 waves: MFMA + epilogue, A from registers (no LDS)", g, out);
    run<0, 8, 1, 4>("V0 8 waves: 1 chain, compiler-placed LDS reads", g, out);
    run<1, 8, 1, 4>("V1 8 waves: 1 chain, PF=4, 2 acc sets (shipped shape)", g, out);
    run<1, 8, 1, 8>("V1 8 waves: 1 chain, PF=8", g, out);
    run<2, 8, 1, 4>("V2 8 waves: 2 interleaved chains, PF=4", g, out);
    run<4, 4, 2, 4>("V4 4 waves x 2 blocks sharing A, PF=4", g, out);
    run<4, 4, 2, 8>("V4 4 waves x 2 blocks sharing A, PF=8", g, out);
    run<4, 4, 3, 8>("V4 4 waves x 3 blocks sharing A, PF=8", g, out);
    return 0;
}
