// Which MFMA shape sustains more bf16 flops under the socket's power cap?  v_mfma_f32_32x32x16_bf16 (what net_kernel uses)
// against v_mfma_f32_16x16x32_bf16: same flop rate on paper (1024 flop per cycle and SIMD), same operand registers per flop for
// A and B taken together, but a quarter of the accumulator registers touched twice as often.  Operands live in registers (no
// LDS, no memory): what differs between the variants is the instruction and its register-file traffic.  Operand statistics as
// in the trunk: A ~ N(0, 0.1) weights, B = relu'd activations (half of them zero).  Four independent accumulator chains each.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_shape_power.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, WAVES / 4) probe(const bf16x8* __restrict__ g, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[16], b[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        a[s] = g[(size_t)(s * 64 + lane)];
        b[s] = g[(size_t)((16 + s) * 64 + lane)];
    }
    float sum = 0.0f;
    if constexpr (SHAPE == 0) {
        f32x16 acc[4] = {f32x16{}, f32x16{}, f32x16{}, f32x16{}};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + c) & 15], b[s], acc[c], 0, 0, 0);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[c][r];
    } else {
        f32x4 acc[8] = {f32x4{}, f32x4{}, f32x4{}, f32x4{}, f32x4{}, f32x4{}, f32x4{}, f32x4{}};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(s + c) & 15], b[s], acc[c], 0, 0, 0);
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) sum += acc[c][r];
    }
    if (sum == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = sum;      // keep the chains alive
}

template <int SHAPE, int WAVES>
double run(const bf16x8* g, float* out, int cus, double seconds) {
    // flops per iteration and wave: 64 MFMAs x 32768 (SHAPE 0) / 128 MFMAs x 16384 (SHAPE 1) = 2.1 MFLOP either way
    const double flop_per_iter = 2097152.0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 2000;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {          // calibrate, then the timed run
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<SHAPE, WAVES>), dim3(cus), dim3(WAVES * 64), 0, 0, g, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 0) iters = (int)(iters * seconds * 1e3 / ms);
    }
    return flop_per_iter * iters * WAVES * cus / (ms * 1e-3) / 1e12;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    std::vector<unsigned short> h(32 * 64 * 8);
    srand(1);
    auto bf16 = [](float v) { unsigned u; std::memcpy(&u, &v, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); };
    auto gauss = [] { float s = 0; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; return s - 6.0f; };
    for (size_t i = 0; i < h.size(); ++i) {
        const bool is_a = i < h.size() / 2;
        const float v = is_a ? 0.1f * gauss() : fmaxf(0.0f, gauss());
        h[i] = bf16(v);
    }
    bf16x8* g; float* out;
    hipMalloc(&g, h.size() * 2); hipMalloc(&out, (size_t)cus * 512 * 4);
    hipMemcpy(g, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("# %d CUs; TFLOP/s sustained over ~2 s per run (dense bf16 peak on paper: 2500)\n", cus);
    for (int round = 0; round < 2; ++round) {
        printf("32x32x16, 1 wave per SIMD: %8.1f\n", run<0, 4>(g, out, cus, 2.0));
        printf("16x16x32, 1 wave per SIMD: %8.1f\n", run<1, 4>(g, out, cus, 2.0));
        printf("32x32x16, 2 waves per SIMD: %8.1f\n", run<0, 8>(g, out, cus, 2.0));
        printf("16x16x32, 2 waves per SIMD: %8.1f\n", run<1, 8>(g, out, cus, 2.0));
    }
    return 0;
}
