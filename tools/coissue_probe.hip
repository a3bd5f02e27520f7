// Can a single wave keep the matrix pipe busy while it also issues VALU work?  (MI355X, gfx950)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/coissue_probe.hip -o /tmp/coissue_probe && /tmp/coissue_probe
// One workgroup of 4 waves (one per SIMD) runs a loop of 4 independent v_mfma_f32_32x32x16_bf16 (8 passes = 32 cycles
// each); K VALU instructions of one kind follow every MFMA.  Reports cycles per MFMA (s_memtime) for K = 0..8.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define REP1(x) x
#define REP2(x) x x
#define REP4(x) x x x x
#define REP6(x) x x x x x x
#define REP8(x) x x x x x x x x
#define BODY(VALU)                                                                            \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n" VALU                            \
                 "v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n" VALU                            \
                 "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n" VALU                            \
                 "v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n" VALU                            \
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a), "+v"(b), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "a"(g0));

template <int MODE, int K>
__global__ void __launch_bounds__(256, 1) probe(unsigned long long* out, int iters) {
    __shared__ u32x4 lds[256];
    lds[threadIdx.x] = u32x4{1, 2, 3, 4};
    __syncthreads();
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    u32x4 a = {threadIdx.x, 1, 2, 3}, b = {4, 5, 6, threadIdx.x};
    unsigned x0 = (threadIdx.x & 63) * 16, x1 = 1, x2 = 2, x3 = 3;
    float g0 = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (K == 0) { BODY("") }
        else if constexpr (MODE == 0) {      // v_pk_max_i16 (1-pass integer)
            if constexpr (K == 2) { BODY(REP2("v_pk_max_i16 %6, %6, 0\n")) }
            if constexpr (K == 4) { BODY(REP4("v_pk_max_i16 %6, %6, 0\n")) }
            if constexpr (K == 6) { BODY(REP6("v_pk_max_i16 %6, %6, 0\n")) }
            if constexpr (K == 8) { BODY(REP8("v_pk_max_i16 %6, %6, 0\n")) }
        } else if constexpr (MODE == 1) {    // v_cvt_pk_bf16_f32
            if constexpr (K == 2) { BODY(REP2("v_cvt_pk_bf16_f32 %6, %7, %8\n")) }
            if constexpr (K == 4) { BODY(REP4("v_cvt_pk_bf16_f32 %6, %7, %8\n")) }
            if constexpr (K == 6) { BODY(REP6("v_cvt_pk_bf16_f32 %6, %7, %8\n")) }
            if constexpr (K == 8) { BODY(REP8("v_cvt_pk_bf16_f32 %6, %7, %8\n")) }
        } else if constexpr (MODE == 2) {    // v_accvgpr_read_b32
            if constexpr (K == 2) { BODY(REP2("v_accvgpr_read_b32 %6, %10\n")) }
            if constexpr (K == 4) { BODY(REP4("v_accvgpr_read_b32 %6, %10\n")) }
            if constexpr (K == 6) { BODY(REP6("v_accvgpr_read_b32 %6, %10\n")) }
            if constexpr (K == 8) { BODY(REP8("v_accvgpr_read_b32 %6, %10\n")) }
        } else if constexpr (MODE == 4) {    // ds_read_b128 (+ the counted wait that goes with it when K is even: K/2 reads, K/2 waits)
            if constexpr (K == 2) { BODY("ds_read_b128 %4, %6 offset:1024\ns_waitcnt lgkmcnt(3)\n") }
            if constexpr (K == 4) { BODY(REP2("ds_read_b128 %4, %6 offset:1024\ns_waitcnt lgkmcnt(3)\n")) }
            if constexpr (K == 6) { BODY("ds_read_b128 %4, %6 offset:1024\n") }
            if constexpr (K == 8) { BODY(REP2("ds_read_b128 %4, %6 offset:1024\n")) }
        } else if constexpr (MODE == 5) {    // s_waitcnt (already satisfied)
            if constexpr (K == 2) { BODY(REP1("s_waitcnt lgkmcnt(3)\n")) }
            if constexpr (K == 4) { BODY(REP2("s_waitcnt lgkmcnt(3)\n")) }
            if constexpr (K == 6) { BODY(REP1("s_nop 0\n")) }
            if constexpr (K == 8) { BODY(REP2("s_nop 0\n")) }
        } else {                              // v_fma_f32
            if constexpr (K == 2) { BODY(REP2("v_fma_f32 %6, %7, %8, %9\n")) }
            if constexpr (K == 4) { BODY(REP4("v_fma_f32 %6, %7, %8, %9\n")) }
            if constexpr (K == 6) { BODY(REP6("v_fma_f32 %6, %7, %8, %9\n")) }
            if constexpr (K == 8) { BODY(REP8("v_fma_f32 %6, %7, %8, %9\n")) }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (c0[0] + c1[1] + c2[2] + c3[3] + (float)(x0 + x1 + x2 + x3) == 123.456f) out[1] = 1;
}

template <int MODE, int K> void run(unsigned long long* d, const char* name) {
    const int iters = 20000;
    hipLaunchKernelGGL((probe<MODE, K>), dim3(1), dim3(256), 0, 0, d, iters);
    hipLaunchKernelGGL((probe<MODE, K>), dim3(1), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-20s K=%d VALU per MFMA: %6.1f cycles per MFMA\n", name, K, (double)h / (4.0 * iters));
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    run<0, 0>(d, "mfma only");
    run<0, 2>(d, "v_pk_max_i16"); run<0, 4>(d, "v_pk_max_i16"); run<0, 6>(d, "v_pk_max_i16"); run<0, 8>(d, "v_pk_max_i16");
    run<1, 2>(d, "v_cvt_pk_bf16_f32"); run<1, 4>(d, "v_cvt_pk_bf16_f32"); run<1, 6>(d, "v_cvt_pk_bf16_f32"); run<1, 8>(d, "v_cvt_pk_bf16_f32");
    run<2, 2>(d, "v_accvgpr_read"); run<2, 4>(d, "v_accvgpr_read"); run<2, 6>(d, "v_accvgpr_read"); run<2, 8>(d, "v_accvgpr_read");
    run<4, 2>(d, "1 ds_read+1 wait"); run<4, 4>(d, "2 ds_read+2 wait"); run<4, 6>(d, "1 ds_read"); run<4, 8>(d, "2 ds_read");
    run<5, 2>(d, "1 s_waitcnt"); run<5, 4>(d, "2 s_waitcnt"); run<5, 6>(d, "1 s_nop 0"); run<5, 8>(d, "2 s_nop 0");
    run<3, 2>(d, "v_fma_f32"); run<3, 4>(d, "v_fma_f32"); run<3, 6>(d, "v_fma_f32"); run<3, 8>(d, "v_fma_f32");
    return 0;
}
