// Can a single wave keep the matrix pipe busy while it also issues VALU work?  (MI355X, gfx950)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/coissue_probe.hip -o /tmp/coissue_probe && /tmp/coissue_probe
// One workgroup of 4 waves (one per SIMD) or 8 waves (two per SIMD) runs a loop of 4 independent
// v_mfma_f32_32x32x16_bf16; K instructions of one kind follow every MFMA.  Reports s_memtime ticks of SIMD time per MFMA.
// The tick is NOT a shader cycle here (tick_rate(): the counter ran at 1447 MHz while 20 ticks = 13.8 ns = one MFMA at
// full rate), so read the table as ratios: 20 = the matrix pipe saturated.
// Measured: two waves per SIMD hide up to 4 VALU instructions (or 2 LDS reads + waits) per MFMA completely (20.0 ticks),
// 6-8 cost ~50 %; a lone wave reaches 57 % of the pipe rate with bare MFMAs (35 ticks), 50 % with 4 VALU in between (40),
// 36 % with 8 (55).  Whole chip, bare loop with constant operands: 2.38-2.40 PFLOP/s by wall clock (95 % of peak).
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

template <int MODE, int K, int THREADS>
__global__ void __launch_bounds__(THREADS, 1) probe(unsigned long long* out, int iters) {
    __shared__ u32x4 lds[512];
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
    double c[2];
    for (int two = 0; two < 2; ++two) {          // one wave per SIMD, then two waves per SIMD running the same loop
        if (two) { hipLaunchKernelGGL((probe<MODE, K, 512>), dim3(1), dim3(512), 0, 0, d, iters); hipLaunchKernelGGL((probe<MODE, K, 512>), dim3(1), dim3(512), 0, 0, d, iters); }
        else { hipLaunchKernelGGL((probe<MODE, K, 256>), dim3(1), dim3(256), 0, 0, d, iters); hipLaunchKernelGGL((probe<MODE, K, 256>), dim3(1), dim3(256), 0, 0, d, iters); }
        hipDeviceSynchronize();
        unsigned long long h = 0;
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        c[two] = (double)h / (4.0 * iters * (two ? 2 : 1));
    }
    printf("%-20s K=%d per MFMA: %6.1f ticks of SIMD time per MFMA with 1 wave/SIMD, %6.1f with 2 waves/SIMD\n", name, K, c[0], c[1]);
}
// whole-chip rate of the bare MFMA loop by wall clock (hipEvents): 1024 workgroups of 8 waves
static void chip_rate(unsigned long long* d) {
    const int iters = 20000, wgs = 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<0, 0, 512>), dim3(wgs), dim3(512), 0, 0, d, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<0, 0, 512>), dim3(wgs), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)wgs * 8 * iters * 4 * 32768.0;
    unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("whole chip, bare MFMA loop, 2 waves/SIMD: %.1f TFLOP/s by wall clock (%.2f ms); s_memtime says %.1f ticks per MFMA of SIMD time\n",
           flop / (ms * 1e-3) / 1e12, ms, (double)h / (4.0 * iters * 2));
}
// frequency of the s_memtime counter: one workgroup, long loop, ticks vs hipEvent wall time
static void tick_rate(unsigned long long* d) {
    const int iters = 400000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<0, 0, 512>), dim3(1), dim3(512), 0, 0, d, 1000);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<0, 0, 512>), dim3(1), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("one workgroup, 2 waves/SIMD, %d iterations: %.3f ms wall, %llu s_memtime ticks -> counter runs at %.1f MHz; %.1f ns per MFMA of SIMD time\n",
           iters, ms, h, (double)h / (ms * 1e3), ms * 1e6 / (4.0 * iters * 2));
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    tick_rate(d);
    chip_rate(d);
    run<0, 0>(d, "mfma only");
    run<0, 2>(d, "v_pk_max_i16"); run<0, 4>(d, "v_pk_max_i16"); run<0, 6>(d, "v_pk_max_i16"); run<0, 8>(d, "v_pk_max_i16");
    run<1, 2>(d, "v_cvt_pk_bf16_f32"); run<1, 4>(d, "v_cvt_pk_bf16_f32"); run<1, 6>(d, "v_cvt_pk_bf16_f32"); run<1, 8>(d, "v_cvt_pk_bf16_f32");
    run<2, 2>(d, "v_accvgpr_read"); run<2, 4>(d, "v_accvgpr_read"); run<2, 6>(d, "v_accvgpr_read"); run<2, 8>(d, "v_accvgpr_read");
    run<4, 2>(d, "1 ds_read+1 wait"); run<4, 4>(d, "2 ds_read+2 wait"); run<4, 6>(d, "1 ds_read"); run<4, 8>(d, "2 ds_read");
    run<5, 2>(d, "1 s_waitcnt"); run<5, 4>(d, "2 s_waitcnt"); run<5, 6>(d, "1 s_nop 0"); run<5, 8>(d, "2 s_nop 0");
    run<3, 2>(d, "v_fma_f32"); run<3, 4>(d, "v_fma_f32"); run<3, 6>(d, "v_fma_f32"); run<3, 8>(d, "v_fma_f32");
    return 0;
}
