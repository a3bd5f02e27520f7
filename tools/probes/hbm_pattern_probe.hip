// hbm_pattern_probe.hip -- what HBM bandwidth do the access patterns of the training kernels get on MI355X?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/hbm_pattern_probe.hip -o /tmp/hbm_probe && /tmp/hbm_probe
// Data: blocks of 16 KiB = [256 features][32 samples] bf16 (64-byte rows), as nrnerf_train.h stores them in bf16 mode.
// Reads, 4 waves per workgroup, one workgroup per CU, every wave 8 x 16-byte-per-lane loads per block, two blocks in flight:
//   rows : lane (h, i) of load (tr, ks) reads 16 bytes at row 32 tr + i, byte 32 ks + 16 h   (the shipped trunk_wgrad: every
//          instruction touches 32 rows, half of each)
//   frag : lane l of load f reads 16 bytes at f * 1024 + l * 16                               (a fragment = 1 KiB contiguous)
// Writes, 8 waves per workgroup, every wave writes whole blocks:
//   rows : sixteen 2-byte stores per 32 x 32 tile, each instruction two whole 64-byte rows    (the shipped forward / backward)
//   frag : sixteen 2-byte stores per tile into fragment order (eight lanes = 16 contiguous bytes per instruction piece)
//   vec  : two 16-byte-per-lane stores per tile (1 KiB contiguous per instruction)             (an upper bound)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool FRAG>
__global__ void __launch_bounds__(256, 1) read_kernel(const char* buf, long long nblocks, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i = lane & 31;
    u32x4 acc = {0, 0, 0, 0};
    auto load = [&](long long blk, u32x4 (&v)[8]) {
        const char* b = buf + blk * 16384;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int f = wave * 8 + q;     // fragment (tr, ks) = (f >> 1, f & 1); the four waves cover a quarter of the block each... twice
            const char* p = FRAG ? b + (f & 15) * 1024 + lane * 16 : b + ((32 * ((f & 15) >> 1) + i) * 64 + 32 * (f & 1) + 16 * h);
            v[q] = *(const u32x4*)p;
        }
    };
    u32x4 v0[8], v1[8];
    long long blk = blockIdx.x;
    if (blk < nblocks) load(blk, v0);
    while (blk < nblocks) {
        const long long n1 = blk + gridDim.x;
        if (n1 < nblocks) load(n1, v1);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc ^= v0[q];
        if (n1 >= nblocks) break;
        const long long n2 = n1 + gridDim.x;
        if (n2 < nblocks) load(n2, v0);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc ^= v1[q];
        blk = n2;
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

template <int MODE>
__global__ void __launch_bounds__(512, 1) write_kernel(char* buf, long long nblocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, j = lane & 31;
    for (long long blk = (long long)blockIdx.x * 8 + wave; blk < nblocks; blk += (long long)gridDim.x * 8) {
        char* b = buf + blk * 16384;
        for (int t = 0; t < 8; ++t) {                         // 8 tiles of 32 features
            if (MODE == 2) {
                const u32x4 v = {(unsigned)blk, (unsigned)t, (unsigned)lane, 7u};
                *(u32x4*)(b + (2 * t) * 1024 + lane * 16) = v;
                *(u32x4*)(b + (2 * t + 1) * 1024 + lane * 16) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;      // feature of register r
                    unsigned short val = (unsigned short)(blk + r);
                    if (MODE == 0) *(unsigned short*)(b + f * 64 + j * 2) = val;            // [feature][32 samples]
                    else {                                                                   // fragment order of trunk_wgrad's operand
                        const int tr = f >> 5, i = f & 31, ks = j >> 4, hh = (j >> 3) & 1, e = j & 7;
                        *(unsigned short*)(b + (2 * tr + ks) * 1024 + (hh * 32 + i) * 16 + e * 2) = val;
                    }
                }
            }
        }
    }
}

int main() {
    const long long bytes = 8ll << 30, nblocks = bytes / 16384;
    char* buf; unsigned* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4); hipMemset(buf, 1, bytes);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto timeit = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
        printf("%-28s %8.3f ms  %6.2f TB/s\n", name, best, bytes / best / 1e9);
    };
    timeit("read  rows (shipped)", [&] { hipLaunchKernelGGL(read_kernel<false>, dim3(cus), dim3(256), 0, 0, buf, nblocks, out); });
    timeit("read  frag (1 KiB/instr)", [&] { hipLaunchKernelGGL(read_kernel<true>, dim3(cus), dim3(256), 0, 0, buf, nblocks, out); });
    timeit("write rows (shipped)", [&] { hipLaunchKernelGGL(write_kernel<0>, dim3(cus), dim3(512), 0, 0, buf, nblocks); });
    timeit("write frag order, 2 B", [&] { hipLaunchKernelGGL(write_kernel<1>, dim3(cus), dim3(512), 0, 0, buf, nblocks); });
    timeit("write 16 B per lane", [&] { hipLaunchKernelGGL(write_kernel<2>, dim3(cus), dim3(512), 0, 0, buf, nblocks); });
    return 0;
}
