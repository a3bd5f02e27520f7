// tr_b16_probe.hip -- what ds_read_b64_tr_b16 returns: LDS halfword i holds the value i; lane l supplies byte address 8 * l
// (chunk l of 4 halfwords); prints the four halfwords every lane receives.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/tr_b16_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned short* out, int stride_bytes) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // chunk of lane l: row (l & 15) / 4 .. : address = row * stride + 8 * (l & 3) + group offset
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds + (unsigned)(((l & 15) >> 2) * stride_bytes + 8 * (l & 3) + (l >> 4) * 4 * stride_bytes);
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16; out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}
int main() {
    unsigned short* d; unsigned short h[256];
    hipMalloc(&d, sizeof(h));
    for (int stride : {32, 96}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("row stride %d bytes (%d halfwords): lane -> 4 halfword indices received (index = row * %d + col)\n", stride, stride / 2, stride / 2);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" (r%d,c%2d)", h[l * 4 + j] / (stride / 2), h[l * 4 + j] % (stride / 2));
            printf("\n");
        }
    }
    return 0;
}
