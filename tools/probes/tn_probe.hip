// tn_probe.hip -- tn_products_kernel (csrc/nrnerf_gen_train.hip) alone, with parts switched off at build time, to see where its time goes:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I nonrigid_nerf_amd/csrc [-DTN_DBG_NOMFMA] [-DTN_DBG_NOFRAG] [-DTN_DBG_NOLOAD] tools/probes/tn_probe.hip -o tn_probe
#include "../../nonrigid_nerf_amd/csrc/nrnerf_gen_train.hip"
#include <cstdio>
#include <vector>
using namespace nrn;
int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 192, D = 8;
    const bool f32 = argc > 2 && atoi(argv[2]) == 4;
    const int ES = f32 ? 4 : 2;
    const long long M = 393216;
    void *A, *B; float *parts, *out;
    hipMalloc(&A, (size_t)D * M * W * ES); hipMalloc(&B, (size_t)D * M * W * ES);
    hipMemset(A, 0x3c, (size_t)D * M * W * ES); hipMemset(B, 0x3c, (size_t)D * M * W * ES);
    TnKernelArgs k{};
    k.kch = 64; k.n_rows = M; k.total = (long long)D * W * W + D * W;
    for (int i = 0; i < D; ++i)
        k.sub[k.n_sub++] = TnSubJob{(char*)A + (size_t)i * M * W * ES, (char*)B + (size_t)i * M * W * ES, W, W, W, W, 0, 0, W, (long long)i * W * W, (long long)D * W * W + i * W};
    hipMalloc(&parts, (size_t)k.kch * k.total * 4); hipMalloc(&out, k.total * 4);
    k.partials = parts;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        launch_tn_products(k, f32, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("W %d %s: %.3f ms  (%.0f GB/s of arrays, %.0f TFLOP/s)\n", W, f32 ? "f32" : "bf16", ms, 2.0 * D * M * W * ES / ms / 1e6, 2.0 * M * D * W * W / ms / 1e9);
    }
    return 0;
}
