/* A plain-C consumer of include/nrnerf.h: proves the boundary is a C ABI (the header compiles as C99, the structures
 * mean what the ctypes mirror in nonrigid_nerf_amd/_lib.py says they mean) without Python or torch in the process.
 *
 *   gcc -std=gnu99 -D__HIP_PLATFORM_AMD__ render_from_c.c -I include -I /opt/rocm/include -L nonrigid_nerf_amd/lib -L /opt/rocm/lib \
 *       -lnrnerf_hip -lamdhip64
 *   ./render_from_c weights.bin rays.bin out.bin
 *
 * weights.bin: int32 header [n_linears] then per nn.Linear: int32 out, int32 in, int32 has_bias, fp32 weight[out*in],
 *              fp32 bias[out]; order: bender network (5), rigidity (3), coarse pts_linears (8), coarse output_linear,
 *              fine pts_linears (8), fine output_linear   -- written by tests/test_c_abi.py from the synthetic scene.
 * rays.bin:    int32 n, fp32 rays[n*8], fp32 latents[n*32].     out.bin: fp32 rgb[n*3], disp[n], acc[n].
 * Renders 64 + 64 samples in exact-fp32 mode (the reference default architecture). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "nrnerf.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "failed: %s (line %d)\n", #x, __LINE__); return 1; } } while (0)

static int read_linear(FILE* f, nrnerf_linear* l) {
    int32_t h[3];
    if (fread(h, 4, 3, f) != 3) return -1;
    float* w = (float*)malloc(sizeof(float) * (size_t)h[0] * (size_t)h[1]);
    float* b = h[2] ? (float*)malloc(sizeof(float) * (size_t)h[0]) : NULL;
    if (fread(w, 4, (size_t)h[0] * h[1], f) != (size_t)h[0] * h[1]) return -1;
    if (b && fread(b, 4, (size_t)h[0], f) != (size_t)h[0]) return -1;
    l->weight = w; l->bias = b; l->out_features = h[0]; l->in_features = h[1];
    return 0;
}

int main(int argc, char** argv) {
    CHECK(argc == 4);
    CHECK(nrnerf_abi_version() == NRNERF_ABI_VERSION);
    FILE* f = fopen(argv[1], "rb");
    CHECK(f);
    int32_t n_lin = 0;
    CHECK(fread(&n_lin, 4, 1, f) == 1 && n_lin == 5 + 3 + 9 + 9);
    nrnerf_linear lin[26];
    for (int i = 0; i < n_lin; ++i) CHECK(read_linear(f, &lin[i]) == 0);
    fclose(f);

    nrnerf_bender_desc bend;
    memset(&bend, 0, sizeof bend);
    bend.latent_size = 32; bend.depth = 5; bend.hidden = 64; bend.rigidity_depth = 3; bend.rigidity_hidden = 32;
    bend.network = &lin[0]; bend.rigidity_network = &lin[5];
    nrnerf_mlp_desc mlp[2];
    memset(mlp, 0, sizeof mlp);
    for (int k = 0; k < 2; ++k) {
        mlp[k].depth = 8; mlp[k].width = 256; mlp[k].skip = 4; mlp[k].output_ch = 5;
        mlp[k].pts_linears = &lin[8 + 9 * k];
        mlp[k].output_linear = lin[8 + 9 * k + 8];
    }
    nrnerf_model_desc desc;
    memset(&desc, 0, sizeof desc);
    desc.struct_size = (uint32_t)sizeof desc;
    desc.precision = NRNERF_PREC_F32; desc.multires = 10; desc.multires_views = 4; desc.device = 0;
    desc.bender = &bend; desc.coarse = &mlp[0]; desc.fine = &mlp[1];
    nrnerf_model* model = NULL;
    int rc = nrnerf_model_create(&desc, &model);
    if (rc) { fprintf(stderr, "nrnerf_model_create: %s\n", nrnerf_strerror(rc)); return 1; }

    f = fopen(argv[2], "rb");
    CHECK(f);
    int32_t n = 0;
    CHECK(fread(&n, 4, 1, f) == 1 && n > 0);
    float* rays = (float*)malloc(sizeof(float) * (size_t)n * 8);
    float* lat = (float*)malloc(sizeof(float) * (size_t)n * 32);
    CHECK(fread(rays, 4, (size_t)n * 8, f) == (size_t)n * 8 && fread(lat, 4, (size_t)n * 32, f) == (size_t)n * 32);
    fclose(f);

    const int S = 64, I = 64;
    float *d_rays, *d_lat, *d_out;                /* d_out: rgb[3n] disp[n] acc[n] rgb0[3n] disp0[n] acc0[n] z_std[n] */
    void* d_ws;
    const size_t ws_bytes = nrnerf_workspace_bytes(model, n, S, I);
    CHECK(hipMalloc((void**)&d_rays, sizeof(float) * (size_t)n * 8) == hipSuccess);
    CHECK(hipMalloc((void**)&d_lat, sizeof(float) * (size_t)n * 32) == hipSuccess);
    CHECK(hipMalloc((void**)&d_out, sizeof(float) * (size_t)n * 11) == hipSuccess);
    CHECK(hipMalloc(&d_ws, ws_bytes) == hipSuccess);                       /* hipMalloc is 256-byte aligned */
    CHECK(hipMemcpy(d_rays, rays, sizeof(float) * (size_t)n * 8, hipMemcpyHostToDevice) == hipSuccess);
    CHECK(hipMemcpy(d_lat, lat, sizeof(float) * (size_t)n * 32, hipMemcpyHostToDevice) == hipSuccess);

    nrnerf_render_args a;
    memset(&a, 0, sizeof a);
    a.struct_size = (uint32_t)sizeof a;
    a.n_rays = n; a.n_samples = S; a.n_importance = I;
    a.rays = d_rays; a.ray_stride = 8; a.latents = d_lat; a.latent_stride = 32;
    a.rgb_map = d_out; a.disp_map = d_out + 3 * (size_t)n; a.acc_map = d_out + 4 * (size_t)n;
    a.rgb0 = d_out + 5 * (size_t)n; a.disp0 = d_out + 8 * (size_t)n; a.acc0 = d_out + 9 * (size_t)n; a.z_std = d_out + 10 * (size_t)n;
    a.workspace = d_ws; a.workspace_bytes = ws_bytes;
    hipStream_t stream;
    CHECK(hipStreamCreate(&stream) == hipSuccess);
    rc = nrnerf_render(model, &a, stream);
    if (rc) { fprintf(stderr, "nrnerf_render: %s\n", nrnerf_strerror(rc)); return 1; }
    CHECK(hipStreamSynchronize(stream) == hipSuccess);

    /* error contract: a too-small workspace is reported, not written through */
    a.workspace_bytes = 16;
    CHECK(nrnerf_render(model, &a, stream) == NRNERF_ERR_WORKSPACE);
    a.workspace_bytes = ws_bytes;
    a.struct_size = 8;
    CHECK(nrnerf_render(model, &a, stream) == NRNERF_ERR_INVALID);

    float* out = (float*)malloc(sizeof(float) * (size_t)n * 5);
    CHECK(hipMemcpy(out, d_out, sizeof(float) * (size_t)n * 5, hipMemcpyDeviceToHost) == hipSuccess);
    f = fopen(argv[3], "wb");
    CHECK(f && fwrite(out, 4, (size_t)n * 5, f) == (size_t)n * 5);
    fclose(f);
    nrnerf_model_destroy(model);
    printf("rendered %d rays from C: rgb[0] = %.6f %.6f %.6f\n", n, out[0], out[1], out[2]);
    return 0;
}
