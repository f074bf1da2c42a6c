// Ablation timing of the chained FFN BACKWARD (ffn_chain_kernel<..., BWD>) and of ln_colsum_kernel at the real shape (development tool).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I speaker_embedding_torch_amd/csrc tools/ffnb_bench.hip -o tools/ffnb_bench
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include "ffn.cuh"
#include "misc.cuh"
using namespace ge2e;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void fill_bf16(bf16_t* p, size_t n, unsigned seed, float amp) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (bf16_t)((((mix32((unsigned)i * 2654435761u + seed) >> 8) * (1.0f / 8388608.0f)) - 1.0f) * amp);
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float base) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = base + (((mix32((unsigned)i * 2654435761u + seed) >> 8) * (1.0f / 8388608.0f)) - 1.0f) * 0.1f;
}
__global__ void fill_u8(unsigned char* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned char)mix32((unsigned)i + seed);
}
template <typename K> float time_kernel(K launch, int iters = 20) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 10; ++i) launch();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); CHECK(hipGetLastError());
    return ms / iters;
}
template <int ABL, int WV> void run(const char* tag, FfnArgs a, int grid_cap) {
    auto kern = ffn_chain_kernel<bf16_t, true, ABL, true, WV, true>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ffn_bwd_smem<WV>()));
    const int npass = (a.M + 32 * WV - 1) / (32 * WV), grid = std::min(grid_cap, npass);
    float ms = time_kernel([&]() { hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WV), ffn_bwd_smem<WV>(), 0, a, npass); });
    printf("%-44s wv=%d abl=%2d grid=%3d  %8.1f us  %7.1f TF/s\n", tag, WV, ABL, grid, ms * 1e3, 4.0 * a.M * 256 * 1024 / ms / 1e9);
}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 153600;
    bf16_t *dH, *Y, *W1, *W2, *F, *C, *dP, *dM; float *ga, *be, *rstd, *dg, *db; unsigned char* Mb;
    CHECK(hipMalloc(&dH, (size_t)M * 256 * 2)); CHECK(hipMalloc(&Y, (size_t)M * 256 * 2)); CHECK(hipMalloc(&W1, 1024 * 256 * 2)); CHECK(hipMalloc(&W2, 1024 * 256 * 2));
    CHECK(hipMalloc(&F, (size_t)M * 1024 * 2)); CHECK(hipMalloc(&Mb, (size_t)M * 128)); CHECK(hipMalloc(&C, (size_t)M * 256 * 2));
    CHECK(hipMalloc(&dP, (size_t)M * 256 * 2)); CHECK(hipMalloc(&dM, (size_t)M * 256 * 2));
    CHECK(hipMalloc(&ga, 1024)); CHECK(hipMalloc(&be, 1024)); CHECK(hipMalloc(&rstd, (size_t)M * 4)); CHECK(hipMalloc(&dg, 1024)); CHECK(hipMalloc(&db, 1024));
    fill_bf16<<<2048, 256>>>(dH, (size_t)M * 256, 1, 1.0f); fill_bf16<<<2048, 256>>>(Y, (size_t)M * 256, 9, 1.0f);
    fill_bf16<<<64, 256>>>(W1, 1024 * 256, 2, 0.1f); fill_bf16<<<64, 256>>>(W2, 1024 * 256, 3, 0.05f);
    fill_f32<<<1, 256>>>(ga, 256, 6, 1.0f); fill_f32<<<1, 256>>>(be, 256, 7, 0.0f); fill_f32<<<2048, 256>>>(rstd, M, 8, 1.0f);
    fill_u8<<<2048, 256>>>(Mb, (size_t)M * 128, 11);
    CHECK(hipDeviceSynchronize());
    FfnArgs a{};
    a.A = dH; a.lda = 256; a.Y = Y; a.W1 = W1; a.W2 = W2; a.Fo = F; a.Mb = Mb; a.ldf = 1024; a.C = C; a.ldc = 256;
    a.gamma = ga; a.beta = be; a.rstd = rstd; a.eps = 1e-5f; a.M = M; a.drow_mul = 1; a.dM = dM;
    a.drop1 = Drop{12345u, 6553u, 1.1111f}; a.drop2 = Drop{54321u, 6553u, 1.1111f};
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 4>("bwd 4-wave, 512 blocks", a, 512);
        run<0, 4>("bwd 4-wave, 400 blocks", a, 400);
    }
    run<8, 4>("bwd 4-wave no barrier", a, 512);
    run<64, 4>("bwd 4-wave no pass epilogue", a, 512);
    run<1, 4>("bwd 4-wave no MFMA", a, 512);
    run<2, 4>("bwd 4-wave no DMA", a, 512);
    run<16, 4>("bwd 4-wave no mask math", a, 512);
    run<32, 4>("bwd 4-wave no stage loop (prologue + epilogue)", a, 512);
    LnBwdArgs l{}; l.dy = dH; l.y = Y; l.gamma = ga; l.beta = be; l.dgamma = dg; l.dbeta = db; l.R = M;
    for (int grid : {128, 256, 512, 1024}) {
        float ms = time_kernel([&]() { hipLaunchKernelGGL(ln_colsum_kernel<bf16_t>, dim3(grid), dim3(256), 0, 0, l); });
        printf("ln_colsum grid %4d  %7.1f us  %6.2f TB/s\n", grid, ms * 1e3, 2.0 * M * 512 / ms / 1e9);
    }
    return 0;
}
