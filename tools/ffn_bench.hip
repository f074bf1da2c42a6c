// Ablation timing of ffn_chain_kernel at the real shape (development tool; results in DESIGN.md).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I speaker_embedding_torch_amd/csrc tools/ffn_bench.hip -o tools/ffn_bench
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include "ffn.cuh"
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
template <typename K> float time_kernel(K launch, int iters = 20) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 10; ++i) launch();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); CHECK(hipGetLastError());
    return ms / iters;
}
template <bool STORE, int ABL, int WV = 8> void run(const char* tag, FfnArgs a, int grid_cap) {
    auto kern = ffn_chain_kernel<bf16_t, STORE, ABL, STORE && WV == 8, WV>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ffn_smem<WV>()));
    const int npass = (a.M + 32 * WV - 1) / (32 * WV), grid = std::min(grid_cap, npass);
    if (!STORE) a.Fo = nullptr;
    float ms = time_kernel([&]() { hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WV), ffn_smem<WV>(), 0, a, npass); });
    printf("%-40s wv=%d store_f=%d abl=%2d grid=%3d  %8.1f us  %7.1f TF/s\n", tag, WV, (int)STORE, ABL, grid, ms * 1e3, 4.0 * a.M * 256 * 1024 / ms / 1e9);
}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 153600;
    bf16_t *A, *W1, *W2, *F, *C; float *b1, *b2, *ga, *be, *rstd; unsigned char* Mb;
    CHECK(hipMalloc(&A, (size_t)M * 256 * 2)); CHECK(hipMalloc(&W1, 1024 * 256 * 2)); CHECK(hipMalloc(&W2, 1024 * 256 * 2));
    CHECK(hipMalloc(&F, (size_t)M * 1024 * 2)); CHECK(hipMalloc(&Mb, (size_t)M * 128)); CHECK(hipMalloc(&C, (size_t)M * 256 * 2));
    CHECK(hipMalloc(&b1, 4096)); CHECK(hipMalloc(&b2, 1024)); CHECK(hipMalloc(&ga, 1024)); CHECK(hipMalloc(&be, 1024)); CHECK(hipMalloc(&rstd, (size_t)M * 4));
    fill_bf16<<<2048, 256>>>(A, (size_t)M * 256, 1, 1.0f); fill_bf16<<<64, 256>>>(W1, 1024 * 256, 2, 0.1f); fill_bf16<<<64, 256>>>(W2, 1024 * 256, 3, 0.05f);
    fill_f32<<<4, 256>>>(b1, 1024, 4, 0.0f); fill_f32<<<1, 256>>>(b2, 256, 5, 0.0f); fill_f32<<<1, 256>>>(ga, 256, 6, 1.0f); fill_f32<<<1, 256>>>(be, 256, 7, 0.0f);
    CHECK(hipDeviceSynchronize());
    FfnArgs a{};
    a.A = A; a.lda = 256; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.Fo = F; a.Mb = Mb; a.ldf = 1024; a.C = C; a.ldc = 256;
    a.gamma = ga; a.beta = be; a.rstd = rstd; a.eps = 1e-5f; a.M = M; a.drow_mul = 1;
    FfnArgs ad = a; ad.drop1 = Drop{12345u, 6553u, 1.1111f}; ad.drop2 = Drop{54321u, 6553u, 1.1111f};
    if (argc > 2) {                               // PMC mode (tools/pmc_tcc_ffn.sh): the two shipped 4-wave kernels only
        run<false, 0, 4>("eval  4-wave, 512 blocks", a, 512);
        run<true, 0, 4>("train 4-wave, 512 blocks", ad, 512);
        return 0;
    }
    for (int rep = 0; rep < 3; ++rep) {          // A/B of the two block shapes, interleaved (the clocks ramp over the first launches)
        run<false, 0>("eval  8-wave, 200 blocks", a, 200);
        run<false, 0, 4>("eval  4-wave, 512 blocks", a, 512);
        run<true, 0>("train 8-wave, 200 blocks", ad, 200);
        run<true, 0, 4>("train 4-wave, 512 blocks", ad, 512);
    }
    run<false, 0>("eval  full", a, 256);
    run<false, 0>("eval  full, 200 blocks", a, 200);
    run<false, 0, 4>("eval  4-wave blocks, 512", a, 512);
    run<false, 0, 4>("eval  4-wave blocks, 400", a, 400);
    run<false, 0, 4>("eval  4-wave blocks, 256 (1 per CU)", a, 256);
    run<true, 0, 4>("train 4-wave blocks, 512", ad, 512);
    run<true, 0, 4>("train 4-wave blocks, 400", ad, 400);
    run<true, 0, 4>("train 4-wave blocks, 256 (1 per CU)", ad, 256);
    run<true, 1, 4>("train 4-wave no MFMA", ad, 400);
    run<true, 2, 4>("train 4-wave no DMA", ad, 400);
    run<true, 8, 4>("train 4-wave no barrier", ad, 400);
    run<true, 16, 4>("train 4-wave no hidden epilogue math", ad, 400);
    run<true, 64, 4>("train 4-wave no pass epilogue", ad, 400);
    run<true, 128, 4>("train 4-wave no hidden store", ad, 512);
    run<true, 256, 4>("train 4-wave hidden stores into a cached 256 KB", ad, 512);
    run<true, 144, 4>("train 4-wave no hidden store, no epilogue math", ad, 512);
    run<true, 16, 4>("train 4-wave no hidden epilogue math", ad, 512);
    run<true, 0, 4>("train 4-wave", ad, 512);
    run<true, 0>("train full (dropout on, hidden stored)", ad, 256);
    run<true, 0>("train full, 200 blocks", ad, 200);
    run<true, 0>("train, dropout off (thr = 0), hidden stored", a, 256);
    run<true, 1>("train no MFMA", ad, 256);
    run<true, 2>("train no DMA", ad, 256);
    run<true, 8>("train no barrier", ad, 256);
    run<true, 16>("train no hidden epilogue math", ad, 256);
    run<true, 64>("train no pass epilogue", ad, 256);
    run<true, 32>("train no stage loop", ad, 256);
    run<false, 1>("eval  no MFMA", a, 256);
    run<false, 2>("eval  no DMA", a, 256);
    run<false, 4>("eval  no fragment reads", a, 256);
    run<false, 8>("eval  no barrier", a, 256);
    run<false, 16>("eval  no hidden epilogue", a, 256);
    run<false, 3>("eval  no MFMA, no DMA", a, 256);
    run<false, 5>("eval  no MFMA, no reads (DMA+sync only)", a, 256);
    run<false, 6>("eval  no DMA, no reads (MFMA+VALU only)", a, 256);
    run<false, 14>("eval  MFMA + VALU only, no barrier", a, 256);
    run<false, 32>("eval  no stage loop (pass prologue + epilogue)", a, 256);
    run<false, 64>("eval  no pass epilogue", a, 256);
    run<false, 96>("eval  neither (A loads + drain only)", a, 256);
    run<false, 0>("eval  full, 200 blocks", a, 200);
    run<false, 0>("eval  full, 128 blocks", a, 128);
    run<false, 0>("eval  full, 64 blocks", a, 64);
    run<false, 0>("eval  full, 8 blocks", a, 8);
    return 0;
}
