// Standalone micro-benchmark of the projection-GEMM kernel variants at the real shapes (development tool).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I speaker_embedding_torch_amd/csrc tools/gemm_bench.hip -o tools/gemm_bench
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "gemm.cuh"
#include "gemm_ws.cuh"
#include "gemm_kl.cuh"
using namespace ge2e;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <typename K> float time_kernel(K launch, int iters = 20) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipGetLastError());
    return ms / iters;
}

__global__ void fill_bf16(bf16_t* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (bf16_t)(((mix32((unsigned)i * 2654435761u + seed) >> 8) * (1.0f / 8388608.0f)) - 1.0f);
}
__global__ void fill_f32(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (((mix32((unsigned)i * 2654435761u + seed) >> 8) * (1.0f / 8388608.0f)) - 1.0f) * 0.1f;
}

__global__ void count_diff(const uint32_t* a, const uint32_t* b, size_t n, unsigned long long* out) {
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(out, c);
}

template <int EPI>
void run_ws(const char* name, GemmArgs a, bf16_t* Cref, bf16_t* C2, int M, double flops, double bytes, int parts) {
    auto kern = gemm_ws_kernel<bf16_t, EPI, 256>;
    const size_t smem = gemm_ws_smem<EPI, 256>();
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    a.C = C2;
    const int ntiles = (M + 15) / 16, CG = a.N / 256;
    const int grid = CG * parts;
    CHECK(hipMemset(C2, 0, (size_t)M * a.N * 2));
    float ms = time_kernel([&]() { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, 0, a, parts, ntiles); });
    unsigned long long* d; CHECK(hipMalloc(&d, 8)); CHECK(hipMemset(d, 0, 8));
    count_diff<<<2048, 256>>>((const uint32_t*)Cref, (const uint32_t*)C2, (size_t)M * a.N / 2, d);
    unsigned long long h = 0; CHECK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost)); CHECK(hipFree(d));
    char tag[64]; snprintf(tag, sizeof tag, "WS parts=%d", parts);
    printf("%-26s %-22s %8.1f us  %7.1f TF/s  %6.2f TB/s(min-bytes)  mismatching words: %llu\n", name, tag, ms * 1e3, flops / ms / 1e9, bytes / ms / 1e9, h);
}

template <int EPI, int K_, int ABL = 0>
void run_kl(const char* name, GemmArgs a, bf16_t* Cref, bf16_t* C2, int M, double flops, double bytes, int grid_cap) {
    auto kern = gemm_kl_kernel<bf16_t, EPI, K_, ABL>;
    const size_t smem = gemm_kl_smem<EPI>();
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    a.C = C2;
    const int ntiles = (M + 127) / 128;
    const int grid = std::min(grid_cap, ntiles);
    CHECK(hipMemset(C2, 0, (size_t)M * a.N * 2));
    float ms = time_kernel([&]() { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, 0, a, ntiles); });
    unsigned long long* d; CHECK(hipMalloc(&d, 8)); CHECK(hipMemset(d, 0, 8));
    count_diff<<<2048, 256>>>((const uint32_t*)Cref, (const uint32_t*)C2, (size_t)M * a.N / 2, d);
    unsigned long long h = 0; CHECK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost)); CHECK(hipFree(d));
    char tag[64]; snprintf(tag, sizeof tag, "KL grid=%d abl=%d", grid, ABL);
    printf("%-26s %-22s %8.1f us  %7.1f TF/s  %6.2f TB/s(min-bytes)  mismatching words: %llu\n", name, tag, ms * 1e3, flops / ms / 1e9, bytes / ms / 1e9, h);
}

int main(int argc, char** argv) {
    const int M = 153600;
    using T = bf16_t;
    T *A, *W, *C, *R, *C2; float* bias; float *gamma, *beta, *rstd;
    CHECK(hipMalloc(&A, (size_t)M * 1024 * 2)); CHECK(hipMalloc(&W, (size_t)1024 * 1024 * 2));
    CHECK(hipMalloc(&C, (size_t)M * 1024 * 2)); CHECK(hipMalloc(&C2, (size_t)M * 1024 * 2)); CHECK(hipMalloc(&R, (size_t)M * 1024 * 2));
    CHECK(hipMalloc(&bias, 4096)); CHECK(hipMalloc(&gamma, 4096)); CHECK(hipMalloc(&beta, 4096)); CHECK(hipMalloc(&rstd, (size_t)M * 4));
    fill_bf16<<<2048, 256>>>(A, (size_t)M * 1024, 1); fill_bf16<<<2048, 256>>>(W, (size_t)1024 * 1024, 2);
    fill_bf16<<<2048, 256>>>(R, (size_t)M * 1024, 3); fill_f32<<<4, 256>>>(bias, 1024, 4);
    fill_f32<<<4, 256>>>(gamma, 1024, 5); fill_f32<<<4, 256>>>(beta, 1024, 6);
    CHECK(hipDeviceSynchronize());

    struct Shape { const char* name; int N, K, epi; };
    const Shape shapes[] = {{"in_proj  N768 K256 bias", 768, 256, EPI_BIAS}, {"ffn1     N1024 K256 relu", 1024, 256, EPI_BIAS_RELU_DROP},
                            {"dF       N1024 K256 mask", 1024, 256, EPI_MASK}, {"dH1      N256 K1024 add", 256, 1024, EPI_ADD},
                            {"dO       N256 K256 none", 256, 256, EPI_NONE}, {"dH       N256 K768 add", 256, 768, EPI_ADD}};
    for (const Shape& s : shapes) {
        GemmArgs a{};
        a.A = A; a.lda = s.K; a.W = W; a.ldw = s.K; a.C = C; a.ldc = s.N; a.M = M; a.N = s.N; a.K = s.K;
        a.bias = bias; a.R = R; a.ldr = s.N; a.mask_scale = 1.1f; a.drop = Drop{12345u, 1677721u, 1.1111f};
        const double flops = 2.0 * M * s.N * s.K;
        const double bytes = 2.0 * ((double)M * s.K + (double)M * s.N + ((s.epi == EPI_MASK || s.epi == EPI_ADD) ? (double)M * s.N : 0));
        auto run = [&](auto kern, int BM, int BN, size_t smem, const char* tag) {
            CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            const int grid = ((M + BM - 1) / BM) * (s.N / BN);
            float ms = time_kernel([&]() { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, 0, a); });
            printf("%-26s %-22s %8.1f us  %7.1f TF/s  %6.2f TB/s(min-bytes)\n", s.name, tag, ms * 1e3, flops / ms / 1e9, bytes / ms / 1e9);
        };
        switch (s.epi) {
#define VARIANTS(EPI_)                                                                                        \
            run(gemm_nt_kernel<T, 128, 128, 64, 64, EPI_, ALOAD_ROW>, 128, 128, std::max<size_t>(2 * 256 * 128, 128 * (128 * sizeof(T) + 16)), "dbuf 64KB"); \
            run(gemm_nt_kernel<T, 128, 128, 64, 64, EPI_, ALOAD_ROW, 1>, 128, 128, std::max<size_t>(256 * 128, 128 * (128 * sizeof(T) + 16)), "single 35KB");
            case EPI_BIAS: VARIANTS(EPI_BIAS) break;
            case EPI_BIAS_RELU_DROP: VARIANTS(EPI_BIAS_RELU_DROP) break;
            case EPI_MASK: VARIANTS(EPI_MASK) break;
            case EPI_ADD: VARIANTS(EPI_ADD) break;
            case EPI_NONE: VARIANTS(EPI_NONE) break;
        }
        if (s.epi == EPI_ADD && s.K == 1024) { run_kl<EPI_ADD, 1024>(s.name, a, C, C2, M, flops, bytes, 256); run_kl<EPI_ADD, 1024, 1>(s.name, a, C, C2, M, flops, bytes, 256); run_kl<EPI_ADD, 1024, 3>(s.name, a, C, C2, M, flops, bytes, 256); run_kl<EPI_ADD, 1024, 5>(s.name, a, C, C2, M, flops, bytes, 256); run_kl<EPI_ADD, 1024, 2>(s.name, a, C, C2, M, flops, bytes, 256); }
        if (s.epi == EPI_ADD && s.K == 768) run_kl<EPI_ADD, 768>(s.name, a, C, C2, M, flops, bytes, 256);
        if (s.K == 256) {
            for (int tot : {512}) {
                const int CG = s.N / 256;
                int parts = (tot / CG) / 8 * 8; if (parts < 8) parts = 8;
                switch (s.epi) {
                    case EPI_BIAS: run_ws<EPI_BIAS>(s.name, a, C, C2, M, flops, bytes, parts); break;
                    case EPI_BIAS_RELU_DROP: run_ws<EPI_BIAS_RELU_DROP>(s.name, a, C, C2, M, flops, bytes, parts); break;
                    case EPI_MASK: run_ws<EPI_MASK>(s.name, a, C, C2, M, flops, bytes, parts); break;
                    case EPI_NONE: run_ws<EPI_NONE>(s.name, a, C, C2, M, flops, bytes, parts); break;
                }
            }
        }
    }
    {   // LN-epilogue GEMMs
        for (int K : {256, 1024}) {
            GemmArgs a{};
            a.A = A; a.lda = K; a.W = W; a.ldw = K; a.C = C; a.ldc = 256; a.M = M; a.N = 256; a.K = K;
            a.bias = bias; a.R = R; a.ldr = 256; a.gamma = gamma; a.beta = beta; a.rstd = rstd; a.eps = 1e-5f;
            a.drop = Drop{12345u, 1677721u, 1.1111f};
            auto kern = gemm_nt_kernel<T, 64, 256, 32, 128, EPI_LN, ALOAD_ROW>;
            const size_t smem = std::max<size_t>(2 * (64 + 256) * 128, 64 * 260 * 4);
            CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            float ms = time_kernel([&]() { hipLaunchKernelGGL(kern, dim3((M + 63) / 64), dim3(256), smem, 0, a); });
            printf("LN gemm N256 K%-4d          %-22s %8.1f us  %7.1f TF/s  %6.2f TB/s\n", K, "v2 64x256 row-epi", ms * 1e3,
                   2.0 * M * 256 * K / ms / 1e9, 2.0 * ((double)M * K + 2.0 * M * 256) / ms / 1e9);
            if (K == 1024) run_kl<EPI_LN, 1024>("LN gemm N256 K1024", a, C, C2, M, 2.0 * M * 256 * K, 2.0 * ((double)M * K + 2.0 * M * 256), 256);
            if (K == 256) for (int parts : {512})
                run_ws<EPI_LN>("LN gemm N256 K256", a, C, C2, M, 2.0 * M * 256 * K, 2.0 * ((double)M * K + 2.0 * M * 256), parts);
        }
    }
    return 0;
}
