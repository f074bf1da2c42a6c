// The fused attention sub-layer kernel (attn_sub.cuh) against the three launches it replaces, at the headline shape, each alone on the chip,
// with the ablations that say where its time goes (development tool; numbers: profiles/r04_ab_log.txt section 3).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I speaker_embedding_torch_amd/csrc tools/attn_sub_bench.hip -o tools/attn_sub_bench
#include <cstdio>
#include <cstdlib>
#include "attn_sub.cuh"
#include "gemm.cuh"
#include "gemm_ws.cuh"
using namespace ge2e;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
template <typename K> float time_kernel(K launch, int iters = 10) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); CHECK(hipGetLastError());
    return ms / iters * 1e3f;
}
__global__ void fill_bf16(bf16_t* p, size_t n, unsigned seed, float amp) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (bf16_t)((((mix32((unsigned)i * 2654435761u + seed) >> 8) * (1.0f / 8388608.0f)) - 1.0f) * amp);
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float base) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = base + (((mix32((unsigned)i * 2654435761u + seed) >> 8) * (1.0f / 8388608.0f)) - 1.0f) * 0.1f;
}
template <int ABL, bool TRAIN> void run(AttnSubArgs a, int n, const char* tag) {
    using T = bf16_t; constexpr int NW = 10;
    if (!TRAIN) { a.qkv = nullptr; a.o = nullptr; a.lse = nullptr; a.rstd = nullptr; a.drop_attn = Drop{0u, 0u, 1.0f}; a.drop_sa = Drop{0u, 0u, 1.0f}; }
    auto k = attn_sub_fwd_kernel<T, NW, false, TRAIN, ABL>;
    constexpr int smem = attn_sub::smem_bytes<NW>();
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    printf("%-5s %-44s %7.1f us\n", TRAIN ? "train" : "eval", tag, time_kernel([&]() { hipLaunchKernelGGL(k, dim3(n), dim3(64 * NW), smem, 0, a); }));
}
int main() {
    const int n = 960, T_ = 160; const size_t R = (size_t)n * T_;
    bf16_t *x, *win, *wo, *qkv, *o, *h1; float *bin, *bo, *ga, *be, *lse, *rstd;
    CHECK(hipMalloc(&x, R * 512)); CHECK(hipMalloc(&win, 768 * 512)); CHECK(hipMalloc(&wo, 256 * 512)); CHECK(hipMalloc(&qkv, R * 1536)); CHECK(hipMalloc(&o, R * 512)); CHECK(hipMalloc(&h1, R * 512));
    CHECK(hipMalloc(&bin, 3072)); CHECK(hipMalloc(&bo, 1024)); CHECK(hipMalloc(&ga, 1024)); CHECK(hipMalloc(&be, 1024)); CHECK(hipMalloc(&lse, R * 16)); CHECK(hipMalloc(&rstd, R * 4));
    fill_bf16<<<2048, 256>>>(x, R * 256, 1, 1.0f); fill_bf16<<<64, 256>>>(win, 768 * 256, 2, 0.06f); fill_bf16<<<64, 256>>>(wo, 256 * 256, 3, 0.06f);
    fill_f32<<<3, 256>>>(bin, 768, 4, 0.0f); fill_f32<<<1, 256>>>(bo, 256, 5, 0.0f); fill_f32<<<1, 256>>>(ga, 256, 6, 1.0f); fill_f32<<<1, 256>>>(be, 256, 7, 0.0f);
    CHECK(hipDeviceSynchronize());
    AttnSubArgs a{};
    a.X = x; a.Win = win; a.bin = bin; a.Wo = wo; a.bo = bo; a.gamma = ga; a.beta = be; a.eps = 1e-5f; a.qkv = qkv; a.o = o; a.lse = lse; a.h1 = h1; a.rstd = rstd;
    a.T = T_; a.scale = 0.125f; a.drop_attn = Drop{12345u, 6553u, 1.0f / 0.9f}; a.drop_sa = Drop{54321u, 6553u, 1.0f / 0.9f};
    for (int rep = 0; rep < 2; ++rep) {
        run<0, true>(a, n, "full");
        run<1, true>(a, n, "no attention");
        run<2, true>(a, n, "no weight streaming (stage never refilled)");
        run<4, true>(a, n, "no projection MFMAs");
        run<3, true>(a, n, "no attention, no weight streaming");
        run<7, true>(a, n, "x load + barriers + epilogue only");
        run<0, false>(a, n, "full");
        run<2, false>(a, n, "no weight streaming");
    }
    {   // the three launches it replaces, alone
        GemmArgs g{}; g.A = x; g.lda = 256; g.W = win; g.ldw = 256; g.C = qkv; g.ldc = 768; g.M = (int)R; g.N = 768; g.K = 256; g.bias = bin;
        constexpr int E1 = EPI_BIAS, E3 = EPI_LN;
        auto k1 = gemm_ws_kernel<bf16_t, E1, 256>;
        const int ntiles = (int)R / 16, cg = 3, parts = (512 / cg) / 8 * 8;
        const size_t sm1 = gemm_ws_smem<E1, 256>(), sm3 = gemm_ws_smem<E3, 256>();
        CHECK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm1));
        const float t1 = time_kernel([&]() { hipLaunchKernelGGL(k1, dim3(cg * parts), dim3(256), sm1, 0, g, parts, ntiles); });
        AttnArgs aa{}; aa.qkv = qkv; aa.o = o; aa.lse = lse; aa.T = T_; aa.H = 4; aa.D = 256; aa.scale = 0.125f; aa.drop = a.drop_attn;
        auto k2 = attn_fwd_kernel<bf16_t, 5, false, true, 5>;
        const size_t s2 = 2 * 160 * 128;
        CHECK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s2));
        const float t2 = time_kernel([&]() { hipLaunchKernelGGL(k2, dim3(n * 4), dim3(320), s2, 0, aa); });
        GemmArgs l{}; l.A = o; l.lda = 256; l.W = wo; l.ldw = 256; l.C = h1; l.ldc = 256; l.M = (int)R; l.N = 256; l.K = 256; l.bias = bo; l.R = x; l.ldr = 256;
        l.gamma = ga; l.beta = be; l.eps = 1e-5f; l.rstd = rstd; l.drop = a.drop_sa; l.drow_mul = 1;
        auto k3 = gemm_ws_kernel<bf16_t, E3, 256>;
        CHECK(hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm3));
        const float t3 = time_kernel([&]() { hipLaunchKernelGGL(k3, dim3(512), dim3(256), sm3, 0, l, 512, ntiles); });
        printf("the three launches it replaces (train): in_proj %.1f + attention %.1f + out_proj / LayerNorm %.1f = %.1f us\n", t1, t2, t3, t1 + t2 + t3);
    }
    return 0;
}
