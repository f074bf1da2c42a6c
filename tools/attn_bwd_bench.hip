// Phase ablation of attn_bwd_kernel at the headline shape (development tool).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I speaker_embedding_torch_amd/csrc tools/attn_bwd_bench.hip -o tools/attn_bwd_bench
#include <cstdio>
#include <cstdlib>
#include "attention.cuh"
#include "experimental/attention_v0.cuh"
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
__global__ void fill_bf16(bf16_t* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (bf16_t)(((mix32((unsigned)i * 2654435761u + seed) >> 8) * (1.0f / 8388608.0f)) - 1.0f);
}
template <int ABL> void run0(const AttnArgs& a0, int n, const char* tag, size_t extra_lds = 0) {
    using T = bf16_t; using G = attn::Geo<T>; constexpr int KT = 5, TP = 160;
    v0::AttnArgs a{}; a.qkv = a0.qkv; a.o = a0.o; a.lse = a0.lse; a.dout = a0.dout; a.dqkv = a0.dqkv; a.T = a0.T; a.H = a0.H; a.D = a0.D; a.scale = a0.scale; a.drop = a0.drop;
    const size_t sb = 2 * (size_t)TP * G::LD + 2 * TP * 4 + TP * (TP / 32) * 4 + extra_lds;
    auto kb = v0::attn_bwd_kernel<T, KT, false, 5, ABL>;
    CHECK(hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb));
    printf("v0  %-40s abl %2d  %7.1f us\n", tag, ABL, time_kernel([&]() { hipLaunchKernelGGL(kb, dim3(n * 4), dim3(320), sb, 0, a); }));
}
template <int ABL> void runf0(const AttnArgs& a0, int n, const char* tag, size_t extra_lds = 0) {
    using T = bf16_t; using G = attn::Geo<T>; constexpr int KT = 5, TP = 160;
    v0::AttnArgs a{}; a.qkv = a0.qkv; a.o = a0.o; a.lse = a0.lse; a.dout = a0.dout; a.dqkv = a0.dqkv; a.T = a0.T; a.H = a0.H; a.D = a0.D; a.scale = a0.scale; a.drop = a0.drop;
    const size_t sf = 2 * (size_t)TP * G::LD + extra_lds;
    auto kf = v0::attn_fwd_kernel<T, KT, false, 1, ABL>;
    CHECK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sf));
    printf("v0 fwd %-37s abl %2d  %7.1f us\n", tag, ABL, time_kernel([&]() { hipLaunchKernelGGL(kf, dim3(n * 4), dim3(320), sf, 0, a); }));
}
template <int ABL, bool DROP = true, int SCH = 5, int MINB = 1> void run(const AttnArgs& a, int n, const char* tag, size_t extra_lds = 0) {
    using T = bf16_t; using G = attn::Geo<T>; constexpr int KT = 5, TP = 160;
    const size_t sb = 2 * (size_t)TP * G::LD + 2 * TP * 4 + TP * (TP / 4) + extra_lds;
    auto kb = attn_bwd_kernel<T, KT, false, DROP, SCH, ABL, MINB>;
    CHECK(hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb));
    printf("sbe%d minb%d %-34s abl %2d  %7.1f us\n", SCH, MINB, tag, ABL, time_kernel([&]() { hipLaunchKernelGGL(kb, dim3(n * 4), dim3(320), sb, 0, a); }));
}
template <int ABL, bool DROP = true, int SCH = 1> void runf(const AttnArgs& a, int n, const char* tag, size_t extra_lds = 0) {
    using T = bf16_t; using G = attn::Geo<T>; constexpr int KT = 5, TP = 160;
    const size_t sf = 2 * (size_t)TP * G::LD + extra_lds;
    auto kf = attn_fwd_kernel<T, KT, false, DROP, SCH, ABL>;
    CHECK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sf));
    printf("sch%d fwd %-36s abl %2d  %7.1f us\n", SCH, tag, ABL, time_kernel([&]() { hipLaunchKernelGGL(kf, dim3(n * 4), dim3(320), sf, 0, a); }));
}
int main(int argc, char** argv) {
    const int n = 960, T_ = 160, D = 256; const size_t R = (size_t)n * T_;
    bf16_t *qkv, *o, *dout, *dqkv; float* lse;
    CHECK(hipMalloc(&qkv, R * 768 * 2)); CHECK(hipMalloc(&o, R * 256 * 2)); CHECK(hipMalloc(&dout, R * 256 * 2)); CHECK(hipMalloc(&dqkv, R * 768 * 2)); CHECK(hipMalloc(&lse, R * 16));
    fill_bf16<<<2048, 256>>>(qkv, R * 768, 1); fill_bf16<<<2048, 256>>>(dout, R * 256, 2); fill_bf16<<<2048, 256>>>(o, R * 256, 3); CHECK(hipMemset(lse, 0, R * 16)); CHECK(hipDeviceSynchronize());
    AttnArgs a{}; a.qkv = qkv; a.o = o; a.dout = dout; a.dqkv = dqkv; a.T = T_; a.H = 4; a.D = D; a.scale = 0.125f; a.lse = lse;
    a.drop = Drop{12345u, 6553u, 1.1111f};
    if (argc > 1) {      // PMC mode: few launches of the variants under study
        run0<0>(a, n, "full"); run<0, true, 5, 1>(a, n, "full"); run<0, true, 5, 2>(a, n, "full");
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        run0<0>(a, n, "full"); run<0, true, 5, 1>(a, n, "full"); run<0, true, 5, 2>(a, n, "full"); run<0, true, 1, 2>(a, n, "full"); run<0, true, 2, 2>(a, n, "full");
        run0<1>(a, n, "phase A only"); run<1, true, 5, 1>(a, n, "phase A only"); run<1, true, 5, 2>(a, n, "phase A only");
        run0<2>(a, n, "phase B only"); run<2, true, 5, 1>(a, n, "phase B only"); run<2, true, 5, 2>(a, n, "phase B only");
        run0<4>(a, n, "no dropout"); run<0, false, 5, 1>(a, n, "no dropout"); run<0, false, 5, 2>(a, n, "no dropout");
        run0<3>(a, n, "tile loads + barriers only");
        runf0<0>(a, n, "full"); runf<0, true, 1>(a, n, "full"); runf<0, true, 2>(a, n, "full"); runf<0, true, 5>(a, n, "full");
        runf0<2>(a, n, "no dropout"); runf<0, false, 1>(a, n, "no dropout");
    }
    return 0;
}
