// Standalone micro-benchmark of the fused attention kernels (development tool).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I speaker_embedding_torch_amd/csrc -I tools tools/attn_bench.hip -o tools/attn_bench
#include <cstdio>
#include <cstdlib>
#include "attention.cuh"
#include "experimental/attn_bwd_onepass.cuh"
using namespace ge2e;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
template <typename K> float time_kernel(K launch, int iters = 10) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) launch();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); CHECK(hipGetLastError());
    return ms / iters;
}
__global__ void fill_bf16(bf16_t* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (bf16_t)(((mix32((unsigned)i * 2654435761u + seed) >> 8) * (1.0f / 8388608.0f)) - 1.0f);
}
template <int SBE> void run(const AttnArgs& a, int n, const char* tag) {
    using T = bf16_t; using G = attn::Geo<T>; constexpr int KT = 5, TP = 160;
    const size_t sf = 2 * (size_t)TP * G::LD, sb = sf + 2 * TP * 4 + TP * (TP / 4);
    auto kf = attn_fwd_kernel<T, KT, false, true, SBE>; auto kb = attn_bwd_kernel<T, KT, false, true, SBE>;
    CHECK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sf));
    CHECK(hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb));
    float f = time_kernel([&]() { hipLaunchKernelGGL(kf, dim3(n * 4), dim3(320), sf, 0, a); });
    float b = time_kernel([&]() { hipLaunchKernelGGL(kb, dim3(n * 4), dim3(320), sb, 0, a); });
    const double ff = 4.0 * 160 * 160 * 64 * n * 4, fb = 14.0 * 160 * 160 * 64 * n * 4;
    printf("%-10s fwd %7.1f us %6.1f TF/s | bwd %7.1f us %6.1f TF/s\n", tag, f * 1e3, ff / f / 1e9, b * 1e3, fb / b / 1e9);
}
template <int ABL> void run1(const AttnArgs& a, int n, const char* tag) {
    using T = bf16_t; constexpr int KT = 5;
    auto kb = attn_bwd1_kernel<T, KT, false, ABL>;
    const size_t sb = attn_bwd1_smem<T, KT>();
    CHECK(hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb));
    int nb = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kb, 320, sb));
    float b = time_kernel([&]() { hipLaunchKernelGGL(kb, dim3(n * 4), dim3(320), sb, 0, a); });
    printf("one-pass %-34s abl=%2d blocks/CU=%d  %7.1f us\n", tag, ABL, nb, b * 1e3);
}
int main() {
    const int n = 960, T_ = 160, D = 256; const size_t R = (size_t)n * T_;
    bf16_t *qkv, *o, *dout, *dqkv;
    CHECK(hipMalloc(&qkv, R * 768 * 2)); CHECK(hipMalloc(&o, R * 256 * 2)); CHECK(hipMalloc(&dout, R * 256 * 2)); CHECK(hipMalloc(&dqkv, R * 768 * 2));
    fill_bf16<<<2048, 256>>>(qkv, R * 768, 1); fill_bf16<<<2048, 256>>>(dout, R * 256, 2); CHECK(hipDeviceSynchronize());
    AttnArgs a{}; a.qkv = qkv; a.o = o; a.dout = dout; a.dqkv = dqkv; a.T = T_; a.H = 4; a.D = D; a.scale = 0.125f;
    a.drop = Drop{12345u, 6553u, 1.1111f};
    float* lse; CHECK(hipMalloc(&lse, R * 4 * 4)); a.lse = lse;
    run<1>(a, n, "sbe=1"); run<5>(a, n, "sbe=5");
    run1<0>(a, n, "full"); run1<1>(a, n, "no dQ atomics"); run1<2>(a, n, "no barrier / write-out"); run1<3>(a, n, "no atomics, no barrier");
    run1<4>(a, n, "no slab, no dQ"); run1<8>(a, n, "no dK/dV MFMAs"); run1<16>(a, n, "no score evaluation"); run1<23>(a, n, "only dK/dV MFMAs");
    AttnArgs nd = a; nd.drop = Drop{0u, 0u, 1.0f}; run1<0>(nd, n, "full, dropout off");
    return 0;
}
