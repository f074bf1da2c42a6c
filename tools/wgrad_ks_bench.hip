// Standalone micro-benchmark of wgrad_ks_kernel + wgrad_ks_reduce_kernel at the real shapes, checked against the 128 x 128 kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I speaker_embedding_torch_amd/csrc tools/wgrad_ks_bench.hip -o tools/wgrad_ks_bench
//   tools/wgrad_ks_bench [blocks]      (default 160)
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "gemm.cuh"
#include "wgrad_ks.cuh"
using namespace ge2e;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void fill_bf16(bf16_t* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (bf16_t)(((mix32((unsigned)i * 2654435761u + seed) >> 8) * (1.0f / 8388608.0f)) - 1.0f);
}
int main(int argc, char** argv) {
    using T = bf16_t; const int R = 153600;
    const int cap = argc > 1 ? atoi(argv[1]) : 160;
    T *Y, *X; float *dW, *dW2, *db, *db2, *part;
    CHECK(hipMalloc(&Y, (size_t)R * 1024 * 2)); CHECK(hipMalloc(&X, (size_t)R * 1024 * 2));
    CHECK(hipMalloc(&dW, 1024 * 1024 * 4)); CHECK(hipMalloc(&dW2, 1024 * 1024 * 4)); CHECK(hipMalloc(&db, 4096)); CHECK(hipMalloc(&db2, 4096));
    CHECK(hipMalloc(&part, (size_t)256 * WK_TILE_FLOATS * 4));
    fill_bf16<<<2048, 256>>>(Y, (size_t)R * 1024, 1); fill_bf16<<<2048, 256>>>(X, (size_t)R * 1024, 2);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    struct S { const char* name; int N, K, ldy; } shapes[] = {{"l1 N1024 K256", 1024, 256, 1024}, {"l2 N256 K1024", 256, 1024, 256}, {"in N768 K256", 768, 256, 768}, {"kv N512 ldy768", 512, 256, 768}, {"out N256 K256", 256, 256, 256}};
    CHECK(hipFuncSetAttribute((const void*)wgrad_ks_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wgrad_ks_smem()));
    for (auto& sh : shapes) {
        const double fl = 2.0 * R * sh.N * sh.K, by = 2.0 * R * (sh.N + sh.K);
        float ms, ms2;
        {   // reference: 128 x 128 tiles
            constexpr int RS = 64; const int LD = 256 + 32;
            auto kern = wgrad_kernel<T, ALOAD_ROW>;
            const size_t smem = std::max<size_t>(4 * (size_t)RS * LD, 128 * (128 * 4 + 16));
            CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            WgradArgs a{}; a.Y = Y; a.ldy = sh.ldy; a.X = X; a.ldx = sh.K; a.dW = dW; a.ldw = sh.K; a.db = db; a.R = R; a.N = sh.N; a.K = sh.K;
            const int tn = sh.N / 128, tk = sh.K / 128;
            int splits = std::max(1, 512 / (tn * tk)); int rps = (R + splits - 1) / splits; rps = (rps + RS - 1) / RS * RS; splits = (R + rps - 1) / rps;
            a.rows_per_split = rps; a.tiles_n = tn; a.tiles_k = tk;
            CHECK(hipMemset(dW, 0, 1024 * 1024 * 4)); CHECK(hipMemset(db, 0, 4096));
            hipLaunchKernelGGL(kern, dim3(tn * tk * splits), dim3(256), smem, 0, a);
            CHECK(hipDeviceSynchronize());
        }
        const int tn = sh.N / 256, tk = sh.K / 256, ntile = tn * tk, stages = R / 32;
        int splits = std::min(cap, 256) / ntile;
        if (splits >= 8) splits = std::min((splits + 4) / 8 * 8, 256 / ntile / 8 * 8);
        splits = std::max(1, std::min(splits, stages / 8));
        const int sps = (stages + splits - 1) / splits;
        splits = (stages + sps - 1) / sps;
        WgradKsArgs k{}; k.Y = Y; k.ldy = sh.ldy; k.X = X; k.ldx = sh.K; k.part = part; k.db = db2; k.R32 = R; k.rows_per_split = sps * 32;
        k.tiles_n = tn; k.tiles_k = tk; k.splits = splits;
        const int grid = 8 * ntile * ((splits + 7) / 8), sgroups = std::max(1, std::min(splits, 256 / (ntile * 32)));
        auto run = [&]() {
            hipLaunchKernelGGL(wgrad_ks_kernel<T>, dim3(grid), dim3(512), wgrad_ks_smem(), 0, k);
            hipLaunchKernelGGL(wgrad_ks_reduce_kernel, dim3(ntile * 32, sgroups), dim3(512), 0, 0, (const float*)part, dW2, sh.K, splits, tn, tk);
        };
        for (int i = 0; i < 2; ++i) run();
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(wgrad_ks_kernel<T>, dim3(grid), dim3(512), wgrad_ks_smem(), 0, k);
        hipEventRecord(e1); for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(wgrad_ks_reduce_kernel, dim3(ntile * 32, sgroups), dim3(512), 0, 0, (const float*)part, dW2, sh.K, splits, tn, tk);
        hipEventRecord(e2); hipEventSynchronize(e2);
        hipEventElapsedTime(&ms, e0, e1); hipEventElapsedTime(&ms2, e1, e2); CHECK(hipGetLastError());
        printf("%-14s grid %3d splits %3d  ks %7.1f us  reduce %6.1f us  %6.1f TF/s  %5.2f TB/s", sh.name, grid, splits, ms * 100, ms2 * 100,
               fl / ((ms + ms2) / 10 * 1e-3) / 1e12, by / ((ms + ms2) / 10 * 1e-3) / 1e12);
        CHECK(hipMemset(dW2, 0, 1024 * 1024 * 4)); CHECK(hipMemset(db2, 0, 4096));
        run();
        CHECK(hipDeviceSynchronize());
        std::vector<float> h1((size_t)sh.N * sh.K), h2((size_t)sh.N * sh.K), b1(sh.N), b2(sh.N);
        CHECK(hipMemcpy(h1.data(), dW, h1.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h2.data(), dW2, h2.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(b1.data(), db, sh.N * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b2.data(), db2, sh.N * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0, bn = 0, bd = 0;
        for (size_t i = 0; i < h1.size(); ++i) { num += (double)(h1[i] - h2[i]) * (h1[i] - h2[i]); den += (double)h1[i] * h1[i]; }
        for (int i = 0; i < sh.N; ++i) { bn += (double)(b1[i] - b2[i]) * (b1[i] - b2[i]); bd += (double)b1[i] * b1[i]; }
        printf("   rel_l2 dW %.2e db %.2e\n", std::sqrt(num / den), std::sqrt(bn / std::max(bd, 1e-30)));
    }
    return 0;
}
