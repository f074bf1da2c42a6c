// Standalone micro-benchmark of the weight-gradient kernel at the real shapes (development tool).
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include "gemm.cuh"
using namespace ge2e;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main() {
    using T = bf16_t; const int R = 153600;
    T *Y, *X; float *dW, *db;
    CHECK(hipMalloc(&Y, (size_t)R * 1024 * 2)); CHECK(hipMalloc(&X, (size_t)R * 1024 * 2)); CHECK(hipMalloc(&dW, 1024 * 1024 * 4)); CHECK(hipMalloc(&db, 4096));
    CHECK(hipMemset(Y, 0x3c, (size_t)R * 1024 * 2)); CHECK(hipMemset(X, 0x3c, (size_t)R * 1024 * 2)); CHECK(hipMemset(dW, 0, 1024 * 1024 * 4)); CHECK(hipMemset(db, 0, 4096));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct S { const char* name; int N, K; } shapes[] = {{"l1 N1024 K256", 1024, 256}, {"in N768 K256", 768, 256}, {"out N256 K256", 256, 256}};
    auto run = [&](auto kern, int RS, const char* tag, int target_blocks) {
        const int LD = 256 + 32;
        const size_t smem = std::max<size_t>(4 * (size_t)RS * LD, 128 * (128 * 4 + 16));
        CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        for (auto& sh : shapes) {
            WgradArgs a{}; a.Y = Y; a.ldy = sh.N; a.X = X; a.ldx = sh.K; a.dW = dW; a.ldw = sh.K; a.db = db; a.R = R; a.N = sh.N; a.K = sh.K;
            const int tn = sh.N / 128, tk = sh.K / 128;
            int splits = (target_blocks + tn * tk - 1) / (tn * tk); int rps = (R + splits - 1) / splits; rps = (rps + RS - 1) / RS * RS; splits = (R + rps - 1) / rps;
            a.rows_per_split = rps; a.tiles_n = tn; a.tiles_k = tk;
            float ms;
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(tn * tk * splits), dim3(256), smem, 0, a);
            hipEventRecord(e0); for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(tn * tk * splits), dim3(256), smem, 0, a); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); CHECK(hipGetLastError());
            const double fl = 2.0 * R * sh.N * sh.K, by = 2.0 * R * (sh.N + sh.K);
            printf("%-22s %-14s blocks %4d  %7.1f us  %6.1f TF/s  %5.2f TB/s\n", tag, sh.name, tn * tk * splits, ms * 100, fl / (ms / 10 * 1e-3) / 1e12, by / (ms / 10 * 1e-3) / 1e12);
        }
    };
    run(wgrad_kernel<T, ALOAD_ROW, 3, 2>, 64, "ns3 rs64 (now)", 512);
    run(wgrad_kernel<T, ALOAD_ROW, 1, 2>, 64, "ns1 rs64", 512);
    run(wgrad_kernel<T, ALOAD_ROW, 1, 1>, 32, "ns1 rs32 4blk/CU", 1024);
    run(wgrad_kernel<T, ALOAD_ROW, 2, 1>, 32, "ns2 rs32", 1024);
    run(wgrad_kernel<T, ALOAD_ROW, 1, 1>, 32, "ns1 rs32 768blk", 768);
    return 0;
}
