// Phase ablation of the last layer's single-query attention kernels at the headline shape (development tool).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I speaker_embedding_torch_amd/csrc tools/attn_last_bench.hip -o tools/attn_last_bench
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <type_traits>
#include "attn_last.cuh"
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
    for (int i = 0; i < 5; ++i) launch();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); CHECK(hipGetLastError());
    return ms / iters * 1e3f;
}
// forward determinism: the same launch five times, outputs compared bit for bit (tools/attn_last_bench det [T] [N])
template <typename T_> int determinism(int N, int T) {
    T_ *x, *q0, *W, *o0; float* bv; unsigned short* ref;
    CHECK(hipMalloc(&x, (size_t)N * T * 256 * 2)); CHECK(hipMalloc(&q0, N * 512)); CHECK(hipMalloc(&W, 768 * 512)); CHECK(hipMalloc(&o0, N * 512)); CHECK(hipMalloc(&bv, 1024));
    fill_bf16<<<2048, 256>>>((bf16_t*)x, (size_t)N * T * 256, 1, 1.0f); fill_bf16<<<64, 256>>>((bf16_t*)q0, N * 256, 2, 1.0f); fill_bf16<<<64, 256>>>((bf16_t*)W, 768 * 256, 3, 0.06f);
    if (sizeof(T_) == 2 && std::is_same<T_, f16_t>::value) {      // the same bit patterns read as halves would be huge: refill with half values
        std::vector<unsigned short> hx((size_t)N * T * 256), hq((size_t)N * 256), hw(768 * 256);
        auto gen = [](std::vector<unsigned short>& v, unsigned seed, float amp) { for (size_t i = 0; i < v.size(); ++i) { _Float16 f = (_Float16)((((mix32((unsigned)i * 2654435761u + seed) >> 8) * (1.0f / 8388608.0f)) - 1.0f) * amp); memcpy(&v[i], &f, 2); } };
        gen(hx, 1, 1.0f); gen(hq, 2, 1.0f); gen(hw, 3, 0.06f);
        CHECK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(q0, hq.data(), hq.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    }
    fill_f32<<<1, 256>>>(bv, 256, 6, 0.0f);
    CHECK(hipDeviceSynchronize());
    AttnLastArgs a{};
    a.x = x; a.q0 = q0; a.Wq = W; a.Wk = W + 256 * 256; a.Wv = W + 512 * 256; a.bv = bv; a.o0 = o0; a.T = T; a.H = 4; a.scale = 0.125f;
    auto f0 = attn_last_fwd_kernel<T_>;
    std::vector<unsigned short> first((size_t)N * 256), cur((size_t)N * 256);
    int bad = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipMemset(o0, 0xFF, N * 512));
        hipLaunchKernelGGL(f0, dim3(N), dim3(256), attn_last_fwd_smem(T), 0, a);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(cur.data(), o0, N * 512, hipMemcpyDeviceToHost));
        if (rep == 0) first = cur;
        else { size_t d = 0, firstrow = (size_t)-1; for (size_t e = 0; e < cur.size(); ++e) if (cur[e] != first[e]) { ++d; if (firstrow == (size_t)-1) firstrow = e / 256; }
               printf("%s T %d N %d launch %d: %zu elements differ from launch 0 (first row %zd)\n", sizeof(T_) == 4 ? "fp32" : (std::is_same<T_, f16_t>::value ? "fp16" : "bf16"), T, N, rep, d, (ssize_t)firstrow); bad += d != 0; }
    }
    return bad;
}
int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "det") {
        const int T = argc > 2 ? atoi(argv[2]) : 64, N = argc > 3 ? atoi(argv[3]) : 1280;
        return determinism<bf16_t>(N, T) + determinism<f16_t>(N, T);
    }
    const int N = 960, T = 160;
    bf16_t *x, *q0, *W, *o0, *do0, *dpre, *dX, *dq0; float *bv, *qk, *prob, *ctx, *sp, *dqk;
    CHECK(hipMalloc(&x, (size_t)N * T * 256 * 2)); CHECK(hipMalloc(&dX, (size_t)N * T * 256 * 2)); CHECK(hipMalloc(&q0, N * 512)); CHECK(hipMalloc(&W, 768 * 512));
    CHECK(hipMalloc(&o0, N * 512)); CHECK(hipMalloc(&do0, N * 512)); CHECK(hipMalloc(&dpre, N * 512)); CHECK(hipMalloc(&dq0, N * 512));
    CHECK(hipMalloc(&bv, 1024)); CHECK(hipMalloc(&qk, (size_t)N * 4096)); CHECK(hipMalloc(&ctx, (size_t)N * 4096)); CHECK(hipMalloc(&dqk, (size_t)N * 4096));
    CHECK(hipMalloc(&prob, (size_t)N * 4 * T * 4)); CHECK(hipMalloc(&sp, N * 16));
    fill_bf16<<<2048, 256>>>(x, (size_t)N * T * 256, 1, 1.0f); fill_bf16<<<64, 256>>>(q0, N * 256, 2, 1.0f); fill_bf16<<<64, 256>>>(W, 768 * 256, 3, 0.06f);
    fill_bf16<<<64, 256>>>(do0, N * 256, 4, 0.1f); fill_bf16<<<64, 256>>>(dpre, N * 256, 5, 0.1f); fill_f32<<<1, 256>>>(bv, 256, 6, 0.0f);
    CHECK(hipDeviceSynchronize());
    AttnLastArgs a{};
    a.x = x; a.q0 = q0; a.Wq = W; a.Wk = W + 256 * 256; a.Wv = W + 512 * 256; a.bv = bv; a.o0 = o0; a.qk = qk; a.prob = prob; a.ctx = ctx; a.sp = sp;
    a.do0 = do0; a.dpre = dpre; a.dX = dX; a.dq0 = dq0; a.dqk = dqk; a.T = T; a.H = 4; a.scale = 0.125f; a.drop = Drop{12345u, 6553u, 1.1111f};
    auto f0 = attn_last_fwd_kernel<bf16_t>; auto b0 = attn_last_bwd_kernel<bf16_t>;
    const size_t sf = attn_last_fwd_smem(T), sb = attn_last_bwd_smem(T);
    for (int abl : {0, 1, 2, 4, 8, 16, 31, 30}) {
        a.abl = abl;
        const float t0 = time_kernel([&]() { hipLaunchKernelGGL(f0, dim3(N), dim3(256), sf, 0, a); });
        printf("fwd abl %2d   %7.1f us\n", abl, t0);
    }
    a.abl = 0;
    printf("bwd          %7.1f us\n", time_kernel([&]() { hipLaunchKernelGGL(b0, dim3(N), dim3(256), sb, 0, a); }));
    auto wg = attn_last_wgrad_kernel<bf16_t>;
    float* dW; CHECK(hipMalloc(&dW, 768 * 256 * 4 + 4096));
    for (int chunks : {4, 8, 16}) {
        const int per = (N + chunks - 1) / chunks;
        printf("wgrad chunks %2d  %7.1f us\n", chunks, time_kernel([&]() { hipLaunchKernelGGL(wg, dim3(128, chunks), dim3(256), 0, 0, (const void*)q0, (const float*)dqk, (const void*)do0,
                                                                       (const float*)ctx, (const float*)sp, dW, dW + 256 * 256, dW + 768 * 256, N, per); }));
    }
    return 0;
}
