// The last layer's compact chain kernels (lastc.cuh) alone on the chip, caches flushed between launches (development tool).
//   hipcc [-DLASTC_NW=4|8|16] --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I speaker_embedding_torch_amd/csrc tools/lastc_bench.hip -o tools/lastc_bench
//   tools/lastc_bench [n = 960]      (then: ablation runs, LastcArgs::abl bits 1 / 2 / 4 / 8 = forward without product 1-4, 16 = no hidden store,
//                                      32 / 64 / 128 / 256 = backward without product 1-4; the last two lines: backward without its column-sum atomics)
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "lastc.cuh"
using namespace ge2e;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void fill_bf16(bf16_t* p, size_t n, unsigned seed, float amp) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (bf16_t)((((mix32((unsigned)i * 2654435761u + seed) >> 8) * (1.0f / 8388608.0f)) - 1.0f) * amp);
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float amp, float off) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = off + (((mix32((unsigned)i * 2654435761u + seed) >> 8) * (1.0f / 8388608.0f)) - 1.0f) * amp;
}
__global__ void null_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void flush_kernel(float* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.0f;
}
template <typename X> X* dev(size_t n) { X* p; CHECK(hipMalloc(&p, n * sizeof(X))); CHECK(hipMemset(p, 0, n * sizeof(X))); return p; }
static bf16_t* rb(size_t n, unsigned seed, float amp) { bf16_t* p = dev<bf16_t>(n); fill_bf16<<<512, 256>>>(p, n, seed, amp); return p; }
static float* rf(size_t n, unsigned seed, float amp, float off = 0.0f) { float* p = dev<float>(n); fill_f32<<<512, 256>>>(p, n, seed, amp, off); return p; }

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 960, t = 160;
    LastcArgs a{};
    a.n = n; a.drow_mul = t; a.eps = 1e-5f;
    a.o = rb((size_t)n * 256, 1, 1.0f); a.x0 = rb((size_t)n * t * 256, 2, 1.0f); a.ldx = 256 * t;
    a.Wo = rb(65536, 3, 0.06f); a.bo = rf(256, 4, 0.1f); a.g1 = rf(256, 5, 0.1f, 1.0f); a.be1 = rf(256, 6, 0.1f);
    a.W1 = rb(262144, 7, 0.06f); a.b1 = rf(1024, 8, 0.1f); a.W2 = rb(262144, 9, 0.03f); a.b2 = rf(256, 10, 0.1f);
    a.g2 = rf(256, 11, 0.1f, 1.0f); a.be2 = rf(256, 12, 0.1f); a.gf = rf(256, 13, 0.1f, 1.0f); a.bf = rf(256, 14, 0.1f);
    a.wq = rf(65536, 15, 0.06f); a.bq = rf(256, 16, 0.1f);
    a.h1 = dev<bf16_t>((size_t)n * 256); a.rstd1 = dev<float>(n); a.f = dev<bf16_t>((size_t)n * 1024); a.h2 = dev<bf16_t>((size_t)n * 256); a.rstd2 = dev<float>(n);
    a.xhat = dev<float>((size_t)n * 256); a.rstd_f = dev<float>(n); a.zm = dev<float>((size_t)n * 256); a.nrm = dev<float>(n);
    a.emb = dev<float>((size_t)n * 256); a.emb_out = dev<float>((size_t)n * 256);
    a.d_sa = Drop{123u, 6553u, 1.0f / 0.9f}; a.d_fh = Drop{456u, 6553u, 1.0f / 0.9f}; a.d_ff = Drop{789u, 6553u, 1.0f / 0.9f};
    a.d_emb = rf((size_t)n * 256, 17, 0.01f); a.wqT = rf(65536, 18, 0.06f);
    a.W2T = rb(262144, 19, 0.03f); a.W1T = rb(262144, 20, 0.06f); a.WoT = rb(65536, 21, 0.06f);
    a.d_raw = dev<float>((size_t)n * 256);
    a.dH = dev<bf16_t>((size_t)n * 256); a.dP = dev<bf16_t>((size_t)n * 256); a.dM = dev<bf16_t>((size_t)n * 256); a.dF = dev<bf16_t>((size_t)n * 1024);
    a.dHb = dev<bf16_t>((size_t)n * 256); a.dP2 = dev<bf16_t>((size_t)n * 256); a.dM2 = dev<bf16_t>((size_t)n * 256); a.dO = dev<bf16_t>((size_t)n * 256);
    a.dgf = dev<float>(256); a.dbf = dev<float>(256); a.dg2 = dev<float>(256); a.db2 = dev<float>(256); a.dg1 = dev<float>(256); a.db1 = dev<float>(256);
    const size_t FL = (size_t)192 << 20;
    float* fl = dev<float>(FL);
    CHECK(hipDeviceSynchronize());
    auto kf = lastc_fwd_kernel<bf16_t, LASTC_NW>; auto kb = lastc_bwd_kernel<bf16_t, LASTC_NW>;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    {   float ms; std::vector<float> tn;
        for (int it = 0; it < 12; ++it) { CHECK(hipEventRecord(e0)); null_kernel<<<60, LASTC_THREADS>>>(nullptr); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); tn.push_back(ms * 1e3f); }
        std::sort(tn.begin() + 2, tn.end()); printf("empty kernel between two events: %.1f us\n", tn[7]); }
    for (int mode = 0; mode < 2 + 14; ++mode) {
        a.abl = mode < 2 ? 0 : (mode == 2 ? 1 : mode == 3 ? 2 : mode == 4 ? 4 : mode == 5 ? 8 : mode == 6 ? 16 : mode == 7 ? 15 : mode == 8 ? 31 : mode == 9 ? 32 : mode == 10 ? 64 : mode == 11 ? 128 : mode == 12 ? 256 : mode == 13 ? 480 : mode == 14 ? 480 : 0);
        if (mode == 14 || mode == 15) { a.dgf = a.dg2 = a.dg1 = nullptr; }            // 0: caches flushed before every launch, 1: back to back
        std::vector<float> tf, tb;
        for (int it = 0; it < 12; ++it) {
            float ms;
            if (mode == 0) flush_kernel<<<2048, 256>>>(fl, FL);
            CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(kf, dim3((n + 15) / 16), dim3(LASTC_THREADS), lastc_smem<LASTC_NW>(), 0, a); CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); tf.push_back(ms * 1e3f);
            if (mode == 0) flush_kernel<<<2048, 256>>>(fl, FL);
            CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(kb, dim3((n + 15) / 16), dim3(LASTC_THREADS), lastc_smem<LASTC_NW>(), 0, a); CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); tb.push_back(ms * 1e3f);
        }
        CHECK(hipGetLastError());
        std::sort(tf.begin() + 2, tf.end()); std::sort(tb.begin() + 2, tb.end());
        printf("abl %2d n %d  %s: forward chain %.1f us (median of 10, min %.1f)   backward chain %.1f us (min %.1f)\n", a.abl, n, mode == 0 ? "caches flushed" : "back to back  ",
               tf[7], tf[2], tb[7], tb[2]);
    }
    std::vector<float> eh((size_t)n * 256);
    CHECK(hipMemcpy(eh.data(), a.emb, eh.size() * 4, hipMemcpyDeviceToHost));
    double s = 0; for (int c = 0; c < 256; ++c) s += (double)eh[c] * eh[c];
    printf("|e_0|^2 = %.6f\n", s);
    return 0;
}
