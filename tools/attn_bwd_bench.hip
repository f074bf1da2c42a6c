// Attention backward variants at the headline shape, same inputs (a real forward's o / lse), outputs compared, each timed alone on the chip
// (development tool): attn_bwd_kernel as shipped and tools/attention_bwd1.cuh (every score evaluated once; measured and NOT shipped:
// profiles/r04_ab_log.txt; the QL / PIPE variants of attn_bwd_kernel measured beside it live in commit c4d4096).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I speaker_embedding_torch_amd/csrc -I tools tools/attn_bwd_bench.hip -o tools/attn_bwd_bench
//   tools/attn_bwd_bench [T] [n]       (T <= 288; default 160 frames, 960 utterances)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include "attention_bwd1.cuh"
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
static float bf(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }

template <int KT, bool PAD, bool DROP> void run(AttnArgs a, int n, bf16_t* dq_old, bf16_t* dq_new, size_t R) {
    using T = bf16_t; using G = attn::Geo<T>; constexpr int TP = 32 * KT;
    {   // a real forward: o and lse
        auto kf = attn_fwd_kernel<T, KT, PAD, DROP, 5>;
        const size_t sf = 2 * (size_t)TP * G::LD;
        CHECK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sf));
        const int nw = std::min(8, ((a.T + 15) / 16 + 1) / 2);
        hipLaunchKernelGGL(kf, dim3(n * 4), dim3(64 * nw), sf, 0, a);
        CHECK(hipDeviceSynchronize());
    }
    const int nw = std::min(8, ((a.T + 15) / 16 + 1) / 2);
    size_t sb = 2 * (size_t)TP * G::LD + 2 * TP * 4 + (DROP ? TP * (TP / 4) : 0);
    if (nw >= 5 && 3 * sb <= (size_t)160 * 1024) sb = (size_t)160 * 1024 / 3 + 1024;
    auto kb = attn_bwd_kernel<T, KT, PAD, DROP, 1, 0, 2>;
    CHECK(hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb));
    a.dqkv = dq_old;
    const float t_old = time_kernel([&]() { hipLaunchKernelGGL(kb, dim3(n * 4), dim3(64 * nw), sb, 0, a); });
    using PL = attn1::Plan<KT>;
    a.dqkv = dq_new;
    auto k1 = attn_bwd1_kernel<T, KT, PAD, DROP>;
    CHECK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, PL::SMEM));
    int occ = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k1, 64 * KT, PL::SMEM));
    const float t_new = time_kernel([&]() { hipLaunchKernelGGL(k1, dim3(n * 4), dim3(64 * KT), PL::SMEM, 0, a); });
    AttnArgs s = a; s.dqkv = dq_old;
    const float t_a = 0, t_b = 0, t_c = 0, t_d = 0;
    if (getenv("ATTN1_ABL")) {
    }
    hipLaunchKernelGGL(kb, dim3(n * 4), dim3(64 * nw), sb, 0, s);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned short> ho(R * 768), hn(R * 768);
    CHECK(hipMemcpy(ho.data(), dq_old, R * 768 * 2, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hn.data(), dq_new, R * 768 * 2, hipMemcpyDeviceToHost));
    const char* nm[3] = {"dQ", "dK", "dV"};
    (void)t_a; (void)t_b; (void)t_c; (void)t_d;
    printf("T %3d KT %d pad %d drop %d | shipped %7.1f us | single-evaluation kernel %7.1f us (QP %d, %d B LDS, %d blocks/CU) |", a.T, KT, (int)PAD, (int)DROP, t_old, t_new, PL::QP, PL::SMEM, occ);
    for (int part = 0; part < 3; ++part) {
        double num = 0, den = 0, mx = 0;
        for (size_t r = 0; r < R; ++r)
            for (int c = 0; c < 256; ++c) {
                const double x = bf(ho[r * 768 + part * 256 + c]), y = bf(hn[r * 768 + part * 256 + c]);
                num += (x - y) * (x - y); den += x * x; mx = std::max(mx, std::fabs(x - y));
            }
        printf(" %s rel %.2e max %.2e", nm[part], std::sqrt(num / std::max(den, 1e-30)), mx);
    }
    printf("\n");
}
template <int KT> void run_kt(const AttnArgs& a, int n, bf16_t* d0, bf16_t* d1, size_t R) {
    const bool pad = a.T % 32 != 0;
    AttnArgs nd = a; nd.drop = Drop{0u, 0u, 1.0f};
    if (pad) { run<KT, true, true>(a, n, d0, d1, R); run<KT, true, false>(nd, n, d0, d1, R); }
    else { run<KT, false, true>(a, n, d0, d1, R); run<KT, false, false>(nd, n, d0, d1, R); }
}
int main(int argc, char** argv) {
    const int T_ = argc > 1 ? atoi(argv[1]) : 160, n = argc > 2 ? atoi(argv[2]) : 960, D = 256; const size_t R = (size_t)n * T_;
    bf16_t *qkv, *o, *dout, *d0, *d1; float* lse;
    CHECK(hipMalloc(&qkv, R * 768 * 2)); CHECK(hipMalloc(&o, R * 256 * 2)); CHECK(hipMalloc(&dout, R * 256 * 2)); CHECK(hipMalloc(&d0, R * 768 * 2)); CHECK(hipMalloc(&d1, R * 768 * 2)); CHECK(hipMalloc(&lse, R * 16));
    fill_bf16<<<2048, 256>>>(qkv, R * 768, 1, 2.0f); fill_bf16<<<2048, 256>>>(dout, R * 256, 2, 1.0f); CHECK(hipMemset(d0, 0xFF, R * 768 * 2)); CHECK(hipMemset(d1, 0xFF, R * 768 * 2)); CHECK(hipDeviceSynchronize());
    AttnArgs a{}; a.qkv = qkv; a.o = o; a.dout = dout; a.T = T_; a.H = 4; a.D = D; a.scale = 0.125f; a.lse = lse;
    a.drop = Drop{12345u, 6553u, 1.0f / 0.9f};
    switch ((T_ + 31) / 32) {
        case 3: run_kt<3>(a, n, d0, d1, R); break;
        case 4: run_kt<4>(a, n, d0, d1, R); break;
        case 5: run_kt<5>(a, n, d0, d1, R); break;
        case 6: run_kt<6>(a, n, d0, d1, R); break;
        case 7: run_kt<7>(a, n, d0, d1, R); break;
        case 8: run_kt<8>(a, n, d0, d1, R); break;
        case 9: run_kt<9>(a, n, d0, d1, R); break;
        default: printf("T out of range\n"); return 1;
    }
    return 0;
}
