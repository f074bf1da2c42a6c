// Split-K GEMM for the LAST layer's compact rows (one row per utterance: M ~ 1e3), 16-bit storage modes, N = 256, K = 1024:
//     forward   h2 = LayerNorm(h1 + drop(f . W2^T + b2))            (reference: torch TransformerEncoderLayer built at Modules.py:25-31)
//     backward  dH1 = dP + dF . W1
// At this height the stage-stream kernel (gemm_kl.cuh) and the tiled kernel (gemm.cuh) run on 8-16 blocks, each walking the whole
// K = 1024 and streaming all of W (512 KB) alone: 35 us forward, 22 us backward for 0.5 GFLOP.  Here the product is cut into
// 64-row x K/KS pieces (60 blocks at 960 rows, KS = 4): a block's four waves take 64 output columns each, read their A and W
// operand fragments straight from global memory / L2 (16 bytes per lane, no LDS: there is no reuse to stage for) and leave an
// fp32 partial tile; a second small launch sums the KS partials of a row and applies the epilogue (bias, dropout, residual,
// LayerNorm -- or the plain addend).
#pragma once
#include "gemm.cuh"

namespace ge2e {

constexpr int SK_KS = 4;          // K slices
constexpr int SK_MAX_M = 8192;    // rows up to which the launcher prefers this path (the partial buffer is sized for the utterance count)

struct GemmSkArgs {
    const void* A; int lda;      // [M, K] of T
    const void* W; int ldw;      // [256, K] of T
    float* part;                 // [KS][M][256]
    int M, K, KS;
};

// grid = (ceil(M / 64), KS), 256 threads
template <typename T>
__global__ void __launch_bounds__(256) gemm_sk_kernel(const GemmSkArgs p) {
    static_assert(sizeof(T) == 2, "16-bit storage modes");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 64, ks = blockIdx.y;
    const int kper = p.K / p.KS, k0 = ks * kper;
    const unsigned char* ar[4];
    const unsigned char* wr[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int m = min(m0 + 16 * rt + i, p.M - 1);
        ar[rt] = (const unsigned char*)p.A + ((size_t)m * p.lda + k0 + 8 * g) * 2;
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wr[nt] = (const unsigned char*)p.W + ((size_t)(64 * wave + 16 * nt + i) * p.ldw + k0 + 8 * g) * 2;
    f32x4 acc[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[nt][rt] = f32x4{0, 0, 0, 0};
#pragma unroll 2
    for (int kk = 0; kk < kper; kk += 32) {
        u32x4 af[4], wf[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) af[rt] = *(const u32x4*)(ar[rt] + kk * 2);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[nt] = *(const u32x4*)(wr[nt] + kk * 2);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[nt][rt] = mma16<T>(wf[nt], af[rt], acc[nt][rt]);     // C^T: rows = output columns, lane column = row m
    }
    // acc[nt][rt][r] = C[m0 + 16 rt + i][64 wave + 16 nt + 4 g + r]
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int m = m0 + 16 * rt + i;
        if (m < p.M) {
            float* dst = p.part + ((size_t)ks * p.M + m) * 256 + 64 * wave + 4 * g;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) *(f32x4*)(dst + 16 * nt) = acc[nt][rt];
        }
    }
}

struct SkEpiArgs {
    const float* part; int M, KS;
    const float* bias;                           // LN: [256]
    const void* R; int ldr;                      // residual (LN) / addend
    void* C; int ldc;                            // [M, 256] of T
    const float* gamma; const float* beta; float* rstd; float eps;
    Drop drop; int drow_mul;
};

// one wave per row (4 rows per block): lane = 4 consecutive columns
template <typename T, bool LN>
__global__ void __launch_bounds__(256) gemm_sk_epi_kernel(const SkEpiArgs p) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;                      // wave-uniform
    const int c0 = 4 * lane;
    f32x4 v = f32x4{0, 0, 0, 0};
    for (int ks = 0; ks < p.KS; ++ks) v += *(const f32x4*)(p.part + ((size_t)ks * p.M + row) * 256 + c0);
    const f32x4 r4 = load4((const T*)p.R + (size_t)row * p.ldr + c0);
    T* const crow = (T*)p.C + (size_t)row * p.ldc + c0;
    if constexpr (LN) {
        v += *(const f32x4*)(p.bias + c0);
        const uint32_t drm = p.drow_mul > 0 ? (uint32_t)p.drow_mul : 1u;
        drop_apply4(p.drop, (uint32_t)row * drm * 256u + (uint32_t)c0, v);
        v += r4;
        const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 256.0f);
        float q2 = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = v[r] - mean; q2 += d * d; }
        const float rs = 1.0f / sqrtf(wave_sum(q2) * (1.0f / 256.0f) + p.eps);
        if (p.rstd && lane == 0) p.rstd[row] = rs;
        const f32x4 ga = *(const f32x4*)(p.gamma + c0), be = *(const f32x4*)(p.beta + c0);
        store4(crow, (v[0] - mean) * rs * ga[0] + be[0], (v[1] - mean) * rs * ga[1] + be[1], (v[2] - mean) * rs * ga[2] + be[2], (v[3] - mean) * rs * ga[3] + be[3]);
    } else {
        v += r4;
        store4(crow, v[0], v[1], v[2], v[3]);
    }
}

}  // namespace ge2e
