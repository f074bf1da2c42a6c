// Streaming k-loop GEMM for the wide-K, 256-column products (bf16 mode).  In the library it runs FFN2 + LayerNorm
// (K = 1024, forward); the addend epilogue (EPI_ADD: dHb = dP + dF.W1 with K = 1024, dHa = dP + dQKV.Wqkv with K = 768) is
// kept for tools/gemm_bench.hip -- alone it ties the tiled kernel (131 / 107 vs 136 / 109 us) and it needs a whole CU's
// LDS, which the backward cannot give it beside the weight gradients.
//
//   C[M, 256] = epilogue(A[M, K] . W[256, K]^T  (+ R[M, 256]))
//
// These read 3-4x more than they write and W (384-512 KB) cannot stay in registers, so every row tile re-streams W
// from L2.  What the tiled kernel in gemm.cuh loses here is (a) W traffic through the CU's load path 4x the A
// traffic at 64-row tiles, (b) one k-step in flight, (c) a cold pipeline at every tile start.  This kernel:
//   * 128-row x 256-column tiles, 8 waves (2 x 4, 64 x 64 each): W traffic per A byte halves;
//   * one persistent block per CU walks its tiles as ONE stream of stages through a 5-slot LDS ring filled by
//     LDS-DMA (global_load_lds, 16 B per lane), four stages ahead, behind counted s_waitcnt vmcnt(N) and one raw
//     s_barrier per stage.  A k-stage is a 32-wide K slice of the A tile (8 KB) and of W (16 KB); after a tile's
//     k-stages come four R-stages carrying its addend rows (residual / upstream gradient) as per-wave slabs, and the
//     next tile's k-stages follow immediately, so the stream never drains at a tile boundary;
//   * the epilogue (bias, dropout, residual, LayerNorm over the row that four waves share through one LDS exchange,
//     or the plain addend) runs from registers; rows leave as 128-byte segments through wave-private LDS slabs.
// LDS images are lane-linear per DMA instruction; XOR swizzles go on the SOURCE address and on the fragment reads.
#pragma once
#include "gemm_ws.cuh"

namespace ge2e {

constexpr int KL_NSTG = 5;     // ring slots (24 KB each)
constexpr int KL_D = 4;        // stages in flight ahead of the one being consumed
constexpr int KL_SLOT = 24 * 1024;

template <int EPI>
constexpr size_t gemm_kl_smem() {
    return (size_t)KL_NSTG * KL_SLOT + 8 * 2048 + (EPI == EPI_LN ? 3 * 1024 + 2 * 2 * 4 * 64 * 8 : 0);
}

// grid = min(CUs, ceil(M / 128)) persistent blocks of 512 threads
template <typename T, int EPI, int K_, int ABL = 0>
__global__ void __launch_bounds__(512) gemm_kl_kernel(const GemmArgs p, const int ntiles) {
    static_assert(sizeof(T) == 2, "16-bit storage modes (bf16_t / f16_t)");
    static_assert(EPI == EPI_LN || EPI == EPI_ADD, "epilogues with an addend tile");
    constexpr int KS = K_ / 32;                  // k-stages per tile
    constexpr int RST = 4;                       // R-stages per tile (32 rows each)
    constexpr int SPT = KS + RST;                // stages per tile
    constexpr int D = KL_D, NSTG = KL_NSTG;
    constexpr int STORES = 8;                    // output store instructions per wave and tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Ring = smem;                                        // [NSTG][24 KB]
    unsigned char* const Os = smem + NSTG * KL_SLOT;                         // [8 waves][2 KB]
    float* const Ls = (float*)(Os + 8 * 2048);                               // EPI_LN: bias, gamma, beta [3][256]
    float* const Xs = Ls + 3 * 256;                                          // EPI_LN: [2][2 wr][4 wc][64 rows] x (mean, M2)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int i = lane & 15, g = lane >> 4;
    const int G = gridDim.x, b = blockIdx.x;
    const int my = b < ntiles ? (ntiles - b + G - 1) / G : 0;
    if (my == 0) return;
    const int total = my * SPT;

    if constexpr (EPI == EPI_LN) {
        if (tid < 256) { Ls[tid] = p.bias[tid]; Ls[256 + tid] = p.gamma[tid]; Ls[512 + tid] = p.beta[tid]; }
        __syncthreads();                          // no DMA in flight yet
    }

    const unsigned char* const Ag = (const unsigned char*)p.A;
    const unsigned char* const Wg = (const unsigned char*)p.W;
    const unsigned char* const Rg = (const unsigned char*)p.R;
    const int last_row = p.M - 1;

    // ---- producer side: stage counter -> (tile, stage in tile), ring slot
    int i_tile = 0, i_s = 0, i_slot = 0;
    auto issue = [&]() {
        const int m0 = (ABL & 2) ? 0 : (b + (i_tile < my ? i_tile : my - 1) * G) * 128;
        unsigned char* const slot = Ring + i_slot * KL_SLOT;
        if constexpr ((ABL & 4) == 0) {
        if (i_s < KS) {
            // k-stage: 24 DMA instructions of 16 rows x 64 B (ids 0-7: A slice, 8-23: W slice); this wave: 3w, 3w+1, 3w+2
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int id = 3 * wave + u;                     // wave-uniform
                const int r = lane >> 2;                         // row inside the 16-row piece
                const int c = (lane & 3) ^ ((lane >> 4) & 3);    // swizzled chunk: pos ^ ((row >> 2) & 3)
                const unsigned char* src;
                if (id < 8) {
                    int gr = m0 + 16 * id + r; gr = gr < last_row ? gr : last_row;
                    src = Ag + ((size_t)gr * p.lda + i_s * 32) * 2 + c * 16;
                } else {
                    src = Wg + ((size_t)(16 * (id - 8) + r) * p.ldw + i_s * 32) * 2 + c * 16;
                }
                glds16(src, slot + id * 1024);
            }
        } else {
            // R-stage j: rows 32j..32j+31 as 8 slabs (m-tile half x column quarter) of 16 rows x 128 B; 16 instructions
            const int j = i_s - KS;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int id = 2 * wave + (u & 1);               // third instruction repeats the first (fixed count per stage)
                const int slab = id >> 1, half = id & 1;
                const int r = 8 * half + (lane >> 3);
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                int gr = m0 + 32 * j + 16 * (slab >> 2) + r; gr = gr < last_row ? gr : last_row;
                glds16(Rg + ((size_t)gr * p.ldr + 64 * (slab & 3)) * 2 + c * 16, slot + id * 1024);
            }
        }
        }
        if (++i_s == SPT) { i_s = 0; ++i_tile; }
        if (++i_slot == NSTG) i_slot = 0;
    };

#pragma unroll 1
    for (int q = 0; q < D; ++q) issue();

    f32x4 acc[4][4];
    unsigned char* const Ow = Os + wave * 2048;
    T* const Cg = (T*)p.C;
    const uint32_t drm = p.drow_mul > 0 ? (uint32_t)p.drow_mul : 1u;
    int c_slot = 0;

    // make the next stage of the stream readable: its DMAs have landed (counted wait: only the D-1 younger stages' 3 DMAs
    // per wave and -- within D stages of a tile end -- that tile's output stores may still be outstanding), everyone is
    // done with the stage before it (its fragments are in registers), whose slot is refilled D stages ahead
    int w_tl = 0, w_s = 0;
    auto advance = [&]() -> const unsigned char* {
        if (w_tl > 0 && w_s < D) wait_vmcnt<3 * (D - 1) + STORES>();
        else wait_vmcnt<3 * (D - 1)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue();
        const unsigned char* const slot = Ring + c_slot * KL_SLOT;
        if (++c_slot == NSTG) c_slot = 0;
        if (++w_s == SPT) { w_s = 0; ++w_tl; }
        return slot;
    };
    const int sw = (g ^ ((i >> 2) & 3)) << 4;
    auto read_frags = [&](const unsigned char* slot, u32x4* af, u32x4* wf) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[mt] = lds16(slot + (64 * wr + 16 * mt + i) * 64 + sw);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[nt] = lds16(slot + 8192 + (64 * wc + 16 * nt + i) * 64 + sw);
    };
    auto mma_stage = [&](const u32x4* af, const u32x4* wf) {
        if constexpr (ABL & 1) return;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mma16<T>(wf[nt], af[mt], acc[mt][nt]);   // [r] = C[row 64wr+16mt+i][col 64wc+16nt+4g+r]
    };

#pragma unroll 1
    for (int tl = 0; tl < my; ++tl) {
        const int m0 = (b + tl * G) * 128;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0, 0, 0, 0};
        // k-stages, software-pipelined inside the wave: the fragment reads of stage s+1 are in flight under the MFMAs of s
        u32x4 fa0[4], fw0[4], fa1[4], fw1[4];
        const unsigned char* slot = advance();
        read_frags(slot, fa0, fw0);
#pragma unroll 1
        for (int s = 0; s < KS; s += 2) {
            // MFMAs first, then the next stage's fragment reads: the reads run under the matrix pipe's backlog (the
            // other order makes hipcc wait for them before the first MFMA)
            slot = advance();
            mma_stage(fa0, fw0);
            __builtin_amdgcn_sched_barrier(0);
            read_frags(slot, fa1, fw1);
            slot = advance();                               // s + 2 == KS: this is the first R-stage
            mma_stage(fa1, fw1);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < KS) read_frags(slot, fa0, fw0);
        }
#pragma unroll 1
        for (int j = 0; j < RST; ++j) {
            if (j > 0) slot = advance();
            if ((j >> 1) == wr) {
                // this wave's m-tiles 2(j&1), 2(j&1)+1: v = acc (+ bias, dropout) + R
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const unsigned char* const slab = slot + (h2 * 4 + wc) * 2048;
#pragma unroll
                    for (int mtl = 0; mtl < 2; ++mtl) {
                        // static register index: mt = 2*(j&1) + h2 must be a compile-time constant -> both parities unrolled
                        if ((j & 1) != mtl) continue;
                        const int mt = 2 * mtl + h2;
                        const int row = m0 + 64 * wr + 16 * mt + i;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const int lc = 64 * wc + 16 * nt + 4 * g;
                            const int so = i * 128 + (((2 * nt + (g >> 1)) ^ ((i >> 1) & 7)) << 4) + (g & 1) * 8;
                            f32x4 v = acc[mt][nt];
                            if constexpr (EPI == EPI_LN) {
                                v += *(const f32x4*)(Ls + lc);
                                drop_apply4(p.drop, (uint32_t)row * drm * 256u + (uint32_t)lc, v);
                            }
                            v += load4((const T*)(slab + so));
                            acc[mt][nt] = v;
                        }
                    }
                }
            }
            if (j < RST - 1) continue;
            // ---- tile complete
            float mean[4], rstd[4];
            if constexpr (EPI == EPI_LN) {
                float* const xs = Xs + (tl & 1) * 1024 + wr * 512;           // [4 wc][64 rows][2]
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    float sm = 0.0f;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) sm += (acc[mt][nt][0] + acc[mt][nt][1]) + (acc[mt][nt][2] + acc[mt][nt][3]);
                    const float mw = cross4_sum(sm) * (1.0f / 64.0f);
                    float qw = 0.0f;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float d = acc[mt][nt][r] - mw; qw += d * d; }
                    qw = cross4_sum(qw);
                    if (g == 0) { xs[(wc * 64 + mt * 16 + i) * 2] = mw; xs[(wc * 64 + mt * 16 + i) * 2 + 1] = qw; }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    float mws[4], m2 = 0.0f, mu = 0.0f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) { mws[w] = xs[(w * 64 + mt * 16 + i) * 2]; mu += mws[w]; m2 += xs[(w * 64 + mt * 16 + i) * 2 + 1]; }
                    mu *= 0.25f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) m2 += 64.0f * (mws[w] - mu) * (mws[w] - mu);
                    mean[mt] = mu;
                    rstd[mt] = 1.0f / sqrtf(m2 * (1.0f / 256.0f) + p.eps);
                }
                if (p.rstd) {
                    // one store instruction per wave: wave (wr, wc) writes the 16 rows of its m-tile wc
                    float rs = rstd[0];
                    rs = wc == 1 ? rstd[1] : rs; rs = wc == 2 ? rstd[2] : rs; rs = wc == 3 ? rstd[3] : rs;
                    const int row = m0 + 64 * wr + 16 * wc + i;
                    if (g == 0 && row < p.M) p.rstd[row] = rs;
                }
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int lc = 64 * wc + 16 * nt + 4 * g;
                    const int so = i * 128 + (((2 * nt + (g >> 1)) ^ ((i >> 1) & 7)) << 4) + (g & 1) * 8;
                    f32x4 v = acc[mt][nt];
                    if constexpr (EPI == EPI_LN) {
                        const f32x4 ga = *(const f32x4*)(Ls + 256 + lc), be = *(const f32x4*)(Ls + 512 + lc);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (v[r] - mean[mt]) * rstd[mt] * ga[r] + be[r];
                    }
                    store4((T*)(Ow + so), v[0], v[1], v[2], v[3]);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int r = u * 8 + (lane >> 3), c = lane & 7;
                    const u32x4 o = lds16(Ow + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
                    const int row = m0 + 64 * wr + 16 * mt + r;
                    if (row < p.M) __builtin_nontemporal_store(o, (u32x4*)((unsigned char*)Cg + ((size_t)row * p.ldc + 64 * wc) * 2 + c * 16));
                }
            }
        }
    }
}

}  // namespace ge2e
