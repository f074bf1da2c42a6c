// Chained FFN forward, 16-bit storage modes: one kernel for
//
//     h2 = LayerNorm( h1 + drop2( drop1(relu(h1 . W1^T + b1)) . W2^T + b2 ) )        (reference: torch TransformerEncoderLayer
//                                                                                      built at Modules.py:25-31, post-LN)
// The [rows, 1024] hidden never round-trips HBM between the two products: unfused, FFN1 writes it (4U) and FFN2 + LN reads it
// back (4U) -- 8 of the 11U the two launches move.  Train mode still emits the hidden once (backward needs it: dW2 = dG^T f and
// the ReLU / dropout mask of dF); eval mode moves 2U instead of 11U and is MFMA-bound.
//
// Structure (one persistent 512-thread block per CU, 8 waves x 32 rows = 256 rows per pass):
//   * a wave keeps ITS 32 rows of h1 in registers as MFMA operand fragments for the whole pass (2 row tiles x 8 k-groups x 16 B
//     = 64 VGPRs) and its 32 x 256 output accumulators (128 VGPRs); nothing of the activations lives in LDS;
//   * the weights stream through a 4-slot LDS ring by LDS-DMA (global_load_lds, 16 B / lane), three stages ahead, behind a
//     counted s_waitcnt vmcnt(N) and one raw s_barrier per stage.  A stage is 32 hidden units: the W1 slice [32][256] (16 KB)
//     and the W2 slice [256][32] (16 KB), shared by the 8 waves; the stream is the same for every pass and never drains;
//   * per stage and wave: 32 MFMAs h[32 rows][32 hidden] = h1 . W1c^T, then bias + ReLU + dropout in registers, then the
//     accumulator tiles are packed straight into the operand of the next 32 MFMAs out += hidden . W2c^T (guide 3, "an
//     accumulator tile as the next MFMA's operand": the second product sums over the accumulator's ROW index);
//   * which hidden unit / output column an accumulator row holds is OURS to choose (it is the weight row a lane reads): the
//     W1 rows are read in the order that makes a lane's two 16x16 tiles 8 CONSECUTIVE hidden units (so the packed operand is
//     in natural k order for plain 16-byte W2 reads, and the hidden leaves as one 16-byte store per lane), and the W2 rows in
//     the order that gives the output accumulators exactly the column <-> lane map of the h1 operand fragments: the residual
//     is then already in this lane's registers and a row's LayerNorm statistics are a lane-quartet reduction;
//   * LDS images are lane-linear per DMA instruction; XOR swizzles go on the SOURCE address and on the fragment reads, chosen
//     so that every ds_read_b128 of the permuted rows is bank-conflict free.
#pragma once
#include "gemm_kl.cuh"

namespace ge2e {

struct FfnArgs {
    const void* A; int lda;        // h1 [M, 256] of T: GEMM operand AND residual
    const void* W1;                // [F][256] of T   (linear1.weight, k-contiguous)
    const float* b1;               // [F]
    const void* W2;                // [256][F] of T   (linear2.weight, k-contiguous)
    const float* b2;               // [256]
    void* Fo; int ldf;             // hidden out [M, F] of T (train), or null
    void* C; int ldc;              // h2 [M, 256] of T
    const float* gamma; const float* beta; float* rstd; float eps;
    Drop drop1, drop2;             // dropout after ReLU (counter row * F + col), dropout2 (counter row * 256 + col)
    int drow_mul;                  // dropout counter row = row * drow_mul (0 = 1)
    int M;
};

constexpr int FFN_NSTG = 4;        // ring slots
constexpr int FFN_D = 3;           // stages in flight ahead of the one being consumed
constexpr int FFN_SLOT = 32 * 1024;
constexpr int FFN_F = 1024;        // hidden width (4 x 256)
constexpr size_t ffn_smem() { return (size_t)FFN_NSTG * FFN_SLOT + (FFN_F + 3 * 256) * 4; }

// grid = min(CUs, ceil(M / 256)) persistent blocks of 512 threads
template <typename T, bool STORE_F>
__global__ void __launch_bounds__(512) ffn_chain_kernel(const FfnArgs p, const int npass) {
    static_assert(sizeof(T) == 2, "16-bit storage modes (bf16_t / f16_t)");
    constexpr int NCH = FFN_F / 32;              // stages per pass
    constexpr int D = FFN_D, NSTG = FFN_NSTG;
    constexpr int NDMA = 4;                      // DMA instructions per wave and stage
    constexpr int NST = STORE_F ? 2 : 0;         // hidden-store instructions per wave and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Ring = smem;                                      // [NSTG][32 KB]: W1 slice | W2 slice
    float* const B1s = (float*)(smem + NSTG * FFN_SLOT);                   // [1024]
    float* const Ls = B1s + FFN_F;                                         // b2, gamma, beta [3][256]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int G = gridDim.x, b = blockIdx.x;
    const int my = b < npass ? (npass - b + G - 1) / G : 0;
    if (my == 0) return;

    for (int q = tid; q < FFN_F; q += 512) B1s[q] = p.b1[q];
    if (tid < 256) { Ls[tid] = p.b2[tid]; Ls[256 + tid] = p.gamma[tid]; Ls[512 + tid] = p.beta[tid]; }
    __syncthreads();                              // no DMA in flight yet: an ordinary barrier

    const unsigned char* const W1g = (const unsigned char*)p.W1;
    const unsigned char* const W2g = (const unsigned char*)p.W2;
    // ---- producer: the weight stream (stage s carries hidden units 32 (s % NCH) .. +31)
    int i_c = 0, i_slot = 0;
    auto issue = [&]() {
        unsigned char* const slot = Ring + i_slot * FFN_SLOT;
#pragma unroll
        for (int u = 0; u < 2; ++u) {            // W1 slice: 16 instructions of 2 rows x 512 B; this wave: 2w, 2w + 1
            const int id = 2 * wave + u;
            const int r = 2 * id + (lane >> 5), pos = lane & 31;
            const int c = pos ^ (((r >> 3) << 2) | (r & 3));          // swizzle = the lane index i that reads row r (below)
            glds16(W1g + ((size_t)(i_c * 32 + r) * 256) * 2 + c * 16, slot + id * 1024);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {            // W2 slice: 16 instructions of 16 rows x 64 B
            const int id = 2 * wave + u;
            const int r = lane >> 2;
            const int c = (lane & 3) ^ ((lane >> 4) & 3);             // pos ^ ((r >> 2) & 3)
            glds16(W2g + ((size_t)(16 * id + r) * FFN_F + i_c * 32) * 2 + c * 16, slot + 16384 + id * 1024);
        }
        if (++i_c == NCH) i_c = 0;
        if (++i_slot == NSTG) i_slot = 0;
    };
#pragma unroll 1
    for (int q = 0; q < D; ++q) issue();

    // fragment read addresses (bytes inside a slot).  Lane (i, g) of hidden tile ht reads W1 row 8 (i >> 2) + 4 ht + (i & 3):
    // its accumulator rows 4g + r are then hidden units 8g + 4 ht + r, i.e. tiles 0 and 1 together hold the 8 CONSECUTIVE
    // units 8g .. 8g + 7.  Lane (i, g) of output tile 2 kg + h reads W2 row 32 kg + 8 (i >> 2) + 4 h + (i & 3): accumulator rows
    // 4g + r are output columns 32 kg + 8g + 4h + r -- the columns of this lane's h1 fragment af[.][kg].
    const int w1row = 8 * (i >> 2) + (i & 3);
    int w2o[2];                                                                       // + 2048 kg
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)       // row r = 8 (i >> 2) + 4 hh + (i & 3) of a 16-row DMA piece sits at chunk position g ^ ((r >> 2) & 3)
        w2o[hh] = 16384 + (w1row + 4 * hh) * 64 + ((g ^ ((2 * ((i >> 2) & 1) + hh) & 3)) << 4);

    const unsigned char* const Ag = (const unsigned char*)p.A;
    T* const Cg = (T*)p.C;
    T* const Fg = (T*)p.Fo;
    const uint32_t drm = p.drow_mul > 0 ? (uint32_t)p.drow_mul : 1u;
    const int last_row = p.M - 1;
    int c_slot = 0;

#pragma unroll 1
    for (int ps = 0; ps < my; ++ps) {
        const int m0 = (b + ps * G) * 256 + 32 * wave;
        // ---- this wave's 32 rows of h1 as operand fragments: af[rt][kg] = h1[row][32 kg + 8g .. + 7]
        u32x4 af[2][8];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            int gr = m0 + 16 * rt + i; gr = gr < last_row ? gr : last_row;
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) af[rt][kg] = *(const u32x4*)(Ag + (size_t)gr * p.lda * 2 + kg * 64 + g * 16);
        }
        // retire these ordinary loads HERE (and with them everything older): the stage loop then contains no load the compiler
        // has to wait for, so its own waits stay out of it, and the counted waits below may assume the steady state from stage 0
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) asm volatile("" : "+v"(af[rt][kg]));
        wait_vmcnt<0>();

        f32x4 oacc[2][16];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) oacc[rt][nt] = f32x4{0, 0, 0, 0};

#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            // stage c has landed once at most the (D-1) younger stages' DMAs and the D stages' worth of hidden stores issued
            // since are outstanding; everyone is done with the slot refilled below (its fragments were consumed a stage ago)
            wait_vmcnt<(D - 1) * NDMA + D * NST>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            issue();
            const unsigned char* const slot = Ring + c_slot * FFN_SLOT;
            if (++c_slot == NSTG) c_slot = 0;

            // ---- product 1: h[rt][ht] = h1 rows . W1 slice^T   (K = 256)
            f32x4 h[2][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) { h[rt][0] = f32x4{0, 0, 0, 0}; h[rt][1] = f32x4{0, 0, 0, 0}; }
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
#pragma unroll
                for (int ht = 0; ht < 2; ++ht) {
                    const u32x4 wf = lds16(slot + (w1row + 4 * ht) * 512 + (((kg * 4 + g) ^ i) << 4));
                    h[0][ht] = mma16<T>(wf, af[0][kg], h[0][ht]);
                    h[1][ht] = mma16<T>(wf, af[1][kg], h[1][ht]);
                }
                if (kg & 1) __builtin_amdgcn_sched_barrier(0);     // bound the live range of the weight fragments (4 reads in flight)
            }
            // ---- bias + ReLU + dropout in registers; the packed tiles are the hidden's storage-type values
            u32x4 hp[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int row = m0 + 16 * rt + i;
#pragma unroll
                for (int ht = 0; ht < 2; ++ht) {
                    const int col = c * 32 + 8 * g + 4 * ht;
                    h[rt][ht] += *(const f32x4*)(B1s + col);
                    (void)relu_drop_apply4(p.drop1, (uint32_t)row * drm * (uint32_t)FFN_F + (uint32_t)col, h[rt][ht]);
                }
                hp[rt] = pack_acc<T>(h[rt][0], h[rt][1]);      // 8 consecutive hidden units 32c + 8g .. + 7 of row (rt, i)
                if constexpr (STORE_F) {
                    if (row < p.M) __builtin_nontemporal_store(hp[rt], (u32x4*)((unsigned char*)Fg + ((size_t)row * p.ldf + c * 32 + 8 * g) * 2));
                }
            }
            // ---- product 2: out[rt][nt] += hidden . W2 slice^T   (K = these 32 hidden units, one MFMA deep)
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) {
                const u32x4 wf = lds16(slot + w2o[nt & 1] + (nt >> 1) * 2048);
                oacc[0][nt] = mma16<T>(wf, hp[0], oacc[0][nt]);
                oacc[1][nt] = mma16<T>(wf, hp[1], oacc[1][nt]);
                if ((nt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- pass epilogue: v = residual + drop2(out + b2); LayerNorm over the row (this lane: 64 of its 256 columns, the
        // other three quarters sit on the lanes with the same i); rows leave as 16-byte pieces per lane (64 B per row and kg)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = m0 + 16 * rt + i;
            float sm = 0.0f;
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
                const T* const res = (const T*)&af[rt][kg];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int col = 32 * kg + 8 * g + 4 * hh;
                    f32x4 v = oacc[rt][2 * kg + hh] + *(const f32x4*)(Ls + col);
                    drop_apply4(p.drop2, (uint32_t)row * drm * 256u + (uint32_t)col, v);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += to_f32(res[4 * hh + r]);
                    oacc[rt][2 * kg + hh] = v;
                    sm += (v[0] + v[1]) + (v[2] + v[3]);
                }
            }
            const float mean = cross4_sum(sm) * (1.0f / 256.0f);
            float q2 = 0.0f;
#pragma unroll
            for (int nt = 0; nt < 16; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float dlt = oacc[rt][nt][r] - mean; q2 += dlt * dlt; }
            const float rs = 1.0f / sqrtf(cross4_sum(q2) * (1.0f / 256.0f) + p.eps);
            if (p.rstd && g == 0 && row < p.M) p.rstd[row] = rs;
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
                f32x4 o2[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int col = 32 * kg + 8 * g + 4 * hh;
                    const f32x4 ga = *(const f32x4*)(Ls + 256 + col), be = *(const f32x4*)(Ls + 512 + col);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o2[hh][r] = (oacc[rt][2 * kg + hh][r] - mean) * rs * ga[r] + be[r];
                }
                if (row < p.M)
                    __builtin_nontemporal_store(pack_acc<T>(o2[0], o2[1]), (u32x4*)((unsigned char*)Cg + ((size_t)row * p.ldc + 32 * kg + 8 * g) * 2));
            }
        }
    }
    wait_vmcnt<0>();                                  // the D stages issued past the end land before the block's LDS is released
}

}  // namespace ge2e
