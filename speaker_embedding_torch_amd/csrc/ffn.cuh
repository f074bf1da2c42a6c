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
//     counted s_waitcnt vmcnt(N) and one raw s_barrier per stage (placed between the two products of a stage: it publishes
//     the NEXT stage, whose first fragments are then prefetched under this stage's last MFMAs).  A stage is 32 hidden units: the W1 slice [32][256] (16 KB)
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
//
// BWD = true (round 3) is the same machine run backwards through the block, one launch for what were three (norm2 backward, the dF
// GEMM, the dH1 GEMM):
//     dP = LayerNorm2'(dH; h2)      dM = drop2'(dP)      dF = (dM . W2) o relu'/drop1' (the forward's mask bits)      dH1 = dP + dF . W1
//   * pass prologue: the wave's 32 rows of dH and h2 arrive in the operand-fragment layout (row per lane, 64 of its 256 columns; the
//     row statistics are lane-quartet reductions as in the forward's LayerNorm); dM leaves as 16-byte pieces (the weight gradient
//     reads it) and IS the first product's operand; dP never leaves the chip: it is the start value of the dH1 accumulators;
//   * "W1" is linear2.weight transposed ([F][256]), "W2" is linear1.weight transposed ([256][F]): the stage stream is unchanged;
//   * hidden epilogue: the fp32 tile is masked by the forward's bits (one byte per lane, row and stage; four stages' bytes arrive as
//     ONE 4-byte-per-lane LDS-DMA per row tile into a wave-private ring -- an ordinary load inside the stage loop would make the
//     compiler drain the weight ring in front of its use), packed, stored once (dF, for the weight gradient) and fed to product 2;
//   * pass epilogue: store dH1.
// The LayerNorm's dgamma / dbeta (column sums over ALL rows) are left to ln_colsum_kernel on the weight-gradient stream.
#pragma once
#include "gemm_kl.cuh"

namespace ge2e {

typedef __attribute__((ext_vector_type(16))) float ffn_f32x16;      // (FFN_MF32_PROBE only)

struct FfnArgs {
    const void* A; int lda;        // h1 [M, 256] of T: GEMM operand AND residual
    const void* W1;                // [F][256] of T   (linear1.weight, k-contiguous)
    const float* b1;               // [F]
    const void* W2;                // [256][F] of T   (linear2.weight, k-contiguous)
    const float* b2;               // [256]
    void* Fo; int ldf;             // hidden out [M, F] of T (train), or null
    unsigned char* Mb;             // train: [M][F / 8] "kept and positive" bits of the hidden, read by the dF GEMM of the backward; hidden
                                   // unit u sits in byte ffn_mask_byte(u), bit u & 7 of its row (gemm_ws.cuh EPI_MASKBITS reads the same map)
    void* C; int ldc;              // h2 [M, 256] of T
    const float* gamma; const float* beta; float* rstd; float eps;
    Drop drop1, drop2;             // dropout after ReLU (counter row * F + col), dropout2 (counter row * 256 + col)
    int drow_mul;                  // dropout counter row = row * drow_mul (0 = 1)
    int M;
    // BWD only.  A / lda: dH (the block output's gradient); Y: h2 (the saved LayerNorm output); rstd: read; dM: [M, 256], drop2' of
    // norm2's input gradient dP (dP itself never leaves the chip); W1 = linear2.weight^T, W2 = linear1.weight^T; Fo = dF; Mb is READ;
    // C = dH1; drop1.scale masks dF, drop2 masks dM.
    const void* Y; void* dM;
};

// Mask-bit layout of a row (128 bytes): the chained kernel's lane (i, g) produces, stage after stage (32 units each), the bits of
// units 32 c + 8 g .. + 7; four consecutive stages are gathered into one 32-bit store, so those four bytes are contiguous:
__host__ __device__ constexpr int ffn_mask_byte(int u) { return 16 * (u >> 7) + 4 * ((u >> 3) & 3) + ((u >> 5) & 3); }
constexpr int FFN_SLOT = 32 * 1024;
constexpr int FFN_F = 1024;        // hidden width (4 x 256)
// Two block shapes:
//   WV = 8: one 512-thread block per CU, 256 rows per pass, 4-slot ring three stages ahead, the barrier between the two products
//           of a stage (it publishes the NEXT stage, whose first fragments are prefetched under this stage's last MFMAs);
//   WV = 4: TWO independent 256-thread blocks per CU, 128 rows per pass each, a 2-slot ring one stage ahead with the barrier at the
//           top of the stage.  The two waves of a SIMD then belong to different blocks: no barrier couples them, so one block's
//           vector-ALU phases, pass prologue (h1 loads) and pass epilogue (LayerNorm, stores) run under the other's MFMAs.  The
//           weights stream twice per CU (2 x 32 KB per stage time: still under half of the CU's L2 ingest rate).
// compile-time loop: f(std::integral_constant<int, B>{}), ..., f(std::integral_constant<int, E - 1>{})
template <int B, int E, typename F> __device__ __forceinline__ void ffn_static_for(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); ffn_static_for<B + 1, E>(f); }
}
// LDS fragment read / counted wait as opaque instructions: the compiler's own wait insertion drains the LDS queue (lgkmcnt(0))
// every few fragments of a read-ahead ring instead of waiting for the oldest read only; these keep FRD reads in flight.  The wait
// names the fragment it releases, so the MFMAs that consume it stay behind it.  (Compiler-placed LDS operations in between only
// make these waits stricter: the counter is in order, and more younger operations mean the oldest retires earlier than counted.)
template <int OFF> __device__ __forceinline__ void ffn_lds_read(u32x4& f, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void ffn_lds_wait(u32x4& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }

template <int WV> constexpr int ffn_nstg() { return WV == 8 ? 4 : 2; }
template <int WV> constexpr size_t ffn_smem() { return (size_t)ffn_nstg<WV>() * FFN_SLOT + (FFN_F + 3 * 256) * 4; }
// BWD: the same table area (gamma, beta, 1 / gamma in place of b2, gamma, beta) + the mask-word ring: [wave][slot 0 / 1][row tile][64 lanes]
template <int WV> constexpr size_t ffn_bwd_smem() { return ffn_smem<WV>() + (size_t)WV * 2 * 2 * 256; }

// grid = persistent blocks of 512 threads, one per CU at most, sized so that every block runs the same number of passes
// ABL (development only, tools/ffn_bench.hip): 1 no MFMA, 2 no DMA, 4 no fragment reads, 8 no barrier, 16 no hidden epilogue, 128 no hidden store
// STAGGER: waves 4-7 lag by one product (see the stage loop).  Measured at 153,600 rows: train (dropout hashes + hidden store in the
// hidden epilogue) 310 -> 279 us with it, eval (a light epilogue) 215 -> 243 us: the launcher staggers train mode only.
template <typename T, bool STORE_F, int ABL = 0, bool STAGGER = STORE_F, int WV = 8, bool BWD = false>
__global__ void __launch_bounds__(64 * WV, WV == 8 ? 1 : 2) ffn_chain_kernel(const FfnArgs p, const int npass) {
    static_assert(sizeof(T) == 2, "16-bit storage modes (bf16_t / f16_t)");
    static_assert(WV == 8 || WV == 4, "waves per block");
    static_assert(!BWD || STORE_F, "the backward chain stores dF");
    constexpr int NCH = FFN_F / 32;              // stages per pass
    constexpr int NSTG = ffn_nstg<WV>(), D = NSTG - 1;
    constexpr int NDMA = 32 / WV;                // DMA instructions per wave and stage
    constexpr int NST = STORE_F && !(ABL & 128) ? 2 : 0;   // hidden-store instructions per wave and stage that the counted waits rely on (the two
                                                 // 16-byte value stores; every fourth stage adds two mask words: a count may be too small)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Ring = smem;                                      // [NSTG][32 KB]: W1 slice | W2 slice
    float* const B1s = (float*)(smem + NSTG * FFN_SLOT);                   // [1024]
    float* const Ls = B1s + FFN_F;                                         // b2, gamma, beta [3][256]   (BWD: gamma, beta, 1 / gamma)
    [[maybe_unused]] unsigned char* const Mr = (unsigned char*)(Ls + 3 * 256);   // BWD: mask words, this wave's part at + wave * 1024

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const bool lag = STAGGER && WV == 8 && wave >= 4;       // wave-uniform (wave comes from readfirstlane)
    const int G = gridDim.x, b = blockIdx.x;
    const int my = b < npass ? (npass - b + G - 1) / G : 0;
    if (my == 0) return;

    if constexpr (BWD) {
        if (tid < 256) { const float gq = p.gamma[tid]; Ls[tid] = gq; Ls[256 + tid] = p.beta[tid]; Ls[512 + tid] = gq != 0.0f ? 1.0f / gq : 0.0f; }
    } else {
    // (train: the bias table is pre-multiplied by dropout1's 1 / (1 - p): the hidden epilogue forms (acc + b) / (1 - p) as ONE fma per element)
    for (int q = tid; q < FFN_F; q += 64 * WV) B1s[q] = p.b1[q] * (STORE_F ? p.drop1.scale : 1.0f);
    if (tid < 256) { Ls[tid] = p.b2[tid]; Ls[256 + tid] = p.gamma[tid]; Ls[512 + tid] = p.beta[tid]; }
    }
    __syncthreads();                              // no DMA in flight yet: an ordinary barrier

    const unsigned char* const W1g = (const unsigned char*)p.W1;
    const unsigned char* const W2g = (const unsigned char*)p.W2;
    // ---- producer: the weight stream (stage s carries hidden units 32 (s % NCH) .. +31)
    // per-lane source offsets (32-bit, loop-invariant) on top of wave-uniform bases.  One VGPR per table: the second
    // instruction of a wave covers W1 rows + 2 (same swizzle with bit 1 of the row flipped: chunk ^ 2, + 1024 B) and W2 rows + 16
    // (same swizzle: a wave-uniform + 16 rows).
    // WV = 8: wave w carries W1 pieces 2w, 2w + 1 and W2 pieces 2w, 2w + 1.  WV = 4: waves 0, 1 carry the 16 W1 pieces (8 each),
    // waves 2, 3 the 16 W2 pieces; piece u of a wave is the first one + a wave-uniform step, the swizzle bit pattern follows the row.
    const int dr1 = (WV == 8 ? 4 * wave : 16 * (wave & 1)) + (lane >> 5);
    const unsigned dma1 = (unsigned)(dr1 * 512 + (((lane & 31) ^ (((dr1 >> 3) << 2) | (dr1 & 3))) << 4));    // W1 slice row dr1, swizzled chunk (= the lane index i that reads it)
    const unsigned dma2 = (unsigned)(((WV == 8 ? 32 * wave : 128 * (wave & 1)) + (lane >> 2)) * (FFN_F * 2) + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));   // W2 row, chunk pos ^ ((r2 >> 2) & 3)
    int i_c = 0, i_slot = 0;
    auto issue = [&]() {
        unsigned char* const slot = Ring + i_slot * FFN_SLOT;
        if constexpr ((ABL & 2) == 0) {
        const unsigned char* const s1 = W1g + (size_t)i_c * (32 * 512);       // wave-uniform: hidden units 32 i_c ..
        const unsigned char* const s2 = W2g + (size_t)i_c * 64;
        unsigned d1 = dma1, d2 = dma2;               // opaque 32-bit copies: the offsets stay ONE register each (not hoisted 64-bit pairs)
        asm volatile("" : "+v"(d1), "+v"(d2));
        if constexpr (WV == 8) {
            glds16(s1 + d1, slot + (2 * wave) * 1024);                                     // rows 4w, 4w + 1 (512 B each)
            glds16(s1 + ((d1 ^ 32u) + 1024u), slot + (2 * wave + 1) * 1024);               // rows 4w + 2, 4w + 3
            glds16(s2 + d2, slot + 16384 + (2 * wave) * 1024);                             // 16 rows x 64 B
            glds16(s2 + 16 * (FFN_F * 2) + d2, slot + 16384 + (2 * wave + 1) * 1024);      // the next 16 rows
        } else if (wave < 2) {
            // W1 rows 16 w + 2 u + (lane >> 5), u = 0..7: the swizzle ((r >> 3) << 2) | (r & 3) of row r0 + 2u differs from r0's in
            // bit 1 (u odd: chunk ^ 2) and in bit 2 (u >= 4: chunk ^ 4)
#pragma unroll
            for (int u = 0; u < 8; ++u)
                glds16(s1 + ((d1 ^ (unsigned)((((u & 1) << 1) | ((u >> 2) << 2)) << 4)) + (unsigned)(u * 1024)), slot + (8 * wave + u) * 1024);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u)              // W2 rows 128 (w - 2) + 16 u + (lane >> 2): the swizzle depends on the row inside the piece only
                glds16(s2 + (size_t)u * (16 * FFN_F * 2) + d2, slot + 16384 + (8 * (wave - 2) + u) * 1024);
        }
        }
        if (++i_c == NCH) i_c = 0;
        if (++i_slot == NSTG) i_slot = 0;
    };
#pragma unroll 1
    for (int q = 0; q < D; ++q) issue();

    // fragment read addresses.  Lane (i, g) of hidden tile ht reads W1 row 8 (i >> 2) + 4 ht + (i & 3): its accumulator rows
    // 4g + r are then hidden units 8g + 4 ht + r, i.e. tiles 0 and 1 together hold the 8 CONSECUTIVE units 8g .. 8g + 7.  Lane
    // (i, g) of output tile 2 kg + h reads W2 row 32 kg + 8 (i >> 2) + 4 h + (i & 3): accumulator rows 4g + r are output columns
    // 32 kg + 8g + 4h + r -- the columns of this lane's h1 fragment af[.][kg].
    // Every address is (one of six lane-dependent bases) + (a compile-time constant that fits the ds_read offset field):
    //   W1 (kg, ht): chunk position (4 kg + g) ^ i = 16 (kg >> 2) + 4 ((kg & 3) ^ (i >> 2)) + (g ^ (i & 3))
    //                -> base w1b[kg & 3] + 2048 ht + 256 (kg >> 2)
    //   W2 (kg, h) : base w2b[h] + 2048 kg
    // (32 separately materialised lane addresses would not fit the register file next to 192 accumulator / operand registers.)
    const int w1row = 8 * (i >> 2) + (i & 3);
    const unsigned w1c = (unsigned)(w1row * 512 + ((g ^ (i & 3)) << 4));                      // W1: + ((kq ^ (i >> 2)) << 6) per k-group class
    // W2 row r = 8 (i >> 2) + 4 hh + (i & 3) of a 16-row DMA piece sits at chunk position g ^ ((r >> 2) & 3) = g ^ (2 ((i >> 2) & 1) + hh):
    // hh = 1 is hh = 0 with chunk bit 0 flipped (^ 16 B) and 4 rows (256 B) further
    const unsigned w2c = (unsigned)(16384 + w1row * 64 + ((g ^ (2 * ((i >> 2) & 1))) << 4));
    const int w1q = i >> 2;

    auto mm = [&](const u32x4& w, const u32x4& a, f32x4 acc) -> f32x4 {
        if constexpr (ABL & 1) { asm volatile("" :: "v"(w), "v"(a)); return acc; }
        else return mma16<T>(w, a, acc);
    };
    const unsigned char* const Ag = (const unsigned char*)p.A;
    T* const Cg = (T*)p.C;
    T* const Fg = (T*)p.Fo;
    const uint32_t drm = p.drow_mul > 0 ? (uint32_t)p.drow_mul : 1u;
    const int last_row = p.M - 1;
    int c_slot = 0;
    float oacc_sink = 0.0f;                           // ABL 64 only

    // W1 fragments of k-group kg from the stage whose bases are a1[]: [0] hidden tile 0, [1] hidden tile 1
    auto read_w1 = [&](const unsigned* a1, int kg, u32x4* w) {
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            if constexpr (ABL & 4) { w[ht] = u32x4{a1[kg & 3], a1[0], a1[1], (unsigned)kg}; asm volatile("" : "+v"(w[ht])); }
            else w[ht] = lds16(smem + a1[kg & 3] + (ht * 2048 + (kg >> 2) * 256));
        }
    };
    // W2 fragments of output tiles 2q, 2q + 1
    auto read_w2 = [&](const unsigned* a2, int q, u32x4* w) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if constexpr (ABL & 4) { w[hh] = u32x4{a2[hh], a2[0], a2[1], (unsigned)q}; asm volatile("" : "+v"(w[hh])); }
            else w[hh] = lds16(smem + a2[hh] + q * 2048);
        }
    };
    // this stage's bases = lane part + slot offset, made opaque so that the constants stay in the instructions' offset fields
    auto stage_bases = [&](int slot_index, unsigned* a1, unsigned* a2) {
        const unsigned so = (unsigned)slot_index * FFN_SLOT;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) { a1[kq] = w1c + so + (unsigned)((kq ^ w1q) << 6); asm volatile("" : "+v"(a1[kq])); }
        a2[0] = w2c + so; a2[1] = ((w2c ^ 16u) + 256u) + so;
        asm volatile("" : "+v"(a2[0]), "+v"(a2[1]));
    };

    // BWD: the forward's mask bits of stage group q (stages 4q .. 4q + 3) for this wave's two row tiles: lane (i, g) fetches the word
    // of row (rt, i), units 128 q + 32 (c & 3) + 8 g .. + 7 in byte c & 3 (ffn_mask_byte), by one 4-byte-per-lane LDS-DMA per row tile
    // into slot q & 1 of the wave's own ring (read back at lane * 4: no barrier, the wave's counted waits cover it)
    [[maybe_unused]] auto issue_mask = [&](int q, int m0w) {
        if constexpr (BWD) {
            int ln = lane; asm volatile("" : "+v"(ln));
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                int gr = m0w + 16 * rt + (ln & 15); gr = gr < last_row ? gr : last_row;
                glds4(p.Mb + (size_t)gr * (FFN_F / 8) + 16 * q + 4 * (ln >> 4), Mr + wave * 1024 + (q & 1) * 512 + rt * 256);
            }
        }
    };

#pragma unroll 1
    for (int ps = 0; ps < my; ++ps) {
        const int m0 = (b + ps * G) * (32 * WV) + 32 * wave;
        // Lane-derived values are re-derived from an OPAQUE copy of the lane id wherever they are needed outside the MFMA loops:
        // otherwise the compiler hoists dozens of loop-invariant addresses / indices out of the pass loop and has to spill them
        // around the stage loop (every reload is a scratch round trip behind s_waitcnt vmcnt(0), which also drains the DMA ring)
        auto lane_ids = [&](int& li, int& lg) { int ln = lane; asm volatile("" : "+v"(ln)); li = ln & 15; lg = ln >> 4; };
        int pi, pg;
        lane_ids(pi, pg);
        // ---- this wave's 32 rows of h1 as operand fragments: af[rt][kg] = h1[row][32 kg + 8g .. + 7]
        u32x4 af[2][8];
        f32x4 oacc[2][16];                           // (BWD: starts as dP, the residual path of dH1 = dP + dF . W1; else 0)
        if constexpr (!BWD) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            int gr = m0 + 16 * rt + pi; gr = gr < last_row ? gr : last_row;
            const unsigned char* arow = Ag + (size_t)gr * p.lda * 2 + pg * 16;
            asm volatile("" : "+v"(arow));
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) af[rt][kg] = *(const u32x4*)(arow + kg * 64);
        }
        } else {
            // ---- norm2 backward of the wave's 32 rows, in the fragment layout (lane (i, g): row i, columns 32 kg + 8 g .. + 7):
            //   xh = (y - beta) / gamma, gy = dy gamma, dP = rstd (gy - mean(gy) - xh mean(gy xh)), dM = drop2'(dP) = product 1's operand
            issue_mask(0, m0);                       // the mask words of stages 0-3 (complete behind the wait below)
            unsigned tb = (unsigned)(NSTG * FFN_SLOT + FFN_F * 4) + (unsigned)(32 * pg);         // &Ls[8 g], opaque
            asm volatile("" : "+v"(tb));
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int row = m0 + 16 * rt + pi;
                const int gr = row < last_row ? row : last_row;
                const unsigned char* drow = Ag + (size_t)gr * p.lda * 2 + pg * 16;
                const unsigned char* yrow = (const unsigned char*)p.Y + (size_t)gr * 512 + pg * 16;
                asm volatile("" : "+v"(drow), "+v"(yrow));
                u32x4 dyf[8], yf[8];
#pragma unroll
                for (int kg = 0; kg < 8; ++kg) { dyf[kg] = *(const u32x4*)(drow + kg * 64); yf[kg] = *(const u32x4*)(yrow + kg * 64); }
                const float rs = p.rstd[gr];
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int kg = 0; kg < 8; ++kg) {
                    const T* const dv = (const T*)&dyf[kg];
                    const T* const yv = (const T*)&yf[kg];
                    asm volatile("" : "+v"(tb));        // (every group re-reads its table entries: hoisted, the 3 x 64 table values per lane spill)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const f32x4 ga = *(const f32x4*)(smem + tb + (32 * kg + 4 * hh) * 4), be = *(const f32x4*)(smem + tb + 1024 + (32 * kg + 4 * hh) * 4);
                        const f32x4 ig = *(const f32x4*)(smem + tb + 2048 + (32 * kg + 4 * hh) * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float gy = to_f32(dv[4 * hh + r]) * ga[r], xh = (to_f32(yv[4 * hh + r]) - be[r]) * ig[r];
                            s1 += gy; s2 += gy * xh;
                        }
                    }
                }
                s1 = cross4_sum(s1) * (1.0f / 256.0f);
                s2 = cross4_sum(s2) * (1.0f / 256.0f);
                // the second sweep widens the PACKED values again (kept as 128 fp32 values between the sweeps they spill)
#pragma unroll
                for (int kg = 0; kg < 8; ++kg) asm volatile("" : "+v"(dyf[kg]), "+v"(yf[kg]));
                unsigned char* mrow = (unsigned char*)p.dM + ((size_t)row * 256 + 8 * pg) * 2;
                asm volatile("" : "+v"(mrow));
#pragma unroll
                for (int kg = 0; kg < 8; ++kg) {
                    const T* const dv = (const T*)&dyf[kg];
                    const T* const yv = (const T*)&yf[kg];
                    asm volatile("" : "+v"(tb));
                    f32x4 dx[2];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const f32x4 ga = *(const f32x4*)(smem + tb + (32 * kg + 4 * hh) * 4), be = *(const f32x4*)(smem + tb + 1024 + (32 * kg + 4 * hh) * 4);
                        const f32x4 ig = *(const f32x4*)(smem + tb + 2048 + (32 * kg + 4 * hh) * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float gy = to_f32(dv[4 * hh + r]) * ga[r], xh = (to_f32(yv[4 * hh + r]) - be[r]) * ig[r];
                            dx[hh][r] = rs * (gy - s1 - xh * s2);
                        }
                        // dP stays on chip: it is the start value of the dH1 accumulators (column 32 kg + 8 g + 4 hh + r of row (rt, i) is
                        // oacc[rt][2 kg + hh][r], the map of the operand fragments), unrounded
                        oacc[rt][2 * kg + hh] = dx[hh];
                    }
                    // dM = drop2'(dP): masked in fp32, rounded once -- the bits the separate norm2-backward kernel stores
                    if (p.drop2.thr != 0) {
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh)
                            drop_apply4(p.drop2, (uint32_t)row * drm * 256u + (uint32_t)(32 * kg + 8 * pg + 4 * hh), dx[hh]);
                    }
                    const u32x4 pk = pack_acc<T>(dx[0], dx[1]);
                    if (row < p.M) __builtin_nontemporal_store(pk, (u32x4*)(mrow + 64 * kg));      // the weight gradient dW2 = dM^T f reads it
                    af[rt][kg] = pk;
                }
            }
        }
        // retire these ordinary loads HERE (and with them everything older): the stage loop then contains no load the compiler
        // has to wait for, so its own waits stay out of it, and the counted wait below may assume the steady state from stage 0
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) asm volatile("" : "+v"(af[rt][kg]));
        wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();             // every wave's pieces of the next D stages have landed
        __builtin_amdgcn_sched_barrier(0);

        if constexpr (!BWD) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) oacc[rt][nt] = f32x4{0, 0, 0, 0};
        }
        uint32_t macc[2] = {0u, 0u};                   // train: mask bytes of four stages (see ffn_mask_byte)

        // The fragment reads run one group (2 fragments = 4 MFMAs) ahead of the matrix pipe: wA / wB alternate, and every
        // product ends by reading the first fragments of the product that follows it in program order into wA.
        //
        // The two waves of a SIMD (w and w + 4) run the three pieces of a stage -- product 1, hidden epilogue, product 2 -- in
        // DIFFERENT phase: waves 4-7 lag by one product (their product 2 of stage c runs at the top of iteration c + 1), so
        // that between two barriers one wave of a SIMD is on the vector ALU / LDS while its partner feeds the matrix pipe.
        // In lockstep both would issue MFMAs at the same time and then both sit in the epilogue (measured: the MFMA time and
        // everything else simply add up).  Barrier protocol and slot reuse are unchanged: every read of a slot still falls
        // between the barrier that published it and the barrier before its refill.
        u32x4 wA[2], wB[2];
        unsigned a1[4], a2[2];
        if constexpr (WV == 8) {
            stage_bases(c_slot, a1, a2);
            read_w1(a1, 0, wA);
        }

        // product 1 of the stage with bases (b1, b2); ends with the first W2 fragments of the same stage in wA
        auto prod1 = [&](f32x4 (&h)[2][2], const unsigned* b1, const unsigned* b2) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) { h[rt][0] = f32x4{0, 0, 0, 0}; h[rt][1] = f32x4{0, 0, 0, 0}; }
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
                u32x4* const cur = (kg & 1) ? wB : wA;
                u32x4* const nxt = (kg & 1) ? wA : wB;
                if (kg < 7) read_w1(b1, kg + 1, nxt);
                else read_w2(b2, 0, nxt);            // kg == 7: nxt == wA
#pragma unroll
                for (int ht = 0; ht < 2; ++ht) {
                    h[0][ht] = mm(cur[ht], af[0][kg], h[0][ht]);
                    h[1][ht] = mm(cur[ht], af[1][kg], h[1][ht]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // bias + ReLU + dropout in registers; the packed tiles are the hidden's storage-type values (and product 2's operand)
        auto hidden_epilogue = [&](f32x4 (&h)[2][2], u32x4* hp, int c) {
            int i, g;                                  // shadow the kernel-scope lane ids: re-derived here (see lane_ids)
            lane_ids(i, g);
            unsigned bb = (unsigned)(NSTG * FFN_SLOT) + (unsigned)(c * 128 + 32 * g);      // &B1s[32 c + 8 g], opaque: constants stay in the offset field
            asm volatile("" : "+v"(bb));
            [[maybe_unused]] unsigned mb = 0;
            if constexpr (BWD) {                       // this lane's mask words of the stage group: Mr + wave KB + slot + lane * 4 (+ 256 for row tile 1)
                int ln = lane; asm volatile("" : "+v"(ln));
                mb = (unsigned)(NSTG * FFN_SLOT + (FFN_F + 3 * 256) * 4) + (unsigned)(wave * 1024 + ((c >> 2) & 1) * 512) + (unsigned)ln * 4u;
                asm volatile("" : "+v"(mb));
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int row = m0 + 16 * rt + i;
                if constexpr (BWD) {
                    // dF = (dM . W2) o mask: bit j of this stage's byte <-> hidden unit 32 c + 8 g + j "kept and positive" in the forward
                    const uint32_t by = *(const uint32_t*)(smem + mb + rt * 256) >> (8 * (c & 3));
                    if constexpr ((ABL & 16) == 0)
#pragma unroll
                    for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[rt][ht][r] = ((by >> (4 * ht + r)) & 1u) ? h[rt][ht][r] * p.drop1.scale : 0.0f;
                } else if constexpr ((ABL & 16) == 0) {
                    // (acc + b1) / (1 - p) in fp32, one fma per element (eval: the factor is 1); the ReLU and the dropout then act on the
                    // PACKED 16-bit pairs: a negative value has its sign bit set, so a packed signed max against 0 is the ReLU (one
                    // instruction per pair), and the pair's two keep decisions are the two 16-bit fields of one hash word (pk_keep)
                    const float sc = STORE_F ? p.drop1.scale : 1.0f;
#pragma unroll
                    for (int ht = 0; ht < 2; ++ht) {
                        const f32x4 b4 = *(const f32x4*)(smem + bb + 16 * ht);
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[rt][ht][r] = __builtin_fmaf(h[rt][ht][r], sc, b4[r]);
                    }
                }
                hp[rt] = pack_acc<T>(h[rt][0], h[rt][1]);      // 8 consecutive hidden units 32c + 8g .. + 7 of row (rt, i)
                if constexpr (!BWD && (ABL & 16) == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) hp[rt][k] = pk_relu16(hp[rt][k]);
                    if constexpr (STORE_F) {
                        if (p.drop1.thr != 0) {            // (uniform; the eval kernel has no dropout: the launcher checks)
                            const uint32_t q0 = ((uint32_t)row * drm * (uint32_t)FFN_F + (uint32_t)(c * 32 + 8 * g)) >> 2;     // two aligned index quads
                            const uint32_t w0 = mix32(q0 ^ p.drop1.key), w2 = mix32((q0 + 1u) ^ p.drop1.key);
                            const uint32_t tt = (p.drop1.thr - 1u) * 0x00010001u;
                            hp[rt][0] = pk_keep16(hp[rt][0], w0, tt); hp[rt][1] = pk_keep16(hp[rt][1], xs32(w0), tt);
                            hp[rt][2] = pk_keep16(hp[rt][2], w2, tt); hp[rt][3] = pk_keep16(hp[rt][3], xs32(w2), tt);
                        }
                    }
                }
                if constexpr (STORE_F && !(ABL & 128)) {
                    if (row < p.M) {       // 32-bit offset in 16-byte units (rows x F x 2 B stays below 2^36 B for every shape the library accepts)
                        unsigned fo = (unsigned)row * (unsigned)(p.ldf >> 3) + (unsigned)(4 * c + g);
                        if constexpr (ABL & 256) fo = (unsigned)(threadIdx.x + 256 * (blockIdx.x & 63));      // ablation: every store into one small cached region
                        asm volatile("" : "+v"(fo));
                        __builtin_nontemporal_store(hp[rt], (u32x4*)((unsigned char*)Fg + (size_t)fo * 16));
                    }
                }
                if constexpr (STORE_F && !BWD && !(ABL & 128)) {
                    // "stored value != 0" of the 8 packed 16-bit values (they are +0 or positive): a packed min against 1 leaves
                    // bit 0 / bit 16 of word k for units 2k / 2k + 1 (building the bits from the compares costs twice the VALU)
                    uint32_t tb = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {      // (as an instruction: the compiler expands min(x, 1) into compares and selects)
                        uint32_t m2;
                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(m2) : "v"(hp[rt][k]), "s"(0x00010001u));
                        tb |= m2 << (2 * k);
                    }
                    macc[rt] |= ((tb | (tb >> 15)) & 0xFFu) << (8 * (c & 3));
                    if ((c & 3) == 3) {                // four stages' bytes = one word: row * 128 + 16 (c >> 2) + 4 g
                        if (row < p.M) {
                            unsigned mo = (unsigned)row * (unsigned)(FFN_F / 32) + (unsigned)(c + g - 3);     // in words: 4 (c >> 2) + g
                            asm volatile("" : "+v"(mo));
                            ((uint32_t*)p.Mb)[mo] = macc[rt];
                        }
                        macc[rt] = 0;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // product 2 with the W2 bases b2; ends with the first W1 fragments of the stage whose bases are nb1 in wA
        auto prod2 = [&](const u32x4* hp, const unsigned* b2, const unsigned* nb1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                u32x4* const cur = (q & 1) ? wB : wA;
                u32x4* const nxt = (q & 1) ? wA : wB;
                if (q < 7) read_w2(b2, q + 1, nxt);
                else if constexpr (WV == 8) read_w1(nb1, 0, nxt);           // q == 7: nxt == wA (WV = 4: the next stage is not published yet)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    oacc[0][2 * q + hh] = mm(cur[hh], hp[0], oacc[0][2 * q + hh]);
                    oacc[1][2 * q + hh] = mm(cur[hh], hp[1], oacc[1][2 * q + hh]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        // stage c + 1 must be complete before anyone reads from it, and the slot of stage c - 1 is refilled: this wave's pieces
        // of stage c + 1 have landed once at most the (D-2) younger stages' DMAs and the (D-1) stages' worth of hidden stores
        // issued since are outstanding
        // (WV = 4, D = 1: the barrier sits at the top of stage c and publishes stage c itself, whose DMAs are older than the previous
        // stage's hidden stores only; the refill that follows goes to the slot of stage c - 1)
        // The count includes the hidden stores issued since -- which a wave whose 16-row tile lies beyond M SKIPS (its store sits
        // behind an exec-zero branch): such a wave (only in the last, partial pass) waits by the DMAs alone, which is correct
        // however many of its stores were issued (a count may be too small, never too large).
        constexpr int PUBW = WV == 8 ? (D - 2) * NDMA + (D - 1) * NST : NST;
        constexpr int PUBW_DMA = WV == 8 ? (D - 2) * NDMA : 0;
        const bool all_rows = m0 + 32 <= p.M;       // wave-uniform
        auto publish_next = [&]() {
            if (all_rows) wait_vmcnt<PUBW>(); else wait_vmcnt<PUBW_DMA>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr ((ABL & 8) == 0) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            issue();
            __builtin_amdgcn_sched_barrier(0);
        };
        constexpr int NCHR = (ABL & 32) ? 0 : NCH;
        if constexpr (WV == 4) {
            // The 32 fragments of a stage -- W1 (kg, ht) for kg = 0..7, then W2 (q, hh) for q = 0..7 -- go through a ring of FRB
            // registers, read FRD = FRB - 1 fragments (2 MFMAs = 32 matrix-pipe cycles each) ahead of their use: an LDS read
            // issued one 4-MFMA group ahead (the 8-wave loops) is waited for after ~32-64 cycles, well inside its latency.
            constexpr int FRB = 6, FRD = FRB - 1;
            u32x4 fr[FRB];
            auto rd = [&](auto J) {
                constexpr int j = decltype(J)::value;
                if constexpr (j < 16) ffn_lds_read<(j & 1) * 2048 + (j >> 3) * 256>(fr[j % FRB], a1[(j >> 1) & 3]);
                else if constexpr (j < 32) ffn_lds_read<((j - 16) >> 1) * 2048>(fr[j % FRB], a2[j & 1]);
            };
#pragma unroll 1
            for (int c = 0; c < NCHR; ++c) {
                f32x4 h[2][2];
                u32x4 hp[2];
                if (c > 0) publish_next();             // stage 0 was published by the pass-start barrier
                else { issue(); __builtin_amdgcn_sched_barrier(0); }
                if constexpr (BWD) { if ((c & 3) == 0 && c + 4 < NCH) issue_mask((c >> 2) + 1, m0); }
                stage_bases(c & 1, a1, a2);            // NCH is even: stage c of every pass sits in slot c & 1
                ffn_static_for<0, FRD>(rd);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) { h[rt][0] = f32x4{0, 0, 0, 0}; h[rt][1] = f32x4{0, 0, 0, 0}; }
                ffn_static_for<0, 16>([&](auto J) {    // product 1: hidden tile j & 1, k-group j >> 1
                    constexpr int j = decltype(J)::value;
                    rd(std::integral_constant<int, j + FRD>{});
                    ffn_lds_wait<FRD>(fr[j % FRB]);
#ifdef FFN_MF32_PROBE
                    // TIMING PROBE ONLY (tools/ffn_bench.hip -DFFN_MF32_PROBE; results wrong): ONE v_mfma_f32_32x32x16 of the same FLOPs for the
                    // two 16x16x32 of this fragment, same operand registers, the stage's hidden accumulators as one 32 x 32 tile
                    *(ffn_f32x16*)&h[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fr[j % FRB]), __builtin_bit_cast(bf16x8_t, af[j & 1][j >> 1]), *(ffn_f32x16*)&h[0][0], 0, 0, 0);
#else
                    h[0][j & 1] = mm(fr[j % FRB], af[0][j >> 1], h[0][j & 1]);
                    h[1][j & 1] = mm(fr[j % FRB], af[1][j >> 1], h[1][j & 1]);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                });
                hidden_epilogue(h, hp, c);
                ffn_static_for<16, 32>([&](auto J) {   // product 2: output tile j - 16
                    constexpr int j = decltype(J)::value;
                    rd(std::integral_constant<int, j + FRD>{});
                    ffn_lds_wait<(j + FRD < 32 ? FRD : 31 - j)>(fr[j % FRB]);
#ifdef FFN_MF32_PROBE
                    ((ffn_f32x16*)&oacc[0][0])[(j - 16) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fr[j % FRB]), __builtin_bit_cast(bf16x8_t, hp[j & 1]), ((ffn_f32x16*)&oacc[0][0])[(j - 16) >> 1], 0, 0, 0);
#else
                    oacc[0][j - 16] = mm(fr[j % FRB], hp[0], oacc[0][j - 16]);
                    oacc[1][j - 16] = mm(fr[j % FRB], hp[1], oacc[1][j - 16]);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        } else if (!lag) {
#pragma unroll 1
            for (int c = 0; c < NCHR; ++c) {
                f32x4 h[2][2];
                u32x4 hp[2];
                unsigned n1[4], n2[2];
                prod1(h, a1, a2);
                publish_next();
                if constexpr (BWD) { if ((c & 3) == 0 && c + 4 < NCH) issue_mask((c >> 2) + 1, m0); }
                hidden_epilogue(h, hp, c);
                c_slot = c_slot + 1 == NSTG ? 0 : c_slot + 1;
                stage_bases(c_slot, n1, n2);
                prod2(hp, a2, n1);
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) a1[kq] = n1[kq];
                a2[0] = n2[0]; a2[1] = n2[1];
            }
        } else if (NCHR > 0) {
            u32x4 hp[2];
            unsigned pa2[2];
            {   // stage 0: no product 2 is pending yet
                f32x4 h[2][2];
                publish_next();
                if constexpr (BWD) issue_mask(1, m0);
                prod1(h, a1, a2);
                hidden_epilogue(h, hp, 0);
                pa2[0] = a2[0]; pa2[1] = a2[1];
                c_slot = c_slot + 1 == NSTG ? 0 : c_slot + 1;
                stage_bases(c_slot, a1, a2);
            }
#pragma unroll 1
            for (int c = 1; c < NCHR; ++c) {
                f32x4 h[2][2];
                prod2(hp, pa2, a1);                // product 2 of stage c - 1, then the first W1 fragments of stage c
                publish_next();
                if constexpr (BWD) { if ((c & 3) == 0 && c + 4 < NCH) issue_mask((c >> 2) + 1, m0); }
                prod1(h, a1, a2);
                hidden_epilogue(h, hp, c);
                pa2[0] = a2[0]; pa2[1] = a2[1];
                c_slot = c_slot + 1 == NSTG ? 0 : c_slot + 1;
                stage_bases(c_slot, a1, a2);
            }
            prod2(hp, pa2, a1);                    // the last stage's product 2 (its trailing fragment read is not used)
        }

        if constexpr (ABL & 64) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int nt = 0; nt < 16; ++nt) oacc_sink += oacc[rt][nt][0] + oacc[rt][nt][1] + oacc[rt][nt][2] + oacc[rt][nt][3];
        }
        // ---- pass epilogue: v = residual + drop2(out + b2); LayerNorm over the row (this lane: 64 of its 256 columns, the
        // other three quarters sit on the lanes with the same i); rows leave as 16-byte pieces per lane (64 B per row and kg)
        // (addresses: ONE lane-dependent base per table / row, made opaque inside the pass loop -- otherwise the compiler hoists
        // ~100 separately materialised addresses out of the pass loop and spills them around the stage loop)
        int ei, eg;
        lane_ids(ei, eg);
        unsigned lb = (unsigned)(NSTG * FFN_SLOT + FFN_F * 4) + (unsigned)(32 * eg);         // &Ls[8 g]
        asm volatile("" : "+v"(lb));
        if constexpr (BWD) {
            // ---- dH1 = dP + dF . W1 (dP was the accumulators' start value)
#pragma unroll
            for (int rt = 0; rt < ((ABL & 64) ? 0 : 2); ++rt) {
                const int i = ei, g = eg;
                const int row = m0 + 16 * rt + i;
                unsigned char* crow = (unsigned char*)Cg + ((size_t)row * p.ldc + 8 * g) * 2;
                asm volatile("" : "+v"(crow));
#pragma unroll
                for (int kg = 0; kg < 8; ++kg)
                    if (row < p.M) __builtin_nontemporal_store(pack_acc<T>(oacc[rt][2 * kg], oacc[rt][2 * kg + 1]), (u32x4*)(crow + 64 * kg));
            }
        } else {
#pragma unroll
        for (int rt = 0; rt < ((ABL & 64) ? 0 : 2); ++rt) {
            const int i = ei, g = eg;                  // (shadow the kernel-scope lane ids)
            const int row = m0 + 16 * rt + i;
            unsigned char* crow = (unsigned char*)Cg + ((size_t)row * p.ldc + 8 * g) * 2;
            asm volatile("" : "+v"(crow));
            float sm = 0.0f;
            // residual: the same h1 values the operand fragments held, RE-READ (L2 / Infinity-Cache resident: this block read them
            // one pass ago) instead of kept: the 64 fragment registers die with the last product 1 and the epilogue runs without
            // spills (kept alive for the residual they cost ~16 spilled registers around the stage loop, each reload an exposed
            // scratch round trip behind s_waitcnt vmcnt(0): pass epilogue 18 -> see DESIGN.md)
            int grr = row < last_row ? row : last_row;
            const unsigned char* rrow = Ag + (size_t)grr * p.lda * 2 + g * 16;
            asm volatile("" : "+v"(rrow));
            u32x4 rres[8];
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) rres[kg] = *(const u32x4*)(rrow + kg * 64);
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
                const T* const res = (const T*)&rres[kg];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int col = 32 * kg + 8 * g + 4 * hh;
                    f32x4 v = oacc[rt][2 * kg + hh] + *(const f32x4*)(smem + lb + (32 * kg + 4 * hh) * 4);
                    if constexpr (STORE_F) drop_apply4(p.drop2, (uint32_t)row * drm * 256u + (uint32_t)col, v);
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
                    const f32x4 ga = *(const f32x4*)(smem + lb + 1024 + (32 * kg + 4 * hh) * 4), be = *(const f32x4*)(smem + lb + 2048 + (32 * kg + 4 * hh) * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o2[hh][r] = (oacc[rt][2 * kg + hh][r] - mean) * rs * ga[r] + be[r];
                }
                if (row < p.M) __builtin_nontemporal_store(pack_acc<T>(o2[0], o2[1]), (u32x4*)(crow + 64 * kg));
            }
        }
        }
    }
    if constexpr (ABL & 64) { if (p.M < 0) { Cg[tid] = from_f32<T>(oacc_sink); } }
    wait_vmcnt<0>();                                  // the D stages issued past the end land before the block's LDS is released
}

}  // namespace ge2e
