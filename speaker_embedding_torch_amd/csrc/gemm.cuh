// Projection GEMMs of the GE2E encoder (SURVEY.md 8a rows a1, a3, a5, a6 and their backward a13).
//
//   gemm_nt_kernel : C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogue)   -- forward and dgrad
//   wgrad_kernel   : dW[N,K] += Y[R,N]^T * X[R,K], db[N] += colsum(Y) -- weight gradients (split over R)
//
// Both run on MFMA 16x16 tiles in either arithmetic mode (common.cuh).  A block owns a BM x BN
// output tile; operands are staged global -> registers -> LDS (register-staged double buffer:
// loads for k-step s+1 are issued before the MFMAs of step s and written to LDS after them).
#pragma once
#include "common.cuh"

namespace ge2e {

enum { EPI_NONE = 0, EPI_BIAS, EPI_BIAS_RELU_DROP, EPI_LN, EPI_MASK, EPI_ADD, EPI_PRENET, EPI_PRENET_BWD, EPI_RESERVED8 /* was EPI_ADD_ROW0: the last layer's K/V dgrad, gone with attn_last.cuh */,
       EPI_MASKBITS /* EPI_MASK with the mask as one BIT per element: R = [M][ldr bytes], bits in the byte order of ffn.cuh ffn_mask_byte (gemm_ws only) */ };
enum { ALOAD_ROW = 0 };     // operand rows are read row-major (the mel batch is packed to rows first: mel_pack_kernel)

struct GemmArgs {
    const void* A; int lda;      // [M, K] of T
    const void* W; int ldw;      // [N, K] of T
    void* C; int ldc;            // [M, N] of T
    int M, N, K;                 // K = padded reduction length (multiple of 128 / sizeof(T))
    const float* bias;           // [N]
    const void* R; int ldr;      // residual (EPI_LN), addend (EPI_ADD), mask source (EPI_MASK), dH0 (EPI_PRENET_BWD)
    const float* gamma; const float* beta; float* rstd; float eps;   // EPI_LN
    Drop drop;                   // dropout site of this epilogue (thr == 0: inactive)
    int drow_mul;                // dropout counter row = row * drow_mul (compact t=0 rows of the last layer: T); 0 = 1
    float mask_scale;            // EPI_MASK: gradient scale of kept elements
    const float* pe_t;           // [T, N] transposed positional table (prenet)
    const float* alpha;          // positional_encoding.alpha (device scalar)
    float* dalpha;               // EPI_PRENET_BWD: scalar accumulator
    int T, mel;                  // frames per utterance (prenet epilogues: row -> frame), real mel dim
    unsigned char* relu_bits;    // EPI_PRENET (train): [M][N / 8] bytes, bit c & 7 of byte c >> 3 = "pre-activation of column c > 0" (prenet_bwd.cuh reads it)
};

// X3 (T = float only): the "fp32x3" mode of common.cuh -- an LDS row holds the bf16 hi halves of its 32 k values in chunks 0-3 and the lo
// halves in chunks 4-7 (same 128 bytes), split on the way in; a k-step is then ONE 16x16x32 slice and three MFMAs per tile.
template <typename T, int BM, int BN, int WM, int WN, int EPI, int ALOAD, int NBUF = 2, bool X3 = false>
__global__ void __launch_bounds__(256) gemm_nt_kernel(const GemmArgs p) {
    static_assert(!X3 || sizeof(T) == 4, "fp32x3 is a mode of fp32 storage");
    constexpr int BK = 128 / (int)sizeof(T);            // 128-byte LDS rows = two k-groups
    constexpr int WAVES_N = BN / WN;
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int NA = BM / 32, NB = BN / 32;            // 16-byte chunks per thread and stage
    static_assert((BM / WM) * WAVES_N == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // NBUF = 2: double-buffered stages (one barrier per k-step, 64 KB -> 2 blocks per CU for the 128x128 tile);
    // NBUF = 1: single stage (two barriers per k-step, 32 KB -> a third resident block hides the serial phases)
    unsigned char* const As = smem;                       // [NBUF][BM][128 B]
    unsigned char* const Bs = smem + NBUF * BM * 128;     // [NBUF][BN][128 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int nbn = p.N / BN;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (L / nbn) * BM, n0 = (L % nbn) * BN;
    const int nk = p.K / BK;

    u32x4 ra[NA], rb[NB];
    auto load_stage = [&](int ks) {
        const int k0 = ks * BK;
        {
            const unsigned char* A = (const unsigned char*)p.A;
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const int id = tid + 256 * q, row = id >> 3, c = id & 7, gr = m0 + row;
                ra[q] = gr < p.M ? *(const u32x4*)(A + ((size_t)gr * p.lda + k0) * sizeof(T) + c * 16) : u32x4{0, 0, 0, 0};
            }
        }
        const unsigned char* W = (const unsigned char*)p.W;
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int id = tid + 256 * q, row = id >> 3, c = id & 7, gr = n0 + row;
            rb[q] = gr < p.N ? *(const u32x4*)(W + ((size_t)gr * p.ldw + k0) * sizeof(T) + c * 16) : u32x4{0, 0, 0, 0};
        }
    };
    auto store_stage = [&](int buf) {
        unsigned char* a = As + buf * BM * 128;
        unsigned char* b = Bs + buf * BN * 128;
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int id = tid + 256 * q;
            if constexpr (X3) {
                u32x2 hi, lo;
                split_bf16x3(ra[q], hi, lo);
                *(u32x2*)(a + swz_off<128>(id >> 3, (id & 7) >> 1) + 8 * (id & 1)) = hi;
                *(u32x2*)(a + swz_off<128>(id >> 3, 4 + ((id & 7) >> 1)) + 8 * (id & 1)) = lo;
            } else
            *(u32x4*)(a + swz_off<128>(id >> 3, id & 7)) = ra[q];
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int id = tid + 256 * q;
            if constexpr (X3) {
                u32x2 hi, lo;
                split_bf16x3(rb[q], hi, lo);
                *(u32x2*)(b + swz_off<128>(id >> 3, (id & 7) >> 1) + 8 * (id & 1)) = hi;
                *(u32x2*)(b + swz_off<128>(id >> 3, 4 + ((id & 7) >> 1)) + 8 * (id & 1)) = lo;
            } else
            *(u32x4*)(b + swz_off<128>(id >> 3, id & 7)) = rb[q];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0, 0, 0, 0};

    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = NBUF == 2 ? (ks & 1) : 0;
        if (ks + 1 < nk) load_stage(ks + 1);
        const unsigned char* a = As + buf * BM * 128;
        const unsigned char* b = Bs + buf * BN * 128;
        if constexpr (X3) {
            u32x4 ah[MT], al[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                ah[mt] = lds16(a + swz_off<128>(wm * WM + mt * 16 + i, g));
                al[mt] = lds16(a + swz_off<128>(wm * WM + mt * 16 + i, 4 + g));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const u32x4 bh = lds16(b + swz_off<128>(wn * WN + nt * 16 + i, g));
                const u32x4 bl = lds16(b + swz_off<128>(wn * WN + nt * 16 + i, 4 + g));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mma16_x3(bh, bl, ah[mt], al[mt], acc[mt][nt]);
            }
        } else
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            u32x4 af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = lds16(a + swz_off<128>(wm * WM + mt * 16 + i, kg * 4 + g));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const u32x4 bf = lds16(b + swz_off<128>(wn * WN + nt * 16 + i, kg * 4 + g));
                // operand order (W rows as MFMA "A", activation rows as MFMA "B") puts an activation row on a
                // lane: acc[mt][nt][r] = C[m0 + wm*WM + mt*16 + i][n0 + wn*WN + nt*16 + 4g + r]
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mma16<T>(bf, af[mt], acc[mt][nt]);
            }
        }
        if constexpr (NBUF == 2) {
            if (ks + 1 < nk) store_stage(buf ^ 1);
            __syncthreads();
        } else {
            __syncthreads();                                  // everyone is done reading the single stage
            if (ks + 1 < nk) { store_stage(0); __syncthreads(); }
        }
    }

    // ------------------------------------------------------------------ epilogue
    // All global traffic of the epilogue is 16 bytes per lane along rows: the residual / mask / addend tile
    // comes in through LDS, each lane transforms its 4-column groups in place, and the finished tile goes out
    // with full-line stores (the MFMA layout itself would give 32-byte segments per row).
    constexpr bool NEEDS_R = (EPI == EPI_MASK || EPI == EPI_ADD || EPI == EPI_PRENET_BWD);
    const uint32_t drm = p.drow_mul > 0 ? (uint32_t)p.drow_mul : 1u;
    if constexpr (EPI == EPI_LN) {
        // Row-complete tile (BN == N == 256): park the raw fp32 accumulators in LDS, then every wave finishes whole
        // rows the way the LayerNorm kernels do -- one row per wave pass, 4 columns per lane, wave-wide reductions --
        // so bias, residual and the output are all full-row (512 B / 1 KB) coalesced accesses.
        static_assert(BN == 256, "LayerNorm epilogue needs the whole row in the tile");
        constexpr int LDF = BN + 4;                        // floats per staged row (16-byte pad)
        float* const Fs = (float*)smem;                    // main loop ended with a barrier: LDS is free
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                *(f32x4*)(Fs + (wm * WM + mt * 16 + i) * LDF + wn * WN + nt * 16 + 4 * g) = acc[mt][nt];
        __syncthreads();
        const uint32_t drm_ = p.drow_mul > 0 ? (uint32_t)p.drow_mul : 1u;
        const int c0 = lane * 4;
        const f32x4 b4 = *(const f32x4*)(p.bias + c0), g4 = *(const f32x4*)(p.gamma + c0), be4 = *(const f32x4*)(p.beta + c0);
        const T* const Rm = (const T*)p.R;
        T* const Cm = (T*)p.C;
        // 4 independent rows per pass: their loads and the two dependent wave reductions interleave (the chain
        // LDS -> residual -> mean -> variance of a single row is pure latency)
        constexpr int RPW = BM / 4;                        // rows per wave
#pragma unroll 1
        for (int it = 0; it < RPW; it += 4) {
            f32x4 v[4];
            float mean[4], rstd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int lr = wave * RPW + it + u, row = m0 + lr;
                v[u] = *(const f32x4*)(Fs + lr * LDF + c0) + b4;
                drop_apply4(p.drop, (uint32_t)row * drm_ * (uint32_t)BN + (uint32_t)c0, v[u]);
                if (row < p.M) v[u] += load4(Rm + (size_t)row * p.ldr + c0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) mean[u] = wave_sum(v[u][0] + v[u][1] + v[u][2] + v[u][3]) * (1.0f / (float)BN);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] -= mean[u];
                rstd[u] = wave_sum(v[u][0] * v[u][0] + v[u][1] * v[u][1] + v[u][2] * v[u][2] + v[u][3] * v[u][3]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = m0 + wave * RPW + it + u;
                const float rs = 1.0f / sqrtf(rstd[u] * (1.0f / (float)BN) + p.eps);
                if (row < p.M) {
                    if (lane == 0 && p.rstd) p.rstd[row] = rs;
                    store4(Cm + (size_t)row * p.ldc + c0, v[u][0] * rs * g4[0] + be4[0], v[u][1] * rs * g4[1] + be4[1],
                           v[u][2] * rs * g4[2] + be4[2], v[u][3] * rs * g4[3] + be4[3]);
                }
            }
        }
        return;
    }
    constexpr int CPRC = BN * (int)sizeof(T) / 16;        // 16-byte chunks per tile row
    constexpr int LDC = BN * (int)sizeof(T) + 16;         // LDS row stride of the staged tile
    constexpr int NCC = BM * CPRC / 256;                  // chunks per thread
    unsigned char* const Cs = smem;                        // main loop ended with a barrier: LDS is free
    T* const C = (T*)p.C;
    if constexpr (NEEDS_R) {
        const unsigned char* Rg = (const unsigned char*)p.R;
#pragma unroll
        for (int q0 = 0; q0 < NCC; q0 += 8) {              // at most 8 x 16 B in flight per thread
            u32x4 rr[8];
#pragma unroll
            for (int q = 0; q < 8 && q0 + q < NCC; ++q) {
                const int id = tid + 256 * (q0 + q), row = id / CPRC, c = id % CPRC, gr = m0 + row;
                rr[q] = gr < p.M ? *(const u32x4*)(Rg + ((size_t)gr * p.ldr + n0) * sizeof(T) + c * 16) : u32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int q = 0; q < 8 && q0 + q < NCC; ++q) {
                const int id = tid + 256 * (q0 + q);
                *(u32x4*)(Cs + (id / CPRC) * LDC + (id % CPRC) * 16) = rr[q];
            }
        }
        __syncthreads();
    }
    float dal = 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int lrow = wm * WM + mt * 16 + i;            // row inside the tile
        const int row = m0 + lrow;
        const int lcolb = wn * WN + 4 * g, colb = n0 + lcolb;
        T* const crow = (T*)(Cs + lrow * LDC);
        {
            int t = 0;
            if constexpr (EPI == EPI_PRENET || EPI == EPI_PRENET_BWD) t = row < p.M ? row % p.T : 0;
            [[maybe_unused]] unsigned long long sbits = 0ull;     // EPI_PRENET: this lane's sign nibbles of the wave's WN columns of the row
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = colb + nt * 16;
                f32x4 v = acc[mt][nt];
                if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU_DROP || EPI == EPI_PRENET || EPI == EPI_PRENET_BWD) {
                    const f32x4 b4 = *(const f32x4*)(p.bias + col);
                    v += b4;
                }
                if constexpr (EPI == EPI_BIAS_RELU_DROP) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
                    drop_apply4(p.drop, (uint32_t)row * drm * (uint32_t)p.N + (uint32_t)col, v);
                }
                if constexpr (EPI == EPI_PRENET) {
                    const f32x4 pe4 = *(const f32x4*)(p.pe_t + (size_t)t * p.N + col);
                    const float al = *p.alpha;
                    if (p.relu_bits) {
                        const unsigned nib = (unsigned)(v[0] > 0.0f) | ((unsigned)(v[1] > 0.0f) << 1) | ((unsigned)(v[2] > 0.0f) << 2) | ((unsigned)(v[3] > 0.0f) << 3);
                        sbits |= (unsigned long long)nib << (16 * nt + 4 * g);       // column lcolb + 16 nt + r - wn WN = 16 nt + 4 g + r
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f) + al * pe4[r];
                    drop_apply4(p.drop, (uint32_t)row * drm * (uint32_t)p.N + (uint32_t)col, v);
                }
                if constexpr (EPI == EPI_PRENET_BWD) {
                    // v = prenet pre-activation (recomputed); staged tile = dL/dh0 -> masked gradient of the pre-activation
                    const f32x4 pe4 = *(const f32x4*)(p.pe_t + (size_t)t * p.N + col);
                    f32x4 d4 = load4(crow + lcolb + nt * 16);
                    drop_apply4(p.drop, (uint32_t)row * drm * (uint32_t)p.N + (uint32_t)col, d4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float gv = row < p.M ? d4[r] : 0.0f;
                        dal += gv * pe4[r];
                        v[r] = v[r] > 0.0f ? gv : 0.0f;
                    }
                }
                if constexpr (EPI == EPI_MASK) {
                    const f32x4 m4 = load4(crow + lcolb + nt * 16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = m4[r] > 0.0f ? v[r] * p.mask_scale : 0.0f;
                }
                if constexpr (EPI == EPI_ADD) v += load4(crow + lcolb + nt * 16);
                store4(crow + lcolb + nt * 16, v[0], v[1], v[2], v[3]);
            }
            if constexpr (EPI == EPI_PRENET) {
                if (p.relu_bits) {      // (wave-uniform) the four lanes of a row hold disjoint nibbles: OR them, one store of WN / 8 bytes per row
                    static_assert(WN == 64, "one 64-bit word per row and wave");
                    unsigned lo = (unsigned)sbits, hi = (unsigned)(sbits >> 32);
                    lo |= __shfl_xor(lo, 16, 64); lo |= __shfl_xor(lo, 32, 64);
                    hi |= __shfl_xor(hi, 16, 64); hi |= __shfl_xor(hi, 32, 64);
                    if (g == 0 && row < p.M) *(u32x2*)(p.relu_bits + (size_t)row * (p.N / 8) + (n0 + wn * WN) / 8) = u32x2{lo, hi};
                }
            }
        }
    }
    __syncthreads();
    {
        unsigned char* Cg = (unsigned char*)C;
#pragma unroll
        for (int q = 0; q < NCC; ++q) {
            const int id = tid + 256 * q, row = id / CPRC, c = id % CPRC, gr = m0 + row;
#ifndef GE2E_NO_NT_STORE   // streaming (non-temporal) stores: the tile is not re-read by this kernel; in_proj 152 -> 119 us
            if (gr < p.M) __builtin_nontemporal_store(*(const u32x4*)(Cs + row * LDC + c * 16), (u32x4*)(Cg + ((size_t)gr * p.ldc + n0) * sizeof(T) + c * 16));
#else
            if (gr < p.M) *(u32x4*)(Cg + ((size_t)gr * p.ldc + n0) * sizeof(T) + c * 16) = *(const u32x4*)(Cs + row * LDC + c * 16);
#endif
        }
    }
    if constexpr (EPI == EPI_PRENET_BWD) {
        __syncthreads();
        float* red = (float*)smem;
        const float s = block256_sum(dal, red);
        if (tid == 0) atomicAdd(p.dalpha, s);
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient: dW[n][k] += sum_r Y[r][n] * X[r][k] over this block's slice of rows.
// Both operands are read TRANSPOSED from plain padded LDS tiles ([row][128 cols]).
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
    const void* Y; int ldy;     // [R, N] of T
    const void* X; int ldx;     // [R, ldx] of T, whole 128-column tiles (K may be smaller: prenet K = mel, ldx = 128)
    float* dW; int ldw;         // [N, ldw] fp32 accumulate (atomics)
    float* db;                  // [N] or null
    int R, N, K;                // K = real width (dW columns written: k < K)
    int rows_per_split;         // multiple of the stage height
    int tiles_n, tiles_k;       // 128x128 output tiles along N and K
    int T, mel;
    int blocks;                 // host side: wgrad_ks blocks for this product (0 = the library's default share of the chip)
};

// X3 (T = float only): the "fp32x3" mode of common.cuh -- a stage is 32 rows = one 16x16x32 slice; every operand tile is kept as two bf16
// planes (hi, lo) in the 16-bit layout (transposed reads with ds_read_b64_tr_b16), split on the way in; three MFMAs per tile and stage.
template <typename T, int XLOAD, int NS_ = 3, int KGS = 2, bool X3 = false>
__global__ void __launch_bounds__(256) wgrad_kernel(const WgradArgs p) {
    static_assert(!X3 || sizeof(T) == 4, "fp32x3 is a mode of fp32 storage");
    constexpr int KG = Prec<T>::KG;
    constexpr int RS = KGS * KG;                            // rows per stage
    constexpr int ROWB = 128 * (int)sizeof(T);              // 128 columns
    // row stride == 32 (mod 256) bytes: the 8 rows x 4 lanes x 8 B of one ds_read_b64_tr_b16 half-wave land on 64
    // distinct banks (a 16-byte pad leaves 2-way conflicts on every transposed read); fp32 scalar reads want +16
    constexpr int LD = X3 ? 256 + 32 : ROWB + (sizeof(T) == 2 ? 32 : 16);
    constexpr int PL = X3 ? 2 : 1;                          // planes per operand tile (X3: hi, lo)
    constexpr int CPR = ROWB / 16;                          // chunks per row
    constexpr int NCH = RS * CPR / 256;                     // chunks per thread (= 4)
    static_assert(!X3 || RS == 32, "fp32x3: one 32-row slice per stage");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Ys = smem;                         // [2][PL][RS][LD]
    unsigned char* const Xs = smem + 2 * PL * RS * LD;      // [2][PL][RS][LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // 1-D grid: consecutive (remapped) ids share an XCD; the tiles of one row slice are consecutive, so the
    // slice's Y and X panels are fetched from HBM once per XCD and re-read from its L2 by the other tiles
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = p.tiles_n * p.tiles_k;
    const int tile = L % ntile, split = L / ntile;
    const int n0 = (tile % p.tiles_n) * 128, k0 = (tile / p.tiles_n) * 128;
    const int rbeg = split * p.rows_per_split;
    const int rend = min(p.R, rbeg + p.rows_per_split);
    const int nst = (rend - rbeg + RS - 1) / RS;

    // NS register sets keep the loads of NS row stages in flight (a block streams ~75 stages: latency, not bytes,
    // bounds a one-stage prefetch)
    constexpr int NS = NS_;
    u32x4 rY[NS][NCH], rX[NS][NCH];
    auto load_stage = [&](int st, u32x4* ry, u32x4* rx) {
#ifdef GE2E_WGRAD_ABL
        const int r0 = (GE2E_WGRAD_ABL & 2) ? 0 : rbeg + st * RS;
#else
        const int r0 = rbeg + st * RS;
#endif
        const unsigned char* Y = (const unsigned char*)p.Y;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int id = tid + 256 * q, row = id / CPR, c = id % CPR, gr = r0 + row;
            ry[q] = gr < rend ? *(const u32x4*)(Y + ((size_t)gr * p.ldy + n0) * sizeof(T) + c * 16) : u32x4{0, 0, 0, 0};
        }
        const unsigned char* X = (const unsigned char*)p.X;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int id = tid + 256 * q, row = id / CPR, c = id % CPR, gr = r0 + row;
            rx[q] = gr < rend ? *(const u32x4*)(X + ((size_t)gr * p.ldx + k0) * sizeof(T) + c * 16) : u32x4{0, 0, 0, 0};
        }
    };
    auto store_stage = [&](int buf, int st, const u32x4* ry, const u32x4* rx) {
        unsigned char* y = Ys + buf * PL * RS * LD;
        unsigned char* x = Xs + buf * PL * RS * LD;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int id = tid + 256 * q, row = id / CPR, c = id % CPR;
            if constexpr (X3) {
                u32x2 hi, lo;
                split_bf16x3(ry[q], hi, lo);
                *(u32x2*)(y + row * LD + c * 8) = hi; *(u32x2*)(y + RS * LD + row * LD + c * 8) = lo;
                split_bf16x3(rx[q], hi, lo);
                *(u32x2*)(x + row * LD + c * 8) = hi; *(u32x2*)(x + RS * LD + row * LD + c * 8) = lo;
            } else {
            *(u32x4*)(y + row * LD + c * 16) = ry[q];
            *(u32x4*)(x + row * LD + c * 16) = rx[q];
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
    float bsum = 0.0f;
    const bool do_bias = (p.db != nullptr) && k0 == 0;

    if (nst == 0) return;
#pragma unroll
    for (int s = 0; s < NS; ++s) load_stage(min(s, nst - 1), rY[s], rX[s]);
    for (int st0 = 0; st0 < nst; st0 += NS) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int st = st0 + s;
            if (st < nst) {                                   // block-uniform
                const int buf = st & 1;
                store_stage(buf, st, rY[s], rX[s]);           // waits for this stage's loads only
                __syncthreads();                              // also: everyone finished reading buf at stage st - 2
                // UNCONDITIONAL refill (stage index clamped; the tail re-reads the last stage, <= NS of ~75 stages):
                // a conditional load makes the number of younger loads path-dependent, and hipcc then falls back
                // from counted s_waitcnt vmcnt(N) to vmcnt(0) at the next use -- which drains the whole ring
                load_stage(min(st + NS, nst - 1), rY[s], rX[s]);
                const unsigned char* y = Ys + buf * PL * RS * LD;
                const unsigned char* x = Xs + buf * PL * RS * LD;
                if constexpr (X3) {
                    u32x4 ah[4], al[4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        ah[mt] = frag_tr16(y, LD, 0, wm * 64 + mt * 16, lane);
                        al[mt] = frag_tr16(y + RS * LD, LD, 0, wm * 64 + mt * 16, lane);
                    }
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const u32x4 bh = frag_tr16(x, LD, 0, wn * 64 + nt * 16, lane);
                        const u32x4 bl = frag_tr16(x + RS * LD, LD, 0, wn * 64 + nt * 16, lane);
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = mma16_x3(ah[mt], al[mt], bh, bl, acc[mt][nt]);
                    }
                } else
#ifdef GE2E_WGRAD_ABL
                if constexpr ((GE2E_WGRAD_ABL & 1) == 0)
#endif
#pragma unroll
                for (int kg = 0; kg < KGS; ++kg) {
                    u32x4 af[4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) af[mt] = frag_tr<T>(y, LD, kg * KG, wm * 64 + mt * 16, lane);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const u32x4 bf = frag_tr<T>(x, LD, kg * KG, wn * 64 + nt * 16, lane);
                        // acc[mt][nt][r] = dW[n0 + wm*64 + mt*16 + 4g + r][k0 + wn*64 + nt*16 + i]
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = mma16<T>(af[mt], bf, acc[mt][nt]);
                    }
                }
                if (do_bias) {
                    const int col = tid & 127, half = tid >> 7;
#pragma unroll 4
                    for (int r = half * (RS / 2); r < (half + 1) * (RS / 2); ++r) {
                        if constexpr (X3) bsum += to_f32(*(const bf16_t*)(y + r * LD + col * 2)) + to_f32(*(const bf16_t*)(y + RS * LD + r * LD + col * 2));
                        else bsum += to_f32(*(const T*)(y + r * LD + col * sizeof(T)));
                    }
                }
            }
        }
    }
    __syncthreads();
#ifdef GE2E_WGRAD_ABL
    if constexpr (GE2E_WGRAD_ABL & 4) { if (p.R > 0) return; }
#endif
    // flush through LDS so that every atomic wave-instruction adds 256 contiguous bytes of one dW row
    // (full-rate shape, MI355X_MICROARCH "Global float atomics"); straight from the MFMA layout it would be 4 x 64 B
    constexpr int LDT = 128 * 4 + 16;
    float* const Ts = (float*)smem;                        // [128][LDT/4]; the stage buffers are dead (barrier above)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Ts[(wm * 64 + mt * 16 + 4 * g + r) * (LDT / 4) + wn * 64 + nt * 16 + i] = acc[mt][nt][r];
    __syncthreads();
    for (int row = wave; row < 128; row += 4) {
        const int n = n0 + row;
        if (n >= p.N) break;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int k = k0 + half * 64 + lane;
            if (k < p.K) atomicAdd(p.dW + (size_t)n * p.ldw + k, Ts[row * (LDT / 4) + half * 64 + lane]);
        }
    }
    if (do_bias) {
        const int col = tid & 127;
        if (n0 + col < p.N) atomicAdd(p.db + n0 + col, bsum);
    }
}

}  // namespace ge2e
