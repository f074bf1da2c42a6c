// Weight-stationary streaming projection GEMM for the short-K shapes (K = 256: in_proj, FFN1, dF, dO; 16-bit storage modes).
//
//   C[M, N] = epilogue(A[M, K] . W[N, K]^T)
//
// With K = 256 these products are streaming kernels (150 FLOP per HBM byte, under the ridge): what bounds them is
// how many bytes a CU keeps in flight, not MFMA.  The tiled kernel in gemm.cuh re-stages a 64 KB W tile and a 64 KB
// A tile per 128x128 output tile (W through L2 1200 times), one k-step in flight.  Here instead
//   * a block owns 256 output columns for its whole life and keeps their weights in REGISTERS as MFMA fragments
//     (wave w: columns 64w..64w+63, 4 n-tiles x 8 k-groups x 16 B = 128 VGPRs): W is read once per block;
//   * the block is persistent over 16-row tiles of A (8 KB each) that arrive by LDS-DMA (global_load_lds, 16 B per
//     lane) into a 4-slot ring, three tiles ahead, behind counted s_waitcnt vmcnt(N) + one raw s_barrier per tile;
//     loads never touch a VGPR and are never waited for together with the output stores;
//   * the epilogue is wave-private: each wave turns its 16x64 accumulator slab into 128-byte row segments through
//     a 2 KB LDS stage (no block barrier) and streams them out with non-temporal stores; the mask / addend tile of
//     the MASK / ADD epilogues arrives by LDS-DMA as well, into a wave-private ring.
// LDS images are lane-linear (the DMA writes base + lane*16), so the XOR swizzle is applied to the SOURCE address
// and again on the fragment reads (guide rule: both sides or neither).
#pragma once
#include "gemm.cuh"

namespace ge2e {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_cvoid_t;

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)g, (lds_void_t*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const void* g, unsigned char* lds_wave_base) {      // 4 B per lane: base + lane * 4
    __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)g, (lds_void_t*)lds_wave_base, 4, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int WS_NSTG = 4;     // ring slots
constexpr int WS_D = 3;        // tiles in flight ahead of the one being consumed

template <int EPI, int K_>
constexpr size_t gemm_ws_smem() {
    constexpr bool HAS_R = (EPI == EPI_MASK || EPI == EPI_ADD || EPI == EPI_LN);
    return (size_t)WS_NSTG * 16 * K_ * 2 + (HAS_R ? (size_t)4 * WS_NSTG * 2048 : EPI == EPI_MASKBITS ? (size_t)4 * WS_NSTG * 256 : 0) +
           4 * 2048 + (EPI == EPI_LN ? 3 * 1024 + 2 * 4 * 16 * 8 : 0);
}

// grid = 8 * (N / 256) * (parts / 8) blocks; `parts` (multiple of 8) row partitions, `ntiles` = ceil(M / 16)
template <typename T, int EPI, int K_>
__global__ void __launch_bounds__(256) gemm_ws_kernel(const GemmArgs p, const int parts, const int ntiles) {
    static_assert(sizeof(T) == 2, "16-bit storage modes (bf16_t / f16_t)");
    constexpr int KGN = K_ / 32;                 // k-groups
    constexpr int ROWB = K_ * 2;                 // bytes per A row
    constexpr int CPR = ROWB / 16;               // 16-byte chunks per A row (32 / 16)
    constexpr int RPI = 64 / CPR;                // A rows per DMA instruction (2 / 4)
    constexpr int NA = 4 / RPI;                  // A DMA instructions per wave and tile (4 rows per wave)
    constexpr bool HAS_R = (EPI == EPI_MASK || EPI == EPI_ADD || EPI == EPI_LN);
    constexpr bool HAS_B = (EPI == EPI_MASKBITS);   // the mask as bits: 16 rows x 16 B per wave and tile (the wave uses 8 B of each row)
    constexpr int NR = HAS_R ? 2 : (HAS_B ? 1 : 0);   // R DMA instructions per wave and tile (16 rows x 128 B; bits: one 4-byte-per-lane piece)
    constexpr int ATILE = 16 * ROWB;
    constexpr int D = WS_D, NSTG = WS_NSTG;
    static_assert(ROWB % 256 == 0, "K must be a multiple of 128");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const As = smem;                                         // [NSTG][16][ROWB]
    unsigned char* const Rs = smem + NSTG * ATILE;                          // [4][NSTG][2048]
    unsigned char* const Os = Rs + (HAS_R ? 4 * NSTG * 2048 : HAS_B ? 4 * NSTG * 256 : 0);   // [4][2048]
    float* const Ls = (float*)(Os + 4 * 2048);                              // EPI_LN: bias, gamma, beta [3][256]
    float* const Xs = Ls + 3 * 256;                                         // EPI_LN: [2][4 waves][16 rows] (mean, M2)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    // block -> (column group, row partition): the column groups of one partition are neighbours on one XCD
    const int CG = p.N / 256;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cg = slot % CG, part = (slot / CG) * 8 + xcd;
    const int n0 = cg * 256 + wave * 64;
    const int my = part < ntiles ? (ntiles - part + parts - 1) / parts : 0;
    if (my == 0) return;

    // ---- stationary operand: this wave's 64 weight rows as MFMA fragments
    u32x4 wf[4][KGN];
    f32x4 b4[4];
    f32x4 ape[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};   // EPI_PRENET: alpha pe[frame][this lane's columns]
    {
        const unsigned char* W = (const unsigned char*)p.W;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int kg = 0; kg < KGN; ++kg)
                wf[nt][kg] = *(const u32x4*)(W + ((size_t)(n0 + nt * 16 + i) * p.ldw + kg * 32 + g * 8) * 2);
            if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU_DROP || EPI == EPI_PRENET) b4[nt] = *(const f32x4*)(p.bias + n0 + nt * 16 + 4 * g);
            else b4[nt] = f32x4{0, 0, 0, 0};
        }
        if constexpr (EPI == EPI_LN) {          // N == 256: the block holds whole rows; per-column constants live in LDS
            Ls[tid] = p.bias[tid]; Ls[256 + tid] = p.gamma[tid]; Ls[512 + tid] = p.beta[tid];
            __syncthreads();                     // no DMA is in flight yet: an ordinary barrier
        }
        if constexpr (EPI == EPI_PRENET) {
            // alpha pe[frame of row i] for this lane's 16 columns, in registers for the block's whole life: the launcher chooses `parts` so that
            // 16 parts is a multiple of T -- every tile of a block then starts at the same frame offset: (16 (part + j parts) + i) % T = (16 part + i) % T
            const float al = *p.alpha;
            const int frame = (int)(((long long)16 * part + i) % p.T);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 pe4 = *(const f32x4*)(p.pe_t + (size_t)frame * p.N + n0 + nt * 16 + 4 * g);
                ape[nt] = f32x4{al * pe4[0], al * pe4[1], al * pe4[2], al * pe4[3]};
            }
        }
        // retire these ordinary loads before the first DMA: the compiler's own waits stay out of the loop
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) asm volatile("" : "+v"(ape[nt]));
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int kg = 0; kg < KGN; ++kg) asm volatile("" : "+v"(wf[nt][kg]));
            asm volatile("" : "+v"(b4[nt]));
        }
    }

    const unsigned char* const Ag = (const unsigned char*)p.A;
    const unsigned char* const Rg = (const unsigned char*)p.R;
    const int last_row = p.M - 1;
    // DMA of the j-th tile of this block into ring slot j % NSTG (j past the end re-fetches the last tile: the
    // per-iteration instruction counts stay fixed, which is what the counted waits rely on)
    auto issue = [&](int j) {
        const int jj = j < my ? j : my - 1;
        const int r0 = (part + jj * parts) * 16, s = j & (NSTG - 1);
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int r = 4 * wave + q * RPI + lane / CPR, pos = lane % CPR;
            const int c = pos ^ (r & 15);
            int gr = r0 + r; gr = gr < last_row ? gr : last_row;
            glds16(Ag + (size_t)gr * p.lda * 2 + c * 16, As + s * ATILE + (4 * wave + q * RPI) * ROWB);
        }
        if constexpr (HAS_R) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = q * 8 + (lane >> 3), pos = lane & 7;
                const int c = pos ^ ((r >> 1) & 7);
                int gr = r0 + r; gr = gr < last_row ? gr : last_row;
                glds16(Rg + ((size_t)gr * p.ldr + n0) * 2 + c * 16, Rs + (wave * NSTG + s) * 2048 + q * 1024);
            }
        }
        if constexpr (HAS_B) {       // row lane >> 2, 32-bit word lane & 3 of the 128 mask bits around this wave's 64 columns
            int gr = r0 + (lane >> 2); gr = gr < last_row ? gr : last_row;
            glds4(Rg + (size_t)gr * p.ldr + (n0 >> 7) * 16 + (lane & 3) * 4, Rs + (wave * NSTG + s) * 256);
        }
    };

#pragma unroll
    for (int j = 0; j < D; ++j) issue(j);

    constexpr int PER = NA + NR;                 // DMA instructions per tile
    unsigned char* const Ow = Os + wave * 2048;
    T* const Cg = (T*)p.C;
    const uint32_t drm = p.drow_mul > 0 ? (uint32_t)p.drow_mul : 1u;

#pragma unroll 1
    for (int t = 0; t < my; ++t) {
        // tile t has landed once at most (D-1) younger tiles' DMAs and the stores issued since are outstanding
        if (t >= D) wait_vmcnt<(D - 1) * PER + 2 * D>();
        else if (t == 0) wait_vmcnt<(D - 1) * PER>();
        else if (t == 1) wait_vmcnt<(D - 1) * PER + 2>();
        else wait_vmcnt<(D - 1) * PER + 4>();
        __builtin_amdgcn_s_barrier();             // everyone's pieces of tile t are in; slot of tile t-1 is free
        __builtin_amdgcn_sched_barrier(0);
        issue(t + D);

        const int s = t & (NSTG - 1);
        const unsigned char* const a = As + s * ATILE;
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int kg = 0; kg < KGN; ++kg) {
            const u32x4 af = lds16(a + swz_off<ROWB>(i, kg * 4 + g));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = mma16<T>(wf[nt][kg], af, acc[nt]);   // acc[nt][r] = C[row i][n0 + 16nt + 4g + r]
        }
        const int r0 = (part + t * parts) * 16;
        const int row = r0 + i;
        if constexpr (EPI == EPI_LN) {
            // v = LN(residual + drop(acc + bias)) over the 256 columns of a row, which the four waves hold 64 each:
            // every wave reduces its slab to (mean, M2) per row, one LDS exchange combines them (Chan's formula)
            const unsigned char* const rsl = Rs + (wave * NSTG + s) * 2048;
            f32x4 v[4];
            float sm = 0.0f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int lc = wave * 64 + nt * 16 + 4 * g;
                const int so = i * 128 + (((2 * nt + (g >> 1)) ^ ((i >> 1) & 7)) << 4) + (g & 1) * 8;
                v[nt] = acc[nt] + *(const f32x4*)(Ls + lc);
                drop_apply4(p.drop, (uint32_t)row * drm * 256u + (uint32_t)lc, v[nt]);
                v[nt] += load4((const T*)(rsl + so));
                sm += (v[nt][0] + v[nt][1]) + (v[nt][2] + v[nt][3]);
            }
            const float mw = cross4_sum(sm) * (1.0f / 64.0f);
            float qw = 0.0f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = v[nt][r] - mw; qw += d * d; }
            qw = cross4_sum(qw);
            float* const xs = Xs + (t & 1) * 128;
            if (g == 0) { xs[(wave * 16 + i) * 2] = mw; xs[(wave * 16 + i) * 2 + 1] = qw; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            float mean = 0.0f, m2 = 0.0f, mws[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) { mws[w] = xs[(w * 16 + i) * 2]; mean += mws[w]; m2 += xs[(w * 16 + i) * 2 + 1]; }
            mean *= 0.25f;
#pragma unroll
            for (int w = 0; w < 4; ++w) m2 += 64.0f * (mws[w] - mean) * (mws[w] - mean);
            const float rs = 1.0f / sqrtf(m2 * (1.0f / 256.0f) + p.eps);
            if (p.rstd && g == 0 && (i >> 2) == wave && row < p.M) p.rstd[row] = rs;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int lc = wave * 64 + nt * 16 + 4 * g;
                const int so = i * 128 + (((2 * nt + (g >> 1)) ^ ((i >> 1) & 7)) << 4) + (g & 1) * 8;
                const f32x4 ga = *(const f32x4*)(Ls + 256 + lc), be = *(const f32x4*)(Ls + 512 + lc);
                store4((T*)(Ow + so), (v[nt][0] - mean) * rs * ga[0] + be[0], (v[nt][1] - mean) * rs * ga[1] + be[1],
                       (v[nt][2] - mean) * rs * ga[2] + be[2], (v[nt][3] - mean) * rs * ga[3] + be[3]);
            }
        } else
        // ---- wave-private epilogue: 16 rows x 64 columns
        {
        [[maybe_unused]] unsigned long long sbits = 0ull;     // EPI_PRENET: this lane's sign nibbles of the wave's 64 columns of the row
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 v = acc[nt] + b4[nt];
            const int col = n0 + nt * 16 + 4 * g;
            // staged 16 x 128 B slab, 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7); this lane: chunk 2nt + (g >> 1), half g & 1
            const int so = i * 128 + (((2 * nt + (g >> 1)) ^ ((i >> 1) & 7)) << 4) + (g & 1) * 8;
            if constexpr (EPI == EPI_BIAS_RELU_DROP) (void)relu_drop_apply4(p.drop, (uint32_t)row * drm * (uint32_t)p.N + (uint32_t)col, v);
            if constexpr (EPI == EPI_PRENET) {      // h0 = drop(relu(x Wp^T + b) + alpha pe[frame]); the pre-activation's signs leave as bits for the backward
                if (p.relu_bits) {
                    const unsigned nib = (unsigned)(v[0] > 0.0f) | ((unsigned)(v[1] > 0.0f) << 1) | ((unsigned)(v[2] > 0.0f) << 2) | ((unsigned)(v[3] > 0.0f) << 3);
                    sbits |= (unsigned long long)nib << (16 * nt + 4 * g);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f) + ape[nt][r];
                drop_apply4(p.drop, (uint32_t)row * drm * (uint32_t)p.N + (uint32_t)col, v);
            }
            if constexpr (EPI == EPI_MASK) {
                const f32x4 m4 = load4((const T*)(Rs + (wave * NSTG + s) * 2048 + so));
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = m4[r] > 0.0f ? v[r] * p.mask_scale : 0.0f;
            }
            if constexpr (EPI == EPI_MASKBITS) {
                // the 128 bits of row i around this wave's columns, in the producer's byte order (ffn.cuh ffn_mask_byte): column u sits in
                // byte 4 ((u >> 3) & 3) + ((u >> 5) & 3), bit u & 7; here u = (n0 & 127) + 16 nt + 4 g + r
                const u32x4 w4 = *(const u32x4*)(Rs + (wave * NSTG + s) * 256 + i * 16);
                const uint32_t wsel = (nt & 1) ? ((g >> 1) ? w4[3] : w4[2]) : ((g >> 1) ? w4[1] : w4[0]);      // word (2 nt + (g >> 1)) & 3
                const uint32_t nib = wsel >> (8 * (((n0 >> 6) & 1) * 2 + (nt >> 1)) + 4 * (g & 1));
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = ((nib >> r) & 1u) ? v[r] * p.mask_scale : 0.0f;
            }
            if constexpr (EPI == EPI_ADD) v += load4((const T*)(Rs + (wave * NSTG + s) * 2048 + so));
            store4((T*)(Ow + so), v[0], v[1], v[2], v[3]);
        }
        if constexpr (EPI == EPI_PRENET) {
            if (p.relu_bits) {      // (wave-uniform) the four lanes of a row hold disjoint nibbles: OR them, one 8-byte store per row.  (The counted waits
                                    // do not count this store: a wave past M skips it, and a count may be too small, never too large)
                unsigned lo = (unsigned)sbits, hi = (unsigned)(sbits >> 32);
                lo |= __shfl_xor(lo, 16, 64); lo |= __shfl_xor(lo, 32, 64);
                hi |= __shfl_xor(hi, 16, 64); hi |= __shfl_xor(hi, 32, 64);
                if (g == 0 && row < p.M) *(u32x2*)(p.relu_bits + (size_t)row * (p.N / 8) + n0 / 8) = u32x2{lo, hi};
            }
        }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = q * 8 + (lane >> 3), c = lane & 7;
            const u32x4 o = lds16(Ow + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
            if (r0 + r < p.M) __builtin_nontemporal_store(o, (u32x4*)((unsigned char*)Cg + ((size_t)(r0 + r) * p.ldc + n0) * 2 + c * 16));
        }
    }
}


// ---------------------------------------------------------------------------------------------
// LayerNorm backward fused into the weight-stationary GEMM that consumes its result (bf16 mode, K = 256 = the model
// width, so an A tile is 16 WHOLE rows):
//     dx    = LN_backward(dy, y, rstd, gamma, beta)         -> dpre  (residual path)
//     dm    = dropout(dx)                                    -> dmask (input gradient of the sub-layer)
//     C     = epilogue(dm . W^T)                             (EPI_MASK: dF = (dG W2) o relu'/dropout;  EPI_NONE: dO = dA Wo)
// replaces ln_bwd_kernel + gemm_ws/gemm_nt: the dm tile never comes back from HBM and one launch disappears from the
// backward chain.  dy and y tiles arrive by LDS-DMA (every wave fetches the 4 rows it will process: its own counted
// wait is enough before the prologue); the prologue is the wave-per-row LayerNorm-backward of misc.cuh, writes dpre /
// dmask rows straight to HBM (512 contiguous bytes per row; column group 0 only) and overwrites the dy tile in LDS
// with dm, which the MFMA phase then reads as its A operand.  dgamma / dbeta accumulate in registers over the block's
// whole life (persistent blocks: one atomic flush per block).  Rings of 4 slots (EPI_NONE, 76 KB) or 3 slots (EPI_MASK with
// its mask ring, 75 KB): two blocks per CU.  In the step only the EPI_NONE form (norm1 backward + dO) is used: with EPI_MASK
// the four column-group blocks of a row partition each redo the prologue and the fused launch lost (327 vs 260 us).
// ---------------------------------------------------------------------------------------------
struct LnFuseArgs {
    const void* y; int ldy;      // saved LayerNorm OUTPUT [M, 256] of T (xhat is rebuilt from it)
    void* dpre;                  // [M, 256] of T
    void* dmask;                 // [M, 256] of T, or null when the dropout is inactive (then A = dpre)
    float* dgamma; float* dbeta; // [256] fp32, atomically accumulated
};

template <int EPI> constexpr int wsf_nstg() { return EPI == EPI_MASK ? 3 : 4; }   // the mask ring costs a slot (two blocks per CU)
template <int EPI>
constexpr size_t gemm_ws_lnbwd_smem() {
    constexpr int NSTG = wsf_nstg<EPI>();
    return (size_t)NSTG * 8192 * 2 + (EPI == EPI_MASK ? 4 * NSTG * 2048 : 4 * 2048) + (size_t)NSTG * 4 * 256;
}

template <int N, int D> struct WsfWait {
    static __device__ __forceinline__ void at(int t, int st) {   // st = stores per tile of this block: 2, 6 or 10
        const int k = t < D ? t : D;                              // tiles whose stores lie behind tile t's DMAs
        // younger operations than tile t's DMAs: (D-1) tiles of N DMAs + k tiles of st stores
        static_assert(D == 2 || D == 3, "ring depth");
        if (st == 2) {
            if (k == 0) wait_vmcnt<(D - 1) * N>(); else if (k == 1) wait_vmcnt<(D - 1) * N + 2>();
            else if (k == 2) wait_vmcnt<(D - 1) * N + 4>(); else wait_vmcnt<(D - 1) * N + 6>();
        } else if (st == 6) {
            if (k == 0) wait_vmcnt<(D - 1) * N>(); else if (k == 1) wait_vmcnt<(D - 1) * N + 6>();
            else if (k == 2) wait_vmcnt<(D - 1) * N + 12>(); else wait_vmcnt<(D - 1) * N + 18>();
        } else {
            if (k == 0) wait_vmcnt<(D - 1) * N>(); else if (k == 1) wait_vmcnt<(D - 1) * N + 10>();
            else if (k == 2) wait_vmcnt<(D - 1) * N + 20>(); else wait_vmcnt<(D - 1) * N + 30>();
        }
    }
};

template <typename T, int EPI>
__global__ void __launch_bounds__(256, 2) gemm_ws_lnbwd_kernel(const GemmArgs p, const LnFuseArgs q, const int parts, const int ntiles) {
    static_assert(sizeof(T) == 2, "16-bit storage modes (bf16_t / f16_t)");
    static_assert(EPI == EPI_MASK || EPI == EPI_NONE, "consumers of a LayerNorm backward");
    constexpr int KGN = 8, ROWB = 512, ATILE = 16 * ROWB;
    constexpr bool HAS_R = (EPI == EPI_MASK);
    constexpr int NSTG = wsf_nstg<EPI>(), D = NSTG - 1;
    constexpr int PER = 2 + 2 + (HAS_R ? 2 : 0) + 1;          // dy, y, mask slab, rstd
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const As = smem;                               // [NSTG][16][512]  dy, overwritten with dm
    unsigned char* const Ys = smem + NSTG * ATILE;                // [NSTG][16][512]  y
    unsigned char* const Rs = Ys + NSTG * ATILE;                  // [4 waves][NSTG][2048] mask slabs (EPI_NONE: [4][2048] output stage)
    float* const Qs = (float*)(Rs + (HAS_R ? 4 * NSTG * 2048 : 4 * 2048));   // [NSTG][4 waves][64] rstd of the wave's rows (lane & 3)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int CG = p.N / 256;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cg = slot % CG, part = (slot / CG) * 8 + xcd;
    const int n0 = cg * 256 + wave * 64;
    const int my = part < ntiles ? (ntiles - part + parts - 1) / parts : 0;
    if (my == 0) return;
    const bool writer = cg == 0;                                  // one column group writes dpre / dmask / dgamma / dbeta
    const int st_per = writer ? (q.dmask ? 10 : 6) : 2;

    u32x4 wf[4][KGN];
    const int c0 = lane * 4;                                      // LayerNorm-backward: 4 columns per lane
    f32x4 ga, be, ig;
    {
        const unsigned char* W = (const unsigned char*)p.W;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kg = 0; kg < KGN; ++kg)
                wf[nt][kg] = *(const u32x4*)(W + ((size_t)(n0 + nt * 16 + i) * p.ldw + kg * 32 + g * 8) * 2);
        ga = *(const f32x4*)(p.gamma + c0); be = *(const f32x4*)(p.beta + c0);
#pragma unroll
        for (int r = 0; r < 4; ++r) ig[r] = ga[r] != 0.0f ? 1.0f / ga[r] : 0.0f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kg = 0; kg < KGN; ++kg) asm volatile("" : "+v"(wf[nt][kg]));
        asm volatile("" : "+v"(ga), "+v"(be), "+v"(ig));
    }

    const unsigned char* const Ag = (const unsigned char*)p.A;
    const unsigned char* const Yg = (const unsigned char*)q.y;
    const unsigned char* const Rg = (const unsigned char*)p.R;
    const int last_row = p.M - 1;
    auto issue = [&](int j) {
        const int jj = j < my ? j : my - 1;
        const int r0 = (part + jj * parts) * 16, s = j % NSTG;
#pragma unroll
        for (int u = 0; u < 2; ++u) {                             // this wave's 4 rows of dy and of y: 2 rows per instruction
            const int r = 4 * wave + 2 * u + (lane >> 5), pos = lane & 31;
            const int c = pos ^ (r & 15);
            int gr = r0 + r; gr = gr < last_row ? gr : last_row;
            glds16(Ag + (size_t)gr * p.lda * 2 + c * 16, As + s * ATILE + (4 * wave + 2 * u) * ROWB);
            glds16(Yg + (size_t)gr * q.ldy * 2 + c * 16, Ys + s * ATILE + (4 * wave + 2 * u) * ROWB);
        }
        if constexpr (HAS_R) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = u * 8 + (lane >> 3), pos = lane & 7;
                const int c = pos ^ ((r >> 1) & 7);
                int gr = r0 + r; gr = gr < last_row ? gr : last_row;
                glds16(Rg + ((size_t)gr * p.ldr + n0) * 2 + c * 16, Rs + (wave * NSTG + s) * 2048 + u * 1024);
            }
        }
        {
            int gr = r0 + 4 * wave + (lane & 3); gr = gr < last_row ? gr : last_row;
            __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(p.rstd + gr), (lds_void_t*)(Qs + (s * 4 + wave) * 64), 4, 0, 0);
        }
    };
#pragma unroll
    for (int j = 0; j < D; ++j) issue(j);

    T* const Cg = (T*)p.C;
    const uint32_t drm = p.drow_mul > 0 ? (uint32_t)p.drow_mul : 1u;
    f32x4 ag = f32x4{0, 0, 0, 0}, ab = f32x4{0, 0, 0, 0};
    // position of this lane's 4 columns (8 bytes) inside a swizzled 512-byte row: chunk lane >> 1, half lane & 1
    const int lch = lane >> 1, lhf = (lane & 1) * 8;

#pragma unroll 1
    for (int t = 0; t < my; ++t) {
        WsfWait<PER, D>::at(t, st_per);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();             // tile t is in (each wave: its own rows; the barrier orders the slot reuse)
        __builtin_amdgcn_sched_barrier(0);
        issue(t + D);
        const int s = t % NSTG;
        const int r0 = (part + t * parts) * 16;
        unsigned char* const a = As + s * ATILE;
        const unsigned char* const yt = Ys + s * ATILE;
        // ---- prologue: LayerNorm backward of this wave's 4 rows, two at a time (two independent rows interleave their
        // reductions; four at once cost 32 more VGPRs and the second resident block)
#pragma unroll 1
        for (int h2 = 0; h2 < 2; ++h2) {
            f32x4 dx[2], xh[2];
            float s1[2], s2[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = 4 * wave + 2 * h2 + u;
                const int off = r * ROWB + ((lch ^ (r & 15)) << 4) + lhf;
                const f32x4 dy = load4((const T*)(a + off));
                const f32x4 y = load4((const T*)(yt + off));
                float t1 = 0.0f, t2 = 0.0f;
                const bool live = r0 + r < p.M;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[u][e] = (y[e] - be[e]) * ig[e];
                    dx[u][e] = dy[e] * ga[e];
                    t1 += dx[u][e];
                    t2 += dx[u][e] * xh[u][e];
                    if (live) { ag[e] += dy[e] * xh[u][e]; ab[e] += dy[e]; }
                }
                s1[u] = t1; s2[u] = t2;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) { s1[u] = wave_sum(s1[u]) * (1.0f / 256.0f); s2[u] = wave_sum(s2[u]) * (1.0f / 256.0f); }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = 4 * wave + 2 * h2 + u, row = r0 + r;
                const int off = r * ROWB + ((lch ^ (r & 15)) << 4) + lhf;
                const float rs = Qs[(s * 4 + wave) * 64 + 2 * h2 + u];
#pragma unroll
                for (int e = 0; e < 4; ++e) dx[u][e] = rs * (dx[u][e] - s1[u] - xh[u][e] * s2[u]);
                if (writer && row < p.M) store4((T*)q.dpre + (size_t)row * 256 + c0, dx[u][0], dx[u][1], dx[u][2], dx[u][3]);
                if (q.dmask) {
                    drop_apply4(p.drop, (uint32_t)row * drm * 256u + (uint32_t)c0, dx[u]);
                    if (writer && row < p.M) store4((T*)q.dmask + (size_t)row * 256 + c0, dx[u][0], dx[u][1], dx[u][2], dx[u][3]);
                }
                store4((T*)(a + off), dx[u][0], dx[u][1], dx[u][2], dx[u][3]);      // the GEMM's A operand
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA phase
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int kg = 0; kg < KGN; ++kg) {
            const u32x4 af = lds16(a + swz_off<ROWB>(i, kg * 4 + g));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = mma16<T>(wf[nt][kg], af, acc[nt]);
        }
        // ---- wave-private epilogue through the wave's mask slab (EPI_MASK: read, then reused as the output stage)
        unsigned char* const Ow = HAS_R ? Rs + (wave * NSTG + s) * 2048 : Rs + wave * 2048;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 v = acc[nt];
            const int so = i * 128 + (((2 * nt + (g >> 1)) ^ ((i >> 1) & 7)) << 4) + (g & 1) * 8;
            if constexpr (EPI == EPI_MASK) {
                const f32x4 m4 = load4((const T*)(Ow + so));
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = m4[r] > 0.0f ? v[r] * p.mask_scale : 0.0f;
            }
            store4((T*)(Ow + so), v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = u * 8 + (lane >> 3), c = lane & 7;
            const u32x4 o = lds16(Ow + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
            if (r0 + r < p.M) __builtin_nontemporal_store(o, (u32x4*)((unsigned char*)Cg + ((size_t)(r0 + r) * p.ldc + n0) * 2 + c * 16));
        }
    }
    // ---- dgamma / dbeta: combine the four waves through LDS, one atomic per column and block
    wait_vmcnt<0>();
    __syncthreads();
    if (writer) {
        float* const red = (float*)smem;                          // [2][4][256]
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[(0 * 4 + wave) * 256 + c0 + e] = ag[e]; red[(1 * 4 + wave) * 256 + c0 + e] = ab[e]; }
        __syncthreads();
        const int c = tid;
        atomicAdd(q.dgamma + c, red[c] + red[256 + c] + red[512 + c] + red[768 + c]);
        atomicAdd(q.dbeta + c, red[1024 + c] + red[1280 + c] + red[1536 + c] + red[1792 + c]);
    }
}

}  // namespace ge2e
