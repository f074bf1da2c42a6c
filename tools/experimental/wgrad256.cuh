// EXPERIMENT (not built into the library): standalone it beats the 128 x 128 kernel on dW2 (131 vs 158 us) and in_proj
// (120 vs 161 us), ties on dW1 and loses on the 256 x 256 shapes (64 MB of atomics per launch); inside the training step,
// beside the main chain, the step time did not move (4.66-4.74 vs 4.70 ms).
// Weight gradient with 256 x 256 output tiles (bf16 mode):  dW[n][k] += sum_r Y[r][n] * X[r][k],  db[n] += sum_r Y[r][n]
//
// The 128 x 128-tile kernel in gemm.cuh moves every operand column slab through the CU once per tile that needs it
// (dW1: 16 U through L2 for 5 U of HBM reads) and that L2 -> CU path, not HBM, is what it saturates (its load /
// LDS-write skeleton alone takes 60 % of its time).  Here a block of 8 waves owns a 256 x 256 tile (wave: 64 n x 128 k,
// 128 accumulator registers), so the traffic through the CU halves, and
//   * 32-row stages (Y slab 32 x 256, X slab 32 x 256, 32 KB) arrive by LDS-DMA into a 4-slot ring, three stages
//     ahead, behind counted s_waitcnt vmcnt(N) and one raw s_barrier per stage: no staging registers, no LDS writes;
//   * both MFMA operands are read transposed (ds_read_b64_tr_b16) from the lane-linear slabs; the bank swizzle
//     (16-byte chunk ^ 2*(row & 7)) is applied to the DMA source address and to the reads;
//   * the bias gradient rides along as one extra MFMA per n-tile against a fragment of ones;
//   * the fp32 tile is flushed with atomics as 256-byte row pieces through per-wave LDS slabs.
// The caller splits the rows into slices of a multiple of 32 rows; rows beyond the last multiple of 32 go to the
// 128 x 128 kernel (one small launch, only for ragged R).
#pragma once
#include "gemm_ws.cuh"

namespace ge2e {

constexpr int WG_NSTG = 4, WG_D = 3;
constexpr int WG_SLOT = 32 * 1024;
constexpr size_t wgrad256_smem() { return (size_t)WG_NSTG * WG_SLOT; }

// transposed fragment from a lane-linear, swizzled slab of 512-byte rows: lane (i, g) gets X[4g + j][c0 + i] (j = 0..3)
// and X[16 + 4g + j][c0 + i].  Issued as inline asm: hipcc puts a full s_waitcnt vmcnt(0) in front of the
// ds_read_tr builtin while an LDS-DMA is in flight (it does not for plain ds_read_b128), which would drain the ring
// every stage.  The caller retires the reads with lds_wait<N>() naming the fragments it is about to use.
__device__ __forceinline__ unsigned tr_addr(const unsigned char* slab, int c0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int row = 4 * g + (i >> 2);
    const int chunk = (c0 >> 3) + ((i & 3) >> 1);
    return (unsigned)(size_t)(lds_void_t*)(slab + row * 512 + ((chunk ^ (2 * (row & 7))) << 4) + (i & 1) * 8);
}
__device__ __forceinline__ void tr_read(u32x4& f, unsigned addr) {
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:8192" : "=&v"(lo), "=&v"(hi) : "v"(addr));
    f = u32x4{lo.x, lo.y, hi.x, hi.y};
}
template <int N> __device__ __forceinline__ void lds_wait4(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

// grid = tiles_n * tiles_k * splits blocks of 512 threads; rows [0, R32) with R32 % 32 == 0, rows_per_split % 32 == 0
__global__ void __launch_bounds__(512) wgrad256_kernel(const WgradArgs p, const int R32) {
    using T = bf16_t;
    constexpr int D = WG_D, NSTG = WG_NSTG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;
    const int i = lane & 15, g = lane >> 4;
    // consecutive (remapped) ids share an XCD and a row slice: the slice's slabs are fetched from HBM once per XCD
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = p.tiles_n * p.tiles_k;
    const int tile = L % ntile, split = L / ntile;
    const int n0 = (tile % p.tiles_n) * 256, k0 = (tile / p.tiles_n) * 256;
    const int rbeg = split * p.rows_per_split;
    const int rend = min(R32, rbeg + p.rows_per_split);
    const int nst = (rend - rbeg) / 32;
    if (nst <= 0) return;

    const unsigned char* const Yg = (const unsigned char*)p.Y;
    const unsigned char* const Xg = (const unsigned char*)p.X;
    // stage st -> ring slot st % NSTG: 32 DMA instructions of 2 rows x 512 B (0-15: Y slab, 16-31: X slab); wave w issues 4w..4w+3
    auto issue = [&](int st) {
        const int sc = st < nst ? st : nst - 1;            // past the end: re-fetch the last stage (fixed instruction count)
        unsigned char* const slot = smem + (st & (NSTG - 1)) * WG_SLOT;
        const int r0 = rbeg + sc * 32;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id = 4 * wave + u;                   // wave-uniform
            const int row = 2 * (id & 15) + (lane >> 5);   // row inside the slab
            const int c = (lane & 31) ^ (2 * (row & 7));
            const unsigned char* src = (id < 16) ? Yg + ((size_t)(r0 + row) * p.ldy + n0) * 2 + c * 16
                                                 : Xg + ((size_t)(r0 + row) * p.ldx + k0) * 2 + c * 16;
            glds16(src, slot + id * 1024);
        }
    };
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s);

    f32x4 acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
    f32x4 accb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) accb[a] = f32x4{0, 0, 0, 0};
    const bool do_bias = (p.db != nullptr) && k0 == 0 && wk == 0;      // wave-uniform
    const u32x4 ones = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

#pragma unroll 1
    for (int st = 0; st < nst; ++st) {
        wait_vmcnt<4 * (D - 1)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue(st + D);
        const unsigned char* const y = smem + (st & (NSTG - 1)) * WG_SLOT;
        const unsigned char* const x = y + 16384;
        // 24 transposed reads (12 fragments) up front; LDS returns in order, so the MFMAs start behind counted lgkmcnt waits
        u32x4 af[4], bf[8];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) tr_read(af[mt], tr_addr(y, wn * 64 + mt * 16, lane));
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) tr_read(bf[nt], tr_addr(x, wk * 128 + nt * 16, lane));
        lds_wait4<14>(af[0], af[1], af[2], af[3]);           // 16 younger reads minus the first two X fragments
        lds_wait4<12>(bf[0], bf[1], bf[0], bf[1]);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            if (nt == 2) lds_wait4<8>(bf[2], bf[3], bf[2], bf[3]);
            if (nt == 4) lds_wait4<4>(bf[4], bf[5], bf[4], bf[5]);
            if (nt == 6) lds_wait4<0>(bf[6], bf[7], bf[6], bf[7]);
            // acc[mt][nt][r] = dW[n0 + 64wn + 16mt + 4g + r][k0 + 128wk + 16nt + i]
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = mma16<T>(af[mt], bf[nt], acc[mt][nt]);
        }
        if (do_bias) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) accb[mt] = mma16<T>(af[mt], ones, accb[mt]);
        }
    }
    // the ring still has (dummy) DMAs in flight: retire them before LDS is reused
    wait_vmcnt<0>();
    __syncthreads();
    // flush: per wave 16 rows x 128 columns fp32 at a time through its own 8 KB slab -> atomics of 256 contiguous bytes
    constexpr int LDT = 128 + 4;                            // floats per staged row
    float* const Ts = (float*)smem + wave * 16 * LDT;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ts[(4 * g + r) * LDT + nt * 16 + i] = acc[mt][nt][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
        for (int row = 0; row < 16; ++row) {
            const int n = n0 + wn * 64 + mt * 16 + row;
            float* const dst = p.dW + (size_t)n * p.ldw + k0 + wk * 128;
            atomicAdd(dst + lane, Ts[row * LDT + lane]);
            atomicAdd(dst + 64 + lane, Ts[row * LDT + 64 + lane]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (do_bias) {
        // accb[mt][r] = sum_r Y[.][n0 + 64wn + 16mt + 4g + r] in every column i: column 0 lanes write
        if (i == 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(p.db + n0 + wn * 64 + mt * 16 + 4 * g + r, accb[mt][r]);
        }
    }
}

}  // namespace ge2e
