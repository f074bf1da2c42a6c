// Weight gradients of the wide products, 16-bit storage modes:  dW[n][k] = sum_r Y[r][n] X[r][k]   (+ db[n] = sum_r Y[r][n])
//
// A weight gradient is a GEMM with a tiny output (256 x 1024) and a huge reduction (R = 153,600 rows), both operands streamed.
// The 128 x 128-tile kernel of gemm.cuh needs 32 KB of operands per 512 MFMA-cycles and CU -- twice what a CU can ingest from
// L2 (~70 GB/s) -- so it runs at the L2 -> CU path's speed, re-reading every operand panel 2-8 times.  This kernel:
//   * one 512-thread block per CU owns a 256 x 256 output tile over a slice of the rows (split-K): 32 KB of operands per 1024
//     MFMA-cycles, the balance point of the ingest path and the matrix pipe; wave = 64 n x 128 k (128 accumulator registers);
//   * 32-row stages (Y slab 32 x 256, X slab 32 x 256 = 32 KB) arrive by LDS-DMA into a 4-slot ring, three stages ahead, behind a
//     counted s_waitcnt vmcnt(N) and one raw s_barrier per stage; both MFMA operands are read transposed
//     (ds_read_b64_tr_b16) from the lane-linear slabs: 8 lane-dependent base addresses + immediates;
//   * the two waves of a SIMD (w and w + 4) run in OPPOSITE phase: between two barriers waves 0-3 read the stage's 12 fragments
//     and then issue its 32 MFMAs, waves 4-7 first issue the MFMAs of the fragments they read in the previous interval and
//     then read this stage's -- one wave of a SIMD is always on the matrix pipe while its partner is on LDS (in lockstep the
//     read phase and the MFMA phase of a stage simply add up);
//   * the fp32 tile leaves as PLAIN 16-byte stores in fragment order into a partial slab (split x tile x 256 KB; coalesced,
//     no atomics: 64 MB of float atomics per launch would take 50 us at the chip's 1.3 TB/s atomic rate), and
//     wgrad_ks_reduce_kernel sums the splits and un-permutes into dW;
//   * block -> (tile, row slice) is XCD-aware: the tiles of ONE row slice share their narrow operand panel (dW2: the 256 columns
//     of dM, dW1 / in_proj: the 256 columns of h), so they sit on ONE XCD (blocks b and b + 8 share an XCD under round-robin
//     dispatch; speed only) at neighbouring positions and stream that panel through the XCD's L2 together: it comes from HBM once
//     per row slice instead of once per tile (round 2 put them on consecutive block ids = different XCDs: PMC 342 MB fetched
//     per launch for 230 MB algorithmic);
//   * the bias gradient rides along as one extra MFMA per n-tile against a fragment of ones (k-tile-0 blocks only).
// Rows beyond the last multiple of 32 and the small shapes (one 256 x 256 tile or less, the prenet's K = 80) stay on the
// 128 x 128 kernel.  (A variant with ONE wave per SIMD, 128 x 128 accumulators per wave in the accumulator file and double-
// buffered fragments was built first: hipcc moved the 256 accumulator registers between the two register files twice per stage
// -- 283 v_accvgpr moves and 153 scratch accesses per stage in the .s, with the MFMA builtin and with "a"-constrained inline
// asm alike -- so the layout that stays within 256 registers per wave is the one that ships.)
#pragma once
#include "gemm_ws.cuh"

namespace ge2e {

constexpr int WK_NSTG = 4, WK_D = 3;
constexpr int WK_SLOT = 32 * 1024;
constexpr size_t wgrad_ks_smem() { return (size_t)WK_NSTG * WK_SLOT; }
constexpr size_t WK_TILE_FLOATS = 256 * 256;

struct WgradKsArgs {
    const void* Y; int ldy;      // [R, N] of T
    const void* X; int ldx;      // [R, K] of T
    float* part;                 // [splits][tiles][256 * 256] fp32 partial tiles in fragment order
    float* db;                   // [N] fp32, atomically accumulated, or null
    int R32;                     // rows handled here (multiple of 32)
    int rows_per_split;          // multiple of 32
    int tiles_n, tiles_k;        // 256 x 256 tiles along N and K
    int splits;                  // row slices; grid = 8 * tiles * ceil(splits / 8) blocks, the surplus ones exit at once
};

template <typename T> __device__ __forceinline__ constexpr unsigned wk_one2();
template <> __device__ __forceinline__ constexpr unsigned wk_one2<bf16_t>() { return 0x3F803F80u; }
template <> __device__ __forceinline__ constexpr unsigned wk_one2<f16_t>() { return 0x3C003C00u; }

// two transposed 8-byte reads (rows 0-15 and 16-31 of the slab) = one 16x16x32 operand fragment.  Inline asm: hipcc puts a full
// s_waitcnt vmcnt(0) in front of the ds_read_tr builtin while an LDS-DMA is in flight, which would drain the ring every stage.
// The caller retires the reads with wk_lds_retire() before the first MFMA that uses them.
__device__ __forceinline__ void wk_tr_read(u32x4& f, unsigned addr) {
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:8192" : "=&v"(lo), "=&v"(hi) : "v"(addr));
    f = u32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ void wk_lds_retire(u32x4* f) {      // all 12 fragments complete (LDS returns in order)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]),
                   "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]));
    __builtin_amdgcn_sched_barrier(0);
}

// grid = 8 * tiles_n * tiles_k * ceil(splits / 8) blocks (<= one per CU) of 512 threads
template <typename T>
__global__ void __launch_bounds__(512) wgrad_ks_kernel(const WgradKsArgs p) {
    static_assert(sizeof(T) == 2, "16-bit storage modes");
    constexpr int D = WK_D, NSTG = WK_NSTG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wk = wave >> 2;                 // waves w and w + 4 (one SIMD) share the n-range and split the k-range
    const bool lag = wave >= 4;
    const int i = lane & 15, g = lane >> 4;
    const int ntile = p.tiles_n * p.tiles_k;
    // XCD x = b & 7 holds row slices x, x + 8, ...; position j = b >> 3 inside the XCD walks the tiles of a slice first
    const int xj = blockIdx.x >> 3;
    const int tile = xj % ntile;
    const int split = (xj / ntile) * 8 + ((int)blockIdx.x & 7);
    if (split >= p.splits) return;                          // (whole block, before any barrier)
    const int n0 = (tile % p.tiles_n) * 256, k0 = (tile / p.tiles_n) * 256;
    const int rbeg = split * p.rows_per_split;
    const int rend = min(p.R32, rbeg + p.rows_per_split);
    const int nst = (rend - rbeg) / 32;                     // may be 0: the block then writes a zero partial tile

    const unsigned char* const Yg = (const unsigned char*)p.Y;
    const unsigned char* const Xg = (const unsigned char*)p.X;
    // stage st -> ring slot st % NSTG: 32 DMA instructions of 2 rows x 512 B (0-15: Y slab, 16-31: X slab); wave w issues 4w .. 4w + 3.
    // 16-byte chunk c of slab row `row` is stored at chunk c ^ (2 (row & 7)) (applied to the SOURCE address; the reads undo it)
    auto issue = [&](int st) {
        const int sc = st < nst ? st : (nst > 0 ? nst - 1 : 0);     // past the end: re-fetch the last stage (fixed instruction count)
        unsigned char* const slot = smem + (st & (NSTG - 1)) * WK_SLOT;
        const size_t r0 = (size_t)(nst > 0 ? rbeg + sc * 32 : 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id = 4 * wave + u;                     // wave-uniform: waves 0-3 carry the Y slab, waves 4-7 the X slab
            const int row = 2 * (id & 15) + (lane >> 5);
            const int c = (lane & 31) ^ (2 * (row & 7));
            const unsigned char* src = (id < 16) ? Yg + ((r0 + row) * p.ldy + n0) * 2 + c * 16
                                                 : Xg + ((r0 + row) * p.ldx + k0) * 2 + c * 16;
            glds16(src, slot + id * 1024);
        }
    };
#pragma unroll 1
    for (int s = 0; s < D; ++s) issue(s);

    // transposed-fragment addresses: lane (i, g) of column tile tt (16 columns) reads row 4g + (i >> 2) (and + 16), the 8 bytes at
    // chunk 2 tt + ((i & 3) >> 1), half i & 1: byte = row * 512 + ((2 tt ^ 2 (row & 7)) + ((i & 3) >> 1)) * 16 + (i & 1) * 8
    //      = [row * 512 + ((i & 3) >> 1) * 16 + (i & 1) * 8] + 32 ((tt & 7) ^ (row & 7)) + 256 (tt >> 3)
    // -> 8 lane-dependent bases (one per tt & 7) + wave-uniform offsets; the X slab is + 16384
    const int trow = 4 * g + (i >> 2);
    unsigned tb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
        tb[q] = (unsigned)(trow * 512 + ((i & 3) >> 1) * 16 + (i & 1) * 8 + 32 * (q ^ (trow & 7)));
    unsigned tby[4];                                         // this wave's Y tiles are column tiles 4 wn + mt: class (4 wn + mt) & 7
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
        tby[mt] = (unsigned)(trow * 512 + ((i & 3) >> 1) * 16 + (i & 1) * 8 + 32 * ((4 * (wn & 1) + mt) ^ (trow & 7)));

    f32x4 acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
    f32x4 accb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) accb[a] = f32x4{0, 0, 0, 0};
    const bool do_bias = (p.db != nullptr) && k0 == 0 && wk == 0;      // wave-uniform
    const unsigned one2 = wk_one2<T>();                      // two packed 1.0 of the storage type
    const u32x4 ones = u32x4{one2, one2, one2, one2};

    // fragments: [0..3] Y tiles (n = 64 wn + 16 mt: column tile 4 wn + mt), [4..11] X tiles (k = 128 wk + 16 nt: column tile 8 wk + nt)
    auto read_frags = [&](int slot_index, u32x4* f) {
        const unsigned soy = (unsigned)(slot_index * WK_SLOT + (wn >> 1) * 256);
        const unsigned sox = (unsigned)(slot_index * WK_SLOT + 16384 + wk * 256);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) wk_tr_read(f[mt], tby[mt] + soy);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) wk_tr_read(f[4 + nt], tb[nt] + sox);
    };
    auto mfmas = [&](const u32x4* f) {
#ifdef WK_MF32_PROBE
        // TIMING PROBE ONLY (tools/wgrad_ks_bench.hip -DWK_MF32_PROBE; the results are wrong): the stage's 32 v_mfma_f32_16x16x32 replaced by the 16
        // v_mfma_f32_32x32x16 of the same FLOPs on the same 12 fragments and 128 accumulator registers -- what the larger shape could buy at
        // best.  Measured: profiles/r04_ab_log.txt.
        typedef __attribute__((ext_vector_type(16))) float f32x16;
        f32x16* const a32 = (f32x16*)&acc[0][0];
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                a32[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, f[2 * (q & 1) + ks]), __builtin_bit_cast(bf16x8_t, f[4 + 2 * (q >> 1) + ks]), a32[q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        return;
#endif
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)       // acc[mt][nt][r] = dW[n0 + 64 wn + 16 mt + 4g + r][k0 + 128 wk + 16 nt + i]
                acc[mt][nt] = mma16<T>(f[mt], f[4 + nt], acc[mt][nt]);
        if (do_bias) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) accb[mt] = mma16<T>(f[mt], ones, accb[mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // publish stage s (everyone's pieces landed: each wave waits for its own, then the barrier) and refill the slot of stage s - 1
    auto publish = [&](int s) {
        wait_vmcnt<4 * (D - 1)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue(s + D);
    };

    u32x4 f[12];
    if (!lag) {
#pragma unroll 1
        for (int s = 0; s < nst; ++s) {
            publish(s);
            read_frags(s & (NSTG - 1), f);
            wk_lds_retire(f);
            mfmas(f);
        }
    } else if (nst > 0) {
        publish(0);
        read_frags(0, f);
        wk_lds_retire(f);
#pragma unroll 1
        for (int s = 1; s < nst; ++s) {
            publish(s);                          // (the fragments in f were read an interval ago: nothing to wait for)
            mfmas(f);
            read_frags(s & (NSTG - 1), f);
            wk_lds_retire(f);
        }
        mfmas(f);
    }
    wait_vmcnt<0>();                             // the stages issued past the end land before the block's LDS is released

    // (Adding the tile straight into dW with float atomics instead -- no partial slab, no reduce pass -- was measured in round 3 and lost:
    // 3.93 vs 3.77 ms per step; 64-byte atomic segments, 128 per wave.)
    // ---- partial tile in fragment order: part[split][tile][wave][mt][nt][lane] (16 B each): plain, fully coalesced stores
    f32x4* const out = (f32x4*)(p.part + ((size_t)split * ntile + tile) * WK_TILE_FLOATS) + (size_t)wave * 32 * 64 + lane;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) __builtin_nontemporal_store(acc[mt][nt], out + (mt * 8 + nt) * 64);
    if (do_bias && i == 0) {                     // accb[mt][r] = sum_rows Y[.][n0 + 64 wn + 16 mt + 4g + r] (the same in every column i)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(p.db + n0 + 64 * wn + 16 * mt + 4 * g + r, accb[mt][r]);
    }
}

// dW[n][k] += sum over splits of the partial tiles.  A (tile, wave, mt) group of the fragment order -- 16 rows n x 128 columns k --
// is ONE contiguous 8 KB piece of every split's partial tile: a 512-thread block sums that piece over its share of the splits
// (16-byte loads, 8 in flight per thread), turns it through LDS and adds it to dW as whole 512-byte row segments, 256 contiguous
// bytes per wave instruction (the full-rate atomic shape; dW is at most 1 MB and lives in the caches, so these few MB of
// atomics cost nothing next to the partial reads).  grid = (tiles * 32, split groups).
// (Round 2: one thread per fragment element summed ALL splits serially and read-modify-wrote dW in 64-byte pieces: 20-60 us.)
// (Round 3, second half: the reduce passes of ONE layer's products run as one launch -- up to WK_MAX_JOBS jobs, each with its own partial
// slab --, deferred to the layer's last product: a reduce pass is a dependent launch of ~9 us that waits for whole CUs next to kernels
// that fill every register of theirs; in the step the 11 passes measured 31 us on average, 200 us at worst, on the stream that ends last.)
constexpr int WK_MAX_JOBS = 4;
struct WkReduceJob { const float* part; float* dW; int ldw, splits, tiles_n, tiles_k, block0, sgroups; };
struct WkReduceArgs { WkReduceJob job[WK_MAX_JOBS]; int njobs; };

__device__ __forceinline__ void wk_reduce_body(const float* part, float* dW, int ldw, int splits, int tiles_n, int tiles_k, int bx, int by, int sg, float (*tr)[132]) {
    const int ntile = tiles_n * tiles_k;
    const int tile = bx >> 5, grp = bx & 31;          // grp = wave * 4 + mt
    const int t = threadIdx.x, lane = t & 63, nt = t >> 6;
    const f32x4* src = (const f32x4*)(part + (size_t)tile * WK_TILE_FLOATS) + grp * 512 + t;
    const size_t stride = (size_t)ntile * (WK_TILE_FLOATS / 4);
    f32x4 s0 = f32x4{0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
    int sp = by;
    for (; sp + 3 * sg < splits; sp += 4 * sg) {
        const f32x4 a = __builtin_nontemporal_load(src + (size_t)sp * stride), b = __builtin_nontemporal_load(src + (size_t)(sp + sg) * stride);
        const f32x4 c = __builtin_nontemporal_load(src + (size_t)(sp + 2 * sg) * stride), d = __builtin_nontemporal_load(src + (size_t)(sp + 3 * sg) * stride);
        s0 += a; s1 += b; s2 += c; s3 += d;
    }
    for (; sp < splits; sp += sg) s0 += __builtin_nontemporal_load(src + (size_t)sp * stride);
    const f32x4 v = (s0 + s1) + (s2 + s3);
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) tr[4 * g + r][16 * nt + i] = v[r];   // v[r] = dW[n0 + 4g + r][k0 + 16 nt + i]
    __syncthreads();
    const int wave = grp >> 2, mt = grp & 3, wn = wave & 3, wk = wave >> 2;
    const int n0 = (tile % tiles_n) * 256 + 64 * wn + 16 * mt, k0 = (tile / tiles_n) * 256 + 128 * wk;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = q * 512 + t, row = e >> 7, col = e & 127;
        atomicAdd(dW + (size_t)(n0 + row) * ldw + k0 + col, tr[row][col]);
    }
}
__global__ void __launch_bounds__(512) wgrad_ks_reduce_kernel(const float* part, float* dW, int ldw, int splits, int tiles_n, int tiles_k) {
    __shared__ float tr[16][132];
    wk_reduce_body(part, dW, ldw, splits, tiles_n, tiles_k, blockIdx.x, blockIdx.y, gridDim.y, tr);
}
// grid = sum over the jobs of tiles * 32 * sgroups blocks; job j owns blocks [block0_j, block0_{j+1})
__global__ void __launch_bounds__(512) wgrad_ks_reduce_multi_kernel(const WkReduceArgs a) {
    __shared__ float tr[16][132];
    int j = 0;
#pragma unroll
    for (int q = 1; q < WK_MAX_JOBS; ++q) if (q < a.njobs && (int)blockIdx.x >= a.job[q].block0) j = q;
    const WkReduceJob& w = a.job[j];
    const int local = blockIdx.x - w.block0, per = w.tiles_n * w.tiles_k * 32;
    wk_reduce_body(w.part, w.dW, w.ldw, w.splits, w.tiles_n, w.tiles_k, local % per, local / per, w.sgroups, tr);
}

}  // namespace ge2e
