// Backward of the prenet (reference Modules.py:10-17,50-52 + the positional encoding's dropout, Modules.py:98-103) in ONE launch.
//
//     h0 = drop( relu(Wp x + bp) + alpha pe )          dD = drop'(dH0)          dalpha = sum dD o pe
//     dZ = dD o [pre > 0]                              dWp = dZ^T x             dbp = sum_r dZ
//
// Nothing downstream needs dZ as a tensor, so it never becomes one: this is the 128 x 128-tile weight-gradient kernel of gemm.cuh whose Y
// operand is MADE on the way from the registers to LDS -- dH0 chunks (16 bytes: 8 columns of a row), the dropout keep bits hashed from
// the element counter, the pre-activation's sign bits from the forward (GemmArgs::relu_bits: 32 bytes per row) -- and which sums dalpha on
// the side (every element of dH0 passes through exactly one block).  Before: a K = 128 GEMM that recomputed the pre-activation to test its
// sign and wrote dZ over dH0 (0.5U + 1U in, 1U out), then the weight gradient reading it back (1U + 0.5U): two launches at the very end of
// the backward's main chain, 110 + 44 us alone.
#pragma once
#include "gemm.cuh"

namespace ge2e {

struct PrenetBwdArgs {
    const void* dH0;             // [R, 256] of T: dL/d(h0)
    const void* X; int ldx;      // [R, ldx] of T: the packed mel rows (ldx = 128)
    const unsigned char* bits;   // [R][32]: sign bits of the pre-activation
    const float* pe_t;           // [T, 256]
    float* dW; int ldw;          // [256, mel] fp32 (atomics)
    float* db;                   // [256]
    float* dalpha;               // scalar
    int R, K;                    // rows; K = mel (dW columns written)
    int nsplit, T;               // blocks per column tile; block `split` takes the 64-row stages split, split + nsplit, ...: RS nsplit is a multiple of T, so a
                                 // thread sees the same FRAME at the same position in every one of its stages (dalpha: see the kernel)
    Drop drop;                   // the positional encoding's dropout (counter row * 256 + col)
};

template <typename T, int NS_ = 3, int KGS = 2>
__global__ void __launch_bounds__(256) prenet_bwd_kernel(const PrenetBwdArgs p) {
    constexpr int KG = Prec<T>::KG, FR = Prec<T>::FRAG;
    constexpr int RS = KGS * KG;                            // rows per stage
    constexpr int ROWB = 128 * (int)sizeof(T);
    constexpr int LD = ROWB + (sizeof(T) == 2 ? 32 : 16);
    constexpr int CPR = ROWB / 16;                          // 16-byte chunks per 128-column row
    constexpr int NCH = RS * CPR / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Ys = smem;                         // [2][RS][LD]
    unsigned char* const Xs = smem + 2 * RS * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = L & 1, split = L >> 1;                 // two 128-column tiles of the 256 output rows, one k tile
    const int n0 = tile * 128;
    const int rend = p.R;
    const int nstg = (p.R + RS - 1) / RS;                   // stages of the whole batch
    const int nst = split < nstg ? (nstg - split + p.nsplit - 1) / p.nsplit : 0;      // ... of this block: split, split + nsplit, ...
    if (nst <= 0) return;
    auto row0 = [&](int st) { return (split + st * p.nsplit) * RS; };

    constexpr int NS = NS_;
    u32x4 rY[NS][NCH], rX[NS][NCH];
    unsigned rB[NS][NCH];
    auto load_stage = [&](int st, u32x4* ry, u32x4* rx, unsigned* rb) {
        const int r0 = row0(st);
        const unsigned char* Y = (const unsigned char*)p.dH0;
        const unsigned char* X = (const unsigned char*)p.X;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int id = tid + 256 * q, row = id / CPR, c = id % CPR, gr = r0 + row;
            const bool ok = gr < rend;
            ry[q] = ok ? *(const u32x4*)(Y + ((size_t)gr * 256 + n0) * sizeof(T) + c * 16) : u32x4{0, 0, 0, 0};
            rx[q] = ok ? *(const u32x4*)(X + (size_t)gr * p.ldx * sizeof(T) + c * 16) : u32x4{0, 0, 0, 0};
            rb[q] = ok ? (unsigned)p.bits[(size_t)gr * 32 + ((n0 + c * FR) >> 3)] : 0u;
        }
    };
    // dalpha = sum dD o pe[frame]: a thread meets the same (frame, columns) at chunk q of every stage (RS nsplit % T == 0), so it sums dD per
    // position and multiplies by pe ONCE at the end (the positional rows fetched per chunk were 8 of the stage loop's 20 memory instructions)
    float sD[NCH][FR];
#pragma unroll
    for (int q = 0; q < NCH; ++q)
#pragma unroll
        for (int e = 0; e < FR; ++e) sD[q][e] = 0.0f;
    // Y chunk -> dZ chunk: dropout' (the same hash as the forward), dalpha's terms, then the ReLU mask
    auto store_stage = [&](int buf, int st, const u32x4* ry, const u32x4* rx, const unsigned* rb) {
        unsigned char* y = Ys + buf * RS * LD;
        unsigned char* x = Xs + buf * RS * LD;
        const int r0 = row0(st);
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int id = tid + 256 * q, row = id / CPR, c = id % CPR, gr = r0 + row;
            const int col = n0 + c * FR;
            const unsigned bits = rb[q] >> (col & 7);        // fp32: 4 columns per chunk = one nibble of the byte
            const T* src = (const T*)&ry[q];
            u32x4 out;
#pragma unroll
            for (int e0 = 0; e0 < FR; e0 += 4) {
                f32x4 d = f32x4{to_f32(src[e0]), to_f32(src[e0 + 1]), to_f32(src[e0 + 2]), to_f32(src[e0 + 3])};
                drop_apply4(p.drop, (uint32_t)gr * 256u + (uint32_t)(col + e0), d);
#pragma unroll
                for (int r = 0; r < 4; ++r) sD[q][e0 + r] += d[r];
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r] = ((bits >> (e0 + r)) & 1u) ? d[r] : 0.0f;
                frag_put4<T>(out, e0, d);              // (pairs leave through ONE packed conversion each)
            }
            *(u32x4*)(y + row * LD + c * 16) = out;
            *(u32x4*)(x + row * LD + c * 16) = rx[q];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
    float bsum = 0.0f;

#pragma unroll
    for (int s = 0; s < NS; ++s) load_stage(min(s, nst - 1), rY[s], rX[s], rB[s]);
    for (int st0 = 0; st0 < nst; st0 += NS) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int st = st0 + s;
            if (st < nst) {                                   // block-uniform
                const int buf = st & 1;
                store_stage(buf, st, rY[s], rX[s], rB[s]);
                __syncthreads();
                load_stage(min(st + NS, nst - 1), rY[s], rX[s], rB[s]);     // (unconditional refill: see wgrad_kernel)
                const unsigned char* y = Ys + buf * RS * LD;
                const unsigned char* x = Xs + buf * RS * LD;
#pragma unroll
                for (int kg = 0; kg < KGS; ++kg) {
                    u32x4 af[4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) af[mt] = frag_tr<T>(y, LD, kg * KG, wm * 64 + mt * 16, lane);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const u32x4 bf = frag_tr<T>(x, LD, kg * KG, wn * 64 + nt * 16, lane);
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = mma16<T>(af[mt], bf, acc[mt][nt]);
                    }
                }
                {   // dbp: column sums of the dZ tile
                    const int col = tid & 127, half = tid >> 7;
#pragma unroll 4
                    for (int r = half * (RS / 2); r < (half + 1) * (RS / 2); ++r) bsum += to_f32(*(const T*)(y + r * LD + col * sizeof(T)));
                }
            }
        }
    }
    __syncthreads();
    constexpr int LDT = 128 * 4 + 16;
    float* const Ts = (float*)smem;
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
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int k = half * 64 + lane;
            if (k < p.K) atomicAdd(p.dW + (size_t)n * p.ldw + k, Ts[row * (LDT / 4) + half * 64 + lane]);
        }
    }
    atomicAdd(p.db + n0 + (tid & 127), bsum);
    __syncthreads();
    {
        float dal = 0.0f;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int id = tid + 256 * q, row = id / CPR, c = id % CPR;
            const float* pe = p.pe_t + (size_t)((int)(((long long)split * RS + row) % p.T)) * 256 + n0 + c * FR;
#pragma unroll
            for (int e = 0; e < FR; ++e) dal += sD[q][e] * pe[e];
        }
        float* red = (float*)smem;
        const float s = block256_sum(dal, red);
        if (tid == 0) atomicAdd(p.dalpha, s);
    }
}

}  // namespace ge2e
