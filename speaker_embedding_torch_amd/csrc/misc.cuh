// Bandwidth-bound pieces of the GE2E hot path: weight preparation, LayerNorm backward, the t=0 tail
// (final LN -> slice mean -> projection -> L2 norm) and the GE2E loss
// (rows a10-a12), each with its backward.  Rows are D = 256 wide: a wave holds a row as 4 floats/lane.
#pragma once
#include "common.cuh"

namespace ge2e {

// ---------------------------------------------------------------------------------------------
// weight preparation: fp32 master [rows][cols] -> T [rows][ldd] (zero padded) and T^T [cols][rows]
// ---------------------------------------------------------------------------------------------
struct PrepJob {
    const float* src; void* dst; void* dstT;     // dst / dstT may be null
    int rows, cols, ldd;        // ldd >= cols
    int tile0;                  // first 32x32 tile index of this job
    int tiles_x;                // tiles along the (padded) column dim
    int lds;                    // source row stride in floats (0: cols)
    float* dstT32;              // fp32 transposed copy [cols][rows], or null (the positional table pe [D][max_pos] -> pe_t [T][D]; Wq -> Wq^T)
};
constexpr int PREP_MAX_JOBS = 16;
struct PrepArgs { PrepJob job[PREP_MAX_JOBS]; int njobs; };

template <typename T>
__global__ void __launch_bounds__(256) prep_weights_kernel(const PrepArgs a) {
    __shared__ float tile[32][33];
    int j = 0;
#pragma unroll 1
    for (int q = 1; q < a.njobs; ++q) if ((int)blockIdx.x >= a.job[q].tile0) j = q;
    const PrepJob jb = a.job[j];
    const int tl = blockIdx.x - jb.tile0, ty = tl / jb.tiles_x, tx = tl % jb.tiles_x;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
#pragma unroll
    for (int r = ly; r < 32; r += 8) {
        const int row = ty * 32 + r, col = tx * 32 + lx;
        const float v = (row < jb.rows && col < jb.cols) ? jb.src[(size_t)row * (jb.lds ? jb.lds : jb.cols) + col] : 0.0f;
        tile[r][lx] = v;
        if (jb.dst && row < jb.rows && col < jb.ldd) ((T*)jb.dst)[(size_t)row * jb.ldd + col] = from_f32<T>(v);
    }
    __syncthreads();
    if (jb.dstT) {
#pragma unroll
        for (int r = ly; r < 32; r += 8) {
            const int col = tx * 32 + r, row = ty * 32 + lx;     // dstT[col][row]
            if (row < jb.rows && col < jb.cols) ((T*)jb.dstT)[(size_t)col * jb.rows + row] = from_f32<T>(tile[lx][r]);
        }
    }
    if (jb.dstT32) {
#pragma unroll
        for (int r = ly; r < 32; r += 8) {
            const int col = tx * 32 + r, row = ty * 32 + lx;
            if (row < jb.rows && col < jb.cols) jb.dstT32[(size_t)col * jb.rows + row] = tile[lx][r];
        }
    }
}

// mel batch fp32 [N][mel][T] (channels-first, contiguous along T) -> row-major T-typed [N*T][KP], zero beyond mel.
// One coalesced streaming pass; afterwards the prenet forward, its backward and its weight gradient all use the ordinary row loaders.
// A block turns 64 frames of one utterance: reads run along t (256 bytes per mel row), every output row leaves as KP * sizeof(T) contiguous
// bytes in 16-byte (fp32: 32-byte) pieces (round 4; the 32 x 32-tile form wrote 64-byte pieces: 32 -> 2x us at 960 x 160).
// grid = (ceil(T / 64), N); KP <= 128, KP % 8 == 0.  TI = float (the reference collater's dtype) or _Float16 (patterns are fp16 on disk: half
// the host-to-device bytes)
template <typename T, typename TI>
__global__ void __launch_bounds__(256) mel_pack_kernel(const TI* x, T* xt, int mel, int T_, int KP) {
    __shared__ float tile[128][65];
    const int t0 = blockIdx.x * 64, n = blockIdx.y;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const bool tok = t0 + lx < T_;
    TI v[32];                                          // all of a thread's loads in flight before the first LDS write
#pragma unroll
    for (int j = 0; j < 32; ++j) { const int k = ly + 4 * j; v[j] = (k < mel && tok) ? x[((size_t)n * mel + k) * T_ + t0 + lx] : (TI)0; }
#pragma unroll
    for (int j = 0; j < 32; ++j) { const int k = ly + 4 * j; if (k < KP) tile[k][lx] = (float)v[j]; }
    __syncthreads();
    const int kc = (threadIdx.x & 15) * 8, rr = threadIdx.x >> 4;
    if (kc < KP) {
#pragma unroll
        for (int r = rr; r < 64; r += 16) {
            const int t = t0 + r;
            if (t >= T_) break;
            T* dst = xt + ((size_t)n * T_ + t) * KP + kc;
            store4(dst, tile[kc][r], tile[kc + 1][r], tile[kc + 2][r], tile[kc + 3][r]);
            store4(dst + 4, tile[kc + 4][r], tile[kc + 5][r], tile[kc + 6][r], tile[kc + 7][r]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm backward over rows of 256 (post-LN residual blocks).  xhat is rebuilt from the saved
// OUTPUT y: xhat = (y - beta) / gamma (gamma == 0 columns carry no xhat information: treated as 0).
// ---------------------------------------------------------------------------------------------
struct LnBwdArgs {
    const void* dy; const void* y;
    const float* gamma; const float* beta; const float* rstd;
    void* dpre;          // gradient wrt the LN input (residual path)
    void* dmask;         // same, through the dropout in front of the sub-layer (null: not needed)
    float* dgamma; float* dbeta;
    int R;
    Drop drop;
    int drow_mul;        // dropout counter row = row * drow_mul (0 = 1)
};

template <typename T>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const LnBwdArgs p) {
    __shared__ float red[2][4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = lane * 4;
    const f32x4 ga = *(const f32x4*)(p.gamma + c0), be = *(const f32x4*)(p.beta + c0);
    f32x4 ig;
#pragma unroll
    for (int r = 0; r < 4; ++r) ig[r] = ga[r] != 0.0f ? 1.0f / ga[r] : 0.0f;
    f32x4 ag = f32x4{0, 0, 0, 0}, ab = f32x4{0, 0, 0, 0};
    for (int row = blockIdx.x * 4 + wave; row < p.R; row += gridDim.x * 4) {
        const f32x4 dy = load4((const T*)p.dy + (size_t)row * 256 + c0);
        const f32x4 y = load4((const T*)p.y + (size_t)row * 256 + c0);
        const float rs = p.rstd[row];
        f32x4 xh, dxh;
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xh[r] = (y[r] - be[r]) * ig[r];
            dxh[r] = dy[r] * ga[r];
            s1 += dxh[r];
            s2 += dxh[r] * xh[r];
            ag[r] += dy[r] * xh[r];
            ab[r] += dy[r];
        }
        s1 = wave_sum(s1) * (1.0f / 256.0f);
        s2 = wave_sum(s2) * (1.0f / 256.0f);
        f32x4 dx;
#pragma unroll
        for (int r = 0; r < 4; ++r) dx[r] = rs * (dxh[r] - s1 - xh[r] * s2);
        store4((T*)p.dpre + (size_t)row * 256 + c0, dx[0], dx[1], dx[2], dx[3]);
        if (p.dmask) {
            drop_apply4(p.drop, (uint32_t)row * (uint32_t)(p.drow_mul > 0 ? p.drow_mul : 1) * 256u + (uint32_t)c0, dx);
            store4((T*)p.dmask + (size_t)row * 256 + c0, dx[0], dx[1], dx[2], dx[3]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { red[0][wave][c0 + r] = ag[r]; red[1][wave][c0 + r] = ab[r]; }
    __syncthreads();
    const int c = threadIdx.x;
    atomicAdd(p.dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(p.dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
}

// gradient buffer <- 0 in ONE launch (hipMemsetAsync splits a buffer whose size is not a multiple of 16 bytes into two fill kernels, each a
// dependent launch at the head of the backward)
// Any 4-byte-aligned buffer: up to three scalar stores bring the pointer to a 16-byte boundary, 16-byte stores do the body, scalar stores the rest.
__global__ void __launch_bounds__(256) zero_f32_kernel(float* p, size_t n) {
    const size_t head = min(n, (size_t)((16 - ((uintptr_t)p & 15)) & 15) >> 2);
    float* const b = p + head;
    const size_t nb = n - head, n4 = nb >> 2;
    for (size_t q = blockIdx.x * (size_t)256 + threadIdx.x; q < n4; q += (size_t)gridDim.x * 256) ((f32x4*)b)[q] = f32x4{0, 0, 0, 0};
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) p[threadIdx.x] = 0.0f;
        if (threadIdx.x < (nb & 3)) b[(n4 << 2) + threadIdx.x] = 0.0f;
    }
}

// dgamma / dbeta of a LayerNorm alone (the row part of its backward rides in the chained FFN backward kernel, ffn.cuh): column sums over all
// rows of dy xhat and dy.  Half a wave per row (16 bytes per lane), 4 row pairs in flight per wave, per-lane accumulators for 8 columns;
// a few hundred blocks, so that the final atomics (every block adds into the same 512 floats) stay a few hundred thousand.  Runs on the
// weight-gradient stream.
template <typename T> __device__ __forceinline__ void colsum_load8(const T* p, float* v) {
    if constexpr (sizeof(T) == 2) {
        const u32x4 w = *(const u32x4*)p;
        const T* e = (const T*)&w;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = to_f32(e[q]);
    } else {
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] = a[q]; v[4 + q] = b[q]; }
    }
}
template <typename T>
__global__ void __launch_bounds__(256) ln_colsum_kernel(const LnBwdArgs p) {
    __shared__ float red[2][8][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, c0 = (lane & 31) * 8;
    float ga[8], be[8], ag[8], ab[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float gq = p.gamma[c0 + q];
        ga[q] = gq != 0.0f ? 1.0f / gq : 0.0f; be[q] = p.beta[c0 + q]; ag[q] = 0.0f; ab[q] = 0.0f;
    }
    constexpr int U = 4;
    const int slot = (blockIdx.x * 4 + wave) * 2 + half, nslot = gridDim.x * 8;
    for (int base = slot; base < p.R; base += U * nslot) {
        float dy[U][8], y[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = base + u * nslot;
            const int rr = row < p.R ? row : base;
            colsum_load8<T>((const T*)p.dy + (size_t)rr * 256 + c0, dy[u]);
            colsum_load8<T>((const T*)p.y + (size_t)rr * 256 + c0, y[u]);
            if (row >= p.R) {
#pragma unroll
                for (int q = 0; q < 8; ++q) dy[u][q] = 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < 8; ++q) { ag[q] += dy[u][q] * ((y[u][q] - be[q]) * ga[q]); ab[q] += dy[u][q]; }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { red[0][wave * 2 + half][c0 + q] = ag[q]; red[1][wave * 2 + half][c0 + q] = ab[q]; }
    __syncthreads();
    const int c = threadIdx.x;
    float sg = 0.0f, sb = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { sg += red[0][w][c]; sb += red[1][w][c]; }
    atomicAdd(p.dgamma + c, sg);
    atomicAdd(p.dbeta + c, sb);
}

// ---------------------------------------------------------------------------------------------
// tail: z = LN_f(h[n, t=0, :]); z' = mean over `samples`; e = Wq z' + bq; e /= max(|e|, 1e-12)
// fp32 arithmetic in both modes (0.13 MFLOP/utt).  One block of 256 threads per output embedding.
// ---------------------------------------------------------------------------------------------
struct TailArgs {
    const void* h;               // [N*T, 256] of T (last layer output)
    int T, samples, N;           // N = utterances (rows used: n*T)
    const float* gf; const float* bf;      // transformer.norm
    const float* wq; const float* wqT; const float* bq;   // projection [256][256], its transpose, bias
    float eps;
    float* xhat; float* rstd;    // [N,256], [N]   saved for backward
    float* zm; float* nrm;       // [N/samples,256], [N/samples]
    float* emb;                  // [N/samples,256] output (also kept for backward)
    float* emb_out;              // forward: the caller's copy of emb (written by the same kernel), or null
    // backward
    const float* d_emb; float* d_raw; void* dH;      // dH: [N*T,256] of T, only rows t=0 written
    float* dgf; float* dbf; float* dwq; float* dbq;
};

// TAIL_RB embeddings per block: the projection matrix is streamed from L2 once per block (32 loads in flight per
// thread) and the LN_f gradient atomics shrink by the same factor.  grid = ceil(M / TAIL_RB), M = N / samples
constexpr int TAIL_RB = 4;
template <typename T>
__global__ void __launch_bounds__(256) tail_fwd_kernel(const TailArgs p) {
    __shared__ float red[4];
    __shared__ float zs[TAIL_RB][256];
    const int c = threadIdx.x, M = p.N / p.samples;
    const int mb = blockIdx.x * TAIL_RB;
#pragma unroll
    for (int u = 0; u < TAIL_RB; ++u) {
        const int m = mb + u;
        float z = 0.0f;
        if (m < M) {                                   // block-uniform
            float zsum = 0.0f;
            for (int s = 0; s < p.samples; ++s) {
                const int row = m * p.samples + s;
                const float v = to_f32(((const T*)p.h)[(size_t)row * p.T * 256 + c]);
                const float mean = block256_sum(v, red) * (1.0f / 256.0f);
                const float d = v - mean;
                const float rs = 1.0f / sqrtf(block256_sum(d * d, red) * (1.0f / 256.0f) + p.eps);
                const float xh = d * rs;
                if (p.xhat) { p.xhat[(size_t)row * 256 + c] = xh; if (c == 0) p.rstd[row] = rs; }
                zsum += xh * p.gf[c] + p.bf[c];
            }
            z = zsum / (float)p.samples;
            if (p.zm) p.zm[(size_t)m * 256 + c] = z;
        }
        zs[u][c] = z;
    }
    __syncthreads();
    float e[TAIL_RB];
#pragma unroll
    for (int u = 0; u < TAIL_RB; ++u) e[u] = p.bq[c];
#pragma unroll 16
    for (int k = 0; k < 256; ++k) {
        const float w = p.wqT[k * 256 + c];
#pragma unroll
        for (int u = 0; u < TAIL_RB; ++u) e[u] += zs[u][k] * w;
    }
#pragma unroll
    for (int u = 0; u < TAIL_RB; ++u) {
        const int m = mb + u;
        if (m >= M) break;                             // block-uniform
        const float nn = fmaxf(sqrtf(block256_sum(e[u] * e[u], red)), 1e-12f);
        p.emb[(size_t)m * 256 + c] = e[u] / nn;
        if (p.emb_out) p.emb_out[(size_t)m * 256 + c] = e[u] / nn;
        if (c == 0 && p.nrm) p.nrm[m] = nn;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) tail_bwd_kernel(const TailArgs p) {
    __shared__ float red[4];
    __shared__ float dr[TAIL_RB][256];
    const int c = threadIdx.x, M = p.N / p.samples;
    const int mb = blockIdx.x * TAIL_RB;
#pragma unroll
    for (int u = 0; u < TAIL_RB; ++u) {
        const int m = mb + u;
        float draw = 0.0f;
        if (m < M) {                                   // block-uniform
            const float de = p.d_emb[(size_t)m * 256 + c], e = p.emb[(size_t)m * 256 + c];
            const float dot = block256_sum(de * e, red);
            draw = (de - e * dot) / p.nrm[m];
            p.d_raw[(size_t)m * 256 + c] = draw;
        }
        dr[u][c] = draw;
    }
    __syncthreads();
    float dzm[TAIL_RB];
#pragma unroll
    for (int u = 0; u < TAIL_RB; ++u) dzm[u] = 0.0f;
#pragma unroll 32
    for (int j = 0; j < 256; ++j) {
        const float w = p.wq[j * 256 + c];
#pragma unroll
        for (int u = 0; u < TAIL_RB; ++u) dzm[u] += dr[u][j] * w;
    }
    const float g = p.gf[c];
    float ag = 0.0f, ab = 0.0f;
#pragma unroll
    for (int u = 0; u < TAIL_RB; ++u) {
        const int m = mb + u;
        if (m >= M) break;                             // block-uniform
        const float dz = dzm[u] / (float)p.samples;
        for (int s = 0; s < p.samples; ++s) {
            const int row = m * p.samples + s;
            const float xh = p.xhat[(size_t)row * 256 + c];
            const float dxh = dz * g;
            const float m1 = block256_sum(dxh, red) * (1.0f / 256.0f);
            const float m2 = block256_sum(dxh * xh, red) * (1.0f / 256.0f);
            ((T*)p.dH)[(size_t)row * p.T * 256 + c] = from_f32<T>(p.rstd[row] * (dxh - m1 - xh * m2));
            ag += dz * xh;
        }
        ab += dz * (float)p.samples;
    }
    atomicAdd(p.dgf + c, ag);
    atomicAdd(p.dbf + c, ab);
}

// dWq[c][k] += sum_m d_raw[m][c] * zm[m][k]; dbq[c] += sum_m d_raw[m][c].  grid = (256 rows c, m-slices);
// the gradient buffer is zeroed at the start of backward, slices combine with one atomic per element.
__global__ void __launch_bounds__(256) tail_wgrad_kernel(const TailArgs p) {
    const int c = blockIdx.x, k = threadIdx.x, M = p.N / p.samples;
    const int per = (M + gridDim.y - 1) / gridDim.y, m0 = blockIdx.y * per, m1 = min(M, m0 + per);
    float acc = 0.0f, bsum = 0.0f;
#pragma unroll 4
    for (int m = m0; m < m1; ++m) {
        const float d = p.d_raw[(size_t)m * 256 + c];
        acc += d * p.zm[(size_t)m * 256 + k];
        bsum += d;
    }
    atomicAdd(p.dwq + c * 256 + k, acc);
    if (k == 0) atomicAdd(p.dbq + c, bsum);
}

// ---------------------------------------------------------------------------------------------
// GE2E loss (reference Modules.py:121-156).  fp32, D = 256.
//   workspace (floats): cent[S*256] cn[S] en[N] rowloss[N] G[N*S] cosm[N*S] dC[S*256]
// ---------------------------------------------------------------------------------------------
struct LossArgs {
    const float* emb; int N, S, P;
    float w, b;
    float* cent; float* cn; float* en; float* rowloss; float* G; float* cosm;
    float* dC; int Y;            // [Y][S][256] partial centroid gradients, one slab per slice of the utterances (plain stores: no zeroing, no atomics)
    float* loss;                 // [1]
    const float* gscale;         // device scalar dL/dloss (backward)
    float* d_emb;                // [N,256]
    float* dwb;                  // [2] dL/dw, dL/db of the criterion's own parameters (reference Modules.py:115-116), or null
    float* rowwb;                // [2][N] per-row terms of the two (summed by loss_wb_reduce_kernel)
};

__global__ void __launch_bounds__(256) loss_centroid_kernel(const LossArgs p) {
    __shared__ float red[4];
    const int s = blockIdx.x, c = threadIdx.x;
    float acc = 0.0f;
    for (int q = 0; q < p.P; ++q) acc += p.emb[((size_t)s * p.P + q) * 256 + c];
    acc /= (float)p.P;
    p.cent[(size_t)s * 256 + c] = acc;
    const float nn = fmaxf(sqrtf(block256_sum(acc * acc, red)), 1e-8f);
    if (c == 0) p.cn[s] = nn;
}

// one block per utterance: cosines against every centroid, softmax cross-entropy, G0 = (softmax - onehot) / N (= dL/dsim).  (The mean over
// the rows stays a launch of its own: 960 float atomics on ONE word serialise at ~12 ns each -- measured +11 us on this 13 us kernel.)
__global__ void __launch_bounds__(256) loss_row_kernel(const LossArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sims = (float*)smem;             // [S]
    float* coss = sims + p.S;               // [S]
    __shared__ float bc[2];
    const int irow = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 e = *(const f32x4*)(p.emb + (size_t)irow * 256 + lane * 4);
    const float en = fmaxf(sqrtf(wave_sum(e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3])), 1e-8f);
    for (int s = wave; s < p.S; s += 4) {
        const f32x4 cv = *(const f32x4*)(p.cent + (size_t)s * 256 + lane * 4);
        const float dot = wave_sum(e[0] * cv[0] + e[1] * cv[1] + e[2] * cv[2] + e[3] * cv[3]);
        const float cs = dot / (en * p.cn[s]);
        if (lane == 0) { coss[s] = cs; sims[s] = p.w * cs - p.b; }
    }
    __syncthreads();
    const int own = irow / p.P;
    if (wave == 0) {
        float mx = -INFINITY;
        for (int s = lane; s < p.S; s += 64) mx = fmaxf(mx, sims[s]);
        mx = wave_max(mx);
        float sum = 0.0f;
        for (int s = lane; s < p.S; s += 64) sum += expf(sims[s] - mx);
        sum = wave_sum(sum);
        const float lse = mx + logf(sum);
        if (lane == 0) {
            bc[0] = lse; p.en[irow] = en;
            p.rowloss[irow] = lse - sims[own];
        }
    }
    __syncthreads();
    const float lse = bc[0];
    const float k = 1.0f / (float)p.N;
    for (int s = threadIdx.x; s < p.S; s += 256) {
        p.G[(size_t)irow * p.S + s] = (expf(sims[s] - lse) - (s == own ? 1.0f : 0.0f)) * k;
        p.cosm[(size_t)irow * p.S + s] = coss[s];
    }
}

__global__ void __launch_bounds__(256) loss_reduce_kernel(const LossArgs p) {
    __shared__ float red[4];
    float acc = 0.0f;
    for (int q = threadIdx.x; q < p.N; q += 256) acc += p.rowloss[q];
    const float t = block256_sum(acc, red);
    if (threadIdx.x == 0) p.loss[0] = t / (float)p.N;
}

// dC_y[s][c] = w ((sum_i G0[i][s] ehat_i[c]) / cn_s - (sum_i G0[i][s] cos[i][s]) c_s[c] / cn_s^2) over slice y of the utterances: the expression
// is linear in the two sums, so the row kernel adds the Y slabs.  grid = (S, Y)
__global__ void __launch_bounds__(256) loss_bwd_centroid_kernel(const LossArgs p) {
    __shared__ float gq[256], gcq[256];     // this slice's G0[q][s] / en_q and G0[q][s] cos[q][s] (the row loop then carries ONE load per iteration)
    const int s = blockIdx.x, c = threadIdx.x;
    const int per = (p.N + gridDim.y - 1) / gridDim.y, q0 = blockIdx.y * per, q1 = min(p.N, q0 + per);
    float a = 0.0f, gc = 0.0f;
    for (int qb = q0; qb < q1; qb += 256) {
        const int nq = min(256, q1 - qb);
        __syncthreads();
        if (c < nq) {
            const float g = p.G[(size_t)(qb + c) * p.S + s];
            gq[c] = g / p.en[qb + c];
            gcq[c] = g * p.cosm[(size_t)(qb + c) * p.S + s];
        }
        __syncthreads();
#pragma unroll 8
        for (int q = 0; q < nq; ++q) {
            a += gq[q] * p.emb[(size_t)(qb + q) * 256 + c];
            gc += gcq[q];
        }
    }
    const float cn = p.cn[s];
    p.dC[((size_t)blockIdx.y * p.S + s) * 256 + c] = p.w * (a / cn - gc * p.cent[(size_t)s * 256 + c] / (cn * cn));
}

// d_emb_i = gscale * ( w (G0_i chat)/en_i - w (G0_i . cos_i) e_i / en_i^2 + dC[spk(i)] / P );   dL/dw += gscale G0_i . cos_i,  dL/db -= gscale sum_s G0[i][s]
__global__ void __launch_bounds__(256) loss_bwd_row_kernel(const LossArgs p) {
    __shared__ float gn[256], gcs[256], gss[256];   // per speaker: G0[i][s] / cn_s, G0[i][s] cos[i][s], G0[i][s]
    const int irow = blockIdx.x, c = threadIdx.x;
    float a = 0.0f, gc = 0.0f, gs = 0.0f;
    for (int sb = 0; sb < p.S; sb += 256) {
        const int ns = min(256, p.S - sb);
        __syncthreads();
        if (c < ns) {
            const float g = p.G[(size_t)irow * p.S + sb + c];
            gn[c] = g / p.cn[sb + c]; gcs[c] = g * p.cosm[(size_t)irow * p.S + sb + c]; gss[c] = g;
        }
        __syncthreads();
#pragma unroll 8
        for (int s = 0; s < ns; ++s) {
            a += gn[s] * p.cent[(size_t)(sb + s) * 256 + c];
            gc += gcs[s];
            gs += gss[s];
        }
    }
    float dc = 0.0f;
    for (int y = 0; y < p.Y; ++y) dc += p.dC[((size_t)y * p.S + irow / p.P) * 256 + c];
    const float en = p.en[irow], gsc = p.gscale[0];
    const float v = p.w * (a / en - gc * p.emb[(size_t)irow * 256 + c] / (en * en)) + dc / (float)p.P;
    p.d_emb[(size_t)irow * 256 + c] = v * gsc;
    if (p.dwb && c == 0) { p.rowwb[irow] = gc * gsc; p.rowwb[p.N + irow] = -gs * gsc; }
}

// dL/dw = sum_i rowwb[0][i], dL/db = sum_i rowwb[1][i] (launched only when the caller wants them)
__global__ void __launch_bounds__(256) loss_wb_reduce_kernel(const LossArgs p) {
    __shared__ float red[4];
    float a = 0.0f, b = 0.0f;
    for (int q = threadIdx.x; q < p.N; q += 256) { a += p.rowwb[q]; b += p.rowwb[p.N + q]; }
    a = block256_sum(a, red);
    b = block256_sum(b, red);
    if (threadIdx.x == 0) { p.dwb[0] = a; p.dwb[1] = b; }
}

// ---------------------------------------------------------------------------------------------
// clip_grad_norm_(max_norm) + AdamW of reference Train.py:154-162 in two launches over all parameters.
//   norm kernel : sumsq += sum g^2                               (one atomic per block)
//   step kernel : coef = min(1, max_norm / (sqrt(sumsq) + 1e-6));  g *= coef (kept, like clip_grad_norm_);
//                 p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//                 p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)          (torch.optim.AdamW)
// Parameters, gradients and moments are separate tensors: a chunk table maps blocks to (tensor, offset).
// ---------------------------------------------------------------------------------------------
constexpr int OPT_MAX_TENSORS = 64;      // per launch (kernel arguments stay under 4 KB); more tensors = more launches
struct OptArgs {
    float* p[OPT_MAX_TENSORS]; float* g[OPT_MAX_TENSORS]; float* m[OPT_MAX_TENSORS]; float* v[OPT_MAX_TENSORS];
    int numel[OPT_MAX_TENSORS];
    int chunk0[OPT_MAX_TENSORS + 1];     // first 4096-element chunk of each tensor
    int ntensors;
    float* sumsq;                        // device scalar, zeroed before the norm kernel
    float max_norm;                      // <= 0: no clipping
    float lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt;
    // mixed-precision loss scaling (null: off).  scaler = { scale, growth_tracker, found_inf, steps_taken } on the device:
    // gradients hold scale x the true gradient; a non-finite squared norm skips the whole update (GradScaler.step) and
    // the bias corrections come from the device-side step count, so a skipped step needs no host round trip
    float* scaler;
    float growth, backoff; int growth_interval;
};
constexpr int OPT_CHUNK = 4096;

__device__ __forceinline__ int opt_find(const OptArgs& a, int chunk) {
    int lo = 0, hi = a.ntensors - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (a.chunk0[mid] <= chunk) lo = mid; else hi = mid - 1; }
    return lo;
}

// 16 bytes per lane and access whatever the tensors' alignment (the gradients are slices of one flat buffer at odd element offsets; global memory
// takes dword-aligned vector accesses): a quarter of the memory instructions of the element-wise form (round 4: 16.8 + 26.6 -> see profiles/r04_ab_log.txt)
typedef f32x4 f32x4u __attribute__((aligned(4)));
__global__ void __launch_bounds__(256) opt_norm_kernel(const OptArgs a) {
    __shared__ float red[4];
    const int t = opt_find(a, blockIdx.x);
    const int base = (blockIdx.x - a.chunk0[t]) * OPT_CHUNK, n = a.numel[t];
    const float* g = a.g[t];
    float acc = 0.0f;
#pragma unroll
    for (int q = threadIdx.x * 4; q < OPT_CHUNK; q += 1024) {
        const int e = base + q;
        if (e + 3 < n) { const f32x4 x = *(const f32x4u*)(g + e); acc += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]); }
        else for (int k = e; k < n && k < e + 4; ++k) acc += g[k] * g[k];
    }
    const float s = block256_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(a.sumsq, s);
}

__global__ void __launch_bounds__(256) opt_adamw_kernel(const OptArgs a) {
    const int t = opt_find(a, blockIdx.x);
    const int base = (blockIdx.x - a.chunk0[t]) * OPT_CHUNK;
    float coef = 1.0f, bc1 = a.bc1, bc2_sqrt = a.bc2_sqrt;
    if (a.scaler) {
        const float ss = *a.sumsq;
        if (!(fabsf(ss) <= 3.0e38f)) return;                 // inf / nan somewhere: GradScaler.step skips the optimizer step
        const float inv = 1.0f / a.scaler[0], tt = a.scaler[3] + 1.0f;
        bc1 = 1.0f - powf(a.beta1, tt);
        bc2_sqrt = sqrtf(1.0f - powf(a.beta2, tt));
        coef = inv;                                          // GradScaler.unscale_
        if (a.max_norm > 0.0f) coef *= fminf(1.0f, a.max_norm / (sqrtf(ss) * inv + 1e-6f));
    } else if (a.max_norm > 0.0f) coef = fminf(1.0f, a.max_norm / (sqrtf(*a.sumsq) + 1e-6f));
    float* __restrict__ p = a.p[t]; float* __restrict__ g = a.g[t]; float* __restrict__ m = a.m[t]; float* __restrict__ v = a.v[t];
    const float decay = 1.0f - a.lr * a.weight_decay, step_size = a.lr / bc1;
    const int n = a.numel[t];
    auto one = [&](float& pe, float& ge, float& me, float& ve) {
        const float gr = ge * coef;
        const float mm = a.beta1 * me + (1.0f - a.beta1) * gr;
        const float vv = a.beta2 * ve + (1.0f - a.beta2) * gr * gr;
        ge = gr; me = mm; ve = vv;
        pe = pe * decay - step_size * (mm / (sqrtf(vv) / bc2_sqrt + a.eps));
    };
#pragma unroll
    for (int q = threadIdx.x * 4; q < OPT_CHUNK; q += 1024) {
        const int e = base + q;
        if (e + 3 < n) {
            f32x4 pp = *(const f32x4u*)(p + e), gg = *(const f32x4u*)(g + e), mm = *(const f32x4u*)(m + e), vv = *(const f32x4u*)(v + e);
#pragma unroll
            for (int r = 0; r < 4; ++r) { float pe = pp[r], ge = gg[r], me = mm[r], ve = vv[r]; one(pe, ge, me, ve); pp[r] = pe; gg[r] = ge; mm[r] = me; vv[r] = ve; }
            *(f32x4u*)(g + e) = gg; *(f32x4u*)(m + e) = mm; *(f32x4u*)(v + e) = vv; *(f32x4u*)(p + e) = pp;
        } else
            for (int k = e; k < n && k < e + 4; ++k) one(p[k], g[k], m[k], v[k]);
    }
}

// GradScaler.update after the step kernel has read the state: one thread
__global__ void opt_scaler_update_kernel(const OptArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float ss = *a.sumsq;
    const bool finite = fabsf(ss) <= 3.0e38f;
    float scale = a.scaler[0], tracker = a.scaler[1];
    if (finite) {
        a.scaler[3] += 1.0f;
        tracker += 1.0f;
        if (tracker >= (float)a.growth_interval) { scale *= a.growth; tracker = 0.0f; }
    } else {
        scale *= a.backoff; tracker = 0.0f;
    }
    a.scaler[0] = scale; a.scaler[1] = tracker; a.scaler[2] = finite ? 0.0f : 1.0f;
}

}  // namespace ge2e
