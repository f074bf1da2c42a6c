// Self-attention of the LAST encoder layer (SURVEY.md 8a rows a4 / a7 and their backward, a13).
//
// Only frame t = 0 of the encoder output is consumed (reference Modules.py:54), so the last layer's attention has ONE query per
// (utterance, head).  With a single query the key / value projections of the other T - 1 frames never have to exist:
//
//     s_t = q0_h . k_t / 8 = q0_h . (Wk_h x_t + bk_h) / 8 = (Wk_h^T q0_h) . x_t / 8 + const      (the constant cancels in the softmax)
//     o_h = sum_t pd_t v_t = sum_t pd_t (Wv_h x_t + bv_h) = Wv_h (sum_t pd_t x_t) + bv_h sum_t pd_t
//
// (x_t = the layer input row, pd = the dropped-out probabilities).  Per utterance that is two 64 x 256 matrix-vector products per
// head and two sweeps over the T x 256 input tile, instead of a [T, 512] K | V projection (40 GFLOP and 3U of HBM traffic per
// step at 64 x 15 x 160), a K / V streaming attention, and in backward the [T, 512] dK | dV tile, its K = 512 dgrad GEMM and its
// weight-gradient product.  Exact in exact arithmetic; in the 16-bit modes K and V are no longer rounded to storage precision (the
// products stay fp32), so the result is closer to the reference's fp32 CPU path, not farther.  Backward:
//
//     dctx_h = Wv_h^T do_h          dpd_t = dctx_h . x_t + do_h . bv_h          ds_t = (pd_t dpd_t - p_t sum_t' pd_t' dpd_t') / 8
//     dqk_h  = sum_t ds_t x_t       dx_t  = sum_h ds_t qk_h + pd_t dctx_h       dq0_h = Wk_h dqk_h
//     dWk_h  = sum_n q0_h (x) dqk_h      dWv_h = sum_n do_h (x) ctx_h      dbv_h = sum_n do_h sum_t pd_t      dbk = 0 exactly
//
// One 256-thread block per utterance; fp32 vector math in every arithmetic mode (0.4 MFLOP per utterance); thread = column for the
// W^T products, wave = head for the softmax and the W products, wave = row subset / lane = 4 columns for the sweeps over x.
#pragma once
#include "common.cuh"

namespace ge2e {

struct AttnLastArgs {
    const void* x;        // [R, 256] of T: the layer input, row n * T + t
    const void* q0;       // [N, 256] of T: q of frame 0 (compact)
    const void* Wk;       // [256][256] of T: in_proj_weight rows 256 .. 511 (k-contiguous)
    const void* Wv;       // rows 512 .. 767
    const float* bv;      // [256]: in_proj_bias + 512
    void* o0;             // fwd out [N, 256] of T (compact)
    float* qk;            // [N][4][256]: Wk_h^T q0_h          (train: saved for backward; eval: null)
    float* prob;          // [N][4][T]  : softmax probabilities (before dropout)
    float* ctx;           // [N][4][256]: sum_t pd_t x_t
    float* sp;            // [N][4]     : sum_t pd_t
    // backward
    const void* do0;      // [N, 256] of T
    const void* Wq;       // [256][256] of T: in_proj_weight rows 0 .. 255
    const void* dpre;     // [N, 256] of T: the residual path's gradient, added to the frame-0 rows
    void* dX;             // [R, 256] of T out: dL/d(layer input), every row
    void* dq0;            // [N, 256] of T out
    float* dqk;           // [N][4][256] out (the weight-gradient kernel reads it)
    int T, H;
    float scale;          // 1 / sqrt(64)
    Drop drop;
};

namespace attn_last {
constexpr int D = 256;
__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

// out[h][t] = (vec_h . x_t) * mul + add[h]  for the block's utterance: wave w takes rows t = w (mod 4), four rows in flight
template <typename T>
__device__ __forceinline__ void sweep_dot(const T* xb, int Tn, const float* vec /* [4][256] LDS */, float* out /* [4][Tn] LDS */,
                                          float mul, const float* add, int w, int lane) {
    f32x4 v4[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) v4[h] = *(const f32x4*)(vec + h * D + 4 * lane);
    for (int t0 = w; t0 < Tn; t0 += 16) {
        float s[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + 4 * u;
            const f32x4 xv = t < Tn ? load4(xb + (size_t)t * D + 4 * lane) : f32x4{0, 0, 0, 0};
#pragma unroll
            for (int h = 0; h < 4; ++h) s[u][h] = dot4(v4[h], xv);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int h = 0; h < 4; ++h) s[u][h] = wave_sum(s[u][h]);
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + 4 * u;
                if (t < Tn) {
#pragma unroll
                    for (int h = 0; h < 4; ++h) out[h * Tn + t] = s[u][h] * mul + (add ? add[h] : 0.0f);
                }
            }
        }
    }
}
// out[64 w + j] = W[64 w + j][:] . vec_w  (j = 0 .. 63) for wave w: lane j ends up holding result j
template <typename T>
__device__ __forceinline__ float rows_dot(const T* W, const float* vec_w /* [256] LDS */, int w, int lane) {
    const f32x4 c4 = *(const f32x4*)(vec_w + 4 * lane);
    float mine = 0.0f;
    for (int j = 0; j < 64; j += 4) {
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = dot4(load4(W + (size_t)(64 * w + j + u) * D + 4 * lane), c4);
#pragma unroll
        for (int u = 0; u < 4; ++u) { d[u] = wave_sum(d[u]); if (lane == j + u) mine = d[u]; }
    }
    return mine;
}
}  // namespace attn_last

// dynamic LDS: (6 * 256 + 16 * 256 + 4 * T) floats
inline size_t attn_last_fwd_smem(int T) { return (size_t)(6 * 256 + 16 * 256 + 4 * T) * 4; }
template <typename T>
__global__ void __launch_bounds__(256) attn_last_fwd_kernel(const AttnLastArgs p) {
    using namespace attn_last;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* const qs = sm;                      // [256] q0
    float* const qt = qs + 256;                // [4][256] qk_h, later ctx_h
    float* const sps = qt + 1024;              // [4] sum of the dropped probabilities (+ padding to 256)
    float* const red = sps + 256;              // [4 waves][4 heads][256]
    float* const sc = red + 4096;              // [4][T] scores, then dropped probabilities
    const int n = blockIdx.x, c = threadIdx.x, lane = c & 63, w = c >> 6;
    const int Tn = p.T;
    const T* const xb = (const T*)p.x + (size_t)n * Tn * D;
    qs[c] = to_f32(((const T*)p.q0)[(size_t)n * D + c]);
    __syncthreads();
    {   // qk_h[c] = sum_j q0[64 h + j] Wk[64 h + j][c]
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const T* wk = (const T*)p.Wk + c;
#pragma unroll 4
        for (int j = 0; j < 64; ++j)
#pragma unroll
            for (int h = 0; h < 4; ++h) a[h] += qs[64 * h + j] * to_f32(wk[(size_t)(64 * h + j) * D]);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            qt[h * D + c] = a[h];
            if (p.qk) p.qk[((size_t)n * 4 + h) * D + c] = a[h];
        }
    }
    __syncthreads();
    sweep_dot<T>(xb, Tn, qt, sc, p.scale, nullptr, w, lane);
    __syncthreads();
    if (w < p.H) {   // softmax over the frames of head w, dropout on the probabilities (counter: query 0 of head row n H + w)
        float* const s = sc + w * Tn;
        float mx = -INFINITY;
        for (int t = lane; t < Tn; t += 64) mx = fmaxf(mx, s[t]);
        mx = wave_max(mx);
        float sum = 0.0f;
        for (int t = lane; t < Tn; t += 64) sum += expf(s[t] - mx);
        const float inv = 1.0f / wave_sum(sum);
        const uint32_t ibase = ((uint32_t)(n * p.H + w) * (uint32_t)Tn) * (uint32_t)((Tn + 3) & ~3);
        float tot = 0.0f;
        for (int t = lane; t < Tn; t += 64) {
            const float pr = expf(s[t] - mx) * inv;
            const bool keep = p.drop.thr == 0 || drop_keep(ibase + (uint32_t)t, p.drop.key, p.drop.thr);
            const float pd = keep ? pr * p.drop.scale : 0.0f;
            if (p.prob) p.prob[((size_t)n * 4 + w) * Tn + t] = pr;
            s[t] = pd;
            tot += pd;
        }
        tot = wave_sum(tot);
        if (lane == 0) { sps[w] = tot; if (p.sp) p.sp[(size_t)n * 4 + w] = tot; }
    } else {
        float* const s = sc + w * Tn;
        for (int t = lane; t < Tn; t += 64) s[t] = 0.0f;
        if (lane == 0) sps[w] = 0.0f;
    }
    __syncthreads();
    {   // ctx_h = sum_t pd_t x_t: wave w takes rows t = w (mod 4), lane 4 columns; then the four waves' partial sums are added
        f32x4 acc[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) acc[h] = f32x4{0, 0, 0, 0};
        for (int t0 = w; t0 < Tn; t0 += 16) {
            f32x4 xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int t = t0 + 4 * u; xv[u] = t < Tn ? load4(xb + (size_t)t * D + 4 * lane) : f32x4{0, 0, 0, 0}; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + 4 * u;
                if (t < Tn) {
#pragma unroll
                    for (int h = 0; h < 4; ++h) acc[h] += sc[h * Tn + t] * xv[u];
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) *(f32x4*)(red + (w * 4 + h) * D + 4 * lane) = acc[h];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float cx = (red[(0 * 4 + h) * D + c] + red[(1 * 4 + h) * D + c]) + (red[(2 * 4 + h) * D + c] + red[(3 * 4 + h) * D + c]);
        qt[h * D + c] = cx;
        if (p.ctx) p.ctx[((size_t)n * 4 + h) * D + c] = cx;
    }
    __syncthreads();
    if (w < p.H) {   // o[64 w + j] = Wv[64 w + j] . ctx_w + bv[64 w + j] sum_t pd_t
        const float r = attn_last::rows_dot<T>((const T*)p.Wv, qt + w * D, w, lane);
        ((T*)p.o0)[(size_t)n * D + 64 * w + lane] = from_f32<T>(r + p.bv[64 * w + lane] * sps[w]);
    } else ((T*)p.o0)[(size_t)n * D + 64 * w + lane] = from_f32<T>(0.0f);
}

inline size_t attn_last_bwd_smem(int T) { return (size_t)(256 + 1024 + 1024 + 4096 + 256 + 256 + 256 + 8 * T) * 4; }
template <typename T>
__global__ void __launch_bounds__(256) attn_last_bwd_kernel(const AttnLastArgs p) {
    using namespace attn_last;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* const dos = sm;                     // [256] do
    float* const dct = dos + 256;              // [4][256] dctx_h = Wv_h^T do_h
    float* const qks = dct + 1024;             // [4][256] qk_h (saved by the forward), later dqk_h
    float* const red = qks + 1024;             // [4][4][256]
    float* const row0 = red + 4096;            // [256] dx of frame 0 before the residual / query terms
    float* const dq0s = row0 + 256;            // [256]
    float* const dob = dq0s + 256;             // [4] do_h . bv_h (+ padding)
    float* const dpt = dob + 256;              // [4][T] dpd_t, then pd_t
    float* const dst = dpt + 4 * p.T;          // [4][T] ds_t
    const int n = blockIdx.x, c = threadIdx.x, lane = c & 63, w = c >> 6;
    const int Tn = p.T;
    const T* const xb = (const T*)p.x + (size_t)n * Tn * D;
    dos[c] = to_f32(((const T*)p.do0)[(size_t)n * D + c]);
    __syncthreads();
    {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const T* wv = (const T*)p.Wv + c;
#pragma unroll 4
        for (int j = 0; j < 64; ++j)
#pragma unroll
            for (int h = 0; h < 4; ++h) a[h] += dos[64 * h + j] * to_f32(wv[(size_t)(64 * h + j) * D]);
#pragma unroll
        for (int h = 0; h < 4; ++h) { dct[h * D + c] = a[h]; qks[h * D + c] = p.qk[((size_t)n * 4 + h) * D + c]; }
        const float b = wave_sum(dos[c] * p.bv[c]);      // wave w covers columns 64 w .. 64 w + 63 = head w
        if (lane == 0) dob[w] = b;
    }
    __syncthreads();
    sweep_dot<T>(xb, Tn, dct, dpt, 1.0f, dob, w, lane);
    __syncthreads();
    if (w < p.H) {
        float* const dp = dpt + w * Tn;
        float* const ds = dst + w * Tn;
        const float* const pr = p.prob + ((size_t)n * 4 + w) * Tn;
        const uint32_t ibase = ((uint32_t)(n * p.H + w) * (uint32_t)Tn) * (uint32_t)((Tn + 3) & ~3);
        float delta = 0.0f;
        for (int t = lane; t < Tn; t += 64) {
            const bool keep = p.drop.thr == 0 || drop_keep(ibase + (uint32_t)t, p.drop.key, p.drop.thr);
            const float pd = keep ? pr[t] * p.drop.scale : 0.0f;
            const float g = pd * dp[t];
            delta += g;
            ds[t] = g;                      // pd_t dpd_t for now
            dp[t] = pd;
        }
        delta = wave_sum(delta);
        for (int t = lane; t < Tn; t += 64) ds[t] = (ds[t] - pr[t] * delta) * p.scale;
    } else {
        for (int t = lane; t < Tn; t += 64) { dpt[w * Tn + t] = 0.0f; dst[w * Tn + t] = 0.0f; }
    }
    __syncthreads();
    {   // one sweep: dqk_h += ds_t x_t (per-wave partial sums) and dx_t = sum_h ds_t qk_h + pd_t dctx_h (stored; frame 0 kept back)
        f32x4 acc[4], q4[4], d4[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) { acc[h] = f32x4{0, 0, 0, 0}; q4[h] = *(const f32x4*)(qks + h * D + 4 * lane); d4[h] = *(const f32x4*)(dct + h * D + 4 * lane); }
        T* const dxb = (T*)p.dX + (size_t)n * Tn * D;
        for (int t0 = w; t0 < Tn; t0 += 16) {
            f32x4 xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int t = t0 + 4 * u; xv[u] = t < Tn ? load4(xb + (size_t)t * D + 4 * lane) : f32x4{0, 0, 0, 0}; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + 4 * u;
                if (t < Tn) {
                    f32x4 o = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const float ds = dst[h * Tn + t], pd = dpt[h * Tn + t];
                        acc[h] += ds * xv[u];
                        o += ds * q4[h] + pd * d4[h];
                    }
                    if (t == 0) *(f32x4*)(row0 + 4 * lane) = o;
                    else store4(dxb + (size_t)t * D + 4 * lane, o[0], o[1], o[2], o[3]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) *(f32x4*)(red + (w * 4 + h) * D + 4 * lane) = acc[h];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float v = (red[(0 * 4 + h) * D + c] + red[(1 * 4 + h) * D + c]) + (red[(2 * 4 + h) * D + c] + red[(3 * 4 + h) * D + c]);
        qks[h * D + c] = v;
        p.dqk[((size_t)n * 4 + h) * D + c] = v;
    }
    __syncthreads();
    {   // dq0[64 w + j] = Wk[64 w + j] . dqk_w
        const float r = w < p.H ? attn_last::rows_dot<T>((const T*)p.Wk, qks + w * D, w, lane) : 0.0f;
        dq0s[64 * w + lane] = r;
        ((T*)p.dq0)[(size_t)n * D + 64 * w + lane] = from_f32<T>(r);
    }
    __syncthreads();
    {   // frame 0 also carries the residual path and the query: dx_0 += dpre + Wq^T dq0
        float a = row0[c] + to_f32(((const T*)p.dpre)[(size_t)n * D + c]);
        const T* wq = (const T*)p.Wq + c;
#pragma unroll 8
        for (int j = 0; j < 256; ++j) a += dq0s[j] * to_f32(wq[(size_t)j * D]);
        ((T*)p.dX)[(size_t)n * Tn * D + c] = from_f32<T>(a);
    }
}

// dWk[r][c] += sum_n q0[n][r] dqk[n][r / 64][c],  dWv[r][c] += sum_n do[n][r] ctx[n][r / 64][c],  dbv[r] += sum_n do[n][r] sp[n][r / 64]
// grid = (512, chunks): blockIdx.x < 256 the Wk rows, >= 256 the Wv rows; blockIdx.y a slice of the utterances
template <typename T>
__global__ void __launch_bounds__(256) attn_last_wgrad_kernel(const void* q0, const float* dqk, const void* do0, const float* ctx,
                                                              const float* sp, float* dWk, float* dWv, float* dbv, int N, int per) {
    __shared__ float red[4];
    const int r = blockIdx.x & 255, isv = blockIdx.x >> 8, h = r >> 6, c = threadIdx.x;
    const T* const a = (const T*)(isv ? do0 : q0) + r;
    const float* const B = (isv ? ctx : dqk) + (size_t)h * 256 + c;
    const int n0 = blockIdx.y * per, n1 = min(N, n0 + per);
    float acc = 0.0f;
#pragma unroll 4
    for (int n = n0; n < n1; ++n) acc += to_f32(a[(size_t)n * 256]) * B[(size_t)n * 1024];
    atomicAdd((isv ? dWv : dWk) + (size_t)r * 256 + c, acc);
    if (isv) {
        float b = 0.0f;
        for (int n = n0 + c; n < n1; n += 256) b += to_f32(a[(size_t)n * 256]) * sp[(size_t)n * 4 + h];
        b = block256_sum(b, red);
        if (c == 0) atomicAdd(dbv + r, b);
    }
}

}  // namespace ge2e
