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
    int abl;              // development only (tools/attn_last_bench.hip): bit k skips phase k of the kernel; 0 in the library
};

namespace attn_last {
constexpr int D = 256;
// Cross-lane reductions are what a vector formulation of these products costs (a wave_sum is six LDS-crossbar permutes: 160 of them
// per wave and sweep made the score sweep 52 of the forward's 95 us), so the two product shapes that would need them run on the
// matrix pipe instead, with 12 of the 16 operand rows / columns left zero -- the FLOPs are negligible either way:
//   sweep_mfma : out[h][t] = vec_h . x_t          (4 x 256 by 256 x T: the four heads are rows 0-3 of the "weight" operand)
//   rows_mfma  : out[r]    = W[r] . vec  (64 rows) (the vector is row 0 of the "activation" operand)
// Operand fragments are the library's usual ones (common.cuh: 16 bytes per lane, lane (i, g) = row i, k-slots of g), so x rows and
// W rows are plain 16-byte global loads.  fp32 mode uses the exact fp32 MFMA.
template <typename T> __device__ __forceinline__ u32x4 frag_from_f32(const float* v) {      // v: this lane's FRAG consecutive k values
    if constexpr (sizeof(T) == 4) return pack_acc<T>(*(const f32x4*)v, f32x4{0, 0, 0, 0});
    else return pack_acc<T>(*(const f32x4*)v, *(const f32x4*)(v + 4));
}
// out[h][t] = (vec_h . x_t) * mul + add[h], h = 0..3, for the rows of the block's utterance; wave w takes the 16-row tiles w, w + 4, ...
template <typename T>
__device__ __forceinline__ void sweep_mfma(const T* xb, int Tn, const float* vec /* [4][256] LDS */, float* out /* [4][Tn] LDS */,
                                           float mul, const float* add, int w, int lane) {
    constexpr int KG = Prec<T>::KG, FR = Prec<T>::FRAG, NKG = D / KG;
    const int i = lane & 15, g = lane >> 4;
    u32x4 wf[NKG];
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg)
        wf[kg] = i < 4 ? frag_from_f32<T>(vec + i * D + KG * kg + FR * g) : u32x4{0, 0, 0, 0};
    for (int j = w; 16 * j < Tn; j += 4) {
        const int t = 16 * j + i;
        const unsigned char* xrow = (const unsigned char*)(xb + (size_t)(t < Tn ? t : Tn - 1) * D) + FR * g * sizeof(T);
        u32x4 xf[NKG];
#pragma unroll
        for (int kg = 0; kg < NKG; ++kg) xf[kg] = *(const u32x4*)(xrow + kg * KG * sizeof(T));
        f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int kg = 0; kg < NKG; ++kg) acc = mma16<T>(wf[kg], xf[kg], acc);       // acc[r] = vec_{4g + r} . x_t
        if (g == 0 && t < Tn) {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[r * Tn + t] = acc[r] * mul + (add ? add[r] : 0.0f);
        }
    }
}
// res[64 w + j] = W[64 w + j][:] . vec  (j = 0 .. 63) for wave w; vec: [256] fp32 in LDS, res: LDS
template <typename T>
__device__ __forceinline__ void rows_mfma(const T* W, const float* vec, float* res, int w, int lane) {
    constexpr int KG = Prec<T>::KG, FR = Prec<T>::FRAG, NKG = D / KG;
    const int i = lane & 15, g = lane >> 4;
    u32x4 vf[NKG];
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg) vf[kg] = i == 0 ? frag_from_f32<T>(vec + KG * kg + FR * g) : u32x4{0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const unsigned char* wrow = (const unsigned char*)(W + (size_t)(64 * w + 16 * m + i) * D) + FR * g * sizeof(T);
        u32x4 wf[NKG];
#pragma unroll
        for (int kg = 0; kg < NKG; ++kg) wf[kg] = *(const u32x4*)(wrow + kg * KG * sizeof(T));
        f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int kg = 0; kg < NKG; ++kg) acc = mma16<T>(wf[kg], vf[kg], acc);       // acc[r] = W[64 w + 16 m + 4 g + r] . X[i]; X[0] = vec
        if (i == 0) *(f32x4*)(res + 64 * w + 16 * m + 4 * g) = acc;
    }
}
// out_w[c] = sum_j vec[64 w + j] W[64 w + j][c] for wave w (= head w): lane = 4 columns, 8 rows in flight (one 8-byte load per lane and row)
template <typename T>
__device__ __forceinline__ f32x4 cols_dot(const T* W, const float* vec /* [256] LDS */, int w, int lane) {
    f32x4 acc = f32x4{0, 0, 0, 0};
    for (int j = 0; j < 64; j += 8) {
        f32x4 r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = load4(W + (size_t)(64 * w + j + u) * D + 4 * lane);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += vec[64 * w + j + u] * r[u];
    }
    return acc;
}
// the wave's rows t = w, w + 4, ... of the input tile, 8 at a time: lane = 4 columns; rows past the end are zero
template <typename T>
__device__ __forceinline__ void rows8(f32x4* xv, const T* xb, int Tn, int lane, int t0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int t = t0 + 4 * u; xv[u] = t < Tn ? load4(xb + (size_t)t * D + 4 * lane) : f32x4{0, 0, 0, 0}; }
}
}  // namespace attn_last

// dynamic LDS: (6 * 256 + 16 * 256 + 4 * T) floats
inline size_t attn_last_fwd_smem(int T) { return (size_t)(6 * 256 + 16 * 256 + 4 * T) * 4; }
template <typename T>
__global__ void __launch_bounds__(256, 4) attn_last_fwd_kernel(const AttnLastArgs p) {
    using namespace attn_last;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* const qs = sm;                      // [256] q0, later the output row
    float* const qt = qs + 256;                // [4][256] qk_h, later ctx_h
    float* const sps = qt + 1024;              // [4] sum of the dropped probabilities (+ padding to 256)
    float* const red = sps + 256;              // [4 waves][4 heads][256]
    float* const sc = red + 4096;              // [4][T] scores, then dropped probabilities
    const int n = blockIdx.x, c = threadIdx.x, lane = c & 63, w = __builtin_amdgcn_readfirstlane(c >> 6);     // (wave-uniform: row addresses become scalar base + lane offset)
    const int Tn = p.T;
    const T* const xb = (const T*)p.x + (size_t)n * Tn * D;
    qs[c] = to_f32(((const T*)p.q0)[(size_t)n * D + c]);
    __syncthreads();
    {   // qk_h[c] = sum_j q0[64 h + j] Wk[64 h + j][c]: wave h = head h
        const f32x4 a = (p.abl & 1) ? f32x4{0, 0, 0, 0} : attn_last::cols_dot<T>((const T*)p.Wk, qs, w, lane);
        *(f32x4*)(qt + w * D + 4 * lane) = a;
        if (p.qk) *(f32x4*)(p.qk + ((size_t)n * 4 + w) * D + 4 * lane) = a;
    }
    __syncthreads();
    if (!(p.abl & 2)) sweep_mfma<T>(xb, Tn, qt, sc, p.scale, nullptr, w, lane);
    __syncthreads();
    if (w < p.H && !(p.abl & 4)) {   // softmax over the frames of head w, dropout on the probabilities (counter: query 0 of head row n H + w)
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
        for (int t0 = w; t0 < Tn && !(p.abl & 8); t0 += 32) {
            f32x4 xv[8];
            rows8<T>(xv, xb, Tn, lane, t0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
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
    // o[64 w + j] = Wv[64 w + j] . ctx_w + bv[64 w + j] sum_t pd_t
    if (w < p.H && !(p.abl & 16)) attn_last::rows_mfma<T>((const T*)p.Wv, qt + w * D, qs, w, lane);
    else qs[c] = 0.0f;
    __syncthreads();
    ((T*)p.o0)[(size_t)n * D + c] = from_f32<T>(qs[c] + p.bv[c] * sps[w]);
}

inline size_t attn_last_bwd_smem(int T) { return (size_t)(256 + 1024 + 1024 + 4096 + 256 + 256 + 256 + 8 * T) * 4; }
template <typename T>
__global__ void __launch_bounds__(256, 3) attn_last_bwd_kernel(const AttnLastArgs p) {
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
    const int n = blockIdx.x, c = threadIdx.x, lane = c & 63, w = __builtin_amdgcn_readfirstlane(c >> 6);     // (wave-uniform: row addresses become scalar base + lane offset)
    const int Tn = p.T;
    const T* const xb = (const T*)p.x + (size_t)n * Tn * D;
    dos[c] = to_f32(((const T*)p.do0)[(size_t)n * D + c]);
    __syncthreads();
    {
        *(f32x4*)(dct + w * D + 4 * lane) = attn_last::cols_dot<T>((const T*)p.Wv, dos, w, lane);
#pragma unroll
        for (int h = 0; h < 4; ++h) qks[h * D + c] = p.qk[((size_t)n * 4 + h) * D + c];
        const float b = wave_sum(dos[c] * p.bv[c]);      // wave w covers columns 64 w .. 64 w + 63 = head w
        if (lane == 0) dob[w] = b;
    }
    __syncthreads();
    sweep_mfma<T>(xb, Tn, dct, dpt, 1.0f, dob, w, lane);
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
        for (int t0 = w; t0 < Tn; t0 += 32) {
            f32x4 xv[8];
            rows8<T>(xv, xb, Tn, lane, t0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
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
    // dq0[64 w + j] = Wk[64 w + j] . dqk_w
    if (w < p.H) attn_last::rows_mfma<T>((const T*)p.Wk, qks + w * D, dq0s, w, lane);
    else dq0s[c] = 0.0f;
    __syncthreads();
    ((T*)p.dq0)[(size_t)n * D + c] = from_f32<T>(dq0s[c]);
    // frame 0 also carries the residual path and the query: dx_0 += dpre + Wq^T dq0 (wave w sums rows 64 w .. 64 w + 63 of Wq)
    *(f32x4*)(red + w * D + 4 * lane) = attn_last::cols_dot<T>((const T*)p.Wq, dq0s, w, lane);
    __syncthreads();
    {
        const float a = row0[c] + to_f32(((const T*)p.dpre)[(size_t)n * D + c]) + (red[c] + red[D + c]) + (red[2 * D + c] + red[3 * D + c]);
        ((T*)p.dX)[(size_t)n * Tn * D + c] = from_f32<T>(a);
    }
}

// dWk[r][c] += sum_n q0[n][r] dqk[n][r / 64][c],  dWv[r][c] += sum_n do[n][r] ctx[n][r / 64][c],  dbv[r] += sum_n do[n][r] sp[n][r / 64]
// grid = (128, chunks): block b < 64 takes rows 4 b .. 4 b + 3 of Wk, b >= 64 the same rows of Wv (one head: one B row feeds four output
// rows); blockIdx.y a slice of the utterances; thread = column
template <typename T>
__global__ void __launch_bounds__(256) attn_last_wgrad_kernel(const void* q0, const float* dqk, const void* do0, const float* ctx,
                                                              const float* sp, float* dWk, float* dWv, float* dbv, int N, int per) {
    __shared__ float red[4];
    const int isv = blockIdx.x >> 6, r0 = (blockIdx.x & 63) * 4, h = r0 >> 6, c = threadIdx.x;
    const T* const a = (const T*)(isv ? do0 : q0) + r0;
    const float* const B = (isv ? ctx : dqk) + (size_t)h * 256 + c;
    const int n0 = blockIdx.y * per, n1 = min(N, n0 + per);
    f32x4 acc = f32x4{0, 0, 0, 0};
    int n = n0;
    for (; n + 8 <= n1; n += 8) {
        float b[8]; f32x4 av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { b[u] = B[(size_t)(n + u) * 1024]; av[u] = load4(a + (size_t)(n + u) * 256); }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += b[u] * av[u];
    }
    for (; n < n1; ++n) acc += B[(size_t)n * 1024] * load4(a + (size_t)n * 256);
    float* const dW = (isv ? dWv : dWk) + (size_t)r0 * 256 + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(dW + r * 256, acc[r]);
    if (isv) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float b = 0.0f;
            for (int m = n0 + c; m < n1; m += 256) b += to_f32(a[(size_t)m * 256 + r]) * sp[(size_t)m * 4 + h];
            b = block256_sum(b, red);
            if (c == 0) atomicAdd(dbv + r0 + r, b);
        }
    }
}

}  // namespace ge2e
