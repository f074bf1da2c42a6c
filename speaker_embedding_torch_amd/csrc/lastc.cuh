// The last layer AFTER its attention, and the tail, on compact rows (one per utterance: only frame 0 of the last layer's output is consumed,
// Modules.py:54), 16-bit storage modes -- ONE launch for what were five:
//
//     h1 = LN1( x0 + drop(o . Wo^T + bo) )                      (out_proj + dropout1 + residual + norm1;     was gemm_ws<EPI_LN>)
//     f  = drop(relu(h1 . W1^T + b1))                           (linear1;                                    was gemm_ws<EPI_BIAS_RELU_DROP>)
//     h2 = LN2( h1 + drop(f . W2^T + b2) )                      (linear2 + dropout2 + residual + norm2;      was gemm_sk + gemm_sk_epi)
//     z  = LN_f(h2);  e = Wq z + bq;  e /= max(|e|, 1e-12)       (transformer.norm, projection, F.normalize;  was tail_fwd_kernel)
//
// (reference: the torch TransformerEncoderLayer built at Modules.py:25-31, post-LN; Modules.py:33-35,54-57 for the tail).  At 960 rows each of
// those launches is a few microseconds of work behind ~5 us of dispatch and drain: 59 us of the step for 1.3 GFLOP.  Here a block owns 16
// rows (one MFMA row tile) for the whole chain; a row never leaves the block, so nothing but the saved activations (h1, f, h2, the LayerNorm
// statistics: the backward and the weight gradients read them) goes to memory:
//   * NW waves; wave w owns NT = 16 / NW column tiles of every 256-column group: acc[q][nt][r] = C[row i][256 q + n0 + 16 nt + 4 g + r], n0 = 16 NT w;
//     the weights are the MFMA's first operand, read from global memory (L2: 60 blocks stream the same 1.3 MB) as 16-byte row fragments in rounds
//     of 16 per wave, the next round in flight under this round's MFMAs (a scheduling barrier keeps the compiler from sinking each load to just
//     before its use) and a product's FIRST round issued before the previous product's epilogue and barrier; the activation tile is the second
//     operand, from a swizzled LDS row tile.  The products run at the rate a CU ingests 64-byte row fragments from L2 (~46 GB/s: 10 us per
//     512 KB weight; 4, 8 or 16 waves alike -- tools/lastc_bench.hip, profiles/r04_ab_log.txt section 6);
//   * every per-column constant (-> LDS) and every saved row is fetched in ONE prologue round trip: fetched where they are used, each epilogue
//     was one more dependent L2 round trip, eleven of them half of the first version's time;
//   * a row's LayerNorm statistics are a lane-quartet reduction + one LDS exchange between the waves (Chan's formula, as gemm_ws.cuh);
//   * the projection runs on the bf16 pipe at fp32 accuracy: z and Wq are split into bf16 hi + lo halves, three MFMAs per product
//     (common.cuh, "fp32x3"; the tail was fp32 arithmetic in every mode and stays at that accuracy: ~1e-5 against 4e-3 of a bf16 rounding).
// Rounding points are those of the five launches (h1, f, h2 are rounded to the storage type before their next use), so the saved tensors and the
// d-vector agree with them to the order of the LayerNorm / dot-product sums.  A tile holds whole utterances: 16 / samples of them (eval mode;
// training runs with samples == 1).
//
// lastc_bwd_kernel is the same chain backwards (see there).
#pragma once
#include "ffn.cuh"

namespace ge2e {

struct LastcArgs {
    int n;                          // compact rows (utterances)
    int abl;                        // (tools/lastc_bench.hip only) ablation bits
    int samples;                    // forward, eval mode: slices per utterance (rows m samples .. + samples - 1 are averaged after transformer.norm, Modules.py:55); 0 / 1: none
    int drow_mul;                   // dropout counter row = row * drow_mul (the frame-0 row of utterance `row` in the full-height numbering)
    float eps;
    // ---- forward
    const void* o;                  // [n, 256] of T: attention output of frame 0
    const void* x0; int ldx;        // layer input, frame-0 rows: row stride ldx elements
    const void* Wo; const float* bo; const float* g1; const float* be1;
    const void* W1; const float* b1; const void* W2; const float* b2; const float* g2; const float* be2;
    const float* gf; const float* bf; const float* wq; const float* bq;       // wq: projection.weight [256 out][256 in] fp32
    void* h1; float* rstd1; void* f; void* h2; float* rstd2;                  // saved ([n, 256 | 1024] of T; rstd: null in eval mode)
    float* xhat; float* rstd_f; float* zm; float* nrm; float* emb; float* emb_out;   // fp32 (tail); xhat / rstd_f / zm / nrm / emb_out may be null
    Drop d_sa, d_fh, d_ff;          // dropout1 (after out_proj), FFN hidden, dropout2
    // ---- backward (h1, f, h2, rstd*, xhat, nrm, emb are READ)
    const float* d_emb;             // [n, 256] fp32: dL/d(d-vector)
    const float* wqT;               // projection.weight transposed ([256 in][256 out] fp32)
    const void* W2T; const void* W1T; const void* WoT;      // linear2.weight^T [1024][256], linear1.weight^T [256][1024], out_proj.weight^T [256][256] of T
    float* d_raw;                   // [n, 256] fp32: dL/d(projection output) (the projection's weight gradient reads it)
    void* dH; void* dP; void* dM; void* dF; void* dHb; void* dP2; void* dM2; void* dO;   // [n, 256 | 1024] of T, as the separate launches leave them
    float* dgf; float* dbf; float* dg2; float* db2; float* dg1; float* db1;              // [256] fp32, atomically accumulated
};

constexpr int LC_XS = 16 * 512;     // [16][256] of T, swizzled row tile
constexpr int LC_FS = 16 * 2048;    // [16][1024] of T (later: the two bf16 planes of z)
#ifndef LASTC_NW
#define LASTC_NW 8
#endif
constexpr int LASTC_THREADS = 64 * LASTC_NW;
template <int NW> constexpr int lc_ex_bytes() { return 2 * NW * 16 * 2 * 4; }      // [parity][wave][row][2] floats
constexpr int LC_CS = 13 * 1024;    // per-column constants, fp32: forward bo g1 be1 b1[1024] b2 g2 be2 gf bf bq; backward gf g2 be2 g1 be1
template <int NW> constexpr size_t lastc_smem() { return LC_XS + LC_FS + lc_ex_bytes<NW>() + LC_CS; }
enum { LC_BO = 0, LC_G1 = 256, LC_BE1 = 512, LC_B1 = 768, LC_B2 = 1792, LC_G2 = 2048, LC_BE2 = 2304, LC_GF = 2560, LC_BF = 2816, LC_BQ = 3072 };   // forward
enum { LCB_GF = 0, LCB_G2 = 256, LCB_BE2 = 512, LCB_G1 = 768, LCB_BE1 = 1024 };                                                                // backward

// byte offset of columns c .. c + 3 (c % 4 == 0) of row r in a swizzled row tile of ROWB bytes per row (16-bit elements)
template <int ROWB> __device__ __forceinline__ int lc_off(int r, int c) { return swz_off<ROWB>(r, c >> 3) + ((c >> 2) & 1) * 8; }

// One product's weight stream:  acc[q][nt] += sum_k W[256 q + n0 + 16 nt + i][k] act[row i][k]   (q < NG column groups, k over KGN k-groups of 32).
// W row-major, ldwb bytes per row; act: swizzled LDS row tile of ROWB bytes per row.  Rounds of NT x RK (= 16) fragments in a two-deep register
// ring: prefetch() issues round 0 (callers put it before the previous product's epilogue), run() issues round r + 1 before the MFMAs of round r.
template <typename T, int NT, int NG, int KGN, int RK, int ROWB>
struct LcStream {
    static constexpr int NR = KGN / RK, ROUNDS = NG * NR;
    u32x4 wb[2][NT][RK];
    const unsigned char* base; int ldwb;
    __device__ __forceinline__ void init(const void* W, const int ldwb_, const int n0, const int i, const int g) {
        base = (const unsigned char*)W + (size_t)(n0 + i) * ldwb_ + g * 16; ldwb = ldwb_;
    }
    template <int RHO> __device__ __forceinline__ void load() {
        constexpr int q = RHO / NR, r = RHO % NR;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int k = 0; k < RK; ++k) wb[RHO & 1][nt][k] = *(const u32x4*)(base + (size_t)(256 * q + 16 * nt) * ldwb + (r * RK + k) * 64);
    }
    __device__ __forceinline__ void prefetch() { load<0>(); __builtin_amdgcn_sched_barrier(0); }
    __device__ __forceinline__ void run(const unsigned char* act, const int i, const int g, f32x4 (&acc)[NG][NT]) {
        ffn_static_for<0, ROUNDS>([&](auto RHO) {
            constexpr int rho = decltype(RHO)::value, q = rho / NR, r = rho % NR;
            if constexpr (rho + 1 < ROUNDS) {
                this->template load<rho + 1>();
                __builtin_amdgcn_sched_barrier(0);   // all of the next round's loads BEFORE this round's MFMAs (left alone, the compiler sinks each load to just before its use)
            }
#pragma unroll
            for (int k = 0; k < RK; ++k) {
                const u32x4 af = lds16(act + swz_off<ROWB>(i, (r * RK + k) * 4 + g));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[q][nt] = mma16<T>(wb[rho & 1][nt][k], af, acc[q][nt]);
            }
        });
    }
};
// The same for an fp32 [256][256] weight on the bf16 pipe at fp32 accuracy ("fp32x3", common.cuh): a fragment is 32 bytes per lane, split into bf16 hi + lo
// halves on arrival; the activation is two bf16 planes (hi, lo) in swizzled LDS row tiles of 512 bytes.  Rounds of NT x RK x 2 (= 16) loads.
template <int NT>
struct LcStreamX3 {
    static constexpr int RK = NT > 1 ? 8 / NT : 4, ROUNDS = 8 / RK;
    u32x4 raw[2][NT][RK][2];
    const float* base;
    __device__ __forceinline__ void init(const float* W, const int n0, const int i, const int g) { base = W + (size_t)(n0 + i) * 256 + g * 8; }
    template <int RHO> __device__ __forceinline__ void load() {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int k = 0; k < RK; ++k) {
                raw[RHO & 1][nt][k][0] = *(const u32x4*)(base + (size_t)(16 * nt) * 256 + (RHO * RK + k) * 32);
                raw[RHO & 1][nt][k][1] = *(const u32x4*)(base + (size_t)(16 * nt) * 256 + (RHO * RK + k) * 32 + 4);
            }
    }
    __device__ __forceinline__ void prefetch() { load<0>(); __builtin_amdgcn_sched_barrier(0); }
    __device__ __forceinline__ void run(const unsigned char* zh, const unsigned char* zl, const int i, const int g, f32x4 (&acc)[NT]) {
        ffn_static_for<0, ROUNDS>([&](auto RHO) {
            constexpr int rho = decltype(RHO)::value;
            if constexpr (rho + 1 < ROUNDS) { this->template load<rho + 1>(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int k = 0; k < RK; ++k) {
                const int ch = (rho * RK + k) * 4 + g;
                const u32x4 ah = lds16(zh + swz_off<512>(i, ch)), al = lds16(zl + swz_off<512>(i, ch));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    u32x2 h0, l0, h1, l1;
                    split_bf16x3(raw[rho & 1][nt][k][0], h0, l0); split_bf16x3(raw[rho & 1][nt][k][1], h1, l1);
                    acc[nt] = mma16_x3(u32x4{h0.x, h0.y, h1.x, h1.y}, u32x4{l0.x, l0.y, l1.x, l1.y}, ah, al, acc[nt]);
                }
            }
        });
    }
};

// (sum a, sum b) over the NW waves for each of the 16 rows; a, b already summed over the lane quartet (cross4_sum).  One block barrier.
template <int NW>
__device__ __forceinline__ void lc_exchange2(float* ex, const int wave, const int i, const int g, float& a, float& b) {
    if (g == 0) { ex[(wave * 16 + i) * 2] = a; ex[(wave * 16 + i) * 2 + 1] = b; }
    __syncthreads();
    float sa = 0.0f, sb = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { sa += ex[(w * 16 + i) * 2]; sb += ex[(w * 16 + i) * 2 + 1]; }
    a = sa; b = sb;
}
// LayerNorm statistics of the 16 rows: v = this wave's 16 NT columns of row i (4 NT per lane); the waves' (mean, M2) pairs combine by Chan's formula.
// One block barrier.
template <int NW, int NT>
__device__ __forceinline__ void lc_ln_stats(const f32x4 (&v)[NT], float* ex, const int wave, const int i, const int g, const float eps, float& mean, float& rs) {
    constexpr float CW = 16.0f * NT;
    float sm = 0.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) sm += (v[nt][0] + v[nt][1]) + (v[nt][2] + v[nt][3]);
    const float mw = cross4_sum(sm) * (1.0f / CW);
    float qw = 0.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = v[nt][r] - mw; qw += d * d; }
    qw = cross4_sum(qw);
    if (g == 0) { ex[(wave * 16 + i) * 2] = mw; ex[(wave * 16 + i) * 2 + 1] = qw; }
    __syncthreads();
    float m2 = 0.0f, mws[NW];
    mean = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { mws[w] = ex[(w * 16 + i) * 2]; mean += mws[w]; m2 += ex[(w * 16 + i) * 2 + 1]; }
    mean *= 1.0f / NW;
#pragma unroll
    for (int w = 0; w < NW; ++w) m2 += CW * (mws[w] - mean) * (mws[w] - mean);
    rs = 1.0f / sqrtf(m2 * (1.0f / 256.0f) + eps);
}
template <typename T> __device__ __forceinline__ f32x4 lc_round4(f32x4 v) {     // what a store in T and a load back give
    return f32x4{to_f32(from_f32<T>(v[0])), to_f32(from_f32<T>(v[1])), to_f32(from_f32<T>(v[2])), to_f32(from_f32<T>(v[3]))};
}
// 4 stored values (8 bytes, in registers) -> floats
template <typename T> __device__ __forceinline__ f32x4 lc_cvt4(u32x2 v);
template <> __device__ __forceinline__ f32x4 lc_cvt4<bf16_t>(u32x2 v) {
    return f32x4{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xFFFF0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xFFFF0000u)};
}
template <> __device__ __forceinline__ f32x4 lc_cvt4<f16_t>(u32x2 v) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
    const f16x4_t h = __builtin_bit_cast(f16x4_t, v);
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
// 4 fp32 values -> 8 bytes in each of the two bf16 planes
__device__ __forceinline__ void lc_put_x3(unsigned char* zh, unsigned char* zl, const int off, const f32x4 v) {
    u32x2 hi, lo;
    split_bf16x3(__builtin_bit_cast(u32x4, v), hi, lo);
    *(u32x2*)(zh + off) = hi; *(u32x2*)(zl + off) = lo;
}

// grid = ceil((n / samples) / (16 / samples)) blocks of 64 NW threads
template <typename T, int NW>
__global__ void __launch_bounds__(64 * NW) lastc_fwd_kernel(const LastcArgs p) {
    static_assert(sizeof(T) == 2, "16-bit storage modes");
    constexpr int NT = 16 / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Xs = smem;
    unsigned char* const Fs = smem + LC_XS;
    float* const Ex = (float*)(smem + LC_XS + LC_FS);
    float* const Ex1 = Ex + NW * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    // a tile holds G = 16 / samples whole utterances (samples = 1: 16 rows; samples = 5: 15 rows, the 16th idles) so that the slice mean stays inside it
    const int smp = p.samples > 1 ? p.samples : 1, G = 16 / smp, RT = G * smp;
    const int r0 = blockIdx.x * RT, row = r0 + i;
    const bool ok = i < RT && row < p.n;
    const int rowc = row < p.n ? row : p.n - 1;
    const int n0 = wave * 16 * NT;
    const uint32_t drow = (uint32_t)row * (uint32_t)(p.drow_mul > 0 ? p.drow_mul : 1);

    // ---- everything that does not depend on a product is fetched NOW, in one round trip: the first weight round, the per-column constants (-> LDS:
    //      fetched where they are used, each epilogue is one more dependent L2 round trip -- eleven of them were half of the kernel's time), the
    //      residual rows, the attention output tile
    LcStream<T, NT, 1, 8, (NT > 1 ? 16 / NT : 8), 512> s1;              // out_proj
    s1.init(p.Wo, 512, n0, i, g);
    s1.prefetch();
    float* const Cs = (float*)(smem + LC_XS + LC_FS + lc_ex_bytes<NW>());
    constexpr int NSEG = 13, SPW = (NSEG + NW - 1) / NW, OPT = (512 + 64 * NW - 1) / (64 * NW);
    f32x4 cseg[SPW];
    u32x4 ot[OPT];
    u32x2 x0r[NT];
#pragma unroll
    for (int q = 0; q < SPW; ++q) {
        const int sg = wave + NW * q;                      // wave-uniform
        const float* src = sg == 0 ? p.bo : sg == 1 ? p.g1 : sg == 2 ? p.be1 : sg < 7 ? p.b1 + 256 * (sg - 3) : sg == 7 ? p.b2 : sg == 8 ? p.g2 : sg == 9 ? p.be2 :
                           sg == 10 ? p.gf : sg == 11 ? p.bf : p.bq;
        if (sg < NSEG) cseg[q] = *(const f32x4*)(src + 4 * lane);
    }
#pragma unroll
    for (int q = 0; q < OPT; ++q) {
        const int ch = (tid + 64 * NW * q) & 511, r = ch >> 5, pos = ch & 31;
        const int gr = r0 + r < p.n ? r0 + r : p.n - 1;
        ot[q] = *(const u32x4*)((const unsigned char*)p.o + ((size_t)gr * 256 + pos * 8) * 2);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) x0r[nt] = *(const u32x2*)((const T*)p.x0 + (size_t)rowc * p.ldx + n0 + 16 * nt + 4 * g);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < SPW; ++q) if (wave + NW * q < NSEG) *(f32x4*)(Cs + 256 * (wave + NW * q) + 4 * lane) = cseg[q];
#pragma unroll
    for (int q = 0; q < OPT; ++q) { const int ch = tid + 64 * NW * q; if (ch < 512) *(u32x4*)(Xs + swz_off<512>(ch >> 5, ch & 31)) = ot[q]; }
    __syncthreads();

    // ---- h1 = LN1(x0 + drop(o Wo^T + bo))
    LcStream<T, NT, 4, 8, (NT > 1 ? 16 / NT : 8), 512> s2;              // linear1: this wave's 64 NT hidden units (16 NT of each 256)
    f32x4 h1v[NT];
    {
        f32x4 acc[1][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[0][nt] = f32x4{0, 0, 0, 0};
        if (!(p.abl & 1)) s1.run(Xs, i, g, acc);
        s2.init(p.W1, 512, n0, i, g);
        s2.prefetch();
        f32x4 v[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int lc = n0 + 16 * nt + 4 * g;
            v[nt] = acc[0][nt] + *(const f32x4*)(Cs + LC_BO + lc);
            drop_apply4(p.d_sa, drow * 256u + (uint32_t)lc, v[nt]);
            v[nt] += lc_cvt4<T>(x0r[nt]);
        }
        float mean, rs;
        lc_ln_stats<NW, NT>(v, Ex, wave, i, g, p.eps, mean, rs);     // (its barrier: every wave is done reading the o tile)
        if (p.rstd1 && g == 0 && wave == 0 && ok) p.rstd1[row] = rs;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int lc = n0 + 16 * nt + 4 * g;
            const f32x4 ga = *(const f32x4*)(Cs + LC_G1 + lc), be = *(const f32x4*)(Cs + LC_BE1 + lc);
#pragma unroll
            for (int r = 0; r < 4; ++r) h1v[nt][r] = (v[nt][r] - mean) * rs * ga[r] + be[r];
            store4((T*)(Xs + lc_off<512>(i, lc)), h1v[nt][0], h1v[nt][1], h1v[nt][2], h1v[nt][3]);
            if (ok) store4((T*)p.h1 + (size_t)row * 256 + lc, h1v[nt][0], h1v[nt][1], h1v[nt][2], h1v[nt][3]);
            h1v[nt] = lc_round4<T>(h1v[nt]);                           // the residual of norm2 is the STORED h1
        }
    }
    __syncthreads();

    // ---- f = drop(relu(h1 W1^T + b1))
    LcStream<T, NT, 1, 32, (NT > 1 ? 16 / NT : 8), 2048> s3;            // linear2
    {
        f32x4 acc[4][NT];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[q][nt] = f32x4{0, 0, 0, 0};
        if (!(p.abl & 2)) s2.run(Xs, i, g, acc);
        s3.init(p.W2, 2048, n0, i, g);
        s3.prefetch();
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = 256 * q + n0 + 16 * nt + 4 * g;
                f32x4 v = acc[q][nt] + *(const f32x4*)(Cs + LC_B1 + col);
                (void)relu_drop_apply4(p.d_fh, drow * 1024u + (uint32_t)col, v);
                store4((T*)(Fs + lc_off<2048>(i, col)), v[0], v[1], v[2], v[3]);
                if (ok && !(p.abl & 16)) store4((T*)p.f + (size_t)row * 1024 + col, v[0], v[1], v[2], v[3]);
            }
    }
    __syncthreads();

    // ---- h2 = LN2(h1 + drop(f W2^T + b2))
    LcStreamX3<NT> s4;                                   // projection
    f32x4 h2v[NT];
    {
        f32x4 acc[1][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[0][nt] = f32x4{0, 0, 0, 0};
        if (!(p.abl & 4)) s3.run(Fs, i, g, acc);
        s4.init(p.wq, n0, i, g);
        s4.prefetch();
        f32x4 v[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int lc = n0 + 16 * nt + 4 * g;
            v[nt] = acc[0][nt] + *(const f32x4*)(Cs + LC_B2 + lc);
            drop_apply4(p.d_ff, drow * 256u + (uint32_t)lc, v[nt]);
            v[nt] += h1v[nt];
        }
        float mean, rs;
        lc_ln_stats<NW, NT>(v, Ex1, wave, i, g, p.eps, mean, rs);    // (its barrier: every wave is done reading the hidden tile)
        if (p.rstd2 && g == 0 && wave == 0 && ok) p.rstd2[row] = rs;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int lc = n0 + 16 * nt + 4 * g;
            const f32x4 ga = *(const f32x4*)(Cs + LC_G2 + lc), be = *(const f32x4*)(Cs + LC_BE2 + lc);
#pragma unroll
            for (int r = 0; r < 4; ++r) h2v[nt][r] = (v[nt][r] - mean) * rs * ga[r] + be[r];
            if (ok) store4((T*)p.h2 + (size_t)row * 256 + lc, h2v[nt][0], h2v[nt][1], h2v[nt][2], h2v[nt][3]);
            h2v[nt] = lc_round4<T>(h2v[nt]);                           // the tail reads the STORED h2
        }
    }

    // ---- z = LN_f(h2) -> the two bf16 planes of z (in the hidden tile's space)
    unsigned char* const Zh = Fs, * const Zl = Fs + LC_XS;
    {
        float mean, rs;
        lc_ln_stats<NW, NT>(h2v, Ex, wave, i, g, p.eps, mean, rs);
        if (p.rstd_f && g == 0 && wave == 0 && ok) p.rstd_f[row] = rs;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int lc = n0 + 16 * nt + 4 * g;
            const f32x4 ga = *(const f32x4*)(Cs + LC_GF + lc), be = *(const f32x4*)(Cs + LC_BF + lc);
            f32x4 xh, z;
#pragma unroll
            for (int r = 0; r < 4; ++r) { xh[r] = (h2v[nt][r] - mean) * rs; z[r] = xh[r] * ga[r] + be[r]; }
            if (p.xhat && ok) *(f32x4*)(p.xhat + (size_t)row * 256 + lc) = xh;
            if (smp == 1) {
                if (p.zm && ok) *(f32x4*)(p.zm + (size_t)row * 256 + lc) = z;
                lc_put_x3(Zh, Zl, lc_off<512>(i, lc), z);
            } else *(f32x4*)(Fs + 2 * LC_XS + i * 1024 + lc * 4) = z;          // fp32 rows for the slice mean (the hidden tile's upper half)
        }
    }
    __syncthreads();
    const int m = blockIdx.x * G + i;                      // output row: utterance
    const bool okm = i < G && m < p.n / smp;
    if (smp > 1) {                                         // (block-uniform) z' = mean over the utterance's slices; rows past G carry zeros
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int lc = n0 + 16 * nt + 4 * g;
            f32x4 zp = f32x4{0, 0, 0, 0};
            if (i < G)
                for (int q = 0; q < smp; ++q) zp += *(const f32x4*)(Fs + 2 * LC_XS + (i * smp + q) * 1024 + lc * 4);
            zp *= 1.0f / (float)smp;
            if (p.zm && okm) *(f32x4*)(p.zm + (size_t)m * 256 + lc) = zp;
            lc_put_x3(Zh, Zl, lc_off<512>(i, lc), zp);
        }
        __syncthreads();
    }

    // ---- e = Wq z + bq;  e /= max(|e|, 1e-12)
    {
        f32x4 e[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) e[nt] = *(const f32x4*)(Cs + LC_BQ + n0 + 16 * nt + 4 * g);
        if (!(p.abl & 8)) s4.run(Zh, Zl, i, g, e);
        float ss = 0.0f, unused = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ss += e[nt][r] * e[nt][r];
        ss = cross4_sum(ss);
        lc_exchange2<NW>(Ex1, wave, i, g, ss, unused);
        const float nn = fmaxf(sqrtf(ss), 1e-12f);
        if (p.nrm && g == 0 && wave == 0 && okm) p.nrm[m] = nn;
        if (okm) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int lc = n0 + 16 * nt + 4 * g;
                const f32x4 o = f32x4{e[nt][0] / nn, e[nt][1] / nn, e[nt][2] / nn, e[nt][3] / nn};
                *(f32x4*)(p.emb + (size_t)m * 256 + lc) = o;
                if (p.emb_out) *(f32x4*)(p.emb_out + (size_t)m * 256 + lc) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same chain backwards, ONE launch for what were six (tail_bwd_kernel, ln_bwd_kernel, gemm_ws<EPI_MASK>, gemm_sk + gemm_sk_epi,
// gemm_ws_lnbwd):
//     d_raw = (d_emb - e <d_emb, e>) / |e_raw|          dz = d_raw . Wq                dH  = LN_f'(dz; xhat)            (+ dgamma_f, dbeta_f)
//     dP  = LN2'(dH; h2)     dM  = drop2'(dP)           dF = (dM . W2) o relu'/drop'   dH1 = dP + dF . W1               (+ dgamma2, dbeta2)
//     dP2 = LN1'(dH1; h1)    dM2 = drop1'(dP2)          dO = dM2 . Wo                                                    (+ dgamma1, dbeta1)
// Every intermediate the weight gradients (dM, dF, dM2, d_raw) and the attention backward (dO, dP2) read is stored as the separate launches
// store it, rounded to the storage type at the same points; the LayerNorm parameter gradients are column sums over the block's 16 rows
// (16-lane shuffles) + one atomic per column and block.  Rows past n enter as zero gradients and contribute nothing.
// ---------------------------------------------------------------------------------------------
// dx = rs (dxh - mean(dxh) - xh mean(dxh xh)) with dxh = dy gamma, per row; one block barrier
template <int NW, int NT>
__device__ __forceinline__ void lc_ln_bwd(const f32x4 (&dy)[NT], const f32x4 (&xh)[NT], const float* gamma, const int n0, const float rs,
                                          float* ex, const int wave, const int i, const int g, f32x4 (&dx)[NT]) {
    f32x4 dxh[NT];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const f32x4 ga = *(const f32x4*)(gamma + n0 + 16 * nt + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) { dxh[nt][r] = dy[nt][r] * ga[r]; s1 += dxh[nt][r]; s2 += dxh[nt][r] * xh[nt][r]; }
    }
    s1 = cross4_sum(s1); s2 = cross4_sum(s2);
    lc_exchange2<NW>(ex, wave, i, g, s1, s2);
    s1 *= 1.0f / 256.0f; s2 *= 1.0f / 256.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dx[nt][r] = rs * (dxh[nt][r] - s1 - xh[nt][r] * s2);
}
// dgamma[c] += sum_rows dy xh, dbeta[c] += sum_rows dy over the block's 16 rows
template <int NT>
__device__ __forceinline__ void lc_colsum(const f32x4 (&dy)[NT], const f32x4 (&xh)[NT], float* dgamma, float* dbeta, const int n0, const int i, const int g) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = group16_sum(dy[nt][r] * xh[nt][r]), b = group16_sum(dy[nt][r]);
            if (i == 0 && dgamma) { atomicAdd(dgamma + n0 + 16 * nt + 4 * g + r, a); atomicAdd(dbeta + n0 + 16 * nt + 4 * g + r, b); }
        }
}
// xhat rebuilt from the saved LayerNorm OUTPUT (as ln_bwd_kernel does): (y - beta) / gamma, 0 where gamma is 0
template <typename T, int NT>
__device__ __forceinline__ void lc_xhat_from_y(const u32x2 (&yr)[NT], const float* gamma, const float* beta, const int n0, const int g, f32x4 (&xh)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int lc = n0 + 16 * nt + 4 * g;
        const f32x4 y = lc_cvt4<T>(yr[nt]), ga = *(const f32x4*)(gamma + lc), be = *(const f32x4*)(beta + lc);
#pragma unroll
        for (int r = 0; r < 4; ++r) xh[nt][r] = (y[r] - be[r]) * (ga[r] != 0.0f ? 1.0f / ga[r] : 0.0f);
    }
}

template <typename T, int NW>
__global__ void __launch_bounds__(64 * NW) lastc_bwd_kernel(const LastcArgs p) {
    static_assert(sizeof(T) == 2, "16-bit storage modes");
    constexpr int NT = 16 / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Xs = smem;
    unsigned char* const Fs = smem + LC_XS;
    unsigned char* const Zh = Fs, * const Zl = Fs + LC_XS;
    float* const Ex = (float*)(smem + LC_XS + LC_FS);
    float* const Ex1 = Ex + NW * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int r0 = blockIdx.x * 16, row = r0 + i;
    const bool ok = row < p.n;
    const int rowc = ok ? row : p.n - 1;
    const int n0 = wave * 16 * NT;
    const uint32_t drow = (uint32_t)row * (uint32_t)(p.drow_mul > 0 ? p.drow_mul : 1);

    // ---- everything that does not depend on a product is fetched NOW, in one round trip (see the forward)
    LcStreamX3<NT> s0;                                   // dz = d_raw Wq
    s0.init(p.wqT, n0, i, g);
    s0.prefetch();
    float* const Cs = (float*)(smem + LC_XS + LC_FS + lc_ex_bytes<NW>());
    f32x4 cseg, de[NT], e[NT], xhf[NT];
    u32x2 y2r[NT], y1r[NT], fm[4][NT];
    {
        const float* src = wave == 0 ? p.gf : wave == 1 ? p.g2 : wave == 2 ? p.be2 : wave == 3 ? p.g1 : p.be1;
        if (wave < 5 || NW == 4) cseg = *(const f32x4*)(src + 4 * lane);
    }
    f32x4 cseg4 = f32x4{0, 0, 0, 0};
    if constexpr (NW == 4) { if (wave == 0) cseg4 = *(const f32x4*)(p.be1 + 4 * lane); }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int lc = n0 + 16 * nt + 4 * g;
        de[nt] = *(const f32x4*)(p.d_emb + (size_t)rowc * 256 + lc);
        e[nt] = *(const f32x4*)(p.emb + (size_t)rowc * 256 + lc);
        xhf[nt] = *(const f32x4*)(p.xhat + (size_t)rowc * 256 + lc);
        y2r[nt] = *(const u32x2*)((const T*)p.h2 + (size_t)rowc * 256 + lc);
        y1r[nt] = *(const u32x2*)((const T*)p.h1 + (size_t)rowc * 256 + lc);
#pragma unroll
        for (int q = 0; q < 4; ++q) fm[q][nt] = *(const u32x2*)((const T*)p.f + (size_t)rowc * 1024 + 256 * q + lc);
    }
    const float nn = p.nrm[rowc], rsf = p.rstd_f[rowc], rs2 = p.rstd2[rowc], rs1 = p.rstd1[rowc];
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NW == 4) {                             // five segments over four waves: wave 3 -> g1, wave 0 also -> be1
        const int sg = wave;
        *(f32x4*)(Cs + 256 * sg + 4 * lane) = cseg;
        if (wave == 0) *(f32x4*)(Cs + LCB_BE1 + 4 * lane) = cseg4;
    } else if (wave < 5) *(f32x4*)(Cs + 256 * wave + 4 * lane) = cseg;
    if (!ok) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) de[nt] = f32x4{0, 0, 0, 0};
    }
    // ---- d_raw = (d_emb - e <d_emb, e>) / |e_raw|
    {
        float dot = 0.0f, unused = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dot += de[nt][r] * e[nt][r];
        dot = cross4_sum(dot);
        lc_exchange2<NW>(Ex, wave, i, g, dot, unused);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int lc = n0 + 16 * nt + 4 * g;
            f32x4 dr;
#pragma unroll
            for (int r = 0; r < 4; ++r) dr[r] = (de[nt][r] - e[nt][r] * dot) / nn;
            if (ok) *(f32x4*)(p.d_raw + (size_t)row * 256 + lc) = dr;
            lc_put_x3(Zh, Zl, lc_off<512>(i, lc), dr);
        }
    }
    __syncthreads();

    // ---- dz = d_raw Wq;  dH = LN_f'(dz)
    LcStream<T, NT, 4, 8, (NT > 1 ? 16 / NT : 8), 512> s1;              // dF = dM W2
    f32x4 dy[NT];
    {
        f32x4 dz[NT], dx[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) dz[nt] = f32x4{0, 0, 0, 0};
        if (!(p.abl & 32)) s0.run(Zh, Zl, i, g, dz);
        s1.init(p.W2T, 512, n0, i, g);
        s1.prefetch();
        lc_ln_bwd<NW, NT>(dz, xhf, Cs + LCB_GF, n0, rsf, Ex1, wave, i, g, dx);
        lc_colsum<NT>(dz, xhf, p.dgf, p.dbf, n0, i, g);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (ok) store4((T*)p.dH + (size_t)row * 256 + n0 + 16 * nt + 4 * g, dx[nt][0], dx[nt][1], dx[nt][2], dx[nt][3]);
            dy[nt] = lc_round4<T>(dx[nt]);
        }
    }

    // ---- dP = LN2'(dH; h2);  dM = drop2'(dP) -> the first product's activation tile
    f32x4 dPr[NT];
    {
        f32x4 xh[NT], dx[NT];
        lc_xhat_from_y<T, NT>(y2r, Cs + LCB_G2, Cs + LCB_BE2, n0, g, xh);
        lc_ln_bwd<NW, NT>(dy, xh, Cs + LCB_G2, n0, rs2, Ex, wave, i, g, dx);
        lc_colsum<NT>(dy, xh, p.dg2, p.db2, n0, i, g);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int lc = n0 + 16 * nt + 4 * g;
            if (ok) store4((T*)p.dP + (size_t)row * 256 + lc, dx[nt][0], dx[nt][1], dx[nt][2], dx[nt][3]);
            dPr[nt] = lc_round4<T>(dx[nt]);
            if (p.d_ff.thr) {
                drop_apply4(p.d_ff, drow * 256u + (uint32_t)lc, dx[nt]);
                if (ok) store4((T*)p.dM + (size_t)row * 256 + lc, dx[nt][0], dx[nt][1], dx[nt][2], dx[nt][3]);
            }
            store4((T*)(Xs + lc_off<512>(i, lc)), dx[nt][0], dx[nt][1], dx[nt][2], dx[nt][3]);
        }
    }
    __syncthreads();

    // ---- dF = (dM W2) o relu'/drop' (the stored hidden is the mask)
    LcStream<T, NT, 1, 32, (NT > 1 ? 16 / NT : 8), 2048> s2;            // dH1 = dP + dF W1
    {
        f32x4 acc[4][NT];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[q][nt] = f32x4{0, 0, 0, 0};
        if (!(p.abl & 64)) s1.run(Xs, i, g, acc);
        s2.init(p.W1T, 2048, n0, i, g);
        s2.prefetch();
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = 256 * q + n0 + 16 * nt + 4 * g;
                const f32x4 m4 = lc_cvt4<T>(fm[q][nt]);
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = m4[r] > 0.0f ? acc[q][nt][r] * p.d_fh.scale : 0.0f;
                store4((T*)(Fs + lc_off<2048>(i, col)), v[0], v[1], v[2], v[3]);
                if (ok) store4((T*)p.dF + (size_t)row * 1024 + col, v[0], v[1], v[2], v[3]);
            }
    }
    __syncthreads();

    // ---- dH1 = dP + dF W1
    LcStream<T, NT, 1, 8, (NT > 1 ? 16 / NT : 8), 512> s3;              // dO = dM2 Wo
    {
        f32x4 acc[1][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[0][nt] = f32x4{0, 0, 0, 0};
        if (!(p.abl & 128)) s2.run(Fs, i, g, acc);
        s3.init(p.WoT, 512, n0, i, g);
        s3.prefetch();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 v = acc[0][nt] + dPr[nt];
            if (ok) store4((T*)p.dHb + (size_t)row * 256 + n0 + 16 * nt + 4 * g, v[0], v[1], v[2], v[3]);
            dy[nt] = lc_round4<T>(v);
        }
    }

    // ---- dP2 = LN1'(dH1; h1);  dM2 = drop1'(dP2) -> the last product's activation tile (every wave is past the first product: the barrier above)
    {
        f32x4 xh[NT], dx[NT];
        lc_xhat_from_y<T, NT>(y1r, Cs + LCB_G1, Cs + LCB_BE1, n0, g, xh);
        lc_ln_bwd<NW, NT>(dy, xh, Cs + LCB_G1, n0, rs1, Ex1, wave, i, g, dx);
        lc_colsum<NT>(dy, xh, p.dg1, p.db1, n0, i, g);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int lc = n0 + 16 * nt + 4 * g;
            if (ok) store4((T*)p.dP2 + (size_t)row * 256 + lc, dx[nt][0], dx[nt][1], dx[nt][2], dx[nt][3]);
            if (p.d_sa.thr) {
                drop_apply4(p.d_sa, drow * 256u + (uint32_t)lc, dx[nt]);
                if (ok) store4((T*)p.dM2 + (size_t)row * 256 + lc, dx[nt][0], dx[nt][1], dx[nt][2], dx[nt][3]);
            }
            store4((T*)(Xs + lc_off<512>(i, lc)), dx[nt][0], dx[nt][1], dx[nt][2], dx[nt][3]);
        }
    }
    __syncthreads();

    // ---- dO = dM2 Wo
    {
        f32x4 acc[1][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[0][nt] = f32x4{0, 0, 0, 0};
        if (!(p.abl & 256)) s3.run(Xs, i, g, acc);
        if (ok) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                store4((T*)p.dO + (size_t)row * 256 + n0 + 16 * nt + 4 * g, acc[0][nt][0], acc[0][nt][1], acc[0][nt][2], acc[0][nt][3]);
        }
    }
}

}  // namespace ge2e
