// Fused self-attention of one (utterance, head): SURVEY.md 8a row a4 and its backward (a13).
//
// The whole key/value sequence of a head (T <= 32*KT frames, 64 dims) is LDS-resident; a wave owns
// 16 query rows at a time.  Scores are computed TRANSPOSED (S^T = K Q^T) so that a query's row of
// probabilities lives on one lane quartet and the probability accumulators feed the P.V MFMAs
// directly as operands (no LDS round trip; guide section 3 "accumulator tile as the next operand").
#pragma once
#include "common.cuh"
#include <type_traits>

namespace ge2e {

struct AttnArgs {
    const void* qkv;     // [R, 3*D] of T : q | k | v, head h at columns h*64
    void* o;             // fwd out   [R, D]   (bwd: the saved forward output, for delta = dO . O)
    float* lse;          // [R, H] log-sum-exp of the scaled scores: written by fwd when non-null, read by bwd
    const void* dout;    // bwd in    [R, D]
    void* dqkv;          // bwd out   [R, 3*D]
    int T, H, D;
    float scale;         // 1/sqrt(64)
    Drop drop;           // dropout on the probabilities
};

template <typename T> __device__ __forceinline__ float exp_prec(float x);
template <> __device__ __forceinline__ float exp_prec<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ float exp_prec<bf16_t>(float x) { return __expf(x); }
template <> __device__ __forceinline__ float exp_prec<f16_t>(float x) { return __expf(x); }
// exp(score * scale - offset) with the constants folded for v_exp_f32 (= 2^x) in bf16 mode: callers pass
// scale * EXPK and offset * EXPK and call exp_k (one fma + one exp per probability); fp32 mode keeps expf.
template <typename T> struct ExpK;
template <> struct ExpK<float> { static constexpr float K = 1.0f; static __device__ __forceinline__ float ex(float x) { return expf(x); } };
template <> struct ExpK<bf16_t> { static constexpr float K = 1.4426950408889634f; static __device__ __forceinline__ float ex(float x) { return __builtin_amdgcn_exp2f(x); } };
template <> struct ExpK<f16_t> : ExpK<bf16_t> {};
template <> struct ExpK<x3_t> : ExpK<bf16_t> {};        // fp32x3: v_exp_f32 (~1 ulp of fp32), as the 16-bit modes

namespace attn {
// LDS image of a [rows][64] head tile, read BOTH by rows (ds_read_b128 operand fragments) and transposed (ds_read_b64_tr_b16):
//   16-bit modes: plain 128-byte rows, 16-byte chunk c of row r stored at chunk c ^ (r & 7).  Both kinds of read are then free of
//     bank conflicts: the four 16-lane groups of a ds_read_b128 each cover the 16 sixteen-byte slots of a 256-byte bank row once,
//     and the 32 eight-byte pieces of a half-wave's transposed read cover it once (rows r0 .. r0 + 7, r0 a multiple of 8).
//     (A 144-byte padded pitch, used before, made the row reads 2-way: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.41 / 0.43 in
//     attention backward / forward, profiles/r02_pmc_mfma_lds_per_kernel.csv.)
//   fp32 mode: 256-byte rows padded by 16 bytes (scalar transposed reads).
template <typename T> struct Geo {
    static constexpr int ROWB = 64 * (int)sizeof(T);
    static constexpr bool SWZ = sizeof(T) == 2;
    static constexpr int LD = SWZ ? ROWB : ROWB + 16;
    static constexpr int CPR = ROWB / 16;
    static constexpr int NKG = 64 / Prec<T>::KG;       // k-groups across the head dim
};
// byte offset of 16-byte chunk `chunk` of row `row`
template <typename T> __device__ __forceinline__ int toff(int row, int chunk) {
    if constexpr (Geo<T>::SWZ) return row * Geo<T>::LD + ((chunk ^ (row & 7)) << 4);
    else return row * Geo<T>::LD + chunk * 16;
}
// transposed fragment of a head tile (rows r0 .. r0 + 31 for 16-bit, columns c0 .. c0 + 15): see frag_tr in common.cuh
template <typename T> __device__ __forceinline__ u32x4 tile_tr(const unsigned char* tile, int r0, int c0, int lane) {
    if constexpr (Geo<T>::SWZ) {
        const int i = lane & 15, g = lane >> 4;
        const int row = r0 + 4 * g + (i >> 2), bc = (c0 + 4 * (i & 3)) * 2;          // this lane's 8-byte piece: row, byte column
        const unsigned char* p = tile + toff<T>(row, bc >> 4) + (bc & 8);
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * Geo<T>::LD));      // row + 16: the same swizzle
        u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        return u32x4{l2.x, l2.y, h2.x, h2.y};
    } else return frag_tr<T>(tile, Geo<T>::LD, r0, c0, lane);
}

// cooperative load of `rows` x 64 head slice into a padded tile; rows >= T are zero
template <typename T>
__device__ __forceinline__ void load_tile(unsigned char* dst, const unsigned char* src, size_t src_ld_bytes,
                                          int T_, int TP) {
    using G = Geo<T>;
    for (int id = threadIdx.x; id < TP * G::CPR; id += blockDim.x) {
        const int row = id / G::CPR, c = id % G::CPR;
        u32x4 v = u32x4{0, 0, 0, 0};
        if (row < T_) v = *(const u32x4*)(src + (size_t)row * src_ld_bytes + c * 16);
        *(u32x4*)(dst + toff<T>(row, c)) = v;
    }
}
// this lane's row fragments (row `row` of a [.,64] slice) straight from global memory
template <typename T>
__device__ __forceinline__ void load_row_frags(u32x4* f, const unsigned char* src, size_t src_ld_bytes, int row,
                                               bool valid, int g) {
    using G = Geo<T>;
#pragma unroll
    for (int k = 0; k < G::NKG; ++k)
        f[k] = valid ? *(const u32x4*)(src + (size_t)row * src_ld_bytes + (k * 4 + g) * 16) : u32x4{0, 0, 0, 0};
}
// 16x16 tile of (tile rows x lane-owned rows): acc[r] = sum_d Tile[16*t + 4g + r][d] * f[lane row i][d]
template <typename T>
__device__ __forceinline__ f32x4 tile_dot(const unsigned char* tile, int t, const u32x4* f, int i, int g) {
    using G = Geo<T>;
    f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < G::NKG; ++k)
        acc = mma16<T>(lds16(tile + toff<T>(16 * t + i, k * 4 + g)), f[k], acc);
    return acc;
}

// ---- the same tile operations with the arithmetic mode as a parameter of the FRAGMENT type (resident kernels) --------------------------------
// fp32x3 (T = x3_t; common.cuh): q | k | v and dO are fp32 in memory; in LDS a head tile is TWO planes of the 16-bit geometry (the bf16 hi
// halves, then -- `po` bytes further -- the lo halves), split once on the way in; an operand fragment is a (hi, lo) pair and a product three
// MFMAs.  The probabilities / dS tiles are split the same way when they become operands.
template <typename T> struct XGeo : Geo<T> { static constexpr int PLANES = 1; };
template <> struct XGeo<x3_t> {
    static constexpr int ROWB = 128, LD = 128, CPR = 8, NKG = 2, PLANES = 2;
    static constexpr bool SWZ = true;
};
template <typename T> constexpr int xtile_bytes(int rows) { return rows * XGeo<T>::LD * XGeo<T>::PLANES; }
template <typename T> struct Frag { u32x4 h; };
template <> struct Frag<x3_t> { u32x4 h, l; };
template <typename T> __device__ __forceinline__ int xtoff(int row, int chunk) {
    if constexpr (std::is_same<T, x3_t>::value) return toff<bf16_t>(row, chunk);
    else return toff<T>(row, chunk);
}
template <typename T> __device__ __forceinline__ f32x4 xmma(const Frag<T>& a, const Frag<T>& b, f32x4 c) {
    if constexpr (std::is_same<T, x3_t>::value) return mma16_x3(a.h, a.l, b.h, b.l, c);
    else return mma16<T>(a.h, b.h, c);
}
// accumulator tiles -> operand fragment ("acc mapping" of pack_acc); fp32x3: hi = bf16(v), lo = bf16(v - hi)
template <typename T> __device__ __forceinline__ Frag<T> xpack(f32x4 t0, f32x4 t1) {
    if constexpr (std::is_same<T, x3_t>::value) {
        Frag<x3_t> f;
        f.h = pack_acc<bf16_t>(t0, t1);
        f32x4 r0, r1;
        r0[0] = t0[0] - __uint_as_float(f.h.x << 16); r0[1] = t0[1] - __uint_as_float(f.h.x & 0xFFFF0000u);
        r0[2] = t0[2] - __uint_as_float(f.h.y << 16); r0[3] = t0[3] - __uint_as_float(f.h.y & 0xFFFF0000u);
        r1[0] = t1[0] - __uint_as_float(f.h.z << 16); r1[1] = t1[1] - __uint_as_float(f.h.z & 0xFFFF0000u);
        r1[2] = t1[2] - __uint_as_float(f.h.w << 16); r1[3] = t1[3] - __uint_as_float(f.h.w & 0xFFFF0000u);
        f.l = pack_acc<bf16_t>(r0, r1);
        return f;
    } else return Frag<T>{pack_acc<T>(t0, t1)};
}
template <typename T> __device__ __forceinline__ Frag<T> xtile_tr(const unsigned char* tile, int po, int r0, int c0, int lane) {
    if constexpr (std::is_same<T, x3_t>::value) return Frag<x3_t>{tile_tr<bf16_t>(tile, r0, c0, lane), tile_tr<bf16_t>(tile + po, r0, c0, lane)};
    else return Frag<T>{tile_tr<T>(tile, r0, c0, lane)};
}
template <typename T>
__device__ __forceinline__ void xload_tile(unsigned char* dst, int po, const unsigned char* src, size_t src_ld_bytes, int T_, int TP) {
    if constexpr (std::is_same<T, x3_t>::value) {
        for (int id = threadIdx.x; id < TP * 16; id += blockDim.x) {      // 16-byte chunks of four floats: chunk c of a row = k values 4c .. 4c + 3
            const int row = id >> 4, c = id & 15;
            u32x4 v = u32x4{0, 0, 0, 0};
            if (row < T_) v = *(const u32x4*)(src + (size_t)row * src_ld_bytes + c * 16);
            u32x2 hi, lo;
            split_bf16x3(v, hi, lo);
            const int o = toff<bf16_t>(row, c >> 1) + 8 * (c & 1);
            *(u32x2*)(dst + o) = hi;
            *(u32x2*)(dst + po + o) = lo;
        }
    } else load_tile<T>(dst, src, src_ld_bytes, T_, TP);
}
template <typename T>
__device__ __forceinline__ void xrow_frags(Frag<T>* f, const unsigned char* src, size_t src_ld_bytes, int row, bool valid, int g) {
    if constexpr (std::is_same<T, x3_t>::value) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {                                     // k values 32 k + 8 g .. + 7: two chunks of four floats
            const unsigned char* q = src + (size_t)row * src_ld_bytes + (32 * k + 8 * g) * 4;
            const u32x4 a = valid ? *(const u32x4*)q : u32x4{0, 0, 0, 0}, b = valid ? *(const u32x4*)(q + 16) : u32x4{0, 0, 0, 0};
            u32x2 ha, la, hb, lb;
            split_bf16x3(a, ha, la); split_bf16x3(b, hb, lb);
            f[k].h = u32x4{ha.x, ha.y, hb.x, hb.y};
            f[k].l = u32x4{la.x, la.y, lb.x, lb.y};
        }
    } else {
        u32x4 t[XGeo<T>::NKG];
        load_row_frags<T>(t, src, src_ld_bytes, row, valid, g);
#pragma unroll
        for (int k = 0; k < XGeo<T>::NKG; ++k) f[k].h = t[k];
    }
}
// this lane's row fragments from an LDS tile
template <typename T> __device__ __forceinline__ Frag<T> xtile_frag(const unsigned char* tile, int po, int row, int chunk) {
    if constexpr (std::is_same<T, x3_t>::value) return Frag<x3_t>{lds16(tile + toff<bf16_t>(row, chunk)), lds16(tile + po + toff<bf16_t>(row, chunk))};
    else return Frag<T>{lds16(tile + toff<T>(row, chunk))};
}
template <typename T>
__device__ __forceinline__ f32x4 xtile_dot(const unsigned char* tile, int po, int t, const Frag<T>* f, int i, int g) {
    f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < XGeo<T>::NKG; ++k) acc = xmma<T>(xtile_frag<T>(tile, po, 16 * t + i, k * 4 + g), f[k], acc);
    return acc;
}
// this lane's share of a . b over the head dims its fragments hold (delta = dO . O)
template <typename T> __device__ __forceinline__ float xrow_dot(const Frag<T>* a, const Frag<T>* b) {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < XGeo<T>::NKG; ++k) {
        if constexpr (std::is_same<T, x3_t>::value) {
            const bf16_t* ah = (const bf16_t*)&a[k].h; const bf16_t* al = (const bf16_t*)&a[k].l;
            const bf16_t* bh = (const bf16_t*)&b[k].h; const bf16_t* bl = (const bf16_t*)&b[k].l;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (to_f32(ah[e]) + to_f32(al[e])) * (to_f32(bh[e]) + to_f32(bl[e]));
        } else {
            const T* x = (const T*)&a[k].h; const T* y = (const T*)&b[k].h;
#pragma unroll
            for (int e = 0; e < Prec<T>::FRAG; ++e) s += to_f32(x[e]) * to_f32(y[e]);
        }
    }
    return s;
}
// keep bytes handed from phase A to phase B through LDS (16-bit geometry; fp32x3 up to 256 frames: at 288 the four planes fill the LDS)
template <typename T, int KT> constexpr bool use_mask() { return sizeof(T) == 2 || (std::is_same<T, x3_t>::value && KT <= 8); }
}  // namespace attn

// DROP is a COMPILE-TIME switch (the launcher tests drop.thr): as a run-time test of drop.thr inside the tile loops it split every
// tile (forward, phase A) or every ELEMENT (phase B: a mask read from LDS, its wait and a branch per probability) into a basic block
// of its own.
// ABL (development only, tools/attn_bwd_bench.hip): 1 no compute (tile loads + barrier only), 4 no exp, 8 no P.V
template <typename T, int KT, bool PAD = true, bool DROP = true, int SBE = 1, int ABL = 0>
__global__ void __launch_bounds__(512) attn_fwd_kernel(const AttnArgs p) {
    using G = attn::XGeo<T>;
    constexpr int TP = 32 * KT, NT16 = 2 * KT, KG = Prec<T>::KG, NG = TP / KG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Ks = smem;
    constexpr int PO = TP * G::LD;                            // fp32x3: offset of a tile's lo plane
    unsigned char* const Vs = smem + attn::xtile_bytes<T>(TP);
    const int n = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const size_t ldq = (size_t)3 * p.D * sizeof(T);
    const unsigned char* base = (const unsigned char*)p.qkv + (size_t)n * p.T * ldq + (size_t)h * 64 * sizeof(T);
    attn::xload_tile<T>(Ks, PO, base + (size_t)p.D * sizeof(T), ldq, p.T, TP);
    attn::xload_tile<T>(Vs, PO, base + (size_t)2 * p.D * sizeof(T), ldq, p.T, TP);
    __syncthreads();

    for (int qt = wave; qt * 16 < p.T && !(ABL & 1); qt += nw) {       // wave-uniform loop: EXEC stays full
        const int qrow = qt * 16 + i;
        const bool vq = !PAD || qrow < p.T;                // PAD == false: T is a multiple of 32, no masking anywhere
        attn::Frag<T> qf[G::NKG];
        attn::xrow_frags<T>(qf, base, ldq, qrow, vq, g);
        f32x4 s[NT16];
        float mx = -INFINITY;                              // maximum of the RAW scores (scale > 0)
#pragma unroll
        for (int t = 0; t < NT16; ++t) { const int sb_i = t;
            s[t] = attn::xtile_dot<T>(Ks, PO, t, qf, i, g);   // S^T[key 16t+4g+r][query qrow]
            if ((sb_i % SBE) == SBE - 1) __builtin_amdgcn_sched_barrier(0);   // keep live ranges per tile (VGPR 173 -> 91)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (PAD && (16 * t + 4 * g + r) >= p.T) s[t][r] = -INFINITY;
                mx = fmaxf(mx, s[t][r]);
            }
        }
        mx = cross4_max(mx);
        const float ck = p.scale * ExpK<T>::K, mk = mx * ck;
        float sum = 0.0f;
#pragma unroll
        for (int t = 0; t < NT16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = (ABL & 4) ? s[t][r] * ck - mk : ExpK<T>::ex(s[t][r] * ck - mk); s[t][r] = e; sum += e; }
        const float tot = cross4_sum(sum);
        const float inv = DROP ? p.drop.scale / tot : 1.0f / tot;       // normalisation and 1 / (1 - p) as ONE factor
        if (p.lse && g == 0 && vq) p.lse[((size_t)n * p.T + qrow) * p.H + h] = mx * p.scale + logf(tot);
        // dropout counter of P[query][key] = (head_row * T + query) * T4 + key, T4 = T rounded up to 4 (aligned quads)
        const uint32_t ibase = ((uint32_t)blockIdx.x * (uint32_t)p.T + (uint32_t)qrow) * (uint32_t)((p.T + 3) & ~3);
#pragma unroll
        for (int t = 0; t < NT16; ++t) {
            if constexpr (DROP) drop_scale4(p.drop, ibase + (uint32_t)(16 * t + 4 * g), s[t], inv);
            else s[t] *= inv;
        }

        f32x4 oacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int gi = 0; gi < ((ABL & 8) ? 1 : NG); ++gi) { const int sb_i = gi;
            const attn::Frag<T> pb = (KG == 32) ? attn::xpack<T>(s[(2 * gi) % NT16], s[(2 * gi + 1) % NT16]) : attn::xpack<T>(s[gi % NT16], s[gi % NT16]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)   // O^T[d = 16dt+4g+r][query] += V^T[d][keys] * P^T[keys][query]
                oacc[dt] = attn::xmma<T>(attn::xtile_tr<T>(Vs, PO, gi * KG, dt * 16, lane), pb, oacc[dt]);
            if ((sb_i % SBE) == SBE - 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (vq) {
            T* orow = (T*)p.o + ((size_t)n * p.T + qrow) * p.D + h * 64 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store4(orow + dt * 16, oacc[dt][0], oacc[dt][1], oacc[dt][2], oacc[dt][3]);
        }
    }
}

// Backward.  The forward saved lse = log sum_k exp(score) per (row, head), and delta_i = sum_k P_ik dP_ik equals
// dO_i . O_i (also with dropout, because O was formed from the dropped P), so no softmax reduction is redone:
// every 16x16 tile of P and dS is a pure function of its own scores.
//   Phase A (wave owns 16 queries; K, V in LDS): delta -> LDS, dQ^T += K^T dS^T.
//   Phase B (wave owns 16 keys;    Q, dO in LDS): dV^T += dO^T Pd, dK^T += Q^T dS.
// With dropout, dS = P (keep dP / (1 - p) - delta) scale is evaluated as P . fma(keep ? dP : 0, scale / (1 - p), - delta scale), and dV
// accumulates the UNSCALED kept probabilities (its 1 / (1 - p) is applied once to the finished accumulators).
// MINB: blocks per CU the register allocation must allow (launch bound).
// ABL (development only, tools/attn_bwd_bench.hip): 1 no phase B, 2 no phase A, 8 no exp
template <typename T, int KT, bool PAD = true, bool DROP = true, int SBE = 5, int ABL = 0, int MINB = 1>
__global__ void __launch_bounds__(512, MINB) attn_bwd_kernel(const AttnArgs p) {
    using G = attn::XGeo<T>;
    constexpr int TP = 32 * KT, KG = Prec<T>::KG, NG = TP / KG, TPG = KG / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const bufA = smem;
    constexpr int PO = TP * G::LD;                            // fp32x3: offset of a tile's lo plane
    unsigned char* const bufB = smem + attn::xtile_bytes<T>(TP);
    float* const st_l = (float*)(smem + 2 * attn::xtile_bytes<T>(TP));     // lse (times ExpK) per query of this head
    float* const st_d = st_l + TP;                            // delta * scale per query
    // The 16-bit modes hash the dropout keep bits once (phase A) and hand them to phase B through LDS; fp32 mode (parity
    // path; its K/V tiles already fill the LDS at 288 frames) re-hashes in phase B instead.  One BYTE per (4 keys, query), key
    // quad major: phase A's lane (keys 16t + 4g .. + 3, query) stores its nibble as one byte, no atomics and no clearing; phase B's
    // lane (queries 16t + 4g .. + 3, key) finds its four bytes in ONE word.
    constexpr bool USE_MASK = attn::use_mask<T, KT>();
    unsigned char* const st_m = (unsigned char*)(st_d + TP);  // [TP / 4][TP]
    const int n = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const size_t ldq = (size_t)3 * p.D * sizeof(T), ldo = (size_t)p.D * sizeof(T);
    const unsigned char* qbase = (const unsigned char*)p.qkv + (size_t)n * p.T * ldq + (size_t)h * 64 * sizeof(T);
    const unsigned char* kbase = qbase + (size_t)p.D * sizeof(T);
    const unsigned char* vbase = qbase + (size_t)2 * p.D * sizeof(T);
    const unsigned char* dobase = (const unsigned char*)p.dout + (size_t)n * p.T * ldo + (size_t)h * 64 * sizeof(T);
    const unsigned char* obase = (const unsigned char*)p.o + (size_t)n * p.T * ldo + (size_t)h * 64 * sizeof(T);
    T* const dq_out = (T*)p.dqkv + (size_t)n * p.T * 3 * p.D + h * 64;
    const uint32_t hbase = (uint32_t)blockIdx.x * (uint32_t)p.T;
    const uint32_t T4 = (uint32_t)((p.T + 3) & ~3);

    // ---------------------------------------------------------------- phase A
    attn::xload_tile<T>(bufA, PO, kbase, ldq, p.T, TP);
    attn::xload_tile<T>(bufB, PO, vbase, ldq, p.T, TP);
    for (int q = threadIdx.x; q < TP; q += blockDim.x) {
        st_l[q] = q < p.T ? p.lse[((size_t)n * p.T + q) * p.H + h] * ExpK<T>::K : 0.0f;
        st_d[q] = 0.0f;          // padded queries: phase B multiplies (dP - delta) by P = 0, so delta must be finite
    }
    const float ck = p.scale * ExpK<T>::K;
    const float sds = DROP ? p.scale * p.drop.scale : p.scale;
    __syncthreads();
    for (int qt = wave; qt * 16 < p.T && !(ABL & 2); qt += nw) {
        const int qrow = qt * 16 + i;
        const bool vq = !PAD || qrow < p.T;
        attn::Frag<T> qf[G::NKG], dof[G::NKG];
        attn::xrow_frags<T>(qf, qbase, ldq, qrow, vq, g);
        attn::xrow_frags<T>(dof, dobase, ldo, qrow, vq, g);
        float delta = 0.0f;
        {
            attn::Frag<T> of[G::NKG];
            attn::xrow_frags<T>(of, obase, ldo, qrow, vq, g);
            delta = attn::xrow_dot<T>(dof, of);
        }
        const float dsc = cross4_sum(delta) * p.scale;
        if (g == 0 && vq) st_d[qrow] = dsc;
        const float lse = st_l[vq ? qrow : 0];
        const uint32_t ibase = (hbase + (uint32_t)qrow) * T4;
        f32x4 qacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) qacc[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int sb_i = gi;
            f32x4 ds[2];
#pragma unroll
            for (int u = 0; u < TPG; ++u) {
                const int t = gi * TPG + u;
                const f32x4 sa = attn::xtile_dot<T>(bufA, PO, t, qf, i, g);     // S^T[key 16t+4g+r][query]
                f32x4 dp = attn::xtile_dot<T>(bufB, PO, t, dof, i, g);          // d(P dropped)^T[key][query]
                if constexpr (DROP) {      // keep bits are hashed once, here; phase B reads them back from LDS
                    const uint32_t m = drop_select4(p.drop, ibase + (uint32_t)(16 * t + 4 * g), dp);
                    if constexpr (USE_MASK) st_m[(4 * t + g) * TP + qrow] = (unsigned char)m;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = (ABL & 8) ? sa[r] * ck - lse : ExpK<T>::ex(sa[r] * ck - lse);
                    if (PAD && (16 * t + 4 * g + r) >= p.T) pr = 0.0f;
                    ds[u][r] = pr * (dp[r] * sds - dsc);
                }
            }
            const attn::Frag<T> sb = attn::xpack<T>(ds[0], ds[TPG - 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)   // dQ^T[d][query] += K^T[d][keys] * dS^T[keys][query]
                qacc[dt] = attn::xmma<T>(attn::xtile_tr<T>(bufA, PO, gi * KG, dt * 16, lane), sb, qacc[dt]);
            if ((sb_i % SBE) == SBE - 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (vq) {
            T* row = dq_out + (size_t)qrow * 3 * p.D + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store4(row + dt * 16, qacc[dt][0], qacc[dt][1], qacc[dt][2], qacc[dt][3]);
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------- phase B
    attn::xload_tile<T>(bufA, PO, qbase, ldq, p.T, TP);
    attn::xload_tile<T>(bufB, PO, dobase, ldo, p.T, TP);
    __syncthreads();
    for (int kt = wave; kt * 16 < p.T && !(ABL & 1); kt += nw) {
        const int krow = kt * 16 + i;
        const bool vk = !PAD || krow < p.T;
        attn::Frag<T> kf[G::NKG], vf[G::NKG];
        attn::xrow_frags<T>(kf, kbase, ldq, krow, vk, g);
        attn::xrow_frags<T>(vf, vbase, ldq, krow, vk, g);
        f32x4 kacc[4], vacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { kacc[dt] = f32x4{0, 0, 0, 0}; vacc[dt] = f32x4{0, 0, 0, 0}; }
        // this key's keep bits of queries 16t + 4g + r: byte r of word [krow / 4][16t + 4g], bit krow % 4
        [[maybe_unused]] const unsigned char* const mrow = st_m + (krow >> 2) * TP + 4 * g;
        [[maybe_unused]] const int mbit = krow & 3;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int sb_i = gi;
            f32x4 pd[2], ds[2];
#pragma unroll
            for (int u = 0; u < TPG; ++u) {
                const int t = gi * TPG + u;
                const f32x4 sa = attn::xtile_dot<T>(bufA, PO, t, kf, i, g);    // S[query 16t+4g+r][key krow]
                const f32x4 da = attn::xtile_dot<T>(bufB, PO, t, vf, i, g);    // d(P dropped)[query][key]
                const f32x4 l4 = *(const f32x4*)(st_l + 16 * t + 4 * g);
                const f32x4 d4 = *(const f32x4*)(st_d + 16 * t + 4 * g);
                [[maybe_unused]] f32x4 keep4 = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
                if constexpr (DROP && USE_MASK) keep4 = ubytes_to_f32x4((*(const uint32_t*)(mrow + 16 * t) >> mbit) & 0x01010101u);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = 16 * t + 4 * g + r;
                    float pr = (ABL & 8) ? sa[r] * ck - l4[r] : ExpK<T>::ex(sa[r] * ck - l4[r]);
                    if (PAD && !((q < p.T) && vk)) pr = 0.0f;
                    float dv = da[r];
                    pd[u][r] = pr;
                    if constexpr (DROP) {
                        // keep as a 0.0 / 1.0 factor (v_cvt_f32_ubyte<r>): one conversion and two multiplies per element
                        const float keep = USE_MASK ? keep4[r]
                                                    : (drop_keep((hbase + (uint32_t)q) * T4 + (uint32_t)krow, p.drop.key, p.drop.thr) ? 1.0f : 0.0f);
                        dv *= keep;
                        pd[u][r] = pr * keep;
                    }
                    ds[u][r] = (PAD && pr == 0.0f) ? 0.0f : pr * (dv * sds - d4[r]);
                }
            }
            const attn::Frag<T> pb = attn::xpack<T>(pd[0], pd[TPG - 1]);
            const attn::Frag<T> sb = attn::xpack<T>(ds[0], ds[TPG - 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                vacc[dt] = attn::xmma<T>(attn::xtile_tr<T>(bufB, PO, gi * KG, dt * 16, lane), pb, vacc[dt]);  // dV^T += dO^T Pd
                kacc[dt] = attn::xmma<T>(attn::xtile_tr<T>(bufA, PO, gi * KG, dt * 16, lane), sb, kacc[dt]);  // dK^T += Q^T dS
            }
            if ((sb_i % SBE) == SBE - 1) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (DROP) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) vacc[dt] *= p.drop.scale;
        }
        if (vk) {
            T* row = dq_out + (size_t)krow * 3 * p.D + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                store4(row + p.D + dt * 16, kacc[dt][0], kacc[dt][1], kacc[dt][2], kacc[dt][3]);
                store4(row + 2 * p.D + dt * 16, vacc[dt][0], vacc[dt][1], vacc[dt][2], vacc[dt][3]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Long sequences (288 < T <= Max_Position 1024; reference Modules.py:107-109 slices pe[:, :, :T] for any T up to Max_Position).
// The kernels above keep a head's whole K / V (or Q / dO) sequence in LDS and a query's whole score row in registers; beyond 288
// frames neither fits.  These stream the other side through LDS in chunks of 64 rows instead:
//   attn_fwd_long_kernel   : block = 64 queries (4 waves x 16) of one (utterance, head); K / V chunks; online softmax
//                            (running raw maximum m, per-lane partial sum l, O rescaled by exp((m_old - m_new) scale)); the
//                            dropout acts on the normalised probabilities, so the unnormalised e . keep / (1 - p) is
//                            accumulated and the division by the final sum comes last (linear).  Writes lse.
//   attn_bwd_long_dq_kernel: block = 64 queries; phase A of attn_bwd_kernel over K / V chunks; also writes delta = dO . O.
//   attn_bwd_long_dkv_kernel: block = 64 keys; phase B over Q / dO chunks (lse, delta per chunk in LDS); dropout keep bits re-hashed.
// Same dropout counters as the resident kernels: ((n H + h) T + query) T4 + key.  Functional path (the reference trains on <= 270
// frames and infers on 64 / 240): not tuned.
// ---------------------------------------------------------------------------------------------
constexpr int ATT_LC = 64;       // chunk rows

template <typename T>
__global__ void __launch_bounds__(256) attn_fwd_long_kernel(const AttnArgs p) {
    using G = attn::Geo<T>;
    constexpr int NT16 = ATT_LC / 16, KG = Prec<T>::KG, NG = ATT_LC / KG;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[ATT_LC * G::LD];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[ATT_LC * G::LD];
    const int bx = blockIdx.x, n = bx / p.H, h = bx % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const size_t ldq = (size_t)3 * p.D * sizeof(T);
    const unsigned char* base = (const unsigned char*)p.qkv + (size_t)n * p.T * ldq + (size_t)h * 64 * sizeof(T);
    const int qrow = blockIdx.y * ATT_LC + 16 * wave + i;
    const bool vq = qrow < p.T;
    u32x4 qf[G::NKG];
    attn::load_row_frags<T>(qf, base, ldq, qrow, vq, g);
    const float ck = p.scale * ExpK<T>::K;
    const uint32_t ibase = ((uint32_t)bx * (uint32_t)p.T + (uint32_t)qrow) * (uint32_t)((p.T + 3) & ~3);
    float m = -INFINITY, lsum = 0.0f;
    f32x4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0, 0, 0, 0};
    for (int c0 = 0; c0 < p.T; c0 += ATT_LC) {
        __syncthreads();                                   // everyone is done with the previous chunk
        attn::load_tile<T>(Ks, base + (size_t)p.D * sizeof(T) + (size_t)c0 * ldq, ldq, p.T - c0, ATT_LC);
        attn::load_tile<T>(Vs, base + (size_t)2 * p.D * sizeof(T) + (size_t)c0 * ldq, ldq, p.T - c0, ATT_LC);
        __syncthreads();
        f32x4 s[NT16];
        float cmax = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT16; ++t) {
            s[t] = attn::tile_dot<T>(Ks, t, qf, i, g);    // S^T[key c0 + 16t + 4g + r][query]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (c0 + 16 * t + 4 * g + r >= p.T) s[t][r] = -INFINITY;
                cmax = fmaxf(cmax, s[t][r]);
            }
        }
        cmax = cross4_max(cmax);
        const float mnew = fmaxf(m, cmax);                 // finite: every chunk holds at least one real key
        const float alpha = ExpK<T>::ex((m - mnew) * ck);  // first chunk: exp(-inf) = 0
        const float mk = mnew * ck;
        float sum = 0.0f;
#pragma unroll
        for (int t = 0; t < NT16; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = ExpK<T>::ex(s[t][r] * ck - mk); s[t][r] = e; sum += e; }
            drop_apply4(p.drop, ibase + (uint32_t)(c0 + 16 * t + 4 * g), s[t]);
        }
        lsum = lsum * alpha + sum;
        m = mnew;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[dt] *= alpha;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const u32x4 pb = (KG == 32) ? pack_acc<T>(s[(2 * gi) % NT16], s[(2 * gi + 1) % NT16]) : pack_acc<T>(s[gi % NT16], s[gi % NT16]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                oacc[dt] = mma16<T>(attn::tile_tr<T>(Vs, gi * KG, dt * 16, lane), pb, oacc[dt]);
        }
    }
    const float tot = cross4_sum(lsum);
    const float inv = 1.0f / tot;
    if (vq) {
        if (p.lse && g == 0) p.lse[((size_t)n * p.T + qrow) * p.H + h] = m * p.scale + logf(tot);
        T* orow = (T*)p.o + ((size_t)n * p.T + qrow) * p.D + h * 64 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4(orow + dt * 16, oacc[dt][0] * inv, oacc[dt][1] * inv, oacc[dt][2] * inv, oacc[dt][3] * inv);
    }
}

struct AttnLongBwd { float* delta; };     // [R, H] fp32: dO . O per (row, head), written by the dQ kernel, read by the dK / dV kernel

template <typename T>
__global__ void __launch_bounds__(256) attn_bwd_long_dq_kernel(const AttnArgs p, const AttnLongBwd x) {
    using G = attn::Geo<T>;
    constexpr int KG = Prec<T>::KG, NG = ATT_LC / KG, TPG = KG / 16;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[ATT_LC * G::LD];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[ATT_LC * G::LD];
    const int bx = blockIdx.x, n = bx / p.H, h = bx % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const size_t ldq = (size_t)3 * p.D * sizeof(T), ldo = (size_t)p.D * sizeof(T);
    const unsigned char* qbase = (const unsigned char*)p.qkv + (size_t)n * p.T * ldq + (size_t)h * 64 * sizeof(T);
    const unsigned char* dobase = (const unsigned char*)p.dout + (size_t)n * p.T * ldo + (size_t)h * 64 * sizeof(T);
    const unsigned char* obase = (const unsigned char*)p.o + (size_t)n * p.T * ldo + (size_t)h * 64 * sizeof(T);
    const int qrow = blockIdx.y * ATT_LC + 16 * wave + i;
    const bool vq = qrow < p.T;
    u32x4 qf[G::NKG], dof[G::NKG];
    attn::load_row_frags<T>(qf, qbase, ldq, qrow, vq, g);
    attn::load_row_frags<T>(dof, dobase, ldo, qrow, vq, g);
    float delta = 0.0f;
    {
        u32x4 of[G::NKG];
        attn::load_row_frags<T>(of, obase, ldo, qrow, vq, g);
#pragma unroll
        for (int k = 0; k < G::NKG; ++k) {
            const T* a = (const T*)&dof[k];
            const T* b = (const T*)&of[k];
#pragma unroll
            for (int e = 0; e < Prec<T>::FRAG; ++e) delta += to_f32(a[e]) * to_f32(b[e]);
        }
    }
    delta = cross4_sum(delta);
    const size_t sidx = ((size_t)n * p.T + (vq ? qrow : 0)) * p.H + h;
    if (vq && g == 0) x.delta[sidx] = delta;
    const float lse = vq ? p.lse[sidx] * ExpK<T>::K : 0.0f;
    const float ck = p.scale * ExpK<T>::K;
    const uint32_t ibase = ((uint32_t)bx * (uint32_t)p.T + (uint32_t)qrow) * (uint32_t)((p.T + 3) & ~3);
    const bool dropping = p.drop.thr != 0;
    f32x4 qacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) qacc[dt] = f32x4{0, 0, 0, 0};
    for (int c0 = 0; c0 < p.T; c0 += ATT_LC) {
        __syncthreads();
        attn::load_tile<T>(Ks, qbase + (size_t)p.D * sizeof(T) + (size_t)c0 * ldq, ldq, p.T - c0, ATT_LC);
        attn::load_tile<T>(Vs, qbase + (size_t)2 * p.D * sizeof(T) + (size_t)c0 * ldq, ldq, p.T - c0, ATT_LC);
        __syncthreads();
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            f32x4 ds[2];
#pragma unroll
            for (int u = 0; u < TPG; ++u) {
                const int t = gi * TPG + u;
                const f32x4 sa = attn::tile_dot<T>(Ks, t, qf, i, g);
                f32x4 dp = attn::tile_dot<T>(Vs, t, dof, i, g);
                if (dropping) {
                    const uint32_t mk = drop_mask4(p.drop, ibase + (uint32_t)(c0 + 16 * t + 4 * g));
#pragma unroll
                    for (int r = 0; r < 4; ++r) dp[r] = ((mk >> r) & 1u) ? dp[r] * p.drop.scale : 0.0f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = ExpK<T>::ex(sa[r] * ck - lse);
                    if (c0 + 16 * t + 4 * g + r >= p.T || !vq) pr = 0.0f;
                    ds[u][r] = pr * (dp[r] - delta) * p.scale;
                }
            }
            const u32x4 sb = pack_acc<T>(ds[0], ds[TPG - 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                qacc[dt] = mma16<T>(attn::tile_tr<T>(Ks, gi * KG, dt * 16, lane), sb, qacc[dt]);
        }
    }
    if (vq) {
        T* row = (T*)p.dqkv + ((size_t)n * p.T + qrow) * 3 * p.D + h * 64 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4(row + dt * 16, qacc[dt][0], qacc[dt][1], qacc[dt][2], qacc[dt][3]);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) attn_bwd_long_dkv_kernel(const AttnArgs p, const AttnLongBwd x) {
    using G = attn::Geo<T>;
    constexpr int KG = Prec<T>::KG, NG = ATT_LC / KG, TPG = KG / 16;
    __shared__ __attribute__((aligned(16))) unsigned char Qs[ATT_LC * G::LD];
    __shared__ __attribute__((aligned(16))) unsigned char Ds[ATT_LC * G::LD];
    __shared__ __attribute__((aligned(16))) float st_l[ATT_LC];
    __shared__ __attribute__((aligned(16))) float st_d[ATT_LC];
    const int bx = blockIdx.x, n = bx / p.H, h = bx % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const size_t ldq = (size_t)3 * p.D * sizeof(T), ldo = (size_t)p.D * sizeof(T);
    const unsigned char* qbase = (const unsigned char*)p.qkv + (size_t)n * p.T * ldq + (size_t)h * 64 * sizeof(T);
    const unsigned char* dobase = (const unsigned char*)p.dout + (size_t)n * p.T * ldo + (size_t)h * 64 * sizeof(T);
    const int krow = blockIdx.y * ATT_LC + 16 * wave + i;
    const bool vk = krow < p.T;
    u32x4 kf[G::NKG], vf[G::NKG];
    attn::load_row_frags<T>(kf, qbase + (size_t)p.D * sizeof(T), ldq, krow, vk, g);
    attn::load_row_frags<T>(vf, qbase + (size_t)2 * p.D * sizeof(T), ldq, krow, vk, g);
    const float ck = p.scale * ExpK<T>::K;
    const uint32_t hbase = (uint32_t)bx * (uint32_t)p.T, T4 = (uint32_t)((p.T + 3) & ~3);
    const bool dropping = p.drop.thr != 0;
    f32x4 kacc[4], vacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { kacc[dt] = f32x4{0, 0, 0, 0}; vacc[dt] = f32x4{0, 0, 0, 0}; }
    for (int c0 = 0; c0 < p.T; c0 += ATT_LC) {
        __syncthreads();
        attn::load_tile<T>(Qs, qbase + (size_t)c0 * ldq, ldq, p.T - c0, ATT_LC);
        attn::load_tile<T>(Ds, dobase + (size_t)c0 * ldo, ldo, p.T - c0, ATT_LC);
        if (threadIdx.x < ATT_LC) {
            const int q = c0 + threadIdx.x;
            const size_t sidx = ((size_t)n * p.T + (q < p.T ? q : 0)) * p.H + h;
            st_l[threadIdx.x] = q < p.T ? p.lse[sidx] * ExpK<T>::K : 0.0f;
            st_d[threadIdx.x] = q < p.T ? x.delta[sidx] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            f32x4 pd[2], ds[2];
#pragma unroll
            for (int u = 0; u < TPG; ++u) {
                const int t = gi * TPG + u;
                const f32x4 sa = attn::tile_dot<T>(Qs, t, kf, i, g);     // S[query c0 + 16t + 4g + r][key krow]
                const f32x4 da = attn::tile_dot<T>(Ds, t, vf, i, g);
                const f32x4 l4 = *(const f32x4*)(st_l + 16 * t + 4 * g);
                const f32x4 d4 = *(const f32x4*)(st_d + 16 * t + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = c0 + 16 * t + 4 * g + r;
                    float pr = ExpK<T>::ex(sa[r] * ck - l4[r]);
                    if (!(q < p.T && vk)) pr = 0.0f;
                    float dv = da[r];
                    pd[u][r] = pr;
                    if (dropping) {
                        const bool keep = drop_keep((hbase + (uint32_t)q) * T4 + (uint32_t)krow, p.drop.key, p.drop.thr);
                        dv = keep ? dv * p.drop.scale : 0.0f;
                        pd[u][r] = keep ? pr * p.drop.scale : 0.0f;
                    }
                    ds[u][r] = pr == 0.0f ? 0.0f : pr * (dv - d4[r]) * p.scale;
                }
            }
            const u32x4 pb = pack_acc<T>(pd[0], pd[TPG - 1]);
            const u32x4 sb = pack_acc<T>(ds[0], ds[TPG - 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                vacc[dt] = mma16<T>(attn::tile_tr<T>(Ds, gi * KG, dt * 16, lane), pb, vacc[dt]);
                kacc[dt] = mma16<T>(attn::tile_tr<T>(Qs, gi * KG, dt * 16, lane), sb, kacc[dt]);
            }
        }
    }
    if (vk) {
        T* row = (T*)p.dqkv + ((size_t)n * p.T + krow) * 3 * p.D + h * 64 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            store4(row + p.D + dt * 16, kacc[dt][0], kacc[dt][1], kacc[dt][2], kacc[dt][3]);
            store4(row + 2 * p.D + dt * 16, vacc[dt][0], vacc[dt][1], vacc[dt][2], vacc[dt][3]);
        }
    }
}

// (The last layer's single-query attention lives in attn_last.cuh: it needs no K / V projection at all.)

}  // namespace ge2e
