// Shared device helpers for the GE2E hot-path kernels (gfx950 / CDNA4 only).
//
// Three arithmetic modes share every kernel template:
//   T = float  : v_mfma_f32_16x16x4_f32   (exact fp32 fma chain; the <=1e-4 parity path)
//   T = bf16_t : v_mfma_f32_16x16x32_bf16 (bf16 storage, fp32 accumulate; the throughput path)
//   T = f16_t  : v_mfma_f32_16x16x32_f16  (IEEE half storage, fp32 accumulate: the reference's own autocast dtype,
//                Train.py:145; 3 more mantissa bits than bf16, 5 exponent bits -> gradients need loss scaling)
// An MFMA operand "fragment" is always 16 bytes per lane (4 f32 / 8 x 16-bit); one "k-group" is the
// 64 bytes of K that the four 16-lane groups of a wave cover together (16 f32 / 32 16-bit values).
// So LDS tiles have the same BYTE layout in all modes and only `mma16` and the conversions differ.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ge2e {

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

template <typename T> struct Prec;
template <> struct Prec<float> {
    static constexpr int FRAG = 4;   // elements per 16-byte fragment
    static constexpr int KG = 16;    // k values per k-group
};
template <> struct Prec<bf16_t> {
    static constexpr int FRAG = 8;
    static constexpr int KG = 32;
};
template <> struct Prec<f16_t> {
    static constexpr int FRAG = 8;
    static constexpr int KG = 32;
};
struct x3_t;
template <> struct Prec<x3_t> {      // fp32x3: the operands reach the matrix pipe as bf16 halves, 32 k values per MFMA
    static constexpr int FRAG = 8;
    static constexpr int KG = 32;
};

// ---------------------------------------------------------------------------------------------
// mma16: C[16x16] += A[16 x KG] * B[KG x 16].   Lane l = 16*g + i supplies A[row i][slots of g]
// and B[slots of g][col i]; returns/accumulates acc[r] = C[row 4g + r][col i].
// Slot -> k mapping is free as long as A and B agree (see frag_* helpers below).
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ f32x4 mma16(u32x4 a, u32x4 b, f32x4 c);

template <> __device__ __forceinline__ f32x4 mma16<float>(u32x4 a, u32x4 b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
}
template <> __device__ __forceinline__ f32x4 mma16<bf16_t>(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

template <> __device__ __forceinline__ f32x4 mma16<f16_t>(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                  __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// "fp32x3": fp32 STORAGE, products on the bf16 matrix pipe (round 4).  x = hi + lo with hi = bf16(x) and lo = bf16(x - hi) (the difference is
// exact in fp32), so a.b = ah.bh + ah.bl + al.bh up to the dropped al.bl term: ~2^-17 relative per product (8 + 8 mantissa bits per
// operand) where an fp32 fma chain has 2^-24 -- three v_mfma_f32_16x16x32_bf16 (48 matrix-pipe cycles per 16 x 16 x 32) instead of eight
// v_mfma_f32_16x16x4_f32 (256).  The split happens ONCE per element and tile, on the way from global memory into the LDS stage (gemm.cuh),
// never per MFMA use; accumulation stays fp32.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void split_bf16x3(u32x4 v, u32x2& hi, u32x2& lo) {
    const float x0 = __uint_as_float(v.x), x1 = __uint_as_float(v.y), x2 = __uint_as_float(v.z), x3 = __uint_as_float(v.w);
    hi.x = __builtin_bit_cast(unsigned, __builtin_convertvector((__attribute__((ext_vector_type(2))) float){x0, x1}, __attribute__((ext_vector_type(2))) __bf16));
    hi.y = __builtin_bit_cast(unsigned, __builtin_convertvector((__attribute__((ext_vector_type(2))) float){x2, x3}, __attribute__((ext_vector_type(2))) __bf16));
    const float r0 = x0 - __uint_as_float(hi.x << 16), r1 = x1 - __uint_as_float(hi.x & 0xFFFF0000u);
    const float r2 = x2 - __uint_as_float(hi.y << 16), r3 = x3 - __uint_as_float(hi.y & 0xFFFF0000u);
    lo.x = __builtin_bit_cast(unsigned, __builtin_convertvector((__attribute__((ext_vector_type(2))) float){r0, r1}, __attribute__((ext_vector_type(2))) __bf16));
    lo.y = __builtin_bit_cast(unsigned, __builtin_convertvector((__attribute__((ext_vector_type(2))) float){r2, r3}, __attribute__((ext_vector_type(2))) __bf16));
}
// Tag type of the fp32x3 mode for kernels whose LDS tiles and fragments change shape with it (attention.cuh): storage is a float.
struct x3_t { float v; };
// C += A.B from the split halves (small terms first)
__device__ __forceinline__ f32x4 mma16_x3(u32x4 ah, u32x4 al, u32x4 bh, u32x4 bl, f32x4 c) {
    c = mma16<bf16_t>(al, bh, c);
    c = mma16<bf16_t>(ah, bl, c);
    return mma16<bf16_t>(ah, bh, c);
}

// ---------------------------------------------------------------------------------------------
// scalar conversions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)v; }     // round to nearest even; overflow -> inf (the loss scaler's cue)

// two floats -> one packed pair, ONE instruction (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, round to nearest even as the scalar casts).
// Written as two scalar casts + shift + or, the compiler emits two conversions, a shift and an or: four instructions per pair in every
// epilogue that packs accumulators.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

__device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, f16x2_t));
}

template <typename T> __device__ __forceinline__ unsigned pack2(float lo, float hi);
template <> __device__ __forceinline__ unsigned pack2<bf16_t>(float lo, float hi) { return pack_bf16x2(lo, hi); }
template <> __device__ __forceinline__ unsigned pack2<f16_t>(float lo, float hi) { return pack_f16x2(lo, hi); }
// elements e0 .. e0 + 3 (e0 % 4 == 0) of a 16-byte fragment of T from 4 floats
template <typename T> __device__ __forceinline__ void frag_put4(u32x4& f, int e0, f32x4 v) {
    if constexpr (sizeof(T) == 2) { f[e0 >> 1] = pack2<T>(v[0], v[1]); f[(e0 >> 1) + 1] = pack2<T>(v[2], v[3]); }
    else { f[0] = __float_as_uint(v[0]); f[1] = __float_as_uint(v[1]); f[2] = __float_as_uint(v[2]); f[3] = __float_as_uint(v[3]); }
}

// pack two accumulator tiles (bf16) / one tile (f32) into an MFMA operand fragment ("acc mapping":
// slot j<4 <-> row 4g+j of tile 0, slot 4+j <-> row 4g+j of tile 1 (bf16); slot s <-> row 4g+s (f32)).
template <typename T> __device__ __forceinline__ u32x4 pack_acc(f32x4 t0, f32x4 t1);
template <> __device__ __forceinline__ u32x4 pack_acc<float>(f32x4 t0, f32x4) {
    u32x4 r;
    r.x = __float_as_uint(t0[0]); r.y = __float_as_uint(t0[1]);
    r.z = __float_as_uint(t0[2]); r.w = __float_as_uint(t0[3]);
    return r;
}
template <> __device__ __forceinline__ u32x4 pack_acc<bf16_t>(f32x4 t0, f32x4 t1) {
    u32x4 r;
    r.x = pack_bf16x2(t0[0], t0[1]); r.y = pack_bf16x2(t0[2], t0[3]);
    r.z = pack_bf16x2(t1[0], t1[1]); r.w = pack_bf16x2(t1[2], t1[3]);
    return r;
}

template <> __device__ __forceinline__ u32x4 pack_acc<f16_t>(f32x4 t0, f32x4 t1) {
    u32x4 r;
    r.x = pack_f16x2(t0[0], t0[1]); r.y = pack_f16x2(t0[2], t0[3]);
    r.z = pack_f16x2(t1[0], t1[1]); r.w = pack_f16x2(t1[2], t1[3]);
    return r;
}

// 4 consecutive output elements of type T from 4 floats (8 B for bf16, 16 B for f32)
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    *(f32x4*)p = f32x4{a, b, c, d};
}
__device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) {
    u32x2 v; v.x = pack_bf16x2(a, b); v.y = pack_bf16x2(c, d);
    *(u32x2*)p = v;
}
__device__ __forceinline__ void store4(f16_t* p, float a, float b, float c, float d) {
    u32x2 v; v.x = pack_f16x2(a, b); v.y = pack_f16x2(c, d);
    *(u32x2*)p = v;
}
__device__ __forceinline__ void store4(x3_t* p, float a, float b, float c, float d) { *(f32x4*)p = f32x4{a, b, c, d}; }
__device__ __forceinline__ f32x4 load4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 load4(const f16_t* p) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
    const f16x4_t v = *(const f16x4_t*)p;
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ f32x4 load4(const bf16_t* p) {
    u32x2 v = *(const u32x2*)p;
    f32x4 r;
    r[0] = __uint_as_float(v.x << 16); r[1] = __uint_as_float(v.x & 0xFFFF0000u);
    r[2] = __uint_as_float(v.y << 16); r[3] = __uint_as_float(v.y & 0xFFFF0000u);
    return r;
}

// ---------------------------------------------------------------------------------------------
// LDS tiles.  "swizzled row tile": rows of ROWB bytes, 16-byte chunk c of row r stored at chunk
// c ^ (r & 7) (ROWB == 128) or c ^ (r & 15) (ROWB % 256 == 0): conflict-free ds_read_b128 for the
// MFMA row-fragment pattern (lane -> row i, chunk g).  "padded tile": plain rows with a 16-byte pad,
// used where a tile is (also) read transposed.
// ---------------------------------------------------------------------------------------------
template <int ROWB> __device__ __forceinline__ int swz_off(int row, int chunk) {
    static_assert(ROWB == 128 || ROWB % 256 == 0, "row bytes");
    if constexpr (ROWB == 128) return row * ROWB + ((chunk ^ (row & 7)) << 4);
    else return row * ROWB + ((chunk ^ (row & 15)) << 4);
}
__device__ __forceinline__ u32x4 lds16(const unsigned char* p) { return *(const u32x4*)p; }

// transposed fragment ("acc mapping") from a plain row-major tile X[r][c] of T, row stride ld bytes:
// lane (i, g) gets the values X[r0 + 4g + j][c0 + i] (j = 0..3) and, for bf16, X[r0 + 16 + 4g + j][c0 + i].
template <typename T> __device__ __forceinline__ u32x4 frag_tr(const unsigned char* tile, int ld, int r0, int c0, int lane);
template <> __device__ __forceinline__ u32x4 frag_tr<float>(const unsigned char* tile, int ld, int r0, int c0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const unsigned char* p = tile + (r0 + 4 * g) * ld + (c0 + i) * 4;
    u32x4 r;
    r.x = *(const unsigned*)(p);
    r.y = *(const unsigned*)(p + ld);
    r.z = *(const unsigned*)(p + 2 * ld);
    r.w = *(const unsigned*)(p + 3 * ld);
    return r;
}
__device__ __forceinline__ u32x4 frag_tr16(const unsigned char* tile, int ld, int r0, int c0, int lane) {
    // ds_read_b64_tr_b16: within a 16-lane group, lane 4q+p supplies the address of (row q, cols 4p..4p+3)
    // of a 4x16 block and receives column (4q+p) of the 4 rows.  EXEC must be all ones.
    const int i = lane & 15, g = lane >> 4;
    const unsigned char* p = tile + (r0 + 4 * g + (i >> 2)) * ld + (c0 + 4 * (i & 3)) * 2;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * ld));
    u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    return u32x4{l2.x, l2.y, h2.x, h2.y};
}
template <> __device__ __forceinline__ u32x4 frag_tr<bf16_t>(const unsigned char* tile, int ld, int r0, int c0, int lane) { return frag_tr16(tile, ld, r0, c0, lane); }
template <> __device__ __forceinline__ u32x4 frag_tr<f16_t>(const unsigned char* tile, int ld, int r0, int c0, int lane) { return frag_tr16(tile, ld, r0, c0, lane); }

// ---------------------------------------------------------------------------------------------
// counter-based dropout, bit-identical to oracle/ge2e_oracle.py: drop_keep.  ONE multiplicative hash serves the four elements of
// an aligned index quad (its two v_mul_lo_u32 are quarter rate: the hashes are what dropout costs, and the kernels that apply it are
// bound by their vector instructions); the quad's second word is one xorshift32 step of the first (six full-rate instructions):
//   w0 = mix32((idx >> 2) ^ key), w1 = xs32(w0);  word = idx & 2 ? w1 : w0;  field = idx & 1 ? word >> 16 : word & 0xFFFF;
//   keep(idx) = field >= thr,  thr = floor(p * 65536)
// (rounds 1-2 hashed every index PAIR; keep rates, the 16 joint keep patterns of a quad and the lag correlations between quads are
// within sampling noise of independent draws over 2^24 quads: tests/test_host_glue.py)
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t xs32(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
__host__ __device__ __forceinline__ bool drop_keep(uint32_t idx, uint32_t key, uint32_t thr) {
    uint32_t w = mix32((idx >> 2) ^ key);
    if (idx & 2u) w = xs32(w);
    return ((idx & 1u) ? (w >> 16) : (w & 0xFFFFu)) >= thr;
}
struct Drop {            // thr == 0  <=>  dropout inactive (eval mode or p == 0)
    uint32_t key, thr;
    float scale;         // 1 / (1 - p)
};
__device__ __forceinline__ float drop_apply(const Drop& d, uint32_t idx, float v) {
    if (d.thr == 0) return v;
    return drop_keep(idx, d.key, d.thr) ? v * d.scale : 0.0f;
}
// keep-mask (bit r <-> element base + r) of 4 consecutive elements, base % 4 == 0
__device__ __forceinline__ uint32_t drop_mask4(const Drop& d, uint32_t base) {
    const uint32_t h0 = mix32((base >> 2) ^ d.key), h1 = xs32(h0);
    return (uint32_t)((h0 & 0xFFFFu) >= d.thr) | ((uint32_t)((h0 >> 16) >= d.thr) << 1) |
           ((uint32_t)((h1 & 0xFFFFu) >= d.thr) << 2) | ((uint32_t)((h1 >> 16) >= d.thr) << 3);
}
// the 16-bit fields are compared in place (the compares land in scalar lane masks and feed v_cndmask directly):
// building the 4-bit mask of drop_mask4 in a VGPR and testing its bits again costs ~2x the VALU of this form
__device__ __forceinline__ void drop_apply4(const Drop& d, uint32_t base, f32x4& v) {
    if (d.thr == 0) return;
    const uint32_t h0 = mix32((base >> 2) ^ d.key), h1 = xs32(h0);
    v[0] = (h0 & 0xFFFFu) >= d.thr ? v[0] * d.scale : 0.0f;
    v[1] = (h0 >> 16) >= d.thr ? v[1] * d.scale : 0.0f;
    v[2] = (h1 & 0xFFFFu) >= d.thr ? v[2] * d.scale : 0.0f;
    v[3] = (h1 >> 16) >= d.thr ? v[3] * d.scale : 0.0f;
}
// v[r] = keep ? v[r] * c : 0 for 4 consecutive elements (c = whatever the caller folds into the 1 / (1 - p)); dropout must be active
__device__ __forceinline__ void drop_scale4(const Drop& d, uint32_t base, f32x4& v, float c) {
    const uint32_t h0 = mix32((base >> 2) ^ d.key), h1 = xs32(h0);
    v[0] = (h0 & 0xFFFFu) >= d.thr ? v[0] * c : 0.0f;
    v[1] = (h0 >> 16) >= d.thr ? v[1] * c : 0.0f;
    v[2] = (h1 & 0xFFFFu) >= d.thr ? v[2] * c : 0.0f;
    v[3] = (h1 >> 16) >= d.thr ? v[3] * c : 0.0f;
}
// v[r] = keep ? v[r] : 0 (unscaled); returns the 4 keep bits (bit r); dropout must be active
__device__ __forceinline__ uint32_t drop_select4(const Drop& d, uint32_t base, f32x4& v) {
    const uint32_t h0 = mix32((base >> 2) ^ d.key), h1 = xs32(h0);
    const bool k0 = (h0 & 0xFFFFu) >= d.thr, k1 = (h0 >> 16) >= d.thr, k2 = (h1 & 0xFFFFu) >= d.thr, k3 = (h1 >> 16) >= d.thr;
    v[0] = k0 ? v[0] : 0.0f; v[1] = k1 ? v[1] : 0.0f; v[2] = k2 ? v[2] : 0.0f; v[3] = k3 ? v[3] : 0.0f;
    return (uint32_t)k0 | ((uint32_t)k1 << 1) | ((uint32_t)k2 << 2) | ((uint32_t)k3 << 3);
}
// the four bytes of w as floats: v_cvt_f32_ubyte0 .. 3, one instruction each (left to itself the compiler shifts and masks first)
__device__ __forceinline__ f32x4 ubytes_to_f32x4(uint32_t w) {
    f32x4 v;
    asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(v[0]) : "v"(w));
    asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(v[1]) : "v"(w));
    asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(v[2]) : "v"(w));
    asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(v[3]) : "v"(w));
    return v;
}
// On PACKED 16-bit pairs (bf16 or fp16 alike: sign bit on top).  ReLU: the signed 16-bit maximum against 0 clears every value whose sign
// bit is set (negative numbers and -0), one instruction per pair.
__device__ __forceinline__ uint32_t pk_relu16(uint32_t v) {
    uint32_t r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(v));
    return r;
}
// Dropout of a pair by the two 16-bit fields of its hash word w (low field <-> low element): field >= thr  <=>  the saturating
// difference field - (thr - 1) is non-zero; clamped to 1 it is the factor of an integer multiply of the pair.  tt = (thr - 1) in both
// halves, thr >= 1.  Three instructions per pair, no compares, no selects.
__device__ __forceinline__ uint32_t pk_keep16(uint32_t v, uint32_t w, uint32_t tt) {
    uint32_t d;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(w), "v"(tt));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(d), "s"(0x00010001u));
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(d) : "v"(v), "v"(d));
    return d;
}
// ReLU + dropout of 4 consecutive elements in one select each; returns the 4 "kept and positive" bits (bit r)
__device__ __forceinline__ uint32_t relu_drop_apply4(const Drop& d, uint32_t base, f32x4& v) {
    bool on[4];
    if (d.thr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { on[r] = v[r] > 0.0f; v[r] = on[r] ? v[r] : 0.0f; }
    } else {
        const uint32_t h0 = mix32((base >> 2) ^ d.key), h1 = xs32(h0);
        on[0] = v[0] > 0.0f && (h0 & 0xFFFFu) >= d.thr;
        on[1] = v[1] > 0.0f && (h0 >> 16) >= d.thr;
        on[2] = v[2] > 0.0f && (h1 & 0xFFFFu) >= d.thr;
        on[3] = v[3] > 0.0f && (h1 >> 16) >= d.thr;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = on[r] ? v[r] * d.scale : 0.0f;
    }
    return (uint32_t)on[0] | ((uint32_t)on[1] << 1) | ((uint32_t)on[2] << 2) | ((uint32_t)on[3] << 3);
}

// ---------------------------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
// sum over the 16 lanes that share g = lane >> 4
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
// sum / max over the 4 lanes that share i = lane & 15
__device__ __forceinline__ float cross4_sum(float v) {
    v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float cross4_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
// block-wide sum for blockDim.x == 256; `red` is 4 floats of LDS; every thread gets the result
__device__ __forceinline__ float block256_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// XCD-aware bijective block remap (guide T1): blocks b and b+8 share an XCD (speed only).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

}  // namespace ge2e
