// Attention backward, every score evaluated ONCE (16-bit storage modes, T <= 288).
//
// attn_bwd_kernel (attention.cuh) evaluates every score twice -- once with the query on the lane (dQ), once with the key on the lane
// (dK, dV) -- and fetches Q / K / V / dO twice (LDS tile + row fragments).  Here a block is one (utterance, head) and
//   * wave w OWNS keys 32 w .. 32 w + 31: their K / V row fragments (16 + 16 registers) and their dK^T / dV^T accumulators (64 registers)
//     stay in registers for the whole block; the wave sweeps the queries in groups of 32;
//   * S = Q K^T and dP = dO V^T are computed with the KEY on the MFMA lane, so the probability / dS accumulator tiles are already the
//     B operands of dV^T += dO^T P and dK^T += Q^T dS (guide: "an accumulator tile as the next MFMA's operand"); exp, dropout and the
//     softmax backward run once per score;
//   * only dS crosses LDS, once, as 16-bit tiles in a [key][query] image (one ds_write_b64 per 16 x 16 tile), and comes back
//     through ds_read_b64_tr_b16 as the operand of dQ^T += K^T dS^T, which the waves share out by (query tile, 16 head dims) units;
//   * Q, K, dO (and O, for delta = dO . O) are fetched from HBM once, into LDS; every fragment comes from there (V: each wave's own rows
//     straight into registers).
// The queries are processed in passes of QP rows (Plan<KT>): the Q / dO tiles and the dS image hold one pass, so that two blocks fit a CU
// at the headline length (160 frames: 75 KB) -- one block's loads run under the other's MFMAs.
//
// Dropout: the counter of P[query][key] is ((n H + h) T + query) T4 + key, hashed per aligned quad of KEYS (common.cuh).  With the key on
// the lane a lane holds four QUERIES of one key, i.e. one field of four different hash words.  The four lanes of a key quad therefore hash
// one query each (lane i & 3 -> query 4 g + (i & 3)), turn their word pair into a keep nibble, and exchange nibbles inside the lane quad
// with two v_or_b32_dpp (quad_perm): 1 hash per 16 x 16 tile and lane, as in the query-on-lane layout.
#pragma once
#include "attention.cuh"

namespace ge2e {

namespace attn1 {
template <int KT> struct Plan {
    static constexpr int TP = 32 * KT;
    static constexpr int BLK = KT <= 6 ? 2 : 1;                           // blocks per CU the LDS budget aims at
    static constexpr int LIM = BLK == 2 ? 78 * 1024 : 156 * 1024;
    // K tile + Q, dO tiles of a pass + dS^T image of a pass + lse [TP] + delta [QP]
    static constexpr int lds(int qp) { return TP * 128 + 2 * qp * 128 + TP * qp * 2 + (TP + qp) * 4; }
    static constexpr int pick() { int best = 32; for (int qp = 32; qp <= TP; qp += 32) if (lds(qp) <= LIM) best = qp; return best; }
    static constexpr int QP = pick();                                     // queries per pass (multiple of 32)
    static constexpr int SMEM = lds(QP);
#ifdef ATTN1_MINW
    static constexpr int MINW = ATTN1_MINW;                               // (development: tools/attn_bwd_bench.hip)
#else
    static constexpr int MINW = (BLK * KT + 3) / 4;                       // waves per SIMD the register allocation must allow
#endif
};
// dS^T image: 16 (keys) x 16 (queries) blocks of 512 bytes, block (key tile, query tile of the pass); inside a block the 8-byte piece
// (key k, query quad c) sits at slot 4 k + (c ^ (k >> 2)): the 16 pieces one ds_write_b64 lane group stores (16 keys, one quad) and the 32
// pieces one half-wave of a transposed read fetches (8 keys x 4 quads) each cover all banks once.
__device__ __forceinline__ int ds_off(int blk, int k, int c) { return blk * 512 + ((4 * k + (c ^ (k >> 2))) << 3); }
__device__ __forceinline__ u32x2 tr8(const unsigned char* p) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p));
}
// OR over the four lanes of a lane quad (two v_or_b32 with quad_perm DPP)
__device__ __forceinline__ uint32_t quad_or(uint32_t w) {
    w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]
    w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x4E, 0xF, 0xF, true);      // quad_perm [2, 3, 0, 1]
    return w;
}
}  // namespace attn1

// ABL (development only, tools/attn_bwd_bench.hip): 1 no dQ phase, 2 no main phase, 4 no dropout work
template <typename T, int KT, bool PAD = true, bool DROP = true, int ABL = 0>
__global__ void __launch_bounds__(64 * KT, attn1::Plan<KT>::MINW) attn_bwd1_kernel(const AttnArgs p) {
    static_assert(sizeof(T) == 2, "16-bit storage modes");
    using G = attn::Geo<T>;
    using PL = attn1::Plan<KT>;
    constexpr int TP = PL::TP, QP = PL::QP, NQT = QP / 16, NTH = 64 * KT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Ks = smem;                               // [TP][64] of T, swizzled rows (attn::toff)
    unsigned char* const Qs = Ks + TP * 128;                      // [QP][64]: this pass's queries
    unsigned char* const Ds = Qs + QP * 128;                      // [QP][64]: dO of this pass's queries
    unsigned char* const St = Ds + QP * 128;                      // dS^T image of this pass: (TP / 16) x NQT blocks of 512 B
    float* const st_l = (float*)(St + TP * QP * 2);               // [TP] lse * ExpK
    float* const st_d = st_l + TP;                                // [QP] delta * scale of this pass's queries
    const int n = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const float ck = p.scale * ExpK<T>::K;
    const float sds = DROP ? p.scale * p.drop.scale : p.scale;

    // ---------------------------------------------------------------- block prologue: K tile, lse; this wave's V rows
    attn::load_tile<T>(Ks, kbase, ldq, p.T, TP);
    for (int q = tid; q < TP; q += NTH) st_l[q] = q < p.T ? p.lse[((size_t)n * p.T + q) * p.H + h] * ExpK<T>::K : 0.0f;
    const int krow0 = 32 * wave + i;                              // this lane's key of tile u: krow0 + 16 u
    u32x4 vf[2][G::NKG];
#pragma unroll
    for (int u = 0; u < 2; ++u) attn::load_row_frags<T>(vf[u], vbase, ldq, krow0 + 16 * u, !PAD || krow0 + 16 * u < p.T, g);
    f32x4 kacc[2][4], vacc[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { kacc[u][dt] = f32x4{0, 0, 0, 0}; vacc[u][dt] = f32x4{0, 0, 0, 0}; }

#pragma unroll 1
    for (int q0 = 0; q0 < TP; q0 += QP) {
        const int nq = min(QP, TP - q0);                          // rows of this pass (multiple of 32)
        // ---- Q and dO tiles of the pass; delta = dO . O rides in the dO load (8 lanes share a row: three quad / row steps)
        // (the Q / dO tiles are free: every wave is past the barrier that ended the previous pass's main phase; the dS image and the K tile may
        // still be read by waves in that pass's dQ phase -- they are not touched before the barrier below)
        attn::load_tile<T>(Qs, qbase + (size_t)q0 * ldq, ldq, p.T - q0, nq);
        for (int id = tid; id < nq * 8; id += NTH) {
            const int row = id >> 3, c = id & 7;
            u32x4 dv = u32x4{0, 0, 0, 0}, ov = u32x4{0, 0, 0, 0};
            if (q0 + row < p.T) {
                dv = *(const u32x4*)(dobase + (size_t)(q0 + row) * ldo + c * 16);
                ov = *(const u32x4*)(obase + (size_t)(q0 + row) * ldo + c * 16);
            }
            *(u32x4*)(Ds + attn::toff<T>(row, c)) = dv;
            const T* a = (const T*)&dv;
            const T* b = (const T*)&ov;
            float s = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += to_f32(a[e]) * to_f32(b[e]);
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            if (c == 0) st_d[row] = s * p.scale;                  // (rows beyond T: 0, finite -- their P is forced to 0)
        }
        __syncthreads();
        // ---------------------------------------------------------------- main phase: this wave's 32 keys x the pass's queries
        if (!(ABL & 2)) {
#pragma unroll 1
        for (int gq = 0; gq < nq / 32; ++gq) {
            // scores first: P and dS of the group's 32 queries x this wave's 32 keys, packed to the storage type as they are made (the
            // packed pairs ARE the operand fragments of the products below and the pieces of the dS^T image)
            u32x4 pb[2], sb[2];                                   // [key tile of the wave]: queries of tile 0 in .xy, of tile 1 in .zw
            u32x4 kf[2][G::NKG];                                  // this wave's K rows: from the K tile every group (16 registers not held across the products below)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < G::NKG; ++k) kf[u][k] = lds16(Ks + attn::toff<T>(krow0 + 16 * u, k * 4 + g));
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const int tq = 2 * gq + tl;                       // query tile of the pass
                u32x4 qa[G::NKG], da[G::NKG];
#pragma unroll
                for (int k = 0; k < G::NKG; ++k) {
                    qa[k] = lds16(Qs + attn::toff<T>(16 * tq + i, k * 4 + g));
                    da[k] = lds16(Ds + attn::toff<T>(16 * tq + i, k * 4 + g));
                }
                const f32x4 l4 = *(const f32x4*)(st_l + q0 + 16 * tq + 4 * g);
                const f32x4 d4 = *(const f32x4*)(st_d + 16 * tq + 4 * g);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 sa = f32x4{0, 0, 0, 0}, dp = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int k = 0; k < G::NKG; ++k) {
                        sa = mma16<T>(qa[k], kf[u][k], sa);       // S[query 16 tq + 4g + r][key krow0 + 16 u]
                        dp = mma16<T>(da[k], vf[u][k], dp);       // d(P dropped)[query][key]
                    }
                    [[maybe_unused]] f32x4 keep4 = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
                    if constexpr (DROP && !(ABL & 4)) {
                        // this lane hashes query 4g + (i & 3) of the tile against the key quad of its lane quad; nibble -> byte i & 3; OR over the quad
                        const uint32_t qh = (uint32_t)(q0 + 16 * tq + 4 * g + (i & 3));
                        const uint32_t kq = (uint32_t)(32 * wave + 16 * u + (i & 12));
                        const uint32_t m = drop_mask4(p.drop, (hbase + qh) * T4 + kq);
                        const uint32_t w = attn1::quad_or(m << (8 * (i & 3)));      // byte r: keep bits of query 4g + r for keys kq .. kq + 3
                        keep4 = ubytes_to_f32x4((w >> (i & 3)) & 0x01010101u);
                    }
                    const bool vk = !PAD || krow0 + 16 * u < p.T;
                    f32x4 pd, ds;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pr = ExpK<T>::ex(sa[r] * ck - l4[r]);
                        if (PAD && !(vk && q0 + 16 * tq + 4 * g + r < p.T)) pr = 0.0f;
                        float dv = dp[r];
                        pd[r] = pr;
                        if constexpr (DROP) { dv *= keep4[r]; pd[r] = pr * keep4[r]; }
                        ds[r] = pr * (dv * sds - d4[r]);
                    }
                    u32x2 piece;
                    piece.x = pack2<T>(ds[0], ds[1]); piece.y = pack2<T>(ds[2], ds[3]);
                    *(u32x2*)(St + attn1::ds_off((2 * wave + u) * NQT + tq, i, g)) = piece;
                    pb[u][2 * tl] = pack2<T>(pd[0], pd[1]); pb[u][2 * tl + 1] = pack2<T>(pd[2], pd[3]);
                    sb[u][2 * tl] = piece.x; sb[u][2 * tl + 1] = piece.y;
                }
            }
            __builtin_amdgcn_sched_barrier(0);                    // (the transposed fragments below are not live across the score math)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const u32x4 trd = attn::tile_tr<T>(Ds, gq * 32, dt * 16, lane);       // dO^T, Q^T of the group's 32 queries, 16 head dims
                const u32x4 trq = attn::tile_tr<T>(Qs, gq * 32, dt * 16, lane);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    vacc[u][dt] = mma16<T>(trd, pb[u], vacc[u][dt]);       // dV^T[d][key] += dO^T[d][queries] Pd[queries][key]
                    kacc[u][dt] = mma16<T>(trq, sb[u], kacc[u][dt]);       // dK^T[d][key] += Q^T[d][queries] dS[queries][key]
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        __syncthreads();
        // ---------------------------------------------------------------- dQ phase: units of (query tile, 16 head dims), all keys
        if (!(ABL & 1)) {
        for (int un = wave; un < (nq / 16) * 4; un += KT) {
            const int tq = un >> 2, dt = un & 3;
            f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll 2
            for (int gi = 0; gi < KT; ++gi) {                    // dQ^T[d][query] += K^T[d][keys 32 gi ..] dS^T[keys][query]
                const u32x4 a = attn::tile_tr<T>(Ks, gi * 32, dt * 16, lane);
                const u32x2 lo = attn1::tr8(St + attn1::ds_off((2 * gi) * NQT + tq, 4 * g + (i >> 2), i & 3));
                const u32x2 hi = attn1::tr8(St + attn1::ds_off((2 * gi + 1) * NQT + tq, 4 * g + (i >> 2), i & 3));
                acc = mma16<T>(a, u32x4{lo.x, lo.y, hi.x, hi.y}, acc);
            }
            const int qrow = q0 + 16 * tq + i;
            if (!PAD || qrow < p.T) store4(dq_out + (size_t)qrow * 3 * p.D + dt * 16 + 4 * g, acc[0], acc[1], acc[2], acc[3]);
        }
        }
    }
    // ---------------------------------------------------------------- this wave's dK, dV rows
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int krow = krow0 + 16 * u;
        if (!PAD || krow < p.T) {
            T* row = dq_out + (size_t)krow * 3 * p.D + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 v = vacc[u][dt];
                if constexpr (DROP) v *= p.drop.scale;
                store4(row + p.D + dt * 16, kacc[u][dt][0], kacc[u][dt][1], kacc[u][dt][2], kacc[u][dt][3]);
                store4(row + 2 * p.D + dt * 16, v[0], v[1], v[2], v[3]);
            }
        }
    }
}

}  // namespace ge2e
