// EXPERIMENT, not part of the library (round 2): one-pass attention backward.  Measured on MI355X at 960 x 4 heads x 160 frames
// (tools/attn_bench.hip) against the two-phase attn_bwd_kernel's 322 us:
//     full                         1365 us
//     without the dQ ds_add_f32     440 us   -> an LDS float atomic costs ~185 cycles per wave instruction here (800 per block)
//     without atomics and barrier   350 us   -> the one-pass skeleton alone (10 waves per CU, a barrier per 32-query group) is
//                                               already slower than the two-phase kernel with its duplicated score evaluation
//     only the dK / dV MFMAs        157 us
// A cross-wave reduction of dQ (or, mirrored, of dK / dV) is inherent to evaluating each score once; through LDS atomics it is
// far too slow, through per-wave fp32 partials it needs 40 KB more LDS (one block per CU).  Results were correct (all parity
// tests passed with it); it stays here for the record.
#pragma once
#include "attention.cuh"

namespace ge2e {

// ---------------------------------------------------------------------------------------------
// One-pass backward (16-bit modes).  The two-phase kernel above evaluates every score, probability, dropout bit and dP twice
// (once per operand layout); here each 16x16 tile is evaluated ONCE.  A wave owns 32 keys (dK, dV accumulate in its
// registers); Q and dO of the head are LDS-resident and visited in groups of 32 queries.  Per group a wave
//   * computes S and dP for its 32 keys x 32 queries (rows = queries, lanes = keys), the probabilities and dS;
//   * feeds P / dS, packed from the accumulators, into dV^T += dO^T P and dK^T += Q^T dS (k-dim = queries);
//   * writes dS (16-bit) into a wave-private [32 keys][32 queries] LDS slab and reads it back TRANSPOSED as the operand of
//     dQ^T += K^T dS^T (k-dim = keys; K^T fragments of the wave's keys live in registers);
//   * adds its dQ^T tiles into a block-shared fp32 [32 queries][64] chunk (ds_add_f32); after the group's barrier the block
//     converts and stores that chunk while the next group accumulates into the other one.
// delta = dO . O comes from a short pre-pass.  The dropout hash of an element pair (two neighbouring keys = two neighbouring
// lanes) is evaluated by ONE of the two lanes and handed over by a DPP quad permute.
// LDS at T = 160: Q, dO 45 KB + dQ chunks 16 KB + slabs 12.5 KB + lse/delta 1.3 KB = 75 KB: two blocks per CU.
// ABL (development, tools/attn_bench.hip): 1 no dQ atomics, 2 no group barrier / chunk write-out, 4 no slab round trip + dQ MFMAs,
// 8 no dK / dV MFMAs, 16 no score / probability evaluation
template <typename T, int KT, bool PAD = true, int ABL = 0>
__global__ void __launch_bounds__(64 * KT, 3) attn_bwd1_kernel(const AttnArgs p) {
    static_assert(sizeof(T) == 2, "16-bit storage modes");
    using G = attn::Geo<T>;
    constexpr int TP = 32 * KT, NW = KT;
    constexpr int SLD = 80;                   // slab row pitch in bytes (32 queries x 2 B + 16)
    constexpr int QP = 65;                    // dQ chunk row pitch in floats (64 + 1: the 16 queries of a ds_add land on 16 banks)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Qs = smem;
    unsigned char* const dOs = smem + TP * G::LD;
    float* const st_l = (float*)(smem + 2 * TP * G::LD);
    float* const st_d = st_l + TP;
    float* const dQb = st_d + TP;                                   // [2][32][QP]
    unsigned char* const slab = (unsigned char*)(dQb + 2 * 32 * QP) + (threadIdx.x >> 6) * (32 * SLD);
    const int n = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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

    // ---- K^T fragments of this wave's 32 keys: through the (not yet used) Q tile
    attn::load_tile<T>(Qs, kbase, ldq, p.T, TP);
    __syncthreads();
    u32x4 kT[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) kT[dt] = attn::tile_tr<T>(Qs, 32 * wave, dt * 16, lane);
    __syncthreads();
    attn::load_tile<T>(Qs, qbase, ldq, p.T, TP);
    attn::load_tile<T>(dOs, dobase, ldo, p.T, TP);
    for (int q = threadIdx.x; q < TP; q += blockDim.x) st_l[q] = q < p.T ? p.lse[((size_t)n * p.T + q) * p.H + h] * ExpK<T>::K : 0.0f;
    for (int q = threadIdx.x; q < 2 * 32 * QP; q += blockDim.x) dQb[q] = 0.0f;
    __syncthreads();
    // ---- delta_q = dO_q . O_q: thread (row, 16-byte chunk), 8 chunks per row reduced over 8 neighbouring lanes
    for (int id = threadIdx.x; id < TP * 8; id += blockDim.x) {       // TP * 8 is a multiple of the block size: whole waves
        const int row = id >> 3, c = id & 7;
        float part = 0.0f;
        if (row < p.T) {
            const u32x4 a = *(const u32x4*)(dOs + attn::toff<T>(row, c));
            const u32x4 b = *(const u32x4*)(obase + (size_t)row * ldo + c * 16);
            const T* pa = (const T*)&a; const T* pb = (const T*)&b;
#pragma unroll
            for (int e = 0; e < 8; ++e) part += to_f32(pa[e]) * to_f32(pb[e]);
        }
        part += __shfl_xor(part, 1); part += __shfl_xor(part, 2); part += __shfl_xor(part, 4);
        if (c == 0) st_d[row] = part;          // padded rows: 0 (finite: phase arithmetic multiplies it by P = 0)
    }
    __syncthreads();

    const bool dropping = p.drop.thr != 0;
    const float ck = p.scale * ExpK<T>::K;
    u32x4 kf[2][G::NKG], vf[2][G::NKG];
    bool vk[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int krow = 32 * wave + 16 * kt + i;
        vk[kt] = !PAD || krow < p.T;
        attn::load_row_frags<T>(kf[kt], kbase, ldq, krow, vk[kt], g);
        attn::load_row_frags<T>(vf[kt], vbase, ldq, krow, vk[kt], g);
    }
    f32x4 kacc[2][4], vacc[2][4];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { kacc[kt][dt] = f32x4{0, 0, 0, 0}; vacc[kt][dt] = f32x4{0, 0, 0, 0}; }
    const int par = i & 1;                    // this lane's key parity = its half of the pair hashes

#pragma unroll 1
    for (int gi = 0; gi < KT; ++gi) {
        u32x4 pb[2], sb[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const uint32_t krow = (uint32_t)(32 * wave + 16 * kt + i);
            f32x4 pd[2], ds[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * gi + u;
                if constexpr (ABL & 16) { pd[u] = f32x4{0.5f, 0.25f, 0.125f, 1.0f}; ds[u] = pd[u]; continue; }
                const f32x4 sa = attn::tile_dot<T>(Qs, t, kf[kt], i, g);     // S[query 16t+4g+r][key krow]
                const f32x4 da = attn::tile_dot<T>(dOs, t, vf[kt], i, g);    // d(P dropped)[query][key]
                const f32x4 l4 = *(const f32x4*)(st_l + 16 * t + 4 * g);
                const f32x4 d4 = *(const f32x4*)(st_d + 16 * t + 4 * g);
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                if (dropping) {
                    // element (q, krow): word = mix32(((hbase + q) T4 + krow) >> 1 ^ key); lanes i and i ^ 1 share it
                    const uint32_t q0 = hbase + (uint32_t)(16 * t + 4 * g + 2 * par);
                    const uint32_t ha = mix32(((q0 * T4 + krow) >> 1) ^ p.drop.key);
                    const uint32_t hb = mix32((((q0 + 1u) * T4 + krow) >> 1) ^ p.drop.key);
                    const uint32_t oa = (uint32_t)__builtin_amdgcn_mov_dpp((int)ha, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                    const uint32_t ob = (uint32_t)__builtin_amdgcn_mov_dpp((int)hb, 0xB1, 0xF, 0xF, true);
                    w[0] = par ? oa : ha; w[1] = par ? ob : hb; w[2] = par ? ha : oa; w[3] = par ? hb : ob;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = 16 * t + 4 * g + r;
                    float pr = ExpK<T>::ex(sa[r] * ck - l4[r]);
                    if (PAD && !((q < p.T) && vk[kt])) pr = 0.0f;
                    float dv = da[r];
                    pd[u][r] = pr;
                    if (dropping) {
                        const bool keep = (par ? (w[r] >> 16) : (w[r] & 0xFFFFu)) >= p.drop.thr;
                        dv = keep ? dv * p.drop.scale : 0.0f;
                        pd[u][r] = keep ? pr * p.drop.scale : 0.0f;
                    }
                    ds[u][r] = (PAD && pr == 0.0f) ? 0.0f : pr * (dv - d4[r]) * p.scale;
                }
                // dS of (key krow, queries 16u + 4g .. + 3 of this group) into the slab: row = key, 4 consecutive queries
                if constexpr ((ABL & 4) == 0) store4((T*)(slab + (16 * kt + i) * SLD + (16 * u + 4 * g) * 2), ds[u][0], ds[u][1], ds[u][2], ds[u][3]);
            }
            pb[kt] = pack_acc<T>(pd[0], pd[1]);
            sb[kt] = pack_acc<T>(ds[0], ds[1]);
        }
#pragma unroll
        for (int dt = 0; dt < ((ABL & 8) ? 0 : 4); ++dt) {
            const u32x4 ao = attn::tile_tr<T>(dOs, 32 * gi, dt * 16, lane);
            const u32x4 aq = attn::tile_tr<T>(Qs, 32 * gi, dt * 16, lane);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                vacc[kt][dt] = mma16<T>(ao, pb[kt], vacc[kt][dt]);      // dV^T += dO^T Pd
                kacc[kt][dt] = mma16<T>(aq, sb[kt], kacc[kt][dt]);      // dK^T += Q^T dS
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // dQ^T[d][query] += K^T[d][32 keys] dS^T[32 keys][query]; the slab is wave-private (program order + the compiler's waits)
        float* const qb = dQb + (gi & 1) * (32 * QP);
#pragma unroll
        for (int u = 0; u < ((ABL & 4) ? 0 : 2); ++u) {
            const u32x4 bq = frag_tr<T>(slab, SLD, 0, 16 * u, lane);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f32x4 a = mma16<T>(kT[dt], bq, f32x4{0, 0, 0, 0});   // [d = 16dt + 4g + r][query 16u + i]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (ABL & 1) asm volatile("" :: "v"(a[r]));
                    else __hip_atomic_fetch_add(qb + (16 * u + i) * QP + 16 * dt + 4 * g + r, a[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if constexpr ((ABL & 2) == 0) __syncthreads();                      // every wave's share of this group's dQ is in the chunk
        // convert + store the chunk (all threads; the other chunk takes the next group's sums meanwhile), and clear it
        for (int id = threadIdx.x; id < ((ABL & 2) ? 0 : 32 * 32); id += blockDim.x) {
            const int q = id >> 5, c2 = (id & 31) * 2, row = 32 * gi + q;
            float* const src = qb + q * QP + c2;
            const float a = src[0], b = src[1];
            src[0] = 0.0f; src[1] = 0.0f;
            if (row < p.T) {
                const u32x4 pk = pack_acc<T>(f32x4{a, b, 0, 0}, f32x4{0, 0, 0, 0});
                *(uint32_t*)(dq_out + (size_t)row * 3 * p.D + c2) = pk.x;
            }
        }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int krow = 32 * wave + 16 * kt + i;
        if (vk[kt] || !PAD) {
            if (krow < p.T) {
                T* row = dq_out + (size_t)krow * 3 * p.D + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    store4(row + p.D + dt * 16, kacc[kt][dt][0], kacc[kt][dt][1], kacc[kt][dt][2], kacc[kt][dt][3]);
                    store4(row + 2 * p.D + dt * 16, vacc[kt][dt][0], vacc[kt][dt][1], vacc[kt][dt][2], vacc[kt][dt][3]);
                }
            }
        }
    }
}
template <typename T, int KT> constexpr size_t attn_bwd1_smem() {
    return 2 * (size_t)(32 * KT) * attn::Geo<T>::LD + 2 * (size_t)(32 * KT) * 4 + 2 * 32 * 65 * 4 + (size_t)KT * 32 * 80;
}


}  // namespace ge2e
