// The attention sub-layer of a full encoder layer as ONE kernel per utterance (the second fusion; 8a rows a3 + a4 + a5):
//
//     h1 = LayerNorm(x + drop(attention(x Win^T + b_in) Wo^T + b_o))                      (reference Modules.py:25-31,53; torch post-LN layer)
//
// replacing in_proj GEMM + attn_fwd_kernel + out_proj / LayerNorm GEMM (11 activation tiles of HBM traffic per layer: x in, q|k|v out and in again,
// o out and in again, x in again, h1 out) by x in once, h1 out once and -- train mode only -- q|k|v and o out once for the backward (6; eval 2).
// 16-bit storage modes, T <= 160 frames (the x tile, one head's K and V and one weight stage fill the LDS: 80 + 40 + 32 KB).
//
// Block = one utterance, one wave per 16 frames (160 frames: 10 waves).  A wave owns its 16 rows end to end:
//   * the x tile sits in LDS (swizzled 512-byte rows); a wave reads its own rows as MFMA activation fragments for the three projections of
//     every head and, at the very end, as the residual; the finished h1 rows go back INTO those LDS rows and leave as whole 512-byte lines;
//   * the weights stream through ONE 32 KB LDS stage: per head q, k, v (64 rows x 256 of Win each) and the head's 256 x 64 column block of Wo,
//     prefetched into registers a stage ahead; the stage's rows are PERMUTED on the way in so that a lane's accumulators are 8 CONSECUTIVE head
//     dims: the packed q tile is then directly the attention's query fragment, the packed k / v tiles are 16-byte pieces of the head's K / V LDS
//     tiles (attention.cuh layout) and of the q|k|v rows in HBM;
//   * attention of the wave's 16 queries against the head's K / V tiles as attn_fwd_kernel does it (S^T = K Q^T, probabilities stay in
//     registers); its O^T accumulators, packed pairwise, ARE the activation fragments of the out-projection (8 registers per head), which runs
//     after the last head against Wo staged in four column blocks with their columns permuted to the fragments' slot order;
//   * epilogue: bias, dropout, residual, LayerNorm (row statistics across the four lanes that share a row), rstd for the backward.
#pragma once
#include "attention.cuh"

namespace ge2e {

struct AttnSubArgs {
    const void* X;                        // [R][256] of T: the layer input (and the residual)
    const void* Win; const float* bin;    // in_proj_weight [768][256] of T (k-contiguous), in_proj_bias [768]
    const void* Wo; const float* bo;      // out_proj.weight [256][256] of T, out_proj.bias [256]
    const float* gamma; const float* beta; float eps;      // norm1
    void* qkv;                            // [R][768] of T out (train: the backward reads it) or null
    void* o;                              // [R][256] of T out (train) or null
    float* lse;                           // [R][H] out (train) or null
    void* h1;                             // [R][256] of T out
    float* rstd;                          // [R] out (train) or null
    int T;
    float scale;                          // 1 / sqrt(64)
    Drop drop_attn, drop_sa;              // dropout on the probabilities / on the sub-layer output
};

namespace attn_sub {
constexpr int WS_BYTES = 32 * 1024;
template <int NW> constexpr int smem_bytes() { return 16 * NW * 512 + 2 * (32 * ((NW + 1) / 2)) * 128 + WS_BYTES; }
// head dim of (projection column tile nt, tile row ir): a lane's accumulators acc[2k][r], acc[2k + 1][r] are dims 32 k + 8 g + r, .. + 4 + r
__device__ __forceinline__ constexpr int perm_row(int d) { return 16 * (2 * (d >> 5) + ((d >> 2) & 1)) + 4 * ((d & 31) >> 3) + (d & 3); }
}  // namespace attn_sub

// ABL (development only, tools/attn_sub_bench.hip): 1 no attention, 2 no weight streaming (the stage is never refilled), 4 no projection MFMAs
template <typename T, int NW, bool PAD, bool DROP, int ABL = 0>
__global__ void __launch_bounds__(64 * NW) attn_sub_fwd_kernel(const AttnSubArgs p) {
    static_assert(sizeof(T) == 2, "16-bit storage modes");
    static_assert(NW >= 1 && NW <= 10, "T <= 160 frames");
    constexpr int KT = (NW + 1) / 2, TP = 32 * KT, NT16 = 2 * KT, NG = KT, NTH = 64 * NW, H = 4, D = 256;
    constexpr int NCH = (2048 + NTH - 1) / NTH;              // 16-byte chunks of a weight stage per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Xs = smem;                           // [16 NW][512 B], chunk c of row r at c ^ (r & 15)
    unsigned char* const Ks = Xs + 16 * NW * 512;             // [TP][128 B] (attn::toff)
    unsigned char* const Vs = Ks + TP * 128;
    unsigned char* const Ws = Vs + TP * 128;                  // weight stage
    const int n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int qrow = 16 * wave + i;                           // this lane's frame
    const bool vq = !PAD || qrow < p.T;
    const size_t row0 = (size_t)n * p.T;                      // first row of the utterance
    const unsigned char* const Xg = (const unsigned char*)p.X + row0 * D * 2;
    const unsigned char* const Wing = (const unsigned char*)p.Win;
    const unsigned char* const Wog = (const unsigned char*)p.Wo;

    // ---------------------------------------------------------------- weight stages: global -> registers (prefetch) -> LDS
    u32x4 wreg[NCH];
    auto fetch_in = [&](int which, int h) {                   // rows 256 which + 64 h .. + 63 of Win, 256 k each: chunk id = row d (6 bits) | chunk c (5 bits)
        if constexpr (ABL & 2) return;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int id = tid + q * NTH;
            if (id < 2048) wreg[q] = *(const u32x4*)(Wing + ((size_t)(256 * which + 64 * h + (id >> 5)) * D) * 2 + (id & 31) * 16);
        }
    };
    auto put_in = [&]() {                                     // row d -> stage row perm_row(d)
        if constexpr (ABL & 2) return;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int id = tid + q * NTH;
            if (id < 2048) *(u32x4*)(Ws + swz_off<512>(attn_sub::perm_row(id >> 5), id & 31)) = wreg[q];
        }
    };
    auto fetch_out = [&](int h) {                             // columns 64 h .. + 63 of all 256 rows of Wo: chunk id = row c (8 bits) | chunk (3 bits)
        if constexpr (ABL & 2) return;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int id = tid + q * NTH;
            if (id < 2048) wreg[q] = *(const u32x4*)(Wog + ((size_t)(id >> 3) * D + 64 * h) * 2 + (id & 7) * 16);
        }
    };
    auto put_out = [&]() {                                    // 8-byte piece q = d / 4 = 8 kk + 4 u + gg  ->  piece 8 kk + 2 gg + u of the row
        if constexpr (ABL & 2) return;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int id = tid + q * NTH;
            if (id < 2048) {
                const int c = id >> 3, ch = id & 7;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int sp = 2 * ch + e, dp = (sp & ~7) | ((sp & 3) << 1) | ((sp >> 2) & 1);
                    *(u32x2*)(Ws + c * 128 + (((dp >> 1) ^ (c & 7)) << 4) + 8 * (dp & 1)) = e ? u32x2{wreg[q].z, wreg[q].w} : u32x2{wreg[q].x, wreg[q].y};
                }
            }
        }
    };

    // ---------------------------------------------------------------- prologue: x tile; unowned K / V rows are zero for good
    fetch_in(0, 0);
    for (int id = tid; id < 16 * NW * 32; id += NTH) {
        const int row = id >> 5, c = id & 31;
        u32x4 v = u32x4{0, 0, 0, 0};
        if (row < p.T) v = *(const u32x4*)(Xg + (size_t)row * D * 2 + c * 16);
        *(u32x4*)(Xs + swz_off<512>(row, c)) = v;
    }
    if constexpr (TP > 16 * NW) {
        for (int id = tid; id < (TP - 16 * NW) * 8; id += NTH) {
            const int row = 16 * NW + (id >> 3), c = id & 7;
            *(u32x4*)(Ks + attn::toff<T>(row, c)) = u32x4{0, 0, 0, 0};
            *(u32x4*)(Vs + attn::toff<T>(row, c)) = u32x4{0, 0, 0, 0};
        }
    }

    // one projection of the wave's 16 rows against the 64 (permuted) rows of the stage, + bias: acc[nt][r] = head dim 32 (nt >> 1) + 8 g + 4 (nt & 1) + r
    auto project = [&](f32x4* acc, const float* bias) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int kg = 0; kg < ((ABL & 4) ? 0 : 8); ++kg) {
            const u32x4 a = lds16(Xs + swz_off<512>(qrow, kg * 4 + g));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = mma16<T>(lds16(Ws + swz_off<512>(16 * nt + i, kg * 4 + g)), a, acc[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] += *(const f32x4*)(bias + 32 * (nt >> 1) + 8 * g + 4 * (nt & 1));
    };
    T* const qkv_row = p.qkv ? (T*)p.qkv + (row0 + qrow) * 3 * D : nullptr;
    // the heads' attention outputs of the wave's 16 rows, as out-projection activation fragments (8 registers per head; the 16 x 256 fp32 output
    // tile -- 64 registers -- only exists after the last head: held across the heads it does not fit beside the attention's score tile at the
    // 168 registers of a 10-wave block)
    u32x4 of[H][2];
#pragma unroll
    for (int hh = 0; hh < H; ++hh) { of[hh][0] = u32x4{0, 0, 0, 0}; of[hh][1] = u32x4{0, 0, 0, 0}; }

#pragma unroll 1
    for (int h = 0; h < H; ++h) {
        // ---- q of the wave's rows: stays in registers as the attention's query fragments
        put_in();
        fetch_in(1, h);
        __syncthreads();
        u32x4 qf[2];
        {
            f32x4 acc[4];
            project(acc, p.bin + 64 * h);
            qf[0] = pack_acc<T>(acc[0], acc[1]); qf[1] = pack_acc<T>(acc[2], acc[3]);
            if (qkv_row && vq) { *(u32x4*)(qkv_row + 64 * h + 8 * g) = qf[0]; *(u32x4*)(qkv_row + 64 * h + 32 + 8 * g) = qf[1]; }
        }
        __syncthreads();
        // ---- k, v of the wave's rows: into the head's K / V tiles
        put_in();
        fetch_in(2, h);
        __syncthreads();
        {
            f32x4 acc[4];
            project(acc, p.bin + 256 + 64 * h);
            const u32x4 k0 = pack_acc<T>(acc[0], acc[1]), k1 = pack_acc<T>(acc[2], acc[3]);
            *(u32x4*)(Ks + attn::toff<T>(qrow, g)) = vq ? k0 : u32x4{0, 0, 0, 0};
            *(u32x4*)(Ks + attn::toff<T>(qrow, 4 + g)) = vq ? k1 : u32x4{0, 0, 0, 0};
            if (qkv_row && vq) { *(u32x4*)(qkv_row + D + 64 * h + 8 * g) = k0; *(u32x4*)(qkv_row + D + 64 * h + 32 + 8 * g) = k1; }
        }
        __syncthreads();
        put_in();
        if (h + 1 < H) fetch_in(0, h + 1); else fetch_out(0);      // the stage after this head's v: the next head's q, or the first block of Wo
        __syncthreads();
        {
            f32x4 acc[4];
            project(acc, p.bin + 512 + 64 * h);
            const u32x4 v0 = pack_acc<T>(acc[0], acc[1]), v1 = pack_acc<T>(acc[2], acc[3]);
            *(u32x4*)(Vs + attn::toff<T>(qrow, g)) = vq ? v0 : u32x4{0, 0, 0, 0};
            *(u32x4*)(Vs + attn::toff<T>(qrow, 4 + g)) = vq ? v1 : u32x4{0, 0, 0, 0};
            if (qkv_row && vq) { *(u32x4*)(qkv_row + 2 * D + 64 * h + 8 * g) = v0; *(u32x4*)(qkv_row + 2 * D + 64 * h + 32 + 8 * g) = v1; }
        }
        __syncthreads();                                      // K, V tiles complete
        // ---- attention of the wave's 16 queries (attn_fwd_kernel's body)
        if constexpr (!(ABL & 1)) {
            f32x4 s[NT16];
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
                s[t] = attn::tile_dot<T>(Ks, t, qf, i, g);     // S^T[key 16t + 4g + r][query qrow]
                if ((t % 5) == 4) __builtin_amdgcn_sched_barrier(0);
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
                for (int r = 0; r < 4; ++r) { const float e = ExpK<T>::ex(s[t][r] * ck - mk); s[t][r] = e; sum += e; }
            const float tot = cross4_sum(sum);
            const float inv = DROP ? p.drop_attn.scale / tot : 1.0f / tot;
            if (p.lse && g == 0 && vq) p.lse[(row0 + qrow) * H + h] = mx * p.scale + logf(tot);
            const uint32_t ibase = ((uint32_t)(n * H + h) * (uint32_t)p.T + (uint32_t)qrow) * (uint32_t)((p.T + 3) & ~3);
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
                if constexpr (DROP) drop_scale4(p.drop_attn, ibase + (uint32_t)(16 * t + 4 * g), s[t], inv);
                else s[t] *= inv;
            }
            f32x4 ov[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) ov[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const u32x4 pb = pack_acc<T>(s[2 * gi], s[2 * gi + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)                // O^T[d = 16 dt + 4g + r][query] += V^T[d][keys] P^T[keys][query]
                    ov[dt] = mma16<T>(attn::tile_tr<T>(Vs, gi * 32, dt * 16, lane), pb, ov[dt]);
                if ((gi % 5) == 4) __builtin_amdgcn_sched_barrier(0);
            }
            if (p.o && vq) {
                T* orow = (T*)p.o + (row0 + qrow) * D + 64 * h + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) store4(orow + dt * 16, ov[dt][0], ov[dt][1], ov[dt][2], ov[dt][3]);
            }
            // the O^T tiles, packed pairwise, are the out-projection's activation fragments: slots 0-3 = dims 32 kk + 4g + j, 4-7 = 32 kk + 16 + 4g + j
            const u32x4 o0 = pack_acc<T>(ov[0], ov[1]), o1 = pack_acc<T>(ov[2], ov[3]);
#pragma unroll
            for (int hh = 0; hh < H; ++hh) {                   // (h is wave-uniform: a select per register, no dynamically indexed array)
                of[hh][0] = h == hh ? o0 : of[hh][0];
                of[hh][1] = h == hh ? o1 : of[hh][1];
            }
        }
        __syncthreads();                                      // everyone is done with K, V (and long done with the stage)
    }

    // ---------------------------------------------------------------- out-projection: Wo in four 256 x 64 column blocks (staged with their columns in the fragments' slot order)
    f32x4 oacc[16];                                           // this wave's 16 x 256 tile: oacc[ct][r] = out[qrow][16 ct + 4 g + r]
#pragma unroll
    for (int ct = 0; ct < 16; ++ct) oacc[ct] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
        put_out();
        if (hh + 1 < H) fetch_out(hh + 1);
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < 16; ++ct)
#pragma unroll
            for (int kk = 0; kk < ((ABL & 4) ? 0 : 2); ++kk) oacc[ct] = mma16<T>(lds16(Ws + attn::toff<T>(16 * ct + i, kk * 4 + g)), of[hh][kk], oacc[ct]);
        if (hh + 1 < H) __syncthreads();
    }

    // ---------------------------------------------------------------- epilogue: bias, dropout, residual, LayerNorm over the row (this lane: 64 of its 256 columns)
    const uint32_t dbase = (uint32_t)(row0 + qrow) * (uint32_t)D;
    float s1 = 0.0f;
#pragma unroll
    for (int ct = 0; ct < 16; ++ct) {
        const int col = 16 * ct + 4 * g;
        f32x4 v = oacc[ct] + *(const f32x4*)(p.bo + col);
        drop_apply4(p.drop_sa, dbase + (uint32_t)col, v);
        const u32x2 xr = *(const u32x2*)(Xs + swz_off<512>(qrow, 2 * ct + (g >> 1)) + 8 * (g & 1));      // x[qrow][col .. col + 3]
        const T* xe = (const T*)&xr;
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] += to_f32(xe[r]); s1 += v[r]; }
        oacc[ct] = v;
    }
    const float mean = cross4_sum(s1) * (1.0f / (float)D);
    float s2 = 0.0f;
#pragma unroll
    for (int ct = 0; ct < 16; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float dlt = oacc[ct][r] - mean; oacc[ct][r] = dlt; s2 += dlt * dlt; }
    const float rs = 1.0f / sqrtf(cross4_sum(s2) * (1.0f / (float)D) + p.eps);
    if (p.rstd && g == 0 && vq) p.rstd[row0 + qrow] = rs;
    // the finished row goes back into the wave's own x rows (nobody else reads them) and leaves as whole 512-byte lines
#pragma unroll
    for (int ct = 0; ct < 16; ++ct) {
        const int col = 16 * ct + 4 * g;
        const f32x4 ga = *(const f32x4*)(p.gamma + col), be = *(const f32x4*)(p.beta + col);
        u32x2 w2;
        w2.x = pack2<T>(oacc[ct][0] * rs * ga[0] + be[0], oacc[ct][1] * rs * ga[1] + be[1]);
        w2.y = pack2<T>(oacc[ct][2] * rs * ga[2] + be[2], oacc[ct][3] * rs * ga[3] + be[3]);
        *(u32x2*)(Xs + swz_off<512>(qrow, 2 * ct + (g >> 1)) + 8 * (g & 1)) = w2;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        unsigned char* const Hg = (unsigned char*)p.h1 + row0 * D * 2;
#pragma unroll
        for (int q = 0; q < 8; ++q) {                         // 16 rows x 32 chunks per wave: lane -> (row 2 q + (lane >> 5), chunk lane & 31)
            const int row = 16 * wave + 2 * q + (lane >> 5), c = lane & 31;
            const u32x4 v = lds16(Xs + swz_off<512>(row, c));
            if (!PAD || row < p.T) __builtin_nontemporal_store(v, (u32x4*)(Hg + (size_t)row * D * 2 + c * 16));
        }
    }
}

}  // namespace ge2e
