// libge2e_hip.so: C-ABI entry points (include/ge2e_hip.h) and the launch sequence of the GE2E hot path.
// One encoder forward = weight prep + mel pack + prenet GEMM + per full layer {in_proj GEMM, fused attention, out_proj GEMM + LN,
// chained FFN (16-bit modes) | FFN1 GEMM, FFN2 GEMM + LN} + the last layer on one row per utterance (single-query attention without
// K / V, attn_last.cuh) + tail; the backward mirrors it (chained FFN backward, norm1 backward + dO, attention backward, dgrad GEMM per
// full layer) with the weight gradients on an internal side stream.  Everything else is enqueued on the caller's stream.
#include "../../include/ge2e_hip.h"

#include <hip/hip_ext.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "attention.cuh"
#include "attn_last.cuh"
#include "attn_sub.cuh"
#include "prenet_bwd.cuh"
#include "gemm.cuh"
#include "gemm_ws.cuh"
#include "gemm_kl.cuh"
#include "gemm_sk.cuh"
#include "ffn.cuh"
#include "lastc.cuh"
#include "wgrad_ks.cuh"
#include "misc.cuh"
#include "melfront.cuh"

using namespace ge2e;

namespace {

constexpr int MAX_LAYERS = 8;
constexpr int MAX_KT = 9;                 // sequence-resident attention instantiated for T <= 32 * MAX_KT = 288 frames
constexpr int WK_MAX_BLOCKS = 256;        // blocks (= partial tiles) of one wgrad_ks launch: one per CU at most (sizes the partial buffer)
constexpr int WK_DEFAULT_BLOCKS = 160;    // what a launch uses by default: see launch_wgrad
constexpr int MAX_FRAMES = 1024;          // longer sequences stream through the chunked kernels (attention.cuh), up to here

struct ParamInfo { std::string name; int64_t numel, offset; };

// Development / diagnostic options (ge2e_set_option).  The library never reads the process environment: the defaults below ARE the
// shipped configuration, an option is process-wide and is read at every call that depends on it (nothing is latched on first use).
// The ctypes loader forwards GE2E_<NAME> environment variables here only when GE2E_DEV_SWITCHES=1 (tools/ab.sh, tools/switch_test.sh).
enum Opt {
    O_NO_OVERLAP, O_NO_WS_GEMM, O_NO_KL_GEMM, O_NO_LNFUSE, O_NO_SK_GEMM, O_NO_FFN_CHAIN, O_FFN_WV, O_NO_FFN_CHAIN_BWD, O_NO_WGRAD_KS,
    O_NO_REDUCE_BATCH, O_WGRAD_KS_BLOCKS, O_NO_EVENT_BIND, O_NO_MASKBITS, O_NO_COLSUM_END, O_NO_PRENET_FUSE, O_NO_LAST_CHAIN, O_ATTN_SUB,
    O_DEBUG_BWD_STOP, O_DEBUG_SIDE_DELAY_US, O_COUNT
};
struct OptDef { const char* name; int def; };
constexpr OptDef OPT_DEFS[O_COUNT] = {
    {"no_overlap", 0},          // weight gradients on the caller's stream (no side stream); read by ge2e_create
    {"no_ws_gemm", 0},          // tiled kernel instead of the weight-stationary K = 256 GEMM
    {"no_kl_gemm", 0},          // tiled kernel instead of the stage-stream K = 1024 GEMM
    {"no_lnfuse", 0},           // norm1 backward as its own launch
    {"no_sk_gemm", 0},          // the last layer's K = 1024 products on the stage-stream / tiled kernels
    {"no_ffn_chain", 0},        // FFN as two GEMM launches
    {"ffn_wv", 0},              // 4 / 8: force one block shape of the chained FFN forward
    {"no_ffn_chain_bwd", 0},    // norm2 backward, dF, dH1 as three launches
    {"no_wgrad_ks", 0},         // 128 x 128 atomic weight-gradient kernel instead of wgrad_ks
    {"no_reduce_batch", 0},     // one reduce pass per weight-gradient product instead of one per layer
    {"wgrad_ks_blocks", 0},     // blocks per wgrad_ks launch (0: WK_DEFAULT_BLOCKS)
    {"no_event_bind", 0},       // fences recorded as separate markers
    {"no_maskbits", 0},         // dF reads the stored hidden as its mask
    {"no_colsum_end", 0},       // norm2 column sums on the weight-gradient stream
    {"no_prenet_fuse", 0},      // prenet backward as recompute GEMM + weight-gradient launch
    {"no_last_chain", 0},       // the last layer after its attention + the tail as five / six launches instead of one each way (lastc.cuh)
    {"attn_sub", 0},            // 1: in_proj, attention, out_proj + LayerNorm of a full layer as ONE launch per utterance (attn_sub.cuh; measured slower, off)
    {"debug_bwd_stop", -1},     // >= 0: backward returns after k layers (parity tests read that layer's scratch through ge2e_debug_tap)
    {"debug_side_delay_us", 0}, // tests: hold the weight-gradient stream back after every fork
};
static_assert(sizeof(OPT_DEFS) / sizeof(OPT_DEFS[0]) == O_COUNT, "one definition per option, in enum order");
std::atomic<int> g_opt[O_COUNT];
struct OptInit { OptInit() { for (int i = 0; i < O_COUNT; ++i) g_opt[i].store(OPT_DEFS[i].def, std::memory_order_relaxed); } } g_opt_init;
inline int opt(int o) { return g_opt[o].load(std::memory_order_relaxed); }

}  // namespace

struct ProfRec { int klass; hipEvent_t start, stop; double work, bytes; int counts; };

struct ge2e_handle_s {
    ge2e_config cfg;
    std::vector<ParamInfo> params;
    int64_t total = 0;
    std::mutex mu;
    std::string err;
    int prof_mask = 0;                     // bench.py roofline leg (ge2e_profile_enable)
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;
    // backward overlap: weight-gradient GEMMs run on an internal side stream, fenced with events
    hipStream_t side = nullptr;
    // One EventSet per in-flight backward.  A set is taken again only after its `done` event -- recorded on the
    // caller's stream after the final join of that backward -- has completed, so no fence event is ever re-recorded
    // while a stream may still have to wait on its previous record (a host thread can run several steps ahead).
    struct EventSet { std::vector<hipEvent_t> ev; hipEvent_t done = nullptr; bool used = false; };
    std::vector<EventSet*> event_sets;
    int overlap = 1;                       // GE2E_NO_OVERLAP=1 turns the side stream off
    int num_cus = 0;                       // of the handle's device, queried at the first persistent launch
    int device = -1;                       // device of the first GPU call; later calls on another current device are refused
    // What the latest training forwards left in their workspaces that the backward's kernel choice depends on (options are read at
    // every call and may change between a forward and its backward): did the FFN leave its ReLU / dropout mask as bits?
    struct FwdNote { const void* ws = nullptr; bool bits = false; };
    FwdNote fwd_notes[8];
    int fwd_note_next = 0;
};

namespace {

int fail(ge2e_handle h, int code, const std::string& msg) {
    if (h) { std::lock_guard<std::mutex> g(h->mu); h->err = msg; }
    return code;
}
int fail_hip(ge2e_handle h, hipError_t e, const char* what) {
    (void)hipGetLastError();       // the error is reported through the return code: do not leave it sticky for later launches
    return fail(h, (int)e, std::string(what) + ": " + hipGetErrorString(e));
}

// parameter table indices (reference GE2E.parameters() order)
enum { P_PRENET_W = 0, P_PRENET_B = 1, P_ALPHA = 2, P_LAYER0 = 3 };
enum { L_IN_W = 0, L_IN_B, L_OUT_W, L_OUT_B, L_L1_W, L_L1_B, L_L2_W, L_L2_B, L_N1_W, L_N1_B, L_N2_W, L_N2_B, L_COUNT };
inline int lp(int l, int which) { return P_LAYER0 + L_COUNT * l + which; }
inline int p_fn_w(const ge2e_config& c) { return P_LAYER0 + L_COUNT * c.layers; }
inline int p_fn_b(const ge2e_config& c) { return p_fn_w(c) + 1; }
inline int p_proj_w(const ge2e_config& c) { return p_fn_w(c) + 2; }
inline int p_proj_b(const ge2e_config& c) { return p_fn_w(c) + 3; }

void build_params(ge2e_handle h) {
    const ge2e_config& c = h->cfg;
    auto add = [&](const std::string& n, int64_t numel) { h->params.push_back({n, numel, h->total}); h->total += numel; };
    const int64_t d = c.emb, f = c.ffn;
    add("prenet.weight", d * c.mel_dim); add("prenet.bias", d); add("positional_encoding.alpha", 1);
    for (int l = 0; l < c.layers; ++l) {
        const std::string p = "transformer.layers." + std::to_string(l) + ".";
        add(p + "self_attn.in_proj_weight", 3 * d * d); add(p + "self_attn.in_proj_bias", 3 * d);
        add(p + "self_attn.out_proj.weight", d * d); add(p + "self_attn.out_proj.bias", d);
        add(p + "linear1.weight", f * d); add(p + "linear1.bias", f);
        add(p + "linear2.weight", d * f); add(p + "linear2.bias", d);
        add(p + "norm1.weight", d); add(p + "norm1.bias", d);
        add(p + "norm2.weight", d); add(p + "norm2.bias", d);
    }
    add("transformer.norm.weight", d); add("transformer.norm.bias", d);
    add("projection.weight", d * d); add("projection.bias", d);
}

// ------------------------------------------------------------------------------------------ workspace
struct Layout {
    size_t esz = 0;
    int R = 0, KP = 0;
    size_t w_prenet = 0, w_in[MAX_LAYERS], w_inT[MAX_LAYERS], w_out[MAX_LAYERS], w_outT[MAX_LAYERS];
    size_t w_l1[MAX_LAYERS], w_l1T[MAX_LAYERS], w_l2[MAX_LAYERS], w_l2T[MAX_LAYERS];
    size_t wqT = 0, pe_t = 0, h0 = 0;
    size_t xt = 0;               // packed mel: [R][KP] of T (row-major copy of the channels-first fp32 input)
    size_t pbits = 0;            // [R][32] sign bits of the prenet's pre-activation (train: written by its forward epilogue, read by prenet_bwd_kernel)
    size_t qkv[MAX_LAYERS], o[MAX_LAYERS], h1[MAX_LAYERS], rstd1[MAX_LAYERS], f[MAX_LAYERS], h2[MAX_LAYERS], rstd2[MAX_LAYERS];
    size_t lse[MAX_LAYERS];      // [R, heads] fp32 log-sum-exp of the attention scores (train only)
    size_t skpart;               // 16-bit modes: [SK_KS][n][256] fp32 partial tiles of the last layer's split-K products (gemm_sk.cuh), or -1
    size_t fbits[MAX_LAYERS];    // [R, ffn / 8] bytes: "stored FFN hidden > 0", one bit per element (train, 16-bit modes, full layers;
                                 // written by the chained FFN kernel, read by the backward's dF GEMM instead of the hidden itself)
    size_t adelta = 0;           // [R, heads] fp32 dO . O of the layer in backward (long-sequence attention only)
    // last layer (attn_last.cuh: one query per utterance, K and V are never materialised): q of frame 0 [n, 256] of T; train mode also keeps
    // qk = Wk_h^T q0_h [n, 4, 256], the probabilities [n, 4, T], ctx = sum_t pd_t x_t [n, 4, 256], sum_t pd_t [n, 4] (all fp32) and, in
    // backward, dqk [n, 4, 256]
    size_t lq0 = 0, lqk = 0, lprob = 0, lctx = 0, lsp = 0, ldqk = 0;
    size_t wpart = 0;            // split-K partial tiles of the 256 x 256 weight-gradient kernel (16-bit modes): WK_MAX_JOBS slabs (one per product of a layer)
    size_t wslab = 0;            // floats per slab: as many 256 x 256 tiles as a launch at this row count can have blocks
    size_t xhat_f = 0, rstd_f = 0, zm = 0, nrm = 0, emb_keep = 0, d_raw = 0;
    size_t dHb = 0, dO = 0;
    // dL/d(output of layer l) lives in dHx[(l + 1) % nH] (l = -1: the prenet output).  The weight-gradient stream reads it too (the
    // norm2 column sums, ln_colsum_kernel), and the layer's last dgrad GEMM writes the NEXT buffer of the rotation: with three buffers a
    // stack of up to 3 layers never waits for that reader (deeper stacks wait for the reader three layers up).
    int nH = 1;
    size_t dHx[3] = {0, 0, 0};
    size_t dH_of(int l) const { return dHx[(l + 1) % nH]; }
    // Scratch the side stream's weight-gradient kernels read after the main chain has moved on: norm2-backward outputs (dP, dM),
    // norm1-backward outputs (dP2, dM2), dF and dQKV.  Layer l uses set l % nset (dQKV: l % nqkv), so with up to 3 layers the main
    // chain never writes a buffer a weight gradient of the same backward may still be reading -- no main-stream wait (each is a
    // barrier packet: ~6 us of idle queue even when already satisfied); deeper stacks wait for the reader two (three) layers up.
    int nset = 1, nqkv = 1;
    size_t dP[2] = {0, 0}, dM[2] = {0, 0}, dF[2] = {0, 0}, dQKV[3] = {0, 0, 0};
    size_t dP2[2] = {0, 0}, dM2[2] = {0, 0}, c_dP2 = 0, c_dM2 = 0;
    // last layer: only frame 0 of its output is consumed, so everything after its K/V projection lives on
    // COMPACT rows (one per utterance): o/h1/f/h2 of that layer and this backward scratch
    size_t c_dH = 0, c_dHb = 0, c_dP = 0, c_dM = 0, c_dF = 0, c_dO = 0, c_dQ0 = 0;
    size_t total = 0;
};

Layout build_layout(const ge2e_config& c, int n, int t, int train, int samples_max = 1) {
    (void)samples_max;
    Layout L;
    L.esz = (c.precision == GE2E_PREC_F32 || c.precision == GE2E_PREC_F32X3) ? 4 : 2;
    L.KP = 128;                  // mel_dim <= 128, padded to one 128-column operand tile
    L.R = n * t;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t d = c.emb, f = c.ffn, R = (size_t)L.R, e = L.esz;
    L.w_prenet = take(d * L.KP * e);
    for (int l = 0; l < c.layers; ++l) {
        L.w_in[l] = take(3 * d * d * e);  L.w_inT[l] = take(3 * d * d * e);
        L.w_out[l] = take(d * d * e);     L.w_outT[l] = take(d * d * e);
        L.w_l1[l] = take(f * d * e);      L.w_l1T[l] = take(f * d * e);
        L.w_l2[l] = take(f * d * e);      L.w_l2T[l] = take(f * d * e);
    }
    L.wqT = take(d * d * 4);
    L.pe_t = take((size_t)t * d * 4);
    L.xt = take(R * (size_t)L.KP * e);
    L.h0 = take(R * d * e);
    L.pbits = train ? take(R * (d / 8)) : (size_t)-1;
    const int last = c.layers - 1;
    if (train) {
        for (int l = 0; l < c.layers; ++l) {
            const size_t Rl = l == last ? (size_t)n : R;
            L.qkv[l] = l == last ? (size_t)-1 : take(R * 3 * d * e); L.o[l] = take(Rl * d * e);
            L.h1[l] = take(Rl * d * e);     L.rstd1[l] = take(Rl * 4);
            L.f[l] = take(Rl * f * e);      L.h2[l] = take(Rl * d * e);
            L.rstd2[l] = take(Rl * 4);
            L.lse[l] = l == last ? (size_t)-1 : take(R * (size_t)c.heads * 4);
            L.fbits[l] = (l == last || e != 2 || f % 128 != 0) ? (size_t)-1 : take(R * (f / 8));
        }
    } else {        // eval: layers reuse one set of buffers; h2 overwrites the layer input
        const size_t qkv = c.layers > 1 ? take(R * 3 * d * e) : (size_t)-1;
        size_t o = 0, h1 = 0, ff = 0;
        if (c.layers > 1) { o = take(R * d * e); h1 = take(R * d * e); ff = take(R * f * e); }
        for (int l = 0; l < last; ++l) {
            L.qkv[l] = qkv; L.o[l] = o; L.h1[l] = h1; L.f[l] = ff; L.h2[l] = L.h0;
            L.rstd1[l] = L.rstd2[l] = L.lse[l] = L.fbits[l] = (size_t)-1;
        }
        L.lse[last] = L.fbits[last] = (size_t)-1;
        L.qkv[last] = (size_t)-1; L.o[last] = take((size_t)n * d * e); L.h1[last] = take((size_t)n * d * e);
        L.f[last] = take((size_t)n * f * e); L.h2[last] = take((size_t)n * d * e);
        L.rstd1[last] = L.rstd2[last] = (size_t)-1;
    }
    L.skpart = e == 2 && n <= SK_MAX_M ? take((size_t)SK_KS * n * 256 * 4) : (size_t)-1;
    L.lq0 = take((size_t)n * d * e);
    if (train) {
        L.lqk = take((size_t)n * 4 * d * 4); L.lprob = take((size_t)n * 4 * (size_t)t * 4); L.lctx = take((size_t)n * 4 * d * 4);
        L.lsp = take((size_t)n * 4 * 4); L.ldqk = take((size_t)n * 4 * d * 4);
    }
    L.xhat_f = take((size_t)n * d * 4); L.rstd_f = take((size_t)n * 4);
    L.zm = take((size_t)n * d * 4);     L.nrm = take((size_t)n * 4);
    L.emb_keep = take((size_t)n * d * 4); L.d_raw = take((size_t)n * d * 4);
    if (train) {
        L.nH = std::max(1, std::min(c.layers, 3));
        for (int s = 0; s < L.nH; ++s) L.dHx[s] = take(R * d * e);
        L.dHb = take(R * d * e); L.dO = take(R * d * e);
        L.nset = c.layers >= 3 ? 2 : 1;              // full-size layers are 0 .. layers - 2 (the last one runs on compact rows)
        L.nqkv = std::min(c.layers, 3);              // dQKV is full-size in every layer
        for (int s = 0; s < L.nset; ++s) {
            L.dP[s] = take(R * d * e); L.dM[s] = take(R * d * e); L.dF[s] = take(R * f * e);
            L.dP2[s] = take(R * d * e); L.dM2[s] = take(R * d * e);
        }
        for (int s = 0; s < L.nqkv; ++s) L.dQKV[s] = take(R * 3 * d * e);
        L.adelta = t > 32 * MAX_KT ? take(R * (size_t)c.heads * 4) : (size_t)-1;
        {   // a launch has tiles x slices blocks, slices <= rows / 256 (8 stages of 32 rows per block at least) and <= WK_MAX_BLOCKS / tiles
            const size_t tiles = std::max<size_t>(3, std::min<size_t>(16, f / 256));
            L.wslab = std::min<size_t>(WK_MAX_BLOCKS, tiles * std::max<size_t>(1, R / 256)) * WK_TILE_FLOATS;
        }
        L.wpart = e == 2 ? take((size_t)WK_MAX_JOBS * L.wslab * 4) : (size_t)-1;
        const size_t nn = (size_t)n;
        L.c_dP2 = take(nn * d * e); L.c_dM2 = take(nn * d * e);
        L.c_dH = take(nn * d * e); L.c_dHb = take(nn * d * e); L.c_dP = take(nn * d * e); L.c_dM = take(nn * d * e);
        L.c_dF = take(nn * f * e); L.c_dO = take(nn * d * e); L.c_dQ0 = take(nn * d * e);
    }
    L.total = off;
    return L;
}

Drop make_drop(bool active, float p, uint64_t seed, uint64_t step, int site) {
    Drop d{0u, 0u, 1.0f};
    if (!active || p <= 0.0f) return d;
    d.key = ge2e_drop_key(seed, step, site);
    d.thr = (uint32_t)std::floor((double)p * 65536.0);
    d.scale = (float)(1.0 / (1.0 - (double)p));
    return d;
}
enum { SITE_PE = 0 };
inline int site_attn(int l) { return 1 + 4 * l; }
inline int site_sa(int l) { return 2 + 4 * l; }
inline int site_ffh(int l) { return 3 + 4 * l; }
inline int site_ff(int l) { return 4 + 4 * l; }

// brackets one launch with events on its stream when its class is being profiled
struct ProfScope {
    ge2e_handle h; hipStream_t st; bool on = false; ProfRec rec{};
    ProfScope(ge2e_handle h_, hipStream_t st_, int klass, double work, double bytes = 0.0, bool counts = true) : h(h_), st(st_) {
        if (!(h->prof_mask & klass)) return;
        auto get = [&]() { hipEvent_t e = nullptr;
            if (!h->ev_pool.empty()) { e = h->ev_pool.back(); h->ev_pool.pop_back(); } else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
            return e; };
        std::lock_guard<std::mutex> g(h->mu);
        rec.klass = klass; rec.work = work; rec.bytes = bytes; rec.counts = counts ? 1 : 0; rec.start = get(); rec.stop = get();
        on = rec.start && rec.stop;
        if (on) hipEventRecord(rec.start, st);
    }
    ~ProfScope() {
        if (!on) return;
        hipEventRecord(rec.stop, st);
        std::lock_guard<std::mutex> g(h->mu);
        h->prof.push_back(rec);
    }
};

// ------------------------------------------------------------------------------------------ launches
// Dynamic LDS above 48 KB needs the function attribute raised first.  The largest size ever requested is remembered per call
// site (a call site launches one kernel instantiation); a later, larger request (runtime-sized LDS) raises it again.
//
// Armed launches (SideCtx::arm): while a fence event is armed for a stream, every kernel launched on that stream carries it as its
// completion ("stop") event, so the event completes with the LAST such kernel and no separate marker packet follows it in the
// queue.  A recorded marker costs the main chain ~6 us of idle queue per fence (rocprofv3 timeline); a bound event costs nothing.
// tl_armed is thread-local (a backward runs on ONE host thread) and is only ever non-empty between SideCtx::arm() and the
// SideCtx::fork() that follows it inside backward_body -- never across a C-ABI call and never while a bucket callback runs
// (callbacks are invoked after a fork(), which clears it), so a re-entrant call from the callback sees it empty.
struct ArmedEvent { hipEvent_t ev = nullptr; hipStream_t on_stream = nullptr; int launches = 0; };
static thread_local ArmedEvent tl_armed;

// tests only (GE2E_DEBUG_SIDE_DELAY_US): holds the weight-gradient stream back after every fork so that the main chain reaches
// its "wait for the reader of this scratch buffer" fences while the reader really is still running.  100 MHz realtime ticks;
// the loop ends when the time is up, whatever else the chip does.
__global__ void debug_delay_kernel(unsigned ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(64);
}
inline int debug_side_delay_us() { return std::max(0, std::min(opt(O_DEBUG_SIDE_DELAY_US), 5000)); }
#define GE2E_LAUNCH(h, kern, grid, block, smem, st, ...)                                                   \
    do {                                                                                                    \
        static std::atomic<size_t> attr_max{(size_t)48 * 1024};                                             \
        if ((size_t)(smem) > attr_max.load(std::memory_order_relaxed)) {                                    \
            hipError_t ea = hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(smem)); \
            if (ea != hipSuccess) return fail_hip(h, ea, "hipFuncSetAttribute " #kern);                     \
            attr_max.store((size_t)(smem), std::memory_order_relaxed);                                      \
        }                                                                                                   \
        if (tl_armed.ev && tl_armed.on_stream == (st)) {                                                           \
            hipExtLaunchKernelGGL(kern, grid, block, smem, st, nullptr, tl_armed.ev, 0, __VA_ARGS__);       \
            ++tl_armed.launches;                                                                            \
        } else                                                                                              \
        hipLaunchKernelGGL(kern, grid, block, smem, st, __VA_ARGS__);                                       \
        hipError_t el = hipGetLastError();                                                                  \
        if (el != hipSuccess) return fail_hip(h, el, #kern);                                                \
    } while (0)

template <typename T, int BM, int BN, int WM, int WN, int EPI, int ALOAD, int NBUF = 2, bool X3 = false>
int launch_gemm(ge2e_handle h, hipStream_t st, const GemmArgs& a) {
    constexpr int BK = 128 / (int)sizeof(T);
    if (a.K % BK != 0 || a.N % BN != 0 || a.M <= 0)
        return fail(h, GE2E_EUNSUPPORTED, "gemm: N must be a multiple of the tile and K of the k-step");
    const int grid = ((a.M + BM - 1) / BM) * (a.N / BN);
    const size_t smem = std::max<size_t>(NBUF * (size_t)(BM + BN) * 128, EPI == EPI_LN ? (size_t)BM * (BN + 4) * 4 : (size_t)BM * (BN * sizeof(T) + 16));
    auto kern = gemm_nt_kernel<T, BM, BN, WM, WN, EPI, ALOAD, NBUF, (X3 && sizeof(T) == 4)>;
    constexpr bool reads_r = (EPI == EPI_LN || EPI == EPI_MASK || EPI == EPI_ADD || EPI == EPI_PRENET_BWD);
    const double kk = a.K;
    // algorithmic HBM bytes of one launch: activations in (fp32 mel for the prenet) + weights + tile out (+ tile in)
    const double abytes = (double)a.M * kk * (double)sizeof(T) + (double)a.N * kk * sizeof(T) +
                          (double)a.M * a.N * sizeof(T) * (reads_r ? 2.0 : 1.0);
    ProfScope ps(h, st, EPI == EPI_LN ? GE2E_K_GEMM_LN : GE2E_K_GEMM, 2.0 * a.M * a.N * kk, abytes);
    GE2E_LAUNCH(h, kern, dim3(grid), dim3(256), smem, st, a);
    return 0;
}
// Short-K bf16 products at full height run on the weight-stationary streaming kernel (gemm_ws.cuh); the fp32 parity
// path and other K keep the tiled kernel.  The choice must not depend on M: a row's result is then independent of the
// batch it sits in (the two kernels round LayerNorm statistics differently).
template <typename T, int EPI>
constexpr bool ws_epilogue() {
    // not EPI_NONE (dO = dA.Wo, backward only): no faster alone (31 vs 33 us), and its 512 persistent blocks cannot share a CU's
    // LDS with a weight-gradient kernel still running on the side stream -- measured 163 us when that happens
    return sizeof(T) == 2 && (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU_DROP || EPI == EPI_MASK || EPI == EPI_LN);
}
inline bool ws_shape(const GemmArgs& a) {
    return !opt(O_NO_WS_GEMM) && a.K == 256 && a.N % 256 == 0 && a.M > 0 && a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldc % 8 == 0;
}
// Row partitions of the prenet's weight-stationary launch: a multiple of 8 (XCD mapping) and of T / gcd(T, 16) (then every 16-row tile of a block starts
// at the same frame offset), the multiple nearest 480 blocks; 0 when no such count is <= 1024 (the tiled kernel then takes the product)
inline int prenet_ws_parts(int t) {
    if (t <= 0) return 0;
    auto gcd = [](long long a, long long b) { while (b) { const long long r = a % b; a = b; b = r; } return a; };
    const long long pd = t / gcd(t, 16), unit = pd / gcd(pd, 8) * 8;
    if (unit > 1024) return 0;
    long long k = (480 + unit / 2) / unit; if (k < 1) k = 1;
    while (k * unit > 1024) --k;
    return (int)(k * unit);
}
template <typename T, int EPI, int KK = 256>
int launch_gemm_ws(ge2e_handle h, hipStream_t st, const GemmArgs& a) {
    constexpr bool reads_r = (EPI == EPI_LN || EPI == EPI_MASK);
    if (EPI == EPI_MASKBITS && (a.N % 128 != 0 || a.ldr % 16 != 0)) return fail(h, GE2E_EUNSUPPORTED, "gemm_ws: mask-bit rows are whole 16-byte words");
    if (EPI == EPI_LN && a.N != 256) return fail(h, GE2E_EUNSUPPORTED, "gemm_ws: the LayerNorm epilogue needs N == 256");
    if (reads_r && a.ldr % 8 != 0) return fail(h, GE2E_EUNSUPPORTED, "gemm_ws: the addend rows must be 16-byte aligned");
    const int cg = a.N / 256, ntiles = (a.M + 15) / 16;
    int parts = (512 / cg) / 8 * 8;                       // two resident blocks per CU
    if (parts > (ntiles + 7) / 8 * 8) parts = (ntiles + 7) / 8 * 8;
    if (parts < 8) parts = 8;
    if constexpr (EPI == EPI_PRENET) {
        parts = prenet_ws_parts(a.T);                     // 16 parts a multiple of T (the kernel keeps its positional rows in registers)
        if (parts <= 0 || (16LL * parts) % a.T != 0 || cg != 1) return fail(h, GE2E_EUNSUPPORTED, "gemm_ws: no frame-aligned partition for the prenet");
    }
    const double abytes = 2.0 * ((double)a.M * a.K + (double)a.N * a.K + (double)a.M * a.N * (reads_r ? 2.0 : 1.0)) +
                          (EPI == EPI_MASKBITS || EPI == EPI_PRENET ? (double)a.M * a.N / 8.0 : 0.0);
    ProfScope ps(h, st, EPI == EPI_LN ? GE2E_K_GEMM_LN : GE2E_K_GEMM, 2.0 * a.M * a.N * (double)a.K, abytes);
    auto kern = gemm_ws_kernel<T, EPI, KK>;
    GE2E_LAUNCH(h, kern, dim3(cg * parts), dim3(256), (gemm_ws_smem<EPI, KK>()), st, a, parts, ntiles);
    return 0;
}

// LayerNorm backward + the K = 256 GEMM that consumes it, one launch (gemm_ws_lnbwd_kernel).  a.A = dy, a.gamma / beta /
// rstd / drop / drow_mul describe the LayerNorm and the dropout in front of the sub-layer, f the rest.
inline bool lnfuse_on() { return !opt(O_NO_LNFUSE); }
template <typename T, int EPI>
int launch_gemm_ws_lnbwd(ge2e_handle h, hipStream_t st, const GemmArgs& a, const LnFuseArgs& f) {
    const int cg = a.N / 256, ntiles = (a.M + 15) / 16;
    int parts = (512 / cg) / 8 * 8;
    if (parts > (ntiles + 7) / 8 * 8) parts = (ntiles + 7) / 8 * 8;
    if (parts < 8) parts = 8;
    const double abytes = 2.0 * ((double)a.M * 256 * (f.dmask ? 4.0 : 3.0) + (double)a.N * a.K + (double)a.M * a.N * (EPI == EPI_MASK ? 2.0 : 1.0));
    ProfScope ps(h, st, GE2E_K_GEMM, 2.0 * a.M * a.N * (double)a.K, abytes);
    auto kern = gemm_ws_lnbwd_kernel<T, EPI>;
    GE2E_LAUNCH(h, kern, dim3(cg * parts), dim3(256), (gemm_ws_lnbwd_smem<EPI>()), st, a, f, parts, ntiles);
    return 0;
}

template <typename T, int EPI, int ALOAD = ALOAD_ROW, bool X3 = false>
int gemm128(ge2e_handle h, hipStream_t st, const GemmArgs& a) {
    if constexpr (ws_epilogue<T, EPI>() && ALOAD == ALOAD_ROW) {
        if (ws_shape(a)) return launch_gemm_ws<T, EPI>(h, st, a);
    }
    // the VALU-heavy ReLU + dropout epilogue gains from a third resident block (single LDS stage: 35 KB), measured
    // 179 -> 144 us for FFN1; the other epilogues measure the same either way and keep the one-barrier double buffer
    constexpr int NBUF = (EPI == EPI_BIAS_RELU_DROP) ? 1 : 2;
    return launch_gemm<T, 128, 128, 64, 64, EPI, ALOAD, NBUF, X3>(h, st, a);
}
// K = 1024 products on the last layer's compact rows: split-K over 64-row pieces + a reduce / epilogue launch (gemm_sk.cuh).  part: [SK_KS][M][256] fp32.
inline bool gemm_sk_on() { return !opt(O_NO_SK_GEMM); }
template <typename T, bool LN>
int launch_gemm_sk(ge2e_handle h, hipStream_t st, const GemmArgs& a, float* part) {
    if constexpr (sizeof(T) == 2) {
        GemmSkArgs k{};
        k.A = a.A; k.lda = a.lda; k.W = a.W; k.ldw = a.ldw; k.part = part; k.M = a.M; k.K = a.K; k.KS = SK_KS;
        ProfScope ps(h, st, LN ? GE2E_K_GEMM_LN : GE2E_K_GEMM, 2.0 * a.M * a.N * (double)a.K, 2.0 * ((double)a.M * a.K + (double)a.N * a.K + 2.0 * (double)a.M * a.N));
        auto kern = gemm_sk_kernel<T>;
        GE2E_LAUNCH(h, kern, dim3((a.M + 63) / 64, SK_KS), dim3(256), 0, st, k);
        SkEpiArgs e{};
        e.part = part; e.M = a.M; e.KS = SK_KS; e.bias = a.bias; e.R = a.R; e.ldr = a.ldr; e.C = a.C; e.ldc = a.ldc;
        e.gamma = a.gamma; e.beta = a.beta; e.rstd = a.rstd; e.eps = a.eps; e.drop = a.drop; e.drow_mul = a.drow_mul;
        auto epi = gemm_sk_epi_kernel<T, LN>;
        GE2E_LAUNCH(h, epi, dim3((a.M + 3) / 4), dim3(256), 0, st, e);
        return 0;
    } else return fail(h, GE2E_EUNSUPPORTED, "gemm_sk: 16-bit storage modes only");
}
template <typename T>
bool gemm_sk_shape(const GemmArgs& a, size_t skpart) {
    return sizeof(T) == 2 && gemm_sk_on() && skpart != (size_t)-1 && a.N == 256 && a.K % (32 * SK_KS) == 0 && a.K >= 512 && a.M > 0 && a.M <= SK_MAX_M &&
           a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldc % 4 == 0 && a.ldr % 4 == 0;
}
template <typename T, bool X3 = false>
int gemm_ln(ge2e_handle h, hipStream_t st, const GemmArgs& a) {
    if constexpr (ws_epilogue<T, EPI_LN>()) {
        if (ws_shape(a)) return launch_gemm_ws<T, EPI_LN>(h, st, a);
        // FFN2 + LayerNorm (K = 1024): persistent stage-stream kernel, one 512-thread block per CU (gemm_kl.cuh).  It needs a
        // whole CU's LDS, so it is used where nothing shares the machine (the forward chain), not beside the weight gradients.
        if (!opt(O_NO_KL_GEMM) && a.K == 1024 && a.N == 256 && a.M > 0 && a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldc % 8 == 0 && a.ldr % 8 == 0) {
            const int ntiles = (a.M + 127) / 128;
            if (h->num_cus <= 0) {
                int dev = h->device >= 0 ? h->device : 0, n = 0;
                if ((h->device < 0 && hipGetDevice(&dev) != hipSuccess) || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
                h->num_cus = n;
            }
            const int grid = std::min(h->num_cus, ntiles);
            const double abytes = 2.0 * ((double)a.M * a.K + (double)a.N * a.K + 2.0 * (double)a.M * a.N);
            ProfScope ps(h, st, GE2E_K_GEMM_LN, 2.0 * a.M * a.N * (double)a.K, abytes);
            auto kern = gemm_kl_kernel<T, EPI_LN, 1024>;
            GE2E_LAUNCH(h, kern, dim3(grid), dim3(512), (gemm_kl_smem<EPI_LN>()), st, a, ntiles);
            return 0;
        }
    }
    return launch_gemm<T, 64, 256, 32, 128, EPI_LN, ALOAD_ROW, 2, X3>(h, st, a);
}

// FFN1 -> ReLU -> dropout -> FFN2 -> dropout -> residual -> LayerNorm in one launch (ffn.cuh): 16-bit modes, full-height layers.
// The choice never depends on M (a row's result must not depend on the batch it sits in): every non-last layer takes it.
inline bool ffn_chain_on() { return !opt(O_NO_FFN_CHAIN); }
inline void note_forward(ge2e_handle h, const void* ws, bool bits) {
    std::lock_guard<std::mutex> g(h->mu);
    for (auto& n : h->fwd_notes) if (n.ws == ws) { n.bits = bits; return; }
    h->fwd_notes[h->fwd_note_next] = {ws, bits};
    h->fwd_note_next = (h->fwd_note_next + 1) % 8;
}
// the backward's view: the note of this workspace's forward, or (more than 8 forwards ago) the option as it stands now
inline bool forward_left_bits(ge2e_handle h, const void* ws) {
    std::lock_guard<std::mutex> g(h->mu);
    for (auto& n : h->fwd_notes) if (n.ws == ws) return n.bits;
    return ffn_chain_on();
}
template <typename T>
int launch_ffn_chain(ge2e_handle h, hipStream_t st, const FfnArgs& a) {
    if constexpr (sizeof(T) != 2) return fail(h, GE2E_EUNSUPPORTED, "ffn chain: 16-bit modes only");
    else {
        if (h->num_cus <= 0) {
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, h->device >= 0 ? h->device : 0) != hipSuccess || n <= 0) n = 256;
            h->num_cus = n;
        }
        // Block shape (ffn.cuh).  In steady state the 8-wave block (one per CU, staggered waves) gets more rows per CU and
        // microsecond -- eval 3.9 vs 3.7, train 3.2 vs 2.7 -- but its 256-row passes quantise: 600 passes (the headline batch) on
        // 256 CUs take three rounds with a quarter of the chip idle, where 1200 half-size passes on 512 half-CU slots flow
        // (measured alone: eval 160 vs 195 us, train 224 vs 238 us; 768,000 rows, 3000 passes: 826 vs 804 us).  Take the 8-wave
        // shape when its rounds are full enough to keep its per-CU advantage.  Option ffn_wv = 4 / 8 forces one (eval).
        const int wv_env = opt(O_FFN_WV);
        const int np8 = (a.M + 255) / 256, rounds8 = (np8 + h->num_cus - 1) / h->num_cus;
        const double fill8 = (double)np8 / ((double)rounds8 * h->num_cus);
        // (train mode always takes the 4-wave shape: its 8-wave instantiation was the one kernel of the library with scratch -- 36 bytes per lane in
        // the pass prologue -- for ~3 % on batches whose 256-row passes fill the chip, e.g. configs[4]: 804 vs 826 us at 768,000 rows)
        const int wv = a.Fo ? 4 : (wv_env == 4 || wv_env == 8 ? wv_env : (fill8 >= 0.95 ? 8 : 4));
        const int rows_pass = wv == 4 ? 128 : 256, slots = wv == 4 ? 2 * h->num_cus : h->num_cus;
        const int npass = (a.M + rows_pass - 1) / rows_pass;
        // every block runs ceil(npass / slots) passes: a grid of ceil(npass / that) blocks finishes at the same time as a full
        // grid would, with fewer blocks competing for the weight stream (600 passes: 200 blocks x 3 measured 189 us, 256 blocks 215 us)
        // (4-wave blocks: a full grid; the blocks with one pass fewer free their half CU for the others -- measured 224 vs 248 us)
        const int per_block = (npass + slots - 1) / slots;
        const int grid = wv == 4 ? std::min(npass, slots) : (npass + per_block - 1) / per_block;
        const double rows = a.M;
        const double abytes = 2.0 * (rows * 256 * 2 + (a.Fo ? rows * FFN_F : 0.0) + 2.0 * 256 * FFN_F) + (a.Fo ? rows * FFN_F / 8 : 0.0);
        ProfScope ps(h, st, GE2E_K_FFN, 2.0 * rows * 256 * FFN_F * 2.0, abytes);
        if (wv == 4) {
            if (a.Fo) { auto kern = ffn_chain_kernel<T, true, 0, false, 4>; GE2E_LAUNCH(h, kern, dim3(grid), dim3(256), ffn_smem<4>(), st, a, npass); }
            else { auto kern = ffn_chain_kernel<T, false, 0, false, 4>; GE2E_LAUNCH(h, kern, dim3(grid), dim3(256), ffn_smem<4>(), st, a, npass); }
        } else { auto kern = ffn_chain_kernel<T, false>; GE2E_LAUNCH(h, kern, dim3(grid), dim3(512), ffn_smem<8>(), st, a, npass); }
        return 0;
    }
}

// norm2 backward + dF + dH1 in one launch (ffn_chain_kernel<..., BWD>): same block shapes and grid policy as the forward's train kernel
template <typename T>
int launch_ffn_chain_bwd(ge2e_handle h, hipStream_t st, const FfnArgs& a) {
    if constexpr (sizeof(T) != 2) return fail(h, GE2E_EUNSUPPORTED, "ffn chain: 16-bit modes only");
    else {
        if (h->num_cus <= 0) {
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, h->device >= 0 ? h->device : 0) != hipSuccess || n <= 0) n = 256;
            h->num_cus = n;
        }
        // the 4-wave shape only (two independent blocks per CU): with dP living in the output accumulators through the prologue the
        // 8-wave instantiation spills inside its stage loop, and alone the 4-wave shape was the faster one anyway (250 vs 280 us)
        const int slots = 2 * h->num_cus;
        const int npass = (a.M + 127) / 128;
        const int grid = std::min(npass, slots);
        const double rows = a.M;
        // algorithmic bytes: dH, h2 in; dM out; mask bits in; dF out; dH1 out; both weight matrices
        const double abytes = 2.0 * (rows * 256 * 4.0 + rows * FFN_F + 2.0 * 256 * FFN_F) + rows * FFN_F / 8;
        ProfScope ps(h, st, GE2E_K_FFN, 2.0 * rows * 256 * FFN_F * 2.0, abytes);
        auto kern = ffn_chain_kernel<T, true, 0, true, 4, true>;
        GE2E_LAUNCH(h, kern, dim3(grid), dim3(256), ffn_bwd_smem<4>(), st, a, npass);
        return 0;
    }
}
inline bool ffn_chain_bwd_on() { return !opt(O_NO_FFN_CHAIN_BWD); }

// 256 x 256-tile split-K weight gradient (wgrad_ks.cuh) for the wide 16-bit products; everything else (and the rows beyond the
// last multiple of 32) on the 128 x 128 kernel below.
inline bool wgrad_ks_on() { return !opt(O_NO_WGRAD_KS); }
template <typename T, int XLOAD, bool X3 = false> int launch_wgrad_tiled(ge2e_handle h, hipStream_t st, WgradArgs a);

// reduce passes of the split-K products launched since the last flush (one slab each): flushed as ONE launch at the end of a layer
struct WkPending {
    WkReduceArgs args{};
    int blocks = 0;
    size_t slab_floats = 0;              // Layout::wslab
    float* slab(float* base) const { return base + (size_t)args.njobs * slab_floats; }
};
inline bool wk_batch_on() { return !opt(O_NO_REDUCE_BATCH); }
int flush_wk_reduce(ge2e_handle h, hipStream_t st, WkPending& pend) {
    if (pend.args.njobs == 0) return 0;
    ProfScope ps(h, st, GE2E_K_WGRAD, 0.0, 0.0, /*counts=*/false);        // class time, not a launch of the class
    GE2E_LAUNCH(h, wgrad_ks_reduce_multi_kernel, dim3(pend.blocks), dim3(512), 0, st, pend.args);
    const size_t keep = pend.slab_floats;
    pend = WkPending{};
    pend.slab_floats = keep;
    return 0;
}

template <typename T, int XLOAD, bool X3 = false>
int launch_wgrad(ge2e_handle h, hipStream_t st, WgradArgs a, float* part = nullptr, WkPending* pend = nullptr) {
    if constexpr (sizeof(T) == 2) {
        const int tn = a.N / 256, tk = a.K / 256;
        if (part && wgrad_ks_on() && a.N % 256 == 0 && a.K % 256 == 0 && tn * tk >= 1 && tn * tk <= 16 && a.R >= 256 &&
            a.ldy % 8 == 0 && a.ldx % 8 == 0) {
            if (h->num_cus <= 0) {
                int n = 0;
                if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, h->device >= 0 ? h->device : 0) != hipSuccess || n <= 0) n = 256;
                h->num_cus = n;
            }
            const int R32 = a.R / 32 * 32, stages = R32 / 32;
            // A block fills a whole CU (128 KB of LDS, every vector register), so the launch is sized to leave CUs to the backward's
            // main chain on the other stream: the two then share the chip in SPACE instead of taking turns (measured, step time:
            // 256 blocks 4.29 ms, 192 4.19, 160 4.16, 128 4.20, 96 4.26).  GE2E_WGRAD_KS_BLOCKS overrides.
            const int blocks_cap = opt(O_WGRAD_KS_BLOCKS) > 0 ? opt(O_WGRAD_KS_BLOCKS) : WK_DEFAULT_BLOCKS;
            const int ntile = tn * tk;
            // (measured alone -- GE2E_K_SERIAL / GE2E_NO_OVERLAP: nothing to share the chip with -- a launch takes every CU)
            bool alone = !h->overlap;
            { std::lock_guard<std::mutex> g(h->mu); alone = alone || (h->prof_mask & GE2E_K_SERIAL) != 0; }
            int splits = std::min(alone ? WK_MAX_BLOCKS : std::min(WK_MAX_BLOCKS, a.blocks > 0 ? a.blocks : blocks_cap), h->num_cus) / ntile;
            // whole XCD rounds: the kernel deals row slices to the 8 XCDs (wgrad_ks.cuh), so a multiple of 8 leaves no XCD a block short
            if (splits >= 8) splits = std::min((splits + 4) / 8 * 8, WK_MAX_BLOCKS / ntile / 8 * 8);
            splits = std::max(1, std::min(splits, stages / 8));          // at least 8 stages per block
            const int sps = (stages + splits - 1) / splits;
            splits = (stages + sps - 1) / sps;
            if (pend && (size_t)ntile * splits * WK_TILE_FLOATS > pend->slab_floats) return fail(h, GE2E_EWORKSPACE, "wgrad_ks: partial slab too small for this launch");
            const bool batch = pend && wk_batch_on() && pend->args.njobs < WK_MAX_JOBS;
            if (batch) part = pend->slab(part);
            WgradKsArgs k{};
            k.Y = a.Y; k.ldy = a.ldy; k.X = a.X; k.ldx = a.ldx; k.part = part; k.db = a.db; k.R32 = R32; k.rows_per_split = sps * 32;
            k.tiles_n = tn; k.tiles_k = tk; k.splits = splits;
            // one scope around the product AND its reduce pass: `roofline.kernel` names both, and the class time is then what rocprof
            // shows for the two rows together
            {
                ProfScope ps(h, st, GE2E_K_WGRAD, 2.0 * R32 * (double)a.N * a.K, (double)R32 * (a.N + a.K) * sizeof(T) + 4.0 * a.N * a.K);
                {
                auto kern = wgrad_ks_kernel<T>;
                GE2E_LAUNCH(h, kern, dim3(8 * ntile * ((splits + 7) / 8)), dim3(512), wgrad_ks_smem(), st, k);
                // split groups: ~256 reduce blocks whatever the tile count (a single-tile product used to sum its 160 partials in 64 blocks)
                const int sgroups = std::max(1, std::min(splits, 256 / (ntile * 32)));
                if (batch) {
                    WkReduceJob& j = pend->args.job[pend->args.njobs++];
                    j.part = part; j.dW = a.dW; j.ldw = a.ldw; j.splits = splits; j.tiles_n = tn; j.tiles_k = tk; j.block0 = pend->blocks; j.sgroups = sgroups;
                    pend->blocks += ntile * 32 * sgroups;
                } else
                GE2E_LAUNCH(h, wgrad_ks_reduce_kernel, dim3(ntile * 32, sgroups), dim3(512), 0, st, (const float*)part, a.dW, a.ldw, splits, tn, tk);
                }
            }
            if (R32 == a.R) return 0;
            WgradArgs tail = a;                                          // < 32 rows left: the tiled kernel adds them atomically
            tail.Y = (const unsigned char*)a.Y + (size_t)R32 * a.ldy * sizeof(T);
            tail.X = (const unsigned char*)a.X + (size_t)R32 * a.ldx * sizeof(T);
            tail.R = a.R - R32;
            return launch_wgrad_tiled<T, XLOAD>(h, st, tail);
        }
    }
    return launch_wgrad_tiled<T, XLOAD, X3>(h, st, a);
}

template <typename T, int XLOAD, bool X3>
int launch_wgrad_tiled(ge2e_handle h, hipStream_t st, WgradArgs a) {
    constexpr bool X = X3 && sizeof(T) == 4;                 // fp32x3: two bf16 planes per operand tile, 16-bit row pitch
    constexpr int RS = 2 * Prec<T>::KG;
    constexpr int LD = X ? 2 * (256 + 32) : 128 * (int)sizeof(T) + (sizeof(T) == 2 ? 32 : 16);
    if (a.N % 128 != 0) return fail(h, GE2E_EUNSUPPORTED, "wgrad: N must be a multiple of 128");
    if (XLOAD == ALOAD_ROW && a.ldx % 128 != 0) return fail(h, GE2E_EUNSUPPORTED, "wgrad: the X operand must be stored in whole 128-column tiles");
    const int tn = a.N / 128, tk = (a.K + 127) / 128;
    // row slices: enough blocks for ~2 per CU, few enough that the fp32 partial tiles (one 64 KB atomic flush per
    // block) stay small next to the operand traffic
    // never more than the 512 blocks that are resident at once (2 per CU): in_proj's 12 tiles x 43 slices = 516 blocks ran the
    // last 4 alone in a second wave, 163 us instead of 125
    int splits = std::max(1, 512 / (tn * tk));
    const int max_splits = (a.R + 4 * RS - 1) / (4 * RS);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int rps = (a.R + splits - 1) / splits;
    rps = (rps + RS - 1) / RS * RS;
    splits = (a.R + rps - 1) / rps;
    a.rows_per_split = rps; a.tiles_n = tn; a.tiles_k = tk;
    const size_t smem = std::max<size_t>(4 * (size_t)RS * LD, 128 * (128 * 4 + 16));
    auto kern = wgrad_kernel<T, XLOAD, 3, 2, X>;
    ProfScope ps(h, st, GE2E_K_WGRAD, 2.0 * a.R * a.N * a.K, (double)a.R * (a.N + a.K) * sizeof(T) + 4.0 * a.N * a.K);
    GE2E_LAUNCH(h, kern, dim3(tn * tk * splits), dim3(256), smem, st, a);
    return 0;
}

template <typename T, int KT, bool PAD, bool DROP>
int launch_attn_kt(ge2e_handle h, hipStream_t st, const AttnArgs& a, int n, bool bwd) {
    constexpr int TP = 32 * KT;
    const int qtiles = (a.T + 15) / 16;
    int nw = (qtiles + 1) / 2;
    // fp32x3: the four LDS planes of a block (88 KB at 160 frames) leave room for ONE block per CU, so the block takes up to 8 waves, one 16-row
    // tile each where that covers the sequence (the 16-bit modes run 2-3 blocks of qtiles / 2 waves per CU)
    if constexpr (std::is_same<T, x3_t>::value) nw = qtiles;
    if (nw > 8) nw = 8;
    if (nw < 1) nw = 1;
    const dim3 grid(n * a.H), block(64 * nw);
    ProfScope ps(h, st, bwd ? GE2E_K_ATTN_BWD : GE2E_K_ATTN_FWD, (bwd ? 14.0 : 4.0) * a.T * a.T * 64.0 * n * a.H,
                 (double)n * a.T * a.D * sizeof(T) * (bwd ? 8.0 : 4.0));
    if (!bwd) {
        const size_t smem = 2 * (size_t)attn::xtile_bytes<T>(TP);
        // scheduling-barrier spacing of the score loop: every 5th tile (measured at 160 frames with the packed conversions: 95 us; every tile 119 us)
        constexpr int SBE = KT >= 3 ? 5 : (KT >= 2 ? 2 : 1);
        auto kern = attn_fwd_kernel<T, KT, PAD, DROP, SBE>;
        GE2E_LAUNCH(h, kern, grid, block, smem, st, a);
    } else {
        size_t smem = 2 * (size_t)attn::xtile_bytes<T>(TP) + 2 * (size_t)TP * 4 + (attn::use_mask<T, KT>() && DROP ? (size_t)TP * (TP / 4) : 0);    // + keep bits: a byte per (4 keys, query)
        // Two blocks of >= 5 waves per CU are enough for this kernel (960 x 160 alone: 310 us at 2 blocks per CU, 318-329 at 3, 415 at 1), and the
        // third block only takes registers and LDS from whatever the weight-gradient stream has in flight: a launch that would fit three asks
        // for a little more LDS than a third of the CU's (step 3.785 -> 3.742 ms).
        if (nw >= 5 && 3 * smem <= (size_t)160 * 1024) smem = (size_t)160 * 1024 / 3 + 1024;
        // a scheduling barrier after every tile group (32 keys / queries): 112-120 registers, three blocks per CU fit; spaced wider the
        // compiler hoists fragment loads (130 registers at 5 groups; 247-256 + spills when a loop never meets a barrier)
        // (Round 4, measured and not kept -- profiles/r04_ab_log.txt, commit c4d4096: the Q tile LDS-resident beside K and V with prefetched dO / O
        // rows, 6 instead of 8 activation tiles fetched: 287 vs 294 us alone, 3.527 vs 3.525-3.549 ms in the step; the score MFMAs of the next
        // group issued by hand ahead of the vector work: 354-386 us; a kernel that evaluates every score once, tools/attention_bwd1.cuh: 396-409 us.)
        auto kern = attn_bwd_kernel<T, KT, PAD, DROP, 1, 0, 2>;
        GE2E_LAUNCH(h, kern, grid, block, smem, st, a);
    }
    return 0;
}
// 288 < T <= 1024: K / V (or Q / dO) stream through LDS in 64-row chunks (attention.cuh, "Long sequences")
template <typename T>
int launch_attn_long(ge2e_handle h, hipStream_t st, const AttnArgs& a, int n, bool bwd, float* delta) {
    const dim3 grid(n * a.H, (a.T + ATT_LC - 1) / ATT_LC), block(256);
    ProfScope ps(h, st, bwd ? GE2E_K_ATTN_BWD : GE2E_K_ATTN_FWD, (bwd ? 14.0 : 4.0) * a.T * a.T * 64.0 * n * a.H,
                 (double)n * a.T * a.D * sizeof(T) * (bwd ? 8.0 : 4.0));
    if (!bwd) {
        auto kern = attn_fwd_long_kernel<T>;
        GE2E_LAUNCH(h, kern, grid, block, 0, st, a);
    } else {
        if (!delta) return fail(h, GE2E_EINVAL, "long-sequence attention backward needs the delta scratch");
        AttnLongBwd x{delta};
        auto k1 = attn_bwd_long_dq_kernel<T>;
        GE2E_LAUNCH(h, k1, grid, block, 0, st, a, x);
        auto k2 = attn_bwd_long_dkv_kernel<T>;
        GE2E_LAUNCH(h, k2, grid, block, 0, st, a, x);
    }
    return 0;
}

template <typename T>
int launch_attn(ge2e_handle h, hipStream_t st, const AttnArgs& a, int n, bool bwd, float* delta = nullptr) {
    // (fp32x3: the chunked long-sequence kernels stay exact fp32 -- a functional path, the same tensors)
    if (a.T > 32 * MAX_KT) return launch_attn_long<std::conditional_t<std::is_same<T, x3_t>::value, float, T>>(h, st, a, n, bwd, delta);
    const bool pad = a.T % 32 != 0;      // multiples of 32 frames need no key / query masking
    const bool drop = a.drop.thr != 0;   // dropout is a compile-time property of the kernels (attention.cuh)
#define launch_attn_kt(TT, KK, PP) (drop ? launch_attn_kt<TT, KK, PP, true>(h, st, a, n, bwd) : launch_attn_kt<TT, KK, PP, false>(h, st, a, n, bwd))
    switch ((a.T + 31) / 32) {
        case 1: return pad ? launch_attn_kt(T, 1, true) : launch_attn_kt(T, 1, false);
        case 2: return pad ? launch_attn_kt(T, 2, true) : launch_attn_kt(T, 2, false);
        case 3: return pad ? launch_attn_kt(T, 3, true) : launch_attn_kt(T, 3, false);
        case 4: return pad ? launch_attn_kt(T, 4, true) : launch_attn_kt(T, 4, false);
        case 5: return pad ? launch_attn_kt(T, 5, true) : launch_attn_kt(T, 5, false);
        case 6: return pad ? launch_attn_kt(T, 6, true) : launch_attn_kt(T, 6, false);
        case 7: return pad ? launch_attn_kt(T, 7, true) : launch_attn_kt(T, 7, false);
        case 8: return pad ? launch_attn_kt(T, 8, true) : launch_attn_kt(T, 8, false);
        case 9: return pad ? launch_attn_kt(T, 9, true) : launch_attn_kt(T, 9, false);
        default: return fail(h, GE2E_EUNSUPPORTED, "attention: frame count not instantiated");
    }
#undef launch_attn_kt
}

// The attention sub-layer of a full layer as one kernel per utterance (attn_sub.cuh): 16-bit modes, 4 heads of 64, 49 .. 160 frames.  OPT-IN
// (option attn_sub = 1): correct and tested, but measured SLOWER than the three launches it replaces (profiles/r04_ab_log.txt section 3).
template <typename T>
bool attn_sub_shape(const ge2e_config& c, int t) {
    return sizeof(T) == 2 && opt(O_ATTN_SUB) && c.emb == 256 && c.heads == 4 && t > 48 && t <= 160;
}
template <typename T, int NW>
int launch_attn_sub_nw(ge2e_handle h, hipStream_t st, const AttnSubArgs& a, int n) {
    if constexpr (sizeof(T) != 2) return fail(h, GE2E_EUNSUPPORTED, "attn_sub: 16-bit modes only");
    else {
        const bool pad = a.T % 32 != 0, drop = a.drop_attn.thr != 0;
        constexpr size_t smem = attn_sub::smem_bytes<NW>();
        const double rows = (double)n * a.T;
        // algorithmic bytes: x in, h1 out, (train) q|k|v and o out, both weight matrices; FLOPs: in_proj + attention + out_proj
        ProfScope ps(h, st, GE2E_K_ATTN_FWD, rows * (2.0 * 256 * 768 + 4.0 * a.T * 256 + 2.0 * 256 * 256),
                     2.0 * (rows * 256 * (a.qkv ? 6.0 : 2.0) + 4.0 * 256 * 256));
#define GE2E_SUB(PP, DD) do { auto kern = attn_sub_fwd_kernel<T, NW, PP, DD>; GE2E_LAUNCH(h, kern, dim3(n), dim3(64 * NW), smem, st, a); } while (0)
        if (pad) { if (drop) GE2E_SUB(true, true); else GE2E_SUB(true, false); }
        else { if (drop) GE2E_SUB(false, true); else GE2E_SUB(false, false); }
#undef GE2E_SUB
        return 0;
    }
}
template <typename T>
int launch_attn_sub(ge2e_handle h, hipStream_t st, const AttnSubArgs& a, int n) {
    switch ((a.T + 15) / 16) {
        case 4: return launch_attn_sub_nw<T, 4>(h, st, a, n);
        case 5: return launch_attn_sub_nw<T, 5>(h, st, a, n);
        case 6: return launch_attn_sub_nw<T, 6>(h, st, a, n);
        case 7: return launch_attn_sub_nw<T, 7>(h, st, a, n);
        case 8: return launch_attn_sub_nw<T, 8>(h, st, a, n);
        case 9: return launch_attn_sub_nw<T, 9>(h, st, a, n);
        case 10: return launch_attn_sub_nw<T, 10>(h, st, a, n);
        default: return fail(h, GE2E_EUNSUPPORTED, "attn_sub: frame count not instantiated");
    }
}

// last layer: one query per utterance, no K / V projection (attn_last.cuh)
// the compact chain of the last layer + the tail (lastc.cuh): what forward and backward share
inline LastcArgs lastc_args(const ge2e_config& c, const Layout& L, unsigned char* ws, const float* const* P, int l, int n, int t, const void* hin) {
    LastcArgs a{};
    a.n = n; a.drow_mul = t; a.eps = c.ln_eps;
    a.o = ws + L.o[l]; a.x0 = hin; a.ldx = c.emb * t;
    a.Wo = ws + L.w_out[l]; a.bo = P[lp(l, L_OUT_B)]; a.g1 = P[lp(l, L_N1_W)]; a.be1 = P[lp(l, L_N1_B)];
    a.W1 = ws + L.w_l1[l]; a.b1 = P[lp(l, L_L1_B)]; a.W2 = ws + L.w_l2[l]; a.b2 = P[lp(l, L_L2_B)]; a.g2 = P[lp(l, L_N2_W)]; a.be2 = P[lp(l, L_N2_B)];
    a.gf = P[p_fn_w(c)]; a.bf = P[p_fn_b(c)]; a.wq = P[p_proj_w(c)]; a.bq = P[p_proj_b(c)];
    a.h1 = ws + L.h1[l]; a.f = ws + L.f[l]; a.h2 = ws + L.h2[l];
    a.xhat = (float*)(ws + L.xhat_f); a.rstd_f = (float*)(ws + L.rstd_f); a.zm = (float*)(ws + L.zm); a.nrm = (float*)(ws + L.nrm);
    a.emb = (float*)(ws + L.emb_keep);
    return a;
}
inline AttnLastArgs attn_last_args(const ge2e_config& c, const Layout& L, unsigned char* ws, const float* const* P, int l, int t, const void* x,
                                   bool train, const Drop& drop) {
    AttnLastArgs a{};
    const size_t dd = (size_t)c.emb * c.emb * L.esz;
    a.x = x; a.q0 = ws + L.lq0; a.Wq = ws + L.w_in[l]; a.Wk = ws + L.w_in[l] + dd; a.Wv = ws + L.w_in[l] + 2 * dd; a.bv = P[lp(l, L_IN_B)] + 2 * c.emb;
    if (train) { a.qk = (float*)(ws + L.lqk); a.prob = (float*)(ws + L.lprob); a.ctx = (float*)(ws + L.lctx); a.sp = (float*)(ws + L.lsp); a.dqk = (float*)(ws + L.ldqk); }
    a.T = t; a.H = c.heads; a.scale = 1.0f / std::sqrt((float)(c.emb / c.heads)); a.drop = drop;
    return a;
}

// A handle's side stream, fence events and CU count belong to ONE device: the one that is current at its first GPU call.
// A later call with another device current would launch on a stream of the wrong device, so it is refused.
int check_device(ge2e_handle h) {
    int dev = -1;
    const hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail_hip(h, e, "hipGetDevice");
    std::lock_guard<std::mutex> g(h->mu);
    if (h->device < 0) h->device = dev;
    else if (h->device != dev) {
        h->err = "this handle was first used on device " + std::to_string(h->device) + " but the current device is " +
                 std::to_string(dev) + " (make the tensors' device current around the call)";
        return GE2E_EINVAL;
    }
    return 0;
}

int check_common(ge2e_handle h, int n, int t, int samples, const void* ws, size_t ws_bytes, const Layout& L) {
    const ge2e_config& c = h->cfg;
    { const int e = check_device(h); if (e) return e; }
    if (n <= 0 || t <= 0 || samples <= 0 || n % samples != 0) return fail(h, GE2E_EINVAL, "n_utts/frames/samples invalid");
    if (t > MAX_FRAMES) return fail(h, GE2E_EUNSUPPORTED, "frames > 1024 not supported by the attention kernels");
    if (t > c.max_position) return fail(h, GE2E_EINVAL, "frames > max_position");
    if ((double)n * t * c.ffn >= 4294967296.0) return fail(h, GE2E_EUNSUPPORTED, "n_utts*frames*ffn must stay below 2^32 (dropout counters)");
    if ((double)n * c.heads * t * ((t + 3) / 4 * 4) >= 4294967296.0)
        return fail(h, GE2E_EUNSUPPORTED, "n_utts*heads*frames^2 must stay below 2^32 (attention dropout counters)");
    if (!ws || ws_bytes < L.total) return fail(h, GE2E_EWORKSPACE, "workspace too small (see ge2e_workspace_bytes)");
    if (((uintptr_t)ws & 255) != 0) return fail(h, GE2E_EINVAL, "workspace must be 256-byte aligned");
    return 0;
}

#define CK(x) do { int _e = (x); if (_e) return _e; } while (0)

// Side-stream fencing for backward: weight-gradient GEMMs only consume tensors the main chain has finished and
// produce gradients nobody reads before the end of backward, so they run on an internal stream next to the
// latency-bound dgrad / attention / LayerNorm kernels of the main stream.
struct SideCtx {
    ge2e_handle h; hipStream_t main_st; hipStream_t side = nullptr; bool on = false; size_t next = 0; int err = 0;
    ge2e_handle_s::EventSet* set = nullptr;
    SideCtx(ge2e_handle h_, hipStream_t m) : h(h_), main_st(m) {
        if (!h->overlap) return;
        { std::lock_guard<std::mutex> g(h->mu); if (h->prof_mask & GE2E_K_SERIAL) return; }   // measuring kernels alone
        if (!h->side) {
            // (default priority: highest / lowest measured 4.46 / 4.54 vs 4.44 ms per step in round 1, +-0.01 ms in round 3)
            if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess) { h->side = nullptr; return; }
        }
        // this backward's own event set: one whose previous backward has completed on the device, else a new one
        std::lock_guard<std::mutex> g(h->mu);
        for (auto* s : h->event_sets)
            if (!s->used || (s->done && hipEventQuery(s->done) == hipSuccess)) { set = s; break; }
        if (!set) {
            set = new (std::nothrow) ge2e_handle_s::EventSet();
            if (!set) return;
            if (hipEventCreateWithFlags(&set->done, hipEventDisableTiming) != hipSuccess) { delete set; set = nullptr; return; }
            h->event_sets.push_back(set);
        }
        set->used = true;
        side = h->side; on = true;
    }
    hipEvent_t ev() {
        if (next == set->ev.size()) {
            hipEvent_t e = nullptr;
            // device-side fences between two streams of ONE device: no system-scope release (the kernels' own agent-scope release /
            // acquire at dispatch boundaries orders the data)
            constexpr unsigned flags = hipEventDisableTiming | (unsigned)hipEventDisableSystemFence;
            if (hipEventCreateWithFlags(&e, flags) != hipSuccess) { err = 1; return nullptr; }
            set->ev.push_back(e);
        }
        return set->ev[next++];
    }
    hipStream_t wstream() const { return on ? side : main_st; }
    // arm(): the main-stream kernels launched from here to the next fork() carry that fork's fence as their completion event
    // (GE2E_LAUNCH), so fork() only has to make the side stream wait for it
    void arm() {
        if (!on || bind_off()) return;
        tl_armed = ArmedEvent{ev(), main_st, 0};
    }
    static bool bind_off() { return opt(O_NO_EVENT_BIND) != 0; }
    void fork() {                           // side waits for everything enqueued on main so far
        if (!on) return;
        const ArmedEvent a = tl_armed;
        tl_armed = ArmedEvent{};
        if (a.ev && a.launches > 0) {       // bound to the last kernel before this point
            if (hipStreamWaitEvent(side, a.ev, 0) != hipSuccess) err = 1;
        } else {
            hipEvent_t e = a.ev ? a.ev : ev();  // (an armed event no kernel took is still unrecorded: use it here)
            if (!e || hipEventRecord(e, main_st) != hipSuccess || hipStreamWaitEvent(side, e, 0) != hipSuccess) err = 1;
        }
        if (const int us = debug_side_delay_us()) hipLaunchKernelGGL(debug_delay_kernel, dim3(1), dim3(64), 0, side, (unsigned)(us * 100));
    }
    hipEvent_t mark() {                     // event after everything enqueued on side so far
        if (!on) return nullptr;
        hipEvent_t e = ev();
        if (!e || hipEventRecord(e, side) != hipSuccess) { err = 1; return nullptr; }
        return e;
    }
    void wait(hipEvent_t& e) {              // main waits for a side-stream reader before overwriting its input
        if (on && e && hipStreamWaitEvent(main_st, e, 0) != hipSuccess) err = 1;
        e = nullptr;
    }
    // End of a backward, on EVERY path (also an error return in the middle): the caller's stream waits for everything the
    // side stream was given -- the caller may free the gradient buffer and the workspace right after the call -- and the
    // set's `done` event marks the point after which its fence events may be recorded again.
    void join() {
        tl_armed = ArmedEvent{};            // (an error return between arm() and fork())
        if (!on) return;
        hipEvent_t e = mark();
        wait(e);
        if (hipEventRecord(set->done, main_st) != hipSuccess) err = 1;
    }
};

template <typename T, bool X3 = false>
int forward_impl(ge2e_handle h, hipStream_t st, const void* mel, bool mel_f16, int n, int t, int samples,
                 const float* const* P, const float* pe, float* out_emb, unsigned char* ws, const Layout& L,
                 bool train, uint64_t seed, uint64_t step, bool prepared = false) {
    const ge2e_config& c = h->cfg;
    const int d = c.emb, R = L.R;
    // ---- weight preparation (fp32 masters -> T copies, transposed copies for dgrad); GE2E_FWD_PREPARED: still in the workspace
    if (!prepared) {
        std::vector<PrepJob> jobs;
        auto add = [&](const float* src, size_t dst, size_t dstT, int rows, int cols, int ldd) {
            PrepJob j{src, ws + dst, (train && dstT != (size_t)-1) ? ws + dstT : nullptr, rows, cols, ldd, 0, (ldd + 31) / 32, 0, nullptr};
            jobs.push_back(j);
        };
        // the two fp32 transposes ride in the same launch: the positional table pe [D][max_pos] -> pe_t [T][D] and projection.weight -> its transpose
        jobs.push_back(PrepJob{pe, nullptr, nullptr, d, t, t, 0, (t + 31) / 32, c.max_position, (float*)(ws + L.pe_t)});
        jobs.push_back(PrepJob{P[p_proj_w(c)], nullptr, nullptr, d, d, d, 0, (d + 31) / 32, 0, (float*)(ws + L.wqT)});
        add(P[P_PRENET_W], L.w_prenet, (size_t)-1, d, c.mel_dim, L.KP);
        for (int l = 0; l < c.layers; ++l) {
            add(P[lp(l, L_IN_W)], L.w_in[l], L.w_inT[l], 3 * d, d, d);
            add(P[lp(l, L_OUT_W)], L.w_out[l], L.w_outT[l], d, d, d);
            add(P[lp(l, L_L1_W)], L.w_l1[l], L.w_l1T[l], c.ffn, d, d);
            add(P[lp(l, L_L2_W)], L.w_l2[l], L.w_l2T[l], d, c.ffn, c.ffn);
        }
        for (size_t b = 0; b < jobs.size(); b += PREP_MAX_JOBS) {
            PrepArgs a{};
            a.njobs = (int)std::min<size_t>(PREP_MAX_JOBS, jobs.size() - b);
            int tiles = 0;
            for (int q = 0; q < a.njobs; ++q) {
                a.job[q] = jobs[b + q];
                a.job[q].tile0 = tiles;
                tiles += a.job[q].tiles_x * ((a.job[q].rows + 31) / 32);
            }
            auto kern = prep_weights_kernel<T>;
            GE2E_LAUNCH(h, kern, dim3(tiles), dim3(256), 0, st, a);
        }
    }
    const bool ffn_chained = sizeof(T) == 2 && c.ffn == FFN_F && d == 256 && ffn_chain_on();     // read ONCE per forward
    const bool last_chained = sizeof(T) == 2 && c.ffn == FFN_F && d == 256 && (samples == 1 || (!train && samples <= 16)) && !opt(O_NO_LAST_CHAIN);
    if (train) note_forward(h, ws, ffn_chained);
    // ---- mel batch: one coalesced pass fp32 [N, mel, T] -> T-typed rows [R, KP]; kept for the prenet backward
    if (mel_f16) {
        auto kern = mel_pack_kernel<T, _Float16>;
        GE2E_LAUNCH(h, kern, dim3((t + 63) / 64, n), dim3(256), 0, st, (const _Float16*)mel, (T*)(ws + L.xt), c.mel_dim, t, L.KP);
    } else {
        auto kern = mel_pack_kernel<T, float>;
        GE2E_LAUNCH(h, kern, dim3((t + 63) / 64, n), dim3(256), 0, st, (const float*)mel, (T*)(ws + L.xt), c.mel_dim, t, L.KP);
    }
    // ---- prenet + ReLU + positional encoding (+ dropout)
    {
        GemmArgs a{};
        a.A = ws + L.xt; a.lda = L.KP; a.W = ws + L.w_prenet; a.ldw = L.KP; a.C = ws + L.h0; a.ldc = d;
        a.M = R; a.N = d; a.K = L.KP; a.bias = P[P_PRENET_B];
        a.drop = make_drop(train, c.pe_dropout, seed, step, SITE_PE);
        a.pe_t = (const float*)(ws + L.pe_t); a.alpha = P[P_ALPHA]; a.T = t; a.mel = c.mel_dim;
        a.relu_bits = train ? ws + L.pbits : nullptr;
        bool ws_done = false;
        if constexpr (sizeof(T) == 2) {       // weight-stationary form (K = 128 = the padded mel width): the 64 weight fragments and the block's 16 positional rows in registers
            if (!opt(O_NO_WS_GEMM) && a.K == 128 && a.N == 256 && a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldc % 8 == 0 && prenet_ws_parts(t) > 0) {
                CK((launch_gemm_ws<T, EPI_PRENET, 128>(h, st, a)));
                ws_done = true;
            }
        }
        if (!ws_done) CK((gemm128<T, EPI_PRENET, ALOAD_ROW, X3>(h, st, a)));
    }
    for (int l = 0; l < c.layers; ++l) {
        unsigned char* hin = ws + (l == 0 ? L.h0 : L.h2[l - 1]);
        // Only frame 0 of the LAST layer's output is consumed (Modules.py:54): that layer needs K and V of every
        // frame but its query, out-projection, FFN and both LayerNorms for frame 0 only -> compact rows, one per
        // utterance, addressed with stride `rmul` where the full-row index matters (residual, dropout counters).
        const bool last = l == c.layers - 1;
        const int Rl = last ? n : R, rmul = last ? t : 1;
        const size_t esz = L.esz;
        if (!last && attn_sub_shape<T>(c, t)) {
            // in_proj -> attention -> out_proj + dropout + residual + norm1 in ONE launch per layer (attn_sub.cuh); q|k|v, o, lse, rstd1 leave for the backward in train mode
            AttnSubArgs a{};
            a.X = hin; a.Win = ws + L.w_in[l]; a.bin = P[lp(l, L_IN_B)]; a.Wo = ws + L.w_out[l]; a.bo = P[lp(l, L_OUT_B)];
            a.gamma = P[lp(l, L_N1_W)]; a.beta = P[lp(l, L_N1_B)]; a.eps = c.ln_eps;
            a.qkv = train ? ws + L.qkv[l] : nullptr; a.o = train ? ws + L.o[l] : nullptr;
            a.lse = train ? (float*)(ws + L.lse[l]) : nullptr; a.h1 = ws + L.h1[l]; a.rstd = train ? (float*)(ws + L.rstd1[l]) : nullptr;
            a.T = t; a.scale = 1.0f / std::sqrt((float)(d / c.heads));
            a.drop_attn = make_drop(train, c.tf_dropout, seed, step, site_attn(l)); a.drop_sa = make_drop(train, c.tf_dropout, seed, step, site_sa(l));
            CK(launch_attn_sub<T>(h, st, a, n));
        } else {
        if (!last) {   // in_proj
            GemmArgs a{};
            a.A = hin; a.lda = d; a.W = ws + L.w_in[l]; a.ldw = d; a.C = ws + L.qkv[l]; a.ldc = 3 * d;
            a.M = R; a.N = 3 * d; a.K = d; a.bias = P[lp(l, L_IN_B)];
            CK((gemm128<T, EPI_BIAS, ALOAD_ROW, X3>(h, st, a)));
        } else {
            GemmArgs q{};   // q of frame 0: rows n * T of hin -> compact rows
            q.A = hin; q.lda = d * t; q.W = ws + L.w_in[l]; q.ldw = d; q.C = ws + L.lq0; q.ldc = d;
            q.M = n; q.N = d; q.K = d; q.bias = P[lp(l, L_IN_B)];
            CK((gemm128<T, EPI_BIAS, ALOAD_ROW, X3>(h, st, q)));
            // one query per (utterance, head): scores and context straight from the layer input, K and V never exist (attn_last.cuh)
            AttnLastArgs a = attn_last_args(c, L, ws, P, l, t, hin, train, make_drop(train, c.tf_dropout, seed, step, site_attn(l)));
            a.o0 = ws + L.o[l];
            ProfScope ps(h, st, GE2E_K_ATTN_FWD, 4.0 * t * 256.0 * 4.0 * n, (double)n * t * d * sizeof(T));
            auto kern = attn_last_fwd_kernel<T>;
            GE2E_LAUNCH(h, kern, dim3(n), dim3(256), attn_last_fwd_smem(t), st, a);
        }
        if (last && last_chained) {
            // out_proj + norm1, the FFN + norm2, transformer.norm, projection and F.normalize of the compact rows in ONE launch (lastc.cuh)
            if constexpr (sizeof(T) == 2) {
                LastcArgs a = lastc_args(c, L, ws, P, l, n, t, hin);
                a.rstd1 = train ? (float*)(ws + L.rstd1[l]) : nullptr; a.rstd2 = train ? (float*)(ws + L.rstd2[l]) : nullptr;
                a.emb_out = out_emb; a.samples = samples;
                a.d_sa = make_drop(train, c.tf_dropout, seed, step, site_sa(l)); a.d_fh = make_drop(train, c.tf_dropout, seed, step, site_ffh(l));
                a.d_ff = make_drop(train, c.tf_dropout, seed, step, site_ff(l));
                auto kern = lastc_fwd_kernel<T, LASTC_NW>;
                const int per_tile = 16 / samples;          // utterances per 16-row tile
                GE2E_LAUNCH(h, kern, dim3((n / samples + per_tile - 1) / per_tile), dim3(LASTC_THREADS), lastc_smem<LASTC_NW>(), st, a);
            }
            return 0;
        }
        if (!last) {   // softmax(q k^T / 8) v per (utterance, head)
            AttnArgs a{};
            a.qkv = ws + L.qkv[l]; a.o = ws + L.o[l]; a.T = t; a.H = c.heads; a.D = d;
            a.lse = train ? (float*)(ws + L.lse[l]) : nullptr;
            a.scale = 1.0f / std::sqrt((float)(d / c.heads));
            a.drop = make_drop(train, c.tf_dropout, seed, step, site_attn(l));
            CK((launch_attn<std::conditional_t<X3 && sizeof(T) == 4, x3_t, T>>(h, st, a, n, false)));
        }
        {   // out_proj + dropout1 + residual + norm1
            GemmArgs a{};
            a.A = ws + L.o[l]; a.lda = d; a.W = ws + L.w_out[l]; a.ldw = d; a.C = ws + L.h1[l]; a.ldc = d;
            a.M = Rl; a.N = d; a.K = d; a.bias = P[lp(l, L_OUT_B)]; a.R = hin; a.ldr = d * rmul;
            a.gamma = P[lp(l, L_N1_W)]; a.beta = P[lp(l, L_N1_B)]; a.eps = c.ln_eps;
            a.rstd = train ? (float*)(ws + L.rstd1[l]) : nullptr;
            a.drop = make_drop(train, c.tf_dropout, seed, step, site_sa(l)); a.drow_mul = rmul;
            CK((gemm_ln<T, X3>(h, st, a)));
        }
        }
        if (!last && ffn_chained) {
            // linear1 + ReLU + dropout + linear2 + dropout2 + residual + norm2, the hidden on chip (written once in train mode)
            FfnArgs a{};
            a.A = ws + L.h1[l]; a.lda = d; a.W1 = ws + L.w_l1[l]; a.b1 = P[lp(l, L_L1_B)]; a.W2 = ws + L.w_l2[l]; a.b2 = P[lp(l, L_L2_B)];
            a.Fo = train ? ws + L.f[l] : nullptr; a.ldf = c.ffn; a.C = ws + L.h2[l]; a.ldc = d;
            a.Mb = train ? ws + L.fbits[l] : nullptr;
            a.gamma = P[lp(l, L_N2_W)]; a.beta = P[lp(l, L_N2_B)]; a.eps = c.ln_eps;
            a.rstd = train ? (float*)(ws + L.rstd2[l]) : nullptr;
            a.drop1 = make_drop(train, c.tf_dropout, seed, step, site_ffh(l));
            a.drop2 = make_drop(train, c.tf_dropout, seed, step, site_ff(l)); a.drow_mul = rmul; a.M = Rl;
            CK(launch_ffn_chain<T>(h, st, a));
            continue;
        }
        {   // linear1 + ReLU + dropout
            GemmArgs a{};
            a.A = ws + L.h1[l]; a.lda = d; a.W = ws + L.w_l1[l]; a.ldw = d; a.C = ws + L.f[l]; a.ldc = c.ffn;
            a.M = Rl; a.N = c.ffn; a.K = d; a.bias = P[lp(l, L_L1_B)];
            a.drop = make_drop(train, c.tf_dropout, seed, step, site_ffh(l)); a.drow_mul = rmul;
            CK((gemm128<T, EPI_BIAS_RELU_DROP, ALOAD_ROW, X3>(h, st, a)));
        }
        {   // linear2 + dropout2 + residual + norm2
            GemmArgs a{};
            a.A = ws + L.f[l]; a.lda = c.ffn; a.W = ws + L.w_l2[l]; a.ldw = c.ffn; a.C = ws + L.h2[l]; a.ldc = d;
            a.M = Rl; a.N = d; a.K = c.ffn; a.bias = P[lp(l, L_L2_B)]; a.R = ws + L.h1[l]; a.ldr = d;
            a.gamma = P[lp(l, L_N2_W)]; a.beta = P[lp(l, L_N2_B)]; a.eps = c.ln_eps;
            a.rstd = train ? (float*)(ws + L.rstd2[l]) : nullptr;
            a.drop = make_drop(train, c.tf_dropout, seed, step, site_ff(l)); a.drow_mul = rmul;
            if (last && gemm_sk_shape<T>(a, L.skpart)) CK((launch_gemm_sk<T, true>(h, st, a, (float*)(ws + L.skpart))));
            else CK((gemm_ln<T, X3>(h, st, a)));
        }
    }
    {   // final LN at t = 0 -> slice mean -> projection -> L2 normalise
        TailArgs a{};
        a.h = ws + L.h2[c.layers - 1]; a.T = 1 /* compact rows */; a.samples = samples; a.N = n;
        a.gf = P[p_fn_w(c)]; a.bf = P[p_fn_b(c)]; a.wq = P[p_proj_w(c)]; a.wqT = (const float*)(ws + L.wqT); a.bq = P[p_proj_b(c)];
        a.eps = c.ln_eps;
        a.xhat = (float*)(ws + L.xhat_f); a.rstd = (float*)(ws + L.rstd_f);
        a.zm = (float*)(ws + L.zm); a.nrm = (float*)(ws + L.nrm);
        a.emb = (float*)(ws + L.emb_keep); a.emb_out = out_emb;       // both copies from the one kernel (no device-to-device copy)
        auto kern = tail_fwd_kernel<T>;
        GE2E_LAUNCH(h, kern, dim3((n / samples + TAIL_RB - 1) / TAIL_RB), dim3(256), 0, st, a);
    }
    return 0;
}

inline bool maskbits_on() { return !opt(O_NO_MASKBITS); }
template <typename T, bool X3>
int backward_body(ge2e_handle h, hipStream_t st, SideCtx& sc, const float* mel, int n, int t, int samples,
                  const float* const* P, const float* d_emb, float* grads, unsigned char* ws, const Layout& L,
                  uint64_t seed, uint64_t step, ge2e_bucket_cb cb, void* user);

template <typename T, bool X3 = false>
int backward_impl(ge2e_handle h, hipStream_t st, const float* mel, int n, int t, int samples,
                  const float* const* P, const float* d_emb, float* grads, unsigned char* ws, const Layout& L,
                  uint64_t seed, uint64_t step, ge2e_bucket_cb cb, void* user) {
    SideCtx sc(h, st);
    const int rc = backward_body<T, X3>(h, st, sc, mel, n, t, samples, P, d_emb, grads, ws, L, seed, step, cb, user);
    sc.join();                                           // also when a launch failed half-way: nothing is left running on the side stream
    if (rc) return rc;
    if (sc.err) return fail(h, GE2E_EINVAL, "side-stream event fencing failed");
    return 0;
}

template <typename T, bool X3>
int backward_body(ge2e_handle h, hipStream_t st, SideCtx& sc, const float* mel, int n, int t, int samples,
                  const float* const* P, const float* d_emb, float* grads, unsigned char* ws, const Layout& L,
                  uint64_t seed, uint64_t step, ge2e_bucket_cb cb, void* user) {
    const ge2e_config& c = h->cfg;
    const int d = c.emb, R = L.R;
    auto G = [&](int idx) { return grads + h->params[idx].offset; };
    auto bucket = [&](int first, int last) {   // parameters [first, last] are final: tell the caller
        if (cb) cb(user, h->params[first].offset, h->params[last].offset + h->params[last].numel - h->params[first].offset);
    };
    if (((uintptr_t)grads & 3) != 0) return fail(h, GE2E_EINVAL, "grads_flat must be 4-byte aligned");
    GE2E_LAUNCH(h, zero_f32_kernel, dim3(512), dim3(256), 0, st, grads, (size_t)h->total);
    hipStream_t wst = sc.wstream();                       // stream of the weight-gradient kernels
    float* const wpart = L.wpart != (size_t)-1 ? (float*)(ws + L.wpart) : nullptr;   // split-K partial tiles (used in stream order on wst)
    WkPending pend;                                       // this layer's reduce passes, flushed as one launch behind its last product
    pend.slab_floats = L.wslab;
    // last side-stream reader of each buffer set (Layout: layer l uses set l % nset, dQKV l % nqkv); null = nobody to wait for
    hipEvent_t g_set1[2] = {nullptr, nullptr}, g_set2[2] = {nullptr, nullptr}, g_dF[2] = {nullptr, nullptr}, g_dQKV[3] = {nullptr, nullptr, nullptr};
    hipEvent_t g_dH[3] = {nullptr, nullptr, nullptr};    // last weight-gradient-stream reader of dHx[i] (the norm2 column sums)
    // The norm2 column sums read buffers that a <= 3-layer stack never reuses (dHx rotates over three), so they can run at ANY later time: they
    // go to the END of the main chain, which otherwise idles ~170 us while the weight-gradient stream finishes (step 3.569 -> 3.541 ms;
    // option no_colsum_end keeps them on the weight-gradient stream, as deeper stacks and runs with bucket callbacks -- whose layer buckets
    // would have to be split -- do anyway)
    // options are read ONCE per backward; what the forward left in this workspace comes from its note, not from the options as they stand now
    const bool fwd_bits = forward_left_bits(h, ws), use_bits = maskbits_on(), chain_bwd_opt = ffn_chain_bwd_on();
    const int stop_after = opt(O_DEBUG_BWD_STOP);        // diagnostics only: >= 0 returns after k layers so ge2e_debug_tap sees that layer's scratch
    const bool defer_colsum = !opt(O_NO_COLSUM_END) && !cb && c.layers <= 3 && stop_after < 0;   // (a stopped backward never reaches its end)
    LnBwdArgs deferred[8]; int ndeferred = 0;
    // the last layer below its attention + the tail as ONE launch (lastc.cuh); what it replaces is skipped in the layer loop
    const bool last_chained = sizeof(T) == 2 && c.ffn == FFN_F && d == 256 && samples == 1 && !opt(O_NO_LAST_CHAIN);
    if (last_chained) {
        if constexpr (sizeof(T) == 2) {
            const int l = c.layers - 1;
            LastcArgs a = lastc_args(c, L, ws, P, l, n, t, nullptr);
            a.rstd1 = (float*)(ws + L.rstd1[l]); a.rstd2 = (float*)(ws + L.rstd2[l]);
            a.d_sa = make_drop(true, c.tf_dropout, seed, step, site_sa(l)); a.d_fh = make_drop(true, c.tf_dropout, seed, step, site_ffh(l));
            a.d_ff = make_drop(true, c.tf_dropout, seed, step, site_ff(l));
            a.d_emb = d_emb; a.wqT = (const float*)(ws + L.wqT); a.W2T = ws + L.w_l2T[l]; a.W1T = ws + L.w_l1T[l]; a.WoT = ws + L.w_outT[l];
            a.d_raw = (float*)(ws + L.d_raw);
            a.dH = ws + L.c_dH; a.dP = ws + L.c_dP; a.dM = ws + L.c_dM; a.dF = ws + L.c_dF; a.dHb = ws + L.c_dHb;
            a.dP2 = ws + L.c_dP2; a.dM2 = ws + L.c_dM2; a.dO = ws + L.c_dO;
            a.dgf = G(p_fn_w(c)); a.dbf = G(p_fn_b(c)); a.dg2 = G(lp(l, L_N2_W)); a.db2 = G(lp(l, L_N2_B)); a.dg1 = G(lp(l, L_N1_W)); a.db1 = G(lp(l, L_N1_B));
            auto kern = lastc_bwd_kernel<T, LASTC_NW>;
            sc.arm();
            GE2E_LAUNCH(h, kern, dim3((n + 15) / 16), dim3(LASTC_THREADS), lastc_smem<LASTC_NW>(), st, a);
            sc.fork();                                    // everything below on the weight-gradient stream until the attention: the four small weight gradients
            TailArgs w{};
            w.T = 1; w.samples = samples; w.N = n; w.zm = (float*)(ws + L.zm); w.d_raw = (float*)(ws + L.d_raw);
            w.dwq = G(p_proj_w(c)); w.dbq = G(p_proj_b(c));
            GE2E_LAUNCH(h, tail_wgrad_kernel, dim3(d, std::max(1, std::min(16, (n / samples + 31) / 32))), dim3(256), 0, wst, w);
        }
    } else
    {
        TailArgs a{};
        a.T = 1 /* compact rows */; a.samples = samples; a.N = n;
        a.gf = P[p_fn_w(c)]; a.bf = P[p_fn_b(c)]; a.wq = P[p_proj_w(c)]; a.bq = P[p_proj_b(c)];
        a.xhat = (float*)(ws + L.xhat_f); a.rstd = (float*)(ws + L.rstd_f);
        a.zm = (float*)(ws + L.zm); a.nrm = (float*)(ws + L.nrm); a.emb = (float*)(ws + L.emb_keep);
        a.d_emb = d_emb; a.d_raw = (float*)(ws + L.d_raw); a.dH = ws + L.c_dH;
        a.dgf = G(p_fn_w(c)); a.dbf = G(p_fn_b(c)); a.dwq = G(p_proj_w(c)); a.dbq = G(p_proj_b(c));
        auto kern = tail_bwd_kernel<T>;
        sc.arm();
        GE2E_LAUNCH(h, kern, dim3((n / samples + TAIL_RB - 1) / TAIL_RB), dim3(256), 0, st, a);
        sc.fork();                                        // projection weight gradient: off the critical chain
        GE2E_LAUNCH(h, tail_wgrad_kernel, dim3(d, std::max(1, std::min(16, (n / samples + 31) / 32))), dim3(256), 0, wst, a);
    }
    bool tail_bucket_pending = true;                      // reported after the first join with the side stream
    const size_t esz = L.esz;
    for (int l = c.layers - 1; l >= 0; --l) {
        if (stop_after >= 0 && c.layers - 1 - l >= stop_after) return 0;
        unsigned char* hin = ws + (l == 0 ? L.h0 : L.h2[l - 1]);
        // the last layer runs on compact rows (one per utterance, frame 0) until its attention; see forward_impl
        const bool last = l == c.layers - 1;
        const int Rl = last ? n : R, rmul = last ? t : 1;
        unsigned char* const b_dH = ws + (last ? L.c_dH : L.dH_of(l));     // dL/d(layer output)
        unsigned char* const b_dHin = ws + L.dH_of(l - 1);                // dL/d(layer input): the next buffer of the rotation
        // norm2 backward + dF + dH1 as ONE launch (ffn.cuh, BWD): full-size layers of the 16-bit modes whose forward left the mask bits
        auto uses_chain_bwd = [&](int ll) {
            if constexpr (sizeof(T) != 2) return false;
            else return ll >= 0 && ll < c.layers - 1 && L.fbits[ll] != (size_t)-1 && c.ffn == FFN_F && d == 256 && fwd_bits && use_bits && chain_bwd_opt;
        };
        const bool chain_bwd = uses_chain_bwd(l);
        // (a larger wgrad_ks share for the products of the layer processed last -- 208 / 256 blocks -- was measured in round 3: no effect)
        auto wk_blocks = [&](int) { return 0; };
        // after this layer's last dgrad GEMM: the norm2 column sums of the layer below, on the weight-gradient stream (they read its dL/d(output))
        bool forked_after_dh = false;
        auto colsum_below = [&](bool already_forked = false) -> int {
            if (!uses_chain_bwd(l - 1)) return 0;
            LnBwdArgs a{};
            a.dy = b_dHin; a.y = ws + L.h2[l - 1]; a.gamma = P[lp(l - 1, L_N2_W)]; a.beta = P[lp(l - 1, L_N2_B)];
            a.dgamma = G(lp(l - 1, L_N2_W)); a.dbeta = G(lp(l - 1, L_N2_B)); a.R = R;
            if (defer_colsum) { deferred[ndeferred++] = a; return 0; }
            if (!already_forked) sc.fork();              // (armed on the dgrad GEMM that wrote b_dHin)
            forked_after_dh = true;
            auto kern = ln_colsum_kernel<T>;
            GE2E_LAUNCH(h, kern, dim3(std::min(512, (R + 31) / 32)), dim3(256), 0, wst, a);
            g_dH[l % L.nH] = sc.mark();                  // b_dHin = dHx[l % nH]
            return 0;
        };
        unsigned char* const b_dHb = ws + (last ? L.c_dHb : L.dHb);
        const int bs = l % L.nset, bq = l % L.nqkv;                   // this layer's buffer sets
        unsigned char* const b_dP = ws + (last ? L.c_dP : L.dP[bs]);
        unsigned char* const b_dM = ws + (last ? L.c_dM : L.dM[bs]);
        unsigned char* const b_dP2 = ws + (last ? L.c_dP2 : L.dP2[bs]);   // norm1-backward outputs (set 2)
        unsigned char* const b_dM2 = ws + (last ? L.c_dM2 : L.dM2[bs]);
        unsigned char* const b_dF = ws + (last ? L.c_dF : L.dF[bs]);
        unsigned char* const b_dQKV = ws + L.dQKV[bq];
        unsigned char* const b_dO = ws + (last ? L.c_dO : L.dO);
        const int ln_grid = std::min(2048, (Rl + 3) / 4);
        const Drop d_ff = make_drop(true, c.tf_dropout, seed, step, site_ff(l));
        const Drop d_fh = make_drop(true, c.tf_dropout, seed, step, site_ffh(l));
        const Drop d_sa = make_drop(true, c.tf_dropout, seed, step, site_sa(l));
        if (!last) sc.wait(g_set1[bs]);                  // (the last layer's compact scratch is written once per backward)
        unsigned char* gm = d_ff.thr ? b_dM : b_dP;
        const bool chained_here = last && last_chained;  // lastc_bwd_kernel left dM, dF, dM2, dP2, dO; only the weight gradients remain (the side stream is forked)
        if (chained_here) {
        } else if (chain_bwd) {
            if constexpr (sizeof(T) == 2) {
                sc.wait(g_dF[bs]);
                FfnArgs a{};
                a.A = b_dH; a.lda = d; a.Y = ws + L.h2[l]; a.gamma = P[lp(l, L_N2_W)]; a.beta = P[lp(l, L_N2_B)];
                a.rstd = (float*)(ws + L.rstd2[l]); a.dM = gm;
                a.W1 = ws + L.w_l2T[l]; a.W2 = ws + L.w_l1T[l]; a.Fo = b_dF; a.ldf = c.ffn; a.Mb = ws + L.fbits[l];
                a.C = b_dHb; a.ldc = d; a.drop1 = d_fh; a.drop2 = d_ff; a.drow_mul = rmul; a.M = Rl; a.eps = c.ln_eps;
                sc.arm();
                CK(launch_ffn_chain_bwd<T>(h, st, a));
            }
        } else {
        {   // norm2 backward
            LnBwdArgs a{};
            a.dy = b_dH; a.y = ws + L.h2[l]; a.gamma = P[lp(l, L_N2_W)]; a.beta = P[lp(l, L_N2_B)];
            a.rstd = (const float*)(ws + L.rstd2[l]); a.dpre = b_dP; a.dmask = d_ff.thr ? b_dM : nullptr;
            a.dgamma = G(lp(l, L_N2_W)); a.dbeta = G(lp(l, L_N2_B)); a.R = Rl; a.drop = d_ff; a.drow_mul = rmul;
            auto kern = ln_bwd_kernel<T>;
            ProfScope ps(h, st, GE2E_K_LN_BWD, 12.0 * Rl * d, (double)Rl * d * L.esz * (d_ff.thr ? 4 : 3));
            GE2E_LAUNCH(h, kern, dim3(ln_grid), dim3(256), 0, st, a);
        }
        if (!last) sc.wait(g_dF[bs]);
        {   // dF = (dG W2) masked by ReLU/dropout of the hidden.  (Fusing norm2's backward into this GEMM as it is fused into
            // dO below was measured and lost: its four column-group blocks each redo the LayerNorm prologue, 327 vs 260 us.)
            GemmArgs a{};
            a.A = gm; a.lda = d; a.W = ws + L.w_l2T[l]; a.ldw = d; a.C = b_dF; a.ldc = c.ffn;
            a.M = Rl; a.N = c.ffn; a.K = d; a.R = ws + L.f[l]; a.ldr = c.ffn; a.mask_scale = d_fh.scale;
            sc.arm();
            // a layer whose forward went through the chained FFN kernel left the mask as bits: 1/16 of the bytes of the hidden
            bool bits = false;
            if constexpr (sizeof(T) == 2) bits = L.fbits[l] != (size_t)-1 && c.ffn == FFN_F && d == 256 && fwd_bits && ws_shape(a) && use_bits;
            if (bits) {
                a.R = ws + L.fbits[l]; a.ldr = c.ffn / 8;
                if constexpr (sizeof(T) == 2) CK((launch_gemm_ws<T, EPI_MASKBITS>(h, st, a)));
            } else
            CK((gemm128<T, EPI_MASK, ALOAD_ROW, X3>(h, st, a)));
        }
        }
        if (!chained_here) sc.fork();
        {   // (started right after norm2's backward instead, next to the dF GEMM that streams the same gm and f: no gain, 4.10 vs 4.10 ms)
            WgradArgs a{};
            a.Y = gm; a.ldy = d; a.X = ws + L.f[l]; a.ldx = c.ffn; a.dW = G(lp(l, L_L2_W)); a.ldw = c.ffn; a.db = G(lp(l, L_L2_B));
            a.R = Rl; a.N = d; a.K = c.ffn; a.blocks = wk_blocks(0);
            CK((launch_wgrad<T, ALOAD_ROW, X3>(h, wst, a, wpart, &pend)));
            if (!last) g_set1[bs] = sc.mark();
        }
        {
            WgradArgs a{};
            a.Y = b_dF; a.ldy = c.ffn; a.X = ws + L.h1[l]; a.ldx = d; a.dW = G(lp(l, L_L1_W)); a.ldw = d; a.db = G(lp(l, L_L1_B));
            a.R = Rl; a.N = c.ffn; a.K = d; a.blocks = wk_blocks(1);
            CK((launch_wgrad<T, ALOAD_ROW, X3>(h, wst, a, wpart, &pend)));
            if (!last) g_dF[bs] = sc.mark();
        }
        if (!chain_bwd && !chained_here) {   // dH1 = dPre2 + dF W1
            GemmArgs a{};
            a.A = b_dF; a.lda = c.ffn; a.W = ws + L.w_l1T[l]; a.ldw = c.ffn; a.C = b_dHb; a.ldc = d;
            a.M = Rl; a.N = d; a.K = c.ffn; a.R = b_dP; a.ldr = d;
            if (last && gemm_sk_shape<T>(a, L.skpart)) CK((launch_gemm_sk<T, false>(h, st, a, (float*)(ws + L.skpart))));
            else CK((gemm128<T, EPI_ADD, ALOAD_ROW, X3>(h, st, a)));
        }
        if (!last) sc.wait(g_set2[bs]);
        gm = d_sa.thr ? b_dM2 : b_dP2;
        GemmArgs ado{};        // dO = dA Wo
        ado.A = gm; ado.lda = d; ado.W = ws + L.w_outT[l]; ado.ldw = d; ado.C = b_dO; ado.ldc = d;
        ado.M = Rl; ado.N = d; ado.K = d;
        if (chained_here) {
        } else {
        sc.arm();
        if (ws_epilogue<T, EPI_MASK>() && ws_shape(ado) && lnfuse_on()) {
            // norm1 backward rides in the prologue of the dO GEMM
            LnFuseArgs f{};
            f.y = ws + L.h1[l]; f.ldy = d; f.dpre = b_dP2; f.dmask = d_sa.thr ? b_dM2 : nullptr;
            f.dgamma = G(lp(l, L_N1_W)); f.dbeta = G(lp(l, L_N1_B));
            ado.A = b_dHb; ado.gamma = P[lp(l, L_N1_W)]; ado.beta = P[lp(l, L_N1_B)];
            ado.rstd = (float*)(ws + L.rstd1[l]); ado.drop = d_sa; ado.drow_mul = rmul;
            if constexpr (sizeof(T) == 2) CK((launch_gemm_ws_lnbwd<T, EPI_NONE>(h, st, ado, f)));
        } else {
        {   // norm1 backward
            LnBwdArgs a{};
            a.dy = b_dHb; a.y = ws + L.h1[l]; a.gamma = P[lp(l, L_N1_W)]; a.beta = P[lp(l, L_N1_B)];
            a.rstd = (const float*)(ws + L.rstd1[l]); a.dpre = b_dP2; a.dmask = d_sa.thr ? b_dM2 : nullptr;
            a.dgamma = G(lp(l, L_N1_W)); a.dbeta = G(lp(l, L_N1_B)); a.R = Rl; a.drop = d_sa; a.drow_mul = rmul;
            auto kern = ln_bwd_kernel<T>;
            ProfScope ps(h, st, GE2E_K_LN_BWD, 12.0 * Rl * d, (double)Rl * d * L.esz * (d_sa.thr ? 4 : 3));
            GE2E_LAUNCH(h, kern, dim3(ln_grid), dim3(256), 0, st, a);
        }
        CK((gemm128<T, EPI_NONE, ALOAD_ROW, X3>(h, st, ado)));
        }
        sc.fork();
        }
        {
            WgradArgs a{};
            a.Y = gm; a.ldy = d; a.X = ws + L.o[l]; a.ldx = d; a.dW = G(lp(l, L_OUT_W)); a.ldw = d; a.db = G(lp(l, L_OUT_B));
            a.R = Rl; a.N = d; a.K = d; a.blocks = wk_blocks(2);
            CK((launch_wgrad<T, ALOAD_ROW, X3>(h, wst, a, wpart, &pend)));
            if (!last) g_set2[bs] = sc.mark();
        }
        sc.wait(g_dQKV[bq]);
        if (!last) {
            AttnArgs a{};
            a.qkv = ws + L.qkv[l]; a.dout = b_dO; a.dqkv = b_dQKV; a.T = t; a.H = c.heads; a.D = d;
            a.o = ws + L.o[l]; a.lse = (float*)(ws + L.lse[l]);
            a.scale = 1.0f / std::sqrt((float)(d / c.heads));
            a.drop = make_drop(true, c.tf_dropout, seed, step, site_attn(l));
            sc.arm();
            CK((launch_attn<std::conditional_t<X3 && sizeof(T) == 4, x3_t, T>>(h, st, a, n, true, L.adelta != (size_t)-1 ? (float*)(ws + L.adelta) : nullptr)));
            sc.fork();
            WgradArgs w{};
            w.Y = b_dQKV; w.ldy = 3 * d; w.X = hin; w.ldx = d; w.dW = G(lp(l, L_IN_W)); w.ldw = d; w.db = G(lp(l, L_IN_B));
            w.R = R; w.N = 3 * d; w.K = d; w.blocks = wk_blocks(3);
            CK((launch_wgrad<T, ALOAD_ROW, X3>(h, wst, w, wpart, &pend)));
            CK(flush_wk_reduce(h, wst, pend));
            g_dQKV[bq] = sc.mark();
            GemmArgs g{};   // dH(layer input) = dPre1 + dQKV Win
            g.A = b_dQKV; g.lda = 3 * d; g.W = ws + L.w_inT[l]; g.ldw = 3 * d; g.C = b_dHin; g.ldc = d;
            g.M = R; g.N = d; g.K = 3 * d; g.R = b_dP2; g.ldr = d;
            sc.wait(g_dH[l % L.nH]);
            if (uses_chain_bwd(l - 1) && !defer_colsum) sc.arm();
            CK((gemm128<T, EPI_ADD, ALOAD_ROW, X3>(h, st, g)));
            CK(colsum_below());
        } else {
            // one query per (utterance, head), K / V never materialised (attn_last.cuh): dL/d(layer input) of every frame -- the K / V
            // path, plus the residual path and the query on the frame-0 rows -- in ONE launch; dq0 and dqk for the weight gradients
            AttnLastArgs a = attn_last_args(c, L, ws, P, l, t, hin, true, make_drop(true, c.tf_dropout, seed, step, site_attn(l)));
            a.do0 = b_dO; a.dpre = b_dP2; a.dX = b_dHin; a.dq0 = ws + L.c_dQ0;
            sc.wait(g_dH[l % L.nH]);
            sc.arm();
            {
                ProfScope ps(h, st, GE2E_K_ATTN_BWD, 10.0 * t * 256.0 * 4.0 * n, (double)n * t * d * sizeof(T) * 2.0);
                auto kern = attn_last_bwd_kernel<T>;
                GE2E_LAUNCH(h, kern, dim3(n), dim3(256), attn_last_bwd_smem(t), st, a);
            }
            sc.fork();
            forked_after_dh = true;
            {   // k | v rows of in_proj_weight (and the v bias; the k bias has no gradient: its score term is constant over the frames)
                // (Parked ~180 us behind the chained FFN backward's persistent blocks here; on the main stream in front of the fork, or at the end of
                // the main chain, it is not -- and the step is no faster: 3.512-3.521 / 3.518-3.525 vs 3.501-3.504 ms, profiles/r04_ab_log.txt section 4.)
                auto kern = attn_last_wgrad_kernel<T>;
                const int chunks = std::max(1, std::min(16, n / 32)), per = (n + chunks - 1) / chunks;
                float* const dW = G(lp(l, L_IN_W));
                GE2E_LAUNCH(h, kern, dim3(128, chunks), dim3(256), 0, wst, (const void*)(ws + L.lq0), (const float*)(ws + L.ldqk), (const void*)b_dO,
                            (const float*)(ws + L.lctx), (const float*)(ws + L.lsp), dW + (size_t)d * d, dW + (size_t)2 * d * d, G(lp(l, L_IN_B)) + 2 * d, n, per);
            }
            WgradArgs wq{};   // q rows from frame 0 of every utterance
            wq.Y = ws + L.c_dQ0; wq.ldy = d; wq.X = hin; wq.ldx = d * t; wq.dW = G(lp(l, L_IN_W)); wq.ldw = d; wq.db = G(lp(l, L_IN_B));
            wq.R = n; wq.N = d; wq.K = d;
            CK((launch_wgrad<T, ALOAD_ROW, X3>(h, wst, wq)));
            CK(flush_wk_reduce(h, wst, pend));
            CK(colsum_below(true));
        }
        if (cb && !forked_after_dh) sc.fork();     // the bucket is final behind the side stream (ge2e_bucket_stream): it now also follows this layer's main-stream kernels
        if (tail_bucket_pending) { bucket(p_fn_w(c), p_proj_b(c)); tail_bucket_pending = false; }
        bucket(lp(l, 0), lp(l, L_COUNT - 1));
    }
    {   // through the PE dropout, alpha * pe and the ReLU, and the prenet's weight / bias gradients, in one launch (prenet_bwd.cuh): the masked
        // gradient of the pre-activation is formed on the way into the weight-gradient kernel's LDS tiles and never stored.  It stays on the
        // MAIN stream: the side stream's own last job (layer 0's in_proj gradient) ends later than this does.
        const bool unfused = opt(O_NO_PRENET_FUSE) != 0;
        unsigned char* const dH0 = ws + L.dH_of(-1);
        if (!unfused) {
            constexpr int RS = 2 * Prec<T>::KG;
            constexpr int LD = 128 * (int)sizeof(T) + (sizeof(T) == 2 ? 32 : 16);
            PrenetBwdArgs a{};
            a.dH0 = dH0; a.X = ws + L.xt; a.ldx = L.KP; a.bits = ws + L.pbits; a.pe_t = (const float*)(ws + L.pe_t);
            a.dW = G(P_PRENET_W); a.ldw = c.mel_dim; a.db = G(P_PRENET_B); a.dalpha = G(P_ALPHA);
            a.R = R; a.K = c.mel_dim; a.T = t; a.drop = make_drop(true, c.pe_dropout, seed, step, SITE_PE);
            // blocks per column tile: ~256, a multiple of T / gcd(T, RS) (every stage of a block then starts at the same frame; T <= max_position keeps it <= 1024)
            int g2 = t, b2 = RS; while (b2) { const int r2 = g2 % b2; g2 = b2; b2 = r2; }
            const int pd = t / g2;
            const int splits = pd * std::max(1, (256 + pd / 2) / pd);
            a.nsplit = splits;
            if (((long long)RS * splits) % t != 0) return fail(h, GE2E_EINVAL, "prenet backward: stages are not frame-aligned");
            const size_t smem = std::max<size_t>(4 * (size_t)RS * LD, 128 * (128 * 4 + 16));
            ProfScope ps(h, st, GE2E_K_WGRAD, 2.0 * R * d * c.mel_dim, (double)R * (d + L.KP) * sizeof(T) + (double)R * d / 8 + 4.0 * d * c.mel_dim);
            auto kern = prenet_bwd_kernel<T>;
            GE2E_LAUNCH(h, kern, dim3(2 * splits), dim3(256), smem, st, a);
        } else {
        GemmArgs a{};
        a.A = ws + L.xt; a.lda = L.KP; a.W = ws + L.w_prenet; a.ldw = L.KP; a.C = dH0; a.ldc = d; a.R = dH0; a.ldr = d;
        a.M = R; a.N = d; a.K = L.KP; a.bias = P[P_PRENET_B];
        a.drop = make_drop(true, c.pe_dropout, seed, step, SITE_PE);
        a.pe_t = (const float*)(ws + L.pe_t); a.dalpha = G(P_ALPHA); a.T = t; a.mel = c.mel_dim;
        CK((gemm128<T, EPI_PRENET_BWD, ALOAD_ROW, X3>(h, st, a)));
        WgradArgs w{};     // dWp[256][mel] from the packed rows; columns mel..127 of the tile are discarded (k < K)
        w.Y = dH0; w.ldy = d; w.X = ws + L.xt; w.ldx = L.KP; w.dW = G(P_PRENET_W); w.ldw = c.mel_dim; w.db = G(P_PRENET_B);
        w.R = R; w.N = d; w.K = c.mel_dim;
        CK((launch_wgrad<T, ALOAD_ROW, X3>(h, st, w)));                 // (no split-K scratch here: wpart belongs to the side stream)
        }
        if (cb) { sc.fork(); bucket(P_PRENET_W, P_ALPHA); }   // final behind the side stream, as the other buckets
        for (int q = 0; q < ndeferred; ++q) {
            auto kern = ln_colsum_kernel<T>;
            GE2E_LAUNCH(h, kern, dim3(std::min(512, (R + 31) / 32)), dim3(256), 0, st, deferred[q]);
        }
    }
    return 0;                                             // backward_impl joins the side stream: the caller's stream owns every gradient again
}

struct LossLayout { size_t cent, cn, en, rowloss, G, cosm, dC, rowwb, total; int Y; };
// slices of the utterances in the centroid-gradient pass (one partial slab each): ~60 rows per block, bounded so that the slabs stay small
inline int loss_slices(int S, int N) { return std::max(1, std::min(std::min(16, 2048 / std::max(S, 1)), N / 48)); }
LossLayout loss_layout(int S, int P, int d) {
    LossLayout L{};
    size_t off = 0;
    auto take = [&](size_t floats) { size_t o = off; off += (floats * 4 + 255) / 256 * 256; return o; };
    const size_t N = (size_t)S * P;
    L.cent = take((size_t)S * d); L.cn = take(S); L.en = take(N); L.rowloss = take(N);
    L.Y = loss_slices(S, (int)N);
    L.G = take(N * S); L.cosm = take(N * S); L.dC = take((size_t)L.Y * S * d); L.rowwb = take(2 * N);
    L.total = off;
    return L;
}
LossArgs loss_args(const float* emb, int S, int P, float w, float b, unsigned char* ws, const LossLayout& L) {
    LossArgs a{};
    a.emb = emb; a.N = S * P; a.S = S; a.P = P; a.w = w; a.b = b;
    a.cent = (float*)(ws + L.cent); a.cn = (float*)(ws + L.cn); a.en = (float*)(ws + L.en);
    a.rowloss = (float*)(ws + L.rowloss); a.G = (float*)(ws + L.G); a.cosm = (float*)(ws + L.cosm); a.dC = (float*)(ws + L.dC); a.Y = L.Y; a.rowwb = (float*)(ws + L.rowwb);
    return a;
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

#ifndef GE2E_SOURCE_HASH
#define GE2E_SOURCE_HASH "unversioned"
#endif
int ge2e_abi_version(void) { return GE2E_ABI_VERSION; }
const char* ge2e_source_hash(void) { return GE2E_SOURCE_HASH; }

int ge2e_create(const ge2e_config* cfg, ge2e_handle* out) {
    if (!cfg || !out) return GE2E_EINVAL;
    *out = nullptr;
    if (cfg->emb != 256 || cfg->heads <= 0 || cfg->emb / cfg->heads != 64 || cfg->emb % cfg->heads != 0) return GE2E_EUNSUPPORTED;
    if (cfg->layers < 1 || cfg->layers > MAX_LAYERS || cfg->mel_dim < 1 || cfg->mel_dim > 128) return GE2E_EUNSUPPORTED;
    if (cfg->ffn % 128 != 0 || cfg->ffn < 128 || cfg->max_position < 1) return GE2E_EUNSUPPORTED;
    if (cfg->precision != GE2E_PREC_F32 && cfg->precision != GE2E_PREC_BF16 && cfg->precision != GE2E_PREC_F16 && cfg->precision != GE2E_PREC_F32X3) return GE2E_EINVAL;
    if (cfg->pe_dropout < 0.f || cfg->pe_dropout >= 1.f || cfg->tf_dropout < 0.f || cfg->tf_dropout >= 1.f) return GE2E_EINVAL;
    ge2e_handle h = new (std::nothrow) ge2e_handle_s();
    if (!h) return GE2E_EINVAL;
    h->cfg = *cfg;
    h->overlap = opt(O_NO_OVERLAP) ? 0 : 1;
    build_params(h);
    *out = h;
    return 0;
}

int ge2e_set_option(const char* name, int value) {
    if (!name) return GE2E_EINVAL;
    for (int i = 0; i < O_COUNT; ++i)
        if (std::strcmp(name, OPT_DEFS[i].name) == 0) { g_opt[i].store(value, std::memory_order_relaxed); return 0; }
    return GE2E_EINVAL;
}
int ge2e_get_option(const char* name, int* value) {
    if (!name || !value) return GE2E_EINVAL;
    for (int i = 0; i < O_COUNT; ++i)
        if (std::strcmp(name, OPT_DEFS[i].name) == 0) { *value = opt(i); return 0; }
    return GE2E_EINVAL;
}
const char* ge2e_option_name(int index) { return index >= 0 && index < O_COUNT ? OPT_DEFS[index].name : nullptr; }

int ge2e_destroy(ge2e_handle h) {
    if (!h) return 0;
    for (auto* s : h->event_sets) { for (hipEvent_t e : s->ev) hipEventDestroy(e); if (s->done) hipEventDestroy(s->done); delete s; }
    for (hipEvent_t e : h->ev_pool) hipEventDestroy(e);
    if (h->side) hipStreamDestroy(h->side);
    delete h;
    return 0;
}

const char* ge2e_last_error(ge2e_handle h) {
    if (!h) return "null handle";
    std::lock_guard<std::mutex> g(h->mu);
    static thread_local std::string copy;
    copy = h->err;
    return copy.c_str();
}

int ge2e_param_count(ge2e_handle h) { return h ? (int)h->params.size() : GE2E_EINVAL; }
const char* ge2e_param_name(ge2e_handle h, int i) { return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].name.c_str() : nullptr; }
int64_t ge2e_param_numel(ge2e_handle h, int i) { return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].numel : -1; }
int64_t ge2e_param_offset(ge2e_handle h, int i) { return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].offset : -1; }
int64_t ge2e_param_total(ge2e_handle h) { return h ? h->total : -1; }
int ge2e_max_frames(ge2e_handle h) { return h ? std::min(MAX_FRAMES, h->cfg.max_position) : MAX_FRAMES; }

size_t ge2e_workspace_bytes(ge2e_handle h, int n_utts, int frames, int train) {
    if (!h || n_utts <= 0 || frames <= 0) return 0;
    return build_layout(h->cfg, n_utts, frames, train).total;
}

static int encoder_forward_any(ge2e_handle h, void* stream, const void* mel, bool mel_f16, int n_utts, int frames, int samples,
                               const float* const* params, const float* pe, float* out_emb,
                               void* workspace, size_t workspace_bytes, int train, uint64_t seed, uint64_t step) {
    if (!h) return GE2E_EINVAL;
    if (!mel || !params || !pe || !out_emb) return fail(h, GE2E_EINVAL, "null pointer argument");
    for (size_t i = 0; i < h->params.size(); ++i)
        if (!params[i]) return fail(h, GE2E_EINVAL, "null parameter pointer: " + h->params[i].name);
    if (train < 0 || train > 2) return fail(h, GE2E_EINVAL, "train: 0, 1 or GE2E_FWD_PREPARED");
    const bool prepared = train == GE2E_FWD_PREPARED;     // eval with the weight copies already in this workspace
    train = train == 1;
    const Layout L = build_layout(h->cfg, n_utts, frames, train);
    CK(check_common(h, n_utts, frames, samples, workspace, workspace_bytes, L));
    hipStream_t st = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)workspace;
    if (h->cfg.precision == GE2E_PREC_BF16)
        return forward_impl<bf16_t>(h, st, mel, mel_f16, n_utts, frames, samples, params, pe, out_emb, ws, L, train != 0, seed, step, prepared);
    if (h->cfg.precision == GE2E_PREC_F16)
        return forward_impl<f16_t>(h, st, mel, mel_f16, n_utts, frames, samples, params, pe, out_emb, ws, L, train != 0, seed, step, prepared);
    if (h->cfg.precision == GE2E_PREC_F32X3)
        return forward_impl<float, true>(h, st, mel, mel_f16, n_utts, frames, samples, params, pe, out_emb, ws, L, train != 0, seed, step, prepared);
    return forward_impl<float>(h, st, mel, mel_f16, n_utts, frames, samples, params, pe, out_emb, ws, L, train != 0, seed, step, prepared);
}

int ge2e_encoder_forward(ge2e_handle h, void* stream, const float* mel, int n_utts, int frames, int samples,
                         const float* const* params, const float* pe, float* out_emb,
                         void* workspace, size_t workspace_bytes, int train, uint64_t seed, uint64_t step) {
    return encoder_forward_any(h, stream, mel, false, n_utts, frames, samples, params, pe, out_emb, workspace, workspace_bytes, train, seed, step);
}

int ge2e_encoder_forward_mel16(ge2e_handle h, void* stream, const void* mel_f16, int n_utts, int frames, int samples,
                               const float* const* params, const float* pe, float* out_emb,
                               void* workspace, size_t workspace_bytes, int train, uint64_t seed, uint64_t step) {
    return encoder_forward_any(h, stream, mel_f16, true, n_utts, frames, samples, params, pe, out_emb, workspace, workspace_bytes, train, seed, step);
}

int ge2e_encoder_backward(ge2e_handle h, void* stream, const float* mel, int n_utts, int frames, int samples,
                          const float* const* params, const float* d_emb, float* grads_flat,
                          void* workspace, size_t workspace_bytes, uint64_t seed, uint64_t step) {
    return ge2e_encoder_backward_cb(h, stream, mel, n_utts, frames, samples, params, d_emb, grads_flat,
                                    workspace, workspace_bytes, seed, step, nullptr, nullptr);
}

int ge2e_encoder_backward_cb(ge2e_handle h, void* stream, const float* mel, int n_utts, int frames, int samples,
                             const float* const* params, const float* d_emb, float* grads_flat,
                             void* workspace, size_t workspace_bytes, uint64_t seed, uint64_t step,
                             ge2e_bucket_cb cb, void* user) {
    if (!h) return GE2E_EINVAL;
    if (!mel || !params || !d_emb || !grads_flat) return fail(h, GE2E_EINVAL, "null pointer argument");
    for (size_t i = 0; i < h->params.size(); ++i)
        if (!params[i]) return fail(h, GE2E_EINVAL, "null parameter pointer: " + h->params[i].name);
    const Layout L = build_layout(h->cfg, n_utts, frames, 1);
    CK(check_common(h, n_utts, frames, samples, workspace, workspace_bytes, L));
    hipStream_t st = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)workspace;
    if (h->cfg.precision == GE2E_PREC_BF16)
        return backward_impl<bf16_t>(h, st, mel, n_utts, frames, samples, params, d_emb, grads_flat, ws, L, seed, step, cb, user);
    if (h->cfg.precision == GE2E_PREC_F16)
        return backward_impl<f16_t>(h, st, mel, n_utts, frames, samples, params, d_emb, grads_flat, ws, L, seed, step, cb, user);
    if (h->cfg.precision == GE2E_PREC_F32X3)
        return backward_impl<float, true>(h, st, mel, n_utts, frames, samples, params, d_emb, grads_flat, ws, L, seed, step, cb, user);
    return backward_impl<float>(h, st, mel, n_utts, frames, samples, params, d_emb, grads_flat, ws, L, seed, step, cb, user);
}

size_t ge2e_loss_workspace_bytes(int speakers, int utts, int emb) {
    if (speakers <= 0 || utts <= 0 || emb <= 0) return 0;
    return loss_layout(speakers, utts, emb).total;
}

int ge2e_loss_forward(ge2e_handle h, void* stream, const float* emb, int speakers, int utts,
                      float w, float b, float* loss, void* loss_ws, size_t loss_ws_bytes) {
    if (!h) return GE2E_EINVAL;
    if (!emb || !loss || !loss_ws || speakers <= 0 || utts <= 0) return fail(h, GE2E_EINVAL, "loss: bad argument");
    if (speakers > 8192) return fail(h, GE2E_EUNSUPPORTED, "loss: more than 8192 speakers per batch");
    const LossLayout L = loss_layout(speakers, utts, h->cfg.emb);
    if (loss_ws_bytes < L.total) return fail(h, GE2E_EWORKSPACE, "loss workspace too small");
    CK(check_device(h));
    hipStream_t st = (hipStream_t)stream;
    LossArgs a = loss_args(emb, speakers, utts, w, b, (unsigned char*)loss_ws, L);
    a.loss = loss;
    GE2E_LAUNCH(h, loss_centroid_kernel, dim3(speakers), dim3(256), 0, st, a);
    GE2E_LAUNCH(h, loss_row_kernel, dim3(a.N), dim3(256), (size_t)2 * speakers * 4, st, a);
    GE2E_LAUNCH(h, loss_reduce_kernel, dim3(1), dim3(256), 0, st, a);
    return 0;
}

int ge2e_loss_backward(ge2e_handle h, void* stream, const float* emb, int speakers, int utts,
                       float w, float b, const float* d_loss, float* d_emb, float* d_weight_bias, void* loss_ws, size_t loss_ws_bytes) {
    if (!h) return GE2E_EINVAL;
    if (!emb || !d_loss || !d_emb || !loss_ws || speakers <= 0 || utts <= 0) return fail(h, GE2E_EINVAL, "loss: bad argument");
    const LossLayout L = loss_layout(speakers, utts, h->cfg.emb);
    if (loss_ws_bytes < L.total) return fail(h, GE2E_EWORKSPACE, "loss workspace too small");
    CK(check_device(h));
    hipStream_t st = (hipStream_t)stream;
    LossArgs a = loss_args(emb, speakers, utts, w, b, (unsigned char*)loss_ws, L);
    a.gscale = d_loss; a.d_emb = d_emb; a.dwb = d_weight_bias;
    GE2E_LAUNCH(h, loss_bwd_centroid_kernel, dim3(speakers, a.Y), dim3(256), 0, st, a);
    GE2E_LAUNCH(h, loss_bwd_row_kernel, dim3(a.N), dim3(256), 0, st, a);
    if (d_weight_bias) GE2E_LAUNCH(h, loss_wb_reduce_kernel, dim3(1), dim3(256), 0, st, a);
    return 0;
}

static int clip_adamw_any(ge2e_handle h, void* stream, int count, float* const* params, float* const* grads,
                          float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                          float* norm_scratch, float max_norm, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int64_t step, float* scaler, float growth, float backoff, int growth_interval) {
    if (!h) return GE2E_EINVAL;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !norm_scratch || count <= 0 || (!scaler && step < 1))
        return fail(h, GE2E_EINVAL, "clip_adamw: bad argument");
    if (scaler && (!(growth >= 1.0f) || !(backoff > 0.0f && backoff <= 1.0f) || growth_interval < 1))
        return fail(h, GE2E_EINVAL, "clip_adamw: bad loss-scaling constants");
    CK(check_device(h));
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < count; ++i)
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] <= 0 || numel[i] > 2147483647LL)
            return fail(h, GE2E_EINVAL, "clip_adamw: null tensor or bad numel");
    const bool need_norm = max_norm > 0.0f || scaler != nullptr;      // the norm doubles as the inf / nan detector
    if (need_norm) {
        hipError_t e = hipMemsetAsync(norm_scratch, 0, 4, st);
        if (e != hipSuccess) return fail_hip(h, e, "zero norm");
    }
    OptArgs last{};
    for (int pass = (need_norm ? 0 : 1); pass < 2; ++pass)          // pass 0: squared norm, pass 1: update
        for (int b = 0; b < count; b += OPT_MAX_TENSORS) {
            OptArgs a{};
            const int cnt = std::min(OPT_MAX_TENSORS, count - b);
            int chunks = 0;
            for (int i = 0; i < cnt; ++i) {
                a.p[i] = params[b + i]; a.g[i] = grads[b + i]; a.m[i] = exp_avg[b + i]; a.v[i] = exp_avg_sq[b + i];
                a.numel[i] = (int)numel[b + i]; a.chunk0[i] = chunks;
                chunks += (int)((numel[b + i] + OPT_CHUNK - 1) / OPT_CHUNK);
            }
            a.chunk0[cnt] = chunks; a.ntensors = cnt; a.sumsq = norm_scratch; a.max_norm = max_norm;
            a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
            a.scaler = scaler; a.growth = growth; a.backoff = backoff; a.growth_interval = growth_interval;
            if (!scaler) {
                a.bc1 = (float)(1.0 - std::pow((double)beta1, (double)step));
                a.bc2_sqrt = (float)std::sqrt(1.0 - std::pow((double)beta2, (double)step));
            }
            if (pass == 0) GE2E_LAUNCH(h, opt_norm_kernel, dim3(chunks), dim3(256), 0, st, a);
            else GE2E_LAUNCH(h, opt_adamw_kernel, dim3(chunks), dim3(256), 0, st, a);
            last = a;
        }
    if (scaler) GE2E_LAUNCH(h, opt_scaler_update_kernel, dim3(1), dim3(64), 0, st, last);
    return 0;
}

int ge2e_clip_adamw_step(ge2e_handle h, void* stream, int count, float* const* params, float* const* grads,
                         float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                         float* norm_scratch, float max_norm, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int64_t step) {
    return clip_adamw_any(h, stream, count, params, grads, exp_avg, exp_avg_sq, numel, norm_scratch, max_norm, lr, beta1, beta2,
                          eps, weight_decay, step, nullptr, 1.0f, 1.0f, 1);
}

int ge2e_clip_adamw_step_scaled(ge2e_handle h, void* stream, int count, float* const* params, float* const* grads,
                                float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                                float* norm_scratch, float max_norm, float lr, float beta1, float beta2, float eps,
                                float weight_decay, float* scaler_state, float growth_factor, float backoff_factor,
                                int growth_interval) {
    if (h && !scaler_state) return fail(h, GE2E_EINVAL, "clip_adamw: null scaler state");
    return clip_adamw_any(h, stream, count, params, grads, exp_avg, exp_avg_sq, numel, norm_scratch, max_norm, lr, beta1, beta2,
                          eps, weight_decay, 0, scaler_state, growth_factor, backoff_factor, growth_interval);
}

void* ge2e_bucket_stream(ge2e_handle h, void* stream) {
    if (!h || !h->overlap || !h->side) return stream;
    { std::lock_guard<std::mutex> g(h->mu); if (h->prof_mask & GE2E_K_SERIAL) return stream; }   // a serialised backward produces everything on the caller's stream
    return (void*)h->side;
}

// ---- wav -> log-mel front-end (reference meldataset.py:73-96)
namespace {
struct MelLayout { int F, bins, n1, k2, rows; size_t W, A, S, Mg, P, Ml, total; };
bool mel_layout(int batch, int samples, int n_fft, int hop, int n_mels, MelLayout& m) {
    if (batch <= 0 || samples <= 0 || n_fft < 128 || n_fft % 32 != 0 || hop <= 0 || hop > n_fft || (n_fft - hop) % 2 != 0 || n_mels < 1 || n_mels > 128) return false;
    const int pad = (n_fft - hop) / 2;
    if (samples <= pad || samples + 2 * pad < n_fft) return false;           // reflect padding needs pad < samples
    m.F = (samples + 2 * pad - n_fft) / hop + 1;
    m.bins = n_fft / 2 + 1; m.n1 = (2 * m.bins + 127) / 128 * 128; m.k2 = (m.bins + 31) / 32 * 32; m.rows = batch * m.F;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    m.W = take((size_t)m.n1 * n_fft * 4); m.A = take((size_t)m.rows * n_fft * 4); m.S = take((size_t)m.rows * m.n1 * 4);
    m.Mg = take((size_t)m.rows * m.k2 * 4); m.P = take((size_t)128 * m.k2 * 4); m.Ml = take((size_t)m.rows * 128 * 4);
    m.total = off;
    return true;
}
}  // namespace

int ge2e_mel_frames(int samples, int n_fft, int hop) {
    MelLayout m;
    return mel_layout(1, samples, n_fft, hop, 1, m) ? m.F : GE2E_EINVAL;
}
size_t ge2e_mel_workspace_bytes(int batch, int samples, int n_fft, int hop, int n_mels) {
    MelLayout m;
    return mel_layout(batch, samples, n_fft, hop, n_mels, m) ? m.total : 0;
}
int ge2e_mel_spectrogram(ge2e_handle h, void* stream, const float* wav, int batch, int samples, int n_fft, int hop, int n_mels,
                         const float* mel_basis, float* out_logmel, void* workspace, size_t workspace_bytes) {
    if (!h) return GE2E_EINVAL;
    if (!wav || !mel_basis || !out_logmel || !workspace) return fail(h, GE2E_EINVAL, "null pointer argument");
    MelLayout m;
    if (!mel_layout(batch, samples, n_fft, hop, n_mels, m)) return fail(h, GE2E_EUNSUPPORTED, "mel front-end: unsupported n_fft / hop / n_mels / length");
    if (workspace_bytes < m.total) return fail(h, GE2E_EINVAL, "mel front-end: workspace too small");
    CK(check_device(h));
    hipStream_t st = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)workspace;
    const int pad = (n_fft - hop) / 2;
    GE2E_LAUNCH(h, dft_basis_kernel, dim3(m.n1), dim3(256), 0, st, (float*)(ws + m.W), n_fft, m.bins, m.n1);
    GE2E_LAUNCH(h, pad_basis_kernel, dim3(128), dim3(256), 0, st, mel_basis, (float*)(ws + m.P), n_mels, m.bins, m.k2);
    GE2E_LAUNCH(h, frame_window_kernel, dim3(m.rows), dim3(256), 0, st, wav, (float*)(ws + m.A), samples, m.F, n_fft, hop, pad);
    {
        GemmArgs a{};
        a.A = ws + m.A; a.lda = n_fft; a.W = ws + m.W; a.ldw = n_fft; a.C = ws + m.S; a.ldc = m.n1; a.M = m.rows; a.N = m.n1; a.K = n_fft;
        CK((launch_gemm<float, 128, 128, 64, 64, EPI_NONE, ALOAD_ROW, 2>(h, st, a)));
    }
    GE2E_LAUNCH(h, mag_kernel, dim3(m.rows), dim3(256), 0, st, (const float*)(ws + m.S), m.n1, (float*)(ws + m.Mg), m.k2, m.bins, m.rows);
    {
        GemmArgs a{};
        a.A = ws + m.Mg; a.lda = m.k2; a.W = ws + m.P; a.ldw = m.k2; a.C = ws + m.Ml; a.ldc = 128; a.M = m.rows; a.N = 128; a.K = m.k2;
        CK((launch_gemm<float, 128, 128, 64, 64, EPI_NONE, ALOAD_ROW, 2>(h, st, a)));
    }
    GE2E_LAUNCH(h, logmel_kernel, dim3((m.F + 31) / 32, batch), dim3(256), 0, st, (const float*)(ws + m.Ml), out_logmel, m.F, n_mels);
    return 0;
}

int ge2e_profile_enable(ge2e_handle h, int class_mask) {
    if (!h) return GE2E_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    h->prof_mask = class_mask;
    return 0;
}

int ge2e_profile_read(ge2e_handle h, int klass, double* total_ms, double* total_work, double* total_bytes, int64_t* launches) {
    if (!h || !total_ms || !total_work || !total_bytes || !launches) return GE2E_EINVAL;
    std::vector<ProfRec> mine, rest;
    {
        std::lock_guard<std::mutex> g(h->mu);
        for (auto& r : h->prof) (r.klass == klass ? mine : rest).push_back(r);
        h->prof.swap(rest);
    }
    double ms = 0.0, work = 0.0, bytes = 0.0;
    int64_t nlaunch = 0;
    for (auto& r : mine) {
        hipError_t e = hipEventSynchronize(r.stop);
        float t = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&t, r.start, r.stop);
        if (e != hipSuccess) return fail_hip(h, e, "profile event");
        ms += t; work += r.work; bytes += r.bytes; nlaunch += r.counts;
    }
    {
        std::lock_guard<std::mutex> g(h->mu);
        for (auto& r : mine) { h->ev_pool.push_back(r.start); h->ev_pool.push_back(r.stop); }
    }
    *total_ms = ms; *total_work = work; *total_bytes = bytes; *launches = nlaunch;
    return 0;
}

int ge2e_debug_tap(ge2e_handle h, const char* name, int n_utts, int frames, int train,
                   size_t* offset_bytes, size_t* size_bytes) {
    if (!h || !name || !offset_bytes || !size_bytes) return GE2E_EINVAL;
    const Layout L = build_layout(h->cfg, n_utts, frames, train);
    const size_t d = h->cfg.emb, e = L.esz;
    size_t R = (size_t)L.R;
    std::string s(name);
    int l = 0;
    const size_t dot = s.find('.');
    std::string base = s;
    if (dot != std::string::npos) { base = s.substr(0, dot); l = std::atoi(s.c_str() + dot + 1); }
    if (l < 0 || l >= h->cfg.layers) return GE2E_EINVAL;
    const bool lastl = l == h->cfg.layers - 1;
    const size_t Rl = lastl ? (size_t)n_utts : R;     // the last layer keeps compact rows (frame 0 only)
    if (base == "h0") { *offset_bytes = L.h0; *size_bytes = R * d * e; }
    else if (base == "qkv" && !lastl) { *offset_bytes = L.qkv[l]; *size_bytes = R * 3 * d * e; }
    else if (base == "q0") { *offset_bytes = L.lq0; *size_bytes = (size_t)n_utts * d * e; }               // the last layer's query, frame 0 (compact)
    else if (base == "o") { *offset_bytes = L.o[l]; *size_bytes = Rl * d * e; }
    else if (base == "h1") { *offset_bytes = L.h1[l]; *size_bytes = Rl * d * e; }
    else if (base == "f") { *offset_bytes = L.f[l]; *size_bytes = Rl * (size_t)h->cfg.ffn * e; }
    else if (base == "h2") { *offset_bytes = L.h2[l]; *size_bytes = Rl * d * e; }
    else if (base == "xt") { *offset_bytes = L.xt; *size_bytes = R * (size_t)L.KP * e; }
    else if (train && base == "rstd1") { *offset_bytes = L.rstd1[l]; *size_bytes = Rl * 4; }          // fp32
    else if (train && base == "rstd2") { *offset_bytes = L.rstd2[l]; *size_bytes = Rl * 4; }          // fp32
    else if (train && base == "lse" && !lastl) { *offset_bytes = L.lse[l]; *size_bytes = R * (size_t)h->cfg.heads * 4; }   // fp32
    // backward scratch of FULL-SIZE layer l (name.l): the buffer set that layer uses, see Layout
    else if (train && base == "dP1") { *offset_bytes = L.dP[l % L.nset]; *size_bytes = R * d * e; }  // norm2-backward outputs (set 1)
    else if (train && base == "dM1") { *offset_bytes = L.dM[l % L.nset]; *size_bytes = R * d * e; }
    else if (train && base == "dHa") { *offset_bytes = L.dH_of(l); *size_bytes = R * d * e; }         // dL/d(output of layer l)
    else if (train && base == "dHin") { *offset_bytes = L.dH_of(l - 1); *size_bytes = R * d * e; }    // dL/d(input of layer l)
    else if (train && base == "dF") { *offset_bytes = L.dF[l % L.nset]; *size_bytes = R * (size_t)h->cfg.ffn * e; }
    else if (train && base == "dHb") { *offset_bytes = L.dHb; *size_bytes = R * d * e; }
    else if (train && base == "dP") { *offset_bytes = L.dP2[l % L.nset]; *size_bytes = R * d * e; }
    else if (train && base == "dM") { *offset_bytes = L.dM2[l % L.nset]; *size_bytes = R * d * e; }
    else if (train && base == "dO") { *offset_bytes = L.dO; *size_bytes = R * d * e; }
    else if (train && base == "dQKV") { *offset_bytes = L.dQKV[l % L.nqkv]; *size_bytes = R * 3 * d * e; }
    else return GE2E_EINVAL;
    return 0;
}

uint32_t ge2e_drop_key(uint64_t seed, uint64_t step, int site) {
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + step * 0xBF58476D1CE4E5B9ull + (uint64_t)(site + 1) * 0x94D049BB133111EBull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z & 0xFFFFFFFFull);
}

int ge2e_drop_keep(uint32_t key, uint32_t index, float p) {
    const uint32_t thr = (uint32_t)std::floor((double)p * 65536.0);
    return drop_keep(index, key, thr) ? 1 : 0;
}

}  // extern "C"
