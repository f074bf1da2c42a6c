/* ge2e_hip.h -- C ABI of libge2e_hip.so: the MI355X (gfx950) GE2E speaker-embedding hot path.
 *
 * The reference (CODEJIN/Speaker_Embedding_Torch) has no FFI; its seam for this path is the Python
 * nn.Module API.  Each entry point below names the reference code it replaces:
 *
 *   ge2e_encoder_forward   <-  Modules.py:46-59   GE2E.forward            (callers Train.py:146,204,239; Inference.py:159)
 *   ge2e_encoder_backward  <-  autograd of the above at Train.py:153      (loss.backward())
 *   ge2e_loss_forward      <-  Modules.py:121-156 GE2E_Loss.forward       (callers Train.py:147-150,205-208)
 *   ge2e_loss_backward     <-  autograd of the above at Train.py:153
 *   ge2e_param_*           <-  the state_dict key set loaded strictly at Train.py:285 / Inference.py:212
 *
 * Conventions: plain pointers and sizes only; all device buffers are allocated and owned by the caller
 * (PyTorch); every call is asynchronous on the hipStream_t it is handed and never synchronises the device;
 * no state is kept per thread ACROSS calls (backward is invoked from the autograd thread): the one thread-local the library
 * has lives strictly inside a ge2e_encoder_backward* call, between two of its own launches, and is empty whenever the bucket
 * callback runs -- the callback may therefore call back into the library on any stream.
 * Every function returns 0 on success, a negative GE2E_E* code for invalid arguments, or a positive
 * hipError_t passed through; ge2e_last_error() gives the text.  No C++ exception crosses this boundary.
 */
#ifndef GE2E_HIP_H
#define GE2E_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GE2E_ABI_VERSION 4

enum {
    GE2E_OK = 0,
    GE2E_EINVAL = -1,        /* bad argument / null pointer */
    GE2E_EUNSUPPORTED = -2,  /* shape outside what the kernels were built for */
    GE2E_EWORKSPACE = -3     /* workspace too small */
};

/* Arithmetic modes.  F32: fp32 MFMA (exact fma chains; the parity path, reference `Use_Mixed_Precision: false`).
 * BF16 / F16: activations and activation gradients stored in 16 bits, fp32 accumulation, fp32 master weights, fp32
 * parameter gradients.  F16 is what the reference's own mixed precision is (torch.cuda.amp.autocast + GradScaler,
 * Train.py:134,145,153-162): its gradients need the loss scaling of ge2e_clip_adamw_step_scaled; BF16 does not.
 * F32X3 (round 4): fp32 STORAGE everywhere, exactly the F32 mode's tensors and workspace, with the projection GEMMs and weight gradients
 * computed on the bf16 matrix pipe from operands split into bf16 hi + lo halves (three products per term, fp32 accumulation: ~2^-17
 * relative per product instead of 2^-24); attention, LayerNorm, tail and loss stay exact fp32.  d-vectors stay within the 1e-4 of
 * north_star at a multiple of the F32 mode's speed. */
enum { GE2E_PREC_F32 = 0, GE2E_PREC_BF16 = 1, GE2E_PREC_F16 = 2, GE2E_PREC_F32X3 = 3 };

/* Mirrors the `Sound.Mel_Dim` / `GE2E.*` block of Hyper_Parameters.yaml:1-18 plus the arithmetic mode. */
typedef struct ge2e_config {
    int32_t mel_dim;         /* Sound.Mel_Dim (80)                                   */
    int32_t emb;             /* GE2E.Embedding_Size (256; the kernels require 256)   */
    int32_t heads;           /* GE2E.Transformer.Head (emb / heads must be 64)       */
    int32_t layers;          /* GE2E.Transformer.Num_Layers                          */
    int32_t ffn;             /* dim_feedforward = 4 * emb (Modules.py:28)            */
    int32_t max_position;    /* GE2E.Positional_Encoding.Max_Position                */
    float pe_dropout;        /* GE2E.Positional_Encoding.Dropout_Rate                */
    float tf_dropout;        /* GE2E.Transformer.Dropout_Rate                        */
    float ln_eps;            /* torch LayerNorm default 1e-5                         */
    int32_t precision;       /* GE2E_PREC_F32 / _BF16 / _F16 / _F32X3 */
} ge2e_config;

typedef struct ge2e_handle_s* ge2e_handle;

int ge2e_abi_version(void);
/* SHA-256 prefix of the sources this binary was compiled from (speaker_embedding_torch_amd/_build.py): the loader refuses
 * a library that does not match the csrc/ it sits beside. */
const char* ge2e_source_hash(void);
int ge2e_create(const ge2e_config* cfg, ge2e_handle* out);     /* no GPU call is made here */
int ge2e_destroy(ge2e_handle h);
const char* ge2e_last_error(ge2e_handle h);

/* Parameter table == reference GE2E.parameters() order (43 tensors, 2,456,321 floats for the default config). */
int ge2e_param_count(ge2e_handle h);
const char* ge2e_param_name(ge2e_handle h, int index);                 /* state_dict key */
int64_t ge2e_param_numel(ge2e_handle h, int index);
int64_t ge2e_param_offset(ge2e_handle h, int index);                   /* element offset in the flat gradient buffer */
int64_t ge2e_param_total(ge2e_handle h);

/* Bytes of caller-owned scratch for n_utts x frames.  train != 0 keeps the activations backward needs. */
size_t ge2e_workspace_bytes(ge2e_handle h, int n_utts, int frames, int train);
/* Longest frame count accepted: min(1024, max_position).  Up to 288 frames a head's sequence is LDS-resident (the tuned path:
 * the reference trains on <= 270 frames and infers on 64 / 240); longer inputs take chunked streaming kernels. */
int ge2e_max_frames(ge2e_handle h);

/* mel:    device fp32 [n_utts, mel_dim, frames]  (channels-first, as Datasets.Collater produces, Datasets.py:84)
 * params: HOST array of ge2e_param_count() DEVICE pointers (fp32), table order
 * pe:     device fp32 [emb, max_position]  -- the `positional_encoding.pe` buffer of the state_dict
 * out_emb:device fp32 [n_utts / samples, emb]  unit-norm d-vectors
 * train:  0 = eval (no dropout, nothing kept); 1 = train (dropout from (seed, step), activations kept in workspace);
 *         eval only, GE2E_FWD_PREPARED (2): the caller states that this workspace still holds the 16-bit weight copies and tables an
 *         earlier EVAL forward of this handle wrote for the same parameter values, n_utts and frames (an inference loop over one
 *         checkpoint), behind `stream`: they are not prepared again (3 launches, ~20 us of a 650 us configs[3] forward) */
#define GE2E_FWD_PREPARED 2
int ge2e_encoder_forward(ge2e_handle h, void* stream, const float* mel, int n_utts, int frames, int samples,
                         const float* const* params, const float* pe, float* out_emb,
                         void* workspace, size_t workspace_bytes, int train, uint64_t seed, uint64_t step);

/* Same forward from an IEEE fp16 mel batch [n_utts, mel_dim, frames] (the reference stores patterns as fp16,
 * Pattern_Generator.py:191-198, and widens them on the host, Datasets.py:84): the batch crosses PCIe at half the size
 * and is widened by the packing pass.  Results are identical to ge2e_encoder_forward on the widened values. */
int ge2e_encoder_forward_mel16(ge2e_handle h, void* stream, const void* mel_f16, int n_utts, int frames, int samples,
                               const float* const* params, const float* pe, float* out_emb,
                               void* workspace, size_t workspace_bytes, int train, uint64_t seed, uint64_t step);

/* Must follow a train-mode forward on the SAME workspace, inputs, seed and step.
 * mel:        the forward's input pointer (either dtype; not read again: the packed rows live in the workspace)
 * d_emb:      device fp32 [n_utts / samples, emb]
 * grads_flat: device fp32 [ge2e_param_total()], OVERWRITTEN with dL/dparam at ge2e_param_offset(i); any 4-byte-aligned address
 *             (an offset view of a larger buffer is fine).
 * Reproducibility: the forward is bitwise reproducible; the parameter gradients are reproducible to fp32 SUMMATION ORDER only
 * (bias gradients, the small-shape weight-gradient kernel and the split-K reduce pass add partial sums with float atomics:
 * two runs of one step agree to ~1e-6 relative, not bit for bit). */
int ge2e_encoder_backward(ge2e_handle h, void* stream, const float* mel, int n_utts, int frames, int samples,
                          const float* const* params, const float* d_emb, float* grads_flat,
                          void* workspace, size_t workspace_bytes, uint64_t seed, uint64_t step);

/* Same, reporting gradient buckets as soon as their last kernel has been ENQUEUED (replaces the post-backward
 * queue_callback of reference distributed.py:114-118): cb(user, element_offset, element_count) is called on the
 * calling host thread, in this order: [final norm + projection], [layer L-1], ..., [layer 0], [prenet + alpha].
 * At the time of the call everything that writes the bucket is enqueued on ge2e_bucket_stream(h, stream) -- the
 * library's weight-gradient stream, which the caller's `stream` does NOT wait for until the call returns -- so the
 * caller orders that bucket's all-reduce behind THAT stream (record an event there / make it current) and the
 * backward chain on `stream` is not held up by the hand-off. */
typedef void (*ge2e_bucket_cb)(void* user, int64_t element_offset, int64_t element_count);
/* The stream behind which a bucket reported by ge2e_encoder_backward_cb(…, stream, …) is final: the internal
 * weight-gradient stream, or `stream` itself when the overlap is off (option "no_overlap").  Valid inside the callback. */
void* ge2e_bucket_stream(ge2e_handle h, void* stream);
int ge2e_encoder_backward_cb(ge2e_handle h, void* stream, const float* mel, int n_utts, int frames, int samples,
                             const float* const* params, const float* d_emb, float* grads_flat,
                             void* workspace, size_t workspace_bytes, uint64_t seed, uint64_t step,
                             ge2e_bucket_cb cb, void* user);

/* GE2E loss over speakers x utts embeddings (row i belongs to speaker i / utts).
 * loss_ws: device scratch of ge2e_loss_workspace_bytes(); kept between forward and backward. */
size_t ge2e_loss_workspace_bytes(int speakers, int utts, int emb);
int ge2e_loss_forward(ge2e_handle h, void* stream, const float* emb, int speakers, int utts,
                      float w, float b, float* loss, void* loss_ws, size_t loss_ws_bytes);
/* d_loss: device fp32 scalar (upstream gradient); d_emb: device fp32 [speakers*utts, emb]; d_weight_bias: device fp32 [2] receiving
 * dL/dw and dL/db of the criterion's own parameters (reference Modules.py:115-116: nn.Parameters that autograd fills although
 * nothing ever optimises them, Train.py:121-127), or NULL.  (dL/db is zero up to rounding: a shift of all logits of a row.) */
int ge2e_loss_backward(ge2e_handle h, void* stream, const float* emb, int speakers, int utts,
                       float w, float b, const float* d_loss, float* d_emb, float* d_weight_bias, void* loss_ws, size_t loss_ws_bytes);

/* clip_grad_norm_(max_norm) followed by one torch.optim.AdamW step (reference Train.py:154-162) over `count`
 * parameter tensors, fused into two launches.  params/grads/exp_avg/exp_avg_sq: HOST arrays of device pointers
 * (fp32, numel[i] elements each); norm_scratch: device fp32 scalar (receives the squared total gradient norm);
 * step: 1-based AdamW step count; max_norm <= 0 disables clipping.  Gradients are left clipped in place, exactly
 * as clip_grad_norm_ leaves them. */
int ge2e_clip_adamw_step(ge2e_handle h, void* stream, int count, float* const* params, float* const* grads,
                         float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                         float* norm_scratch, float max_norm, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int64_t step);

/* The same step under mixed-precision loss scaling, with NO host synchronisation (replaces GradScaler.unscale_ /
 * clip_grad_norm_ / GradScaler.step / GradScaler.update of reference Train.py:154-162).  `grads` hold scale x the true
 * gradient (the caller multiplied the loss by scaler_state[0] before backward, as GradScaler.scale does).
 * scaler_state: device fp32[4] = { scale, growth_tracker, found_inf of the last call (0/1), AdamW steps taken so far }.
 * If any gradient is inf / nan the parameters, moments and step count stay untouched and scale *= backoff_factor;
 * otherwise gradients are unscaled in place, clipped, applied (bias correction from the DEVICE step count), and after
 * growth_interval clean steps in a row scale *= growth_factor  (torch.cuda.amp.GradScaler semantics). */
int ge2e_clip_adamw_step_scaled(ge2e_handle h, void* stream, int count, float* const* params, float* const* grads,
                                float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                                float* norm_scratch, float max_norm, float lr, float beta1, float beta2, float eps,
                                float weight_decay, float* scaler_state, float growth_factor, float backoff_factor,
                                int growth_interval);

/* wav -> log-mel front-end (reference meldataset.py:73-96 `mel_spectrogram`, used by Inference.py:71-81
 * and Pattern_Generator.py:96-106): reflect-pad (n_fft - hop)/2, STFT with a periodic Hann window of n_fft samples
 * (the reference's Frame_Length == N_FFT, center=False), magnitude sqrt(re^2 + im^2 + 1e-9), mel filterbank,
 * log(clamp(., 1e-5)).
 * wav:        device fp32 [batch, samples]         (zero-padded to a common length, as Pattern_Generator.Audio_Stack does)
 * mel_basis:  device fp32 [n_mels, n_fft/2 + 1]    (librosa.filters.mel in the reference)
 * out_logmel: device fp32 [batch, n_mels, ge2e_mel_frames()]  -- channels-first, the layout ge2e_encoder_forward reads */
int ge2e_mel_frames(int samples, int n_fft, int hop);                       /* negative: unsupported geometry */
size_t ge2e_mel_workspace_bytes(int batch, int samples, int n_fft, int hop, int n_mels);   /* 0: unsupported */
int ge2e_mel_spectrogram(ge2e_handle h, void* stream, const float* wav, int batch, int samples, int n_fft, int hop, int n_mels,
                         const float* mel_basis, float* out_logmel, void* workspace, size_t workspace_bytes);

/* Live per-kernel timing for the roofline leg of bench.py.  While a class bit is enabled every launch of that
 * kernel class is bracketed by hipEvents ON THE LAUNCH STREAM; ge2e_profile_read() synchronises those events,
 * returns the summed duration, the summed algorithmic FLOPs, the summed algorithmic HBM bytes (operands in + tile
 * out, each counted once per launch) and the launch count, and resets the class.  Disabled (mask 0) there is no overhead. */
enum {
    GE2E_K_GEMM = 1,       /* 128x128-tile projection GEMMs (prenet, in_proj, FFN1, dgrads)   work = 2*M*N*K   */
    GE2E_K_GEMM_LN = 2,    /* 128x256-tile GEMMs with the residual+LayerNorm epilogue         work = 2*M*N*K   */
    GE2E_K_WGRAD = 4,      /* weight-gradient GEMMs                                            work = 2*R*N*K   */
    GE2E_K_ATTN_FWD = 8,   /* fused attention forward                                          work = 4*T*T*64 per head */
    GE2E_K_ATTN_BWD = 16,  /* fused attention backward                                         work = 14*T*T*64 per head */
    GE2E_K_LN_BWD = 32,    /* LayerNorm backward                                               work = bytes moved */
    GE2E_K_FFN = 64,       /* chained FFN1 -> ReLU -> FFN2 -> LayerNorm (16-bit modes)         work = 4*M*256*1024 */
    GE2E_K_SERIAL = 1 << 30  /* not a class: while set, a backward keeps its weight-gradient kernels on the caller's stream, so the
                              * timings of every class are those of kernels running ALONE (no overlap with the other stream)   */
};
int ge2e_profile_enable(ge2e_handle h, int class_mask);
int ge2e_profile_read(ge2e_handle h, int klass, double* total_ms, double* total_work, double* total_bytes,
                      int64_t* launches);

/* Diagnostics: byte offset/size inside the workspace of a named intermediate of the last forward
 * ("h0", "qkv.<l>", "o.<l>", "h1.<l>", "f.<l>", "h2.<l>", and after a backward the scratch of FULL layer l: "dF.<l>",
 * "dP1.<l>", "dM1.<l>" (norm2 backward), "dP.<l>", "dM.<l>" (norm1 backward), "dQKV.<l>" -- layers alternate between buffer
 * sets, so a layer's scratch survives the next layer's backward -- and of the last processed layer "dHb", "dO", "dHa";
 * element type follows cfg.precision -- except "rstd1.<l>", "rstd2.<l>" and "lse.<l>", which are fp32; "xt" = the packed mel rows).
 * The LAST layer is evaluated for frame 0 only (nothing else of it is consumed, Modules.py:54), so its
 * "o", "h1", "f", "h2" taps are compact [n_utts, width] and its "qkv" holds q in frame-0 rows only.
 * Used by the parity tests to localise a failing kernel; returns GE2E_EINVAL for unknown names. */
int ge2e_debug_tap(ge2e_handle h, const char* name, int n_utts, int frames, int train,
                   size_t* offset_bytes, size_t* size_bytes);

/* Development / diagnostic options.  The library never reads the process environment; the defaults are the shipped configuration.
 * An option is process-wide and takes effect at the next call that depends on it ("no_overlap": at the next ge2e_create); a call reads
 * its options once, at entry, and a backward takes what its forward left in the workspace (the FFN mask as bits or not) from that
 * forward's own record, so an option may change between a forward and its backward.  Names
 * (ge2e_option_name(0 .. ) enumerates them, NULL past the end): kernel-selection ablations "no_overlap", "no_ws_gemm", "no_kl_gemm",
 * "no_lnfuse", "no_sk_gemm", "no_ffn_chain", "ffn_wv" (4 | 8), "no_ffn_chain_bwd", "no_wgrad_ks", "no_reduce_batch",
 * "wgrad_ks_blocks" (n), "no_event_bind", "no_maskbits", "no_colsum_end", "no_prenet_fuse", "no_last_chain", and the opt-in "attn_sub"
 * (1: the attention sub-layer of a full layer as one launch per utterance; measured slower than its three launches, default 0); test hooks
 * "debug_bwd_stop" (k >= 0: a backward returns after k layers so that ge2e_debug_tap shows that layer's scratch; -1 off) and
 * "debug_side_delay_us" (hold the weight-gradient stream back after every fork).  Unknown name: GE2E_EINVAL.
 * (The ctypes loader forwards GE2E_<NAME>=value environment variables to this call only when GE2E_DEV_SWITCHES=1 is set.) */
int ge2e_set_option(const char* name, int value);
int ge2e_get_option(const char* name, int* value);
const char* ge2e_option_name(int index);

/* Host-side helpers shared with the CPU oracle (dropout stream definition; no GPU needed). */
uint32_t ge2e_drop_key(uint64_t seed, uint64_t step, int site);
int ge2e_drop_keep(uint32_t key, uint32_t index, float p);

#ifdef __cplusplus
}
#endif
#endif /* GE2E_HIP_H */
