"""ctypes binding of libge2e_hip.so (include/ge2e_hip.h).  There is no fallback: if the library is
missing, or a call fails, a RuntimeError is raised."""
import ctypes as C
import os
import threading

from . import _build
from ._build import LIB

_lock = threading.Lock()
_lib = None

BUCKET_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int64)

PREC_F32, PREC_BF16, PREC_F16, PREC_F32X3 = 0, 1, 2, 3
PRECISIONS = {"fp32": PREC_F32, "bf16": PREC_BF16, "fp16": PREC_F16, "fp32x3": PREC_F32X3}
ABI_VERSION = 4
K_GEMM, K_GEMM_LN, K_WGRAD, K_ATTN_FWD, K_ATTN_BWD, K_LN_BWD, K_FFN = 1, 2, 4, 8, 16, 32, 64
FWD_PREPARED = 2       # ge2e_encoder_forward's `train` argument: eval forward, weight copies already in the workspace
K_SERIAL = 1 << 30      # with a class bit: the backward keeps its weight gradients on the caller's stream (kernels timed alone)


class Config(C.Structure):
    _fields_ = [("mel_dim", C.c_int32), ("emb", C.c_int32), ("heads", C.c_int32), ("layers", C.c_int32),
                ("ffn", C.c_int32), ("max_position", C.c_int32), ("pe_dropout", C.c_float),
                ("tf_dropout", C.c_float), ("ln_eps", C.c_float), ("precision", C.c_int32)]


# every symbol include/ge2e_hip.h declares: name -> (restype, argtypes)
_PF = C.POINTER(C.c_float)
_SIG = {
    "ge2e_abi_version": (C.c_int, []),
    "ge2e_source_hash": (C.c_char_p, []),
    "ge2e_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "ge2e_destroy": (C.c_int, [C.c_void_p]),
    "ge2e_last_error": (C.c_char_p, [C.c_void_p]),
    "ge2e_param_count": (C.c_int, [C.c_void_p]),
    "ge2e_param_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "ge2e_param_numel": (C.c_int64, [C.c_void_p, C.c_int]),
    "ge2e_param_offset": (C.c_int64, [C.c_void_p, C.c_int]),
    "ge2e_param_total": (C.c_int64, [C.c_void_p]),
    "ge2e_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "ge2e_max_frames": (C.c_int, [C.c_void_p]),
    "ge2e_encoder_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                       C.c_int, C.c_uint64, C.c_uint64]),
    "ge2e_encoder_forward_mel16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                             C.c_int, C.c_uint64, C.c_uint64]),
    "ge2e_encoder_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.c_uint64, C.c_uint64]),
    "ge2e_encoder_backward_cb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_uint64, C.c_uint64, BUCKET_CB, C.c_void_p]),
    "ge2e_bucket_stream": (C.c_void_p, [C.c_void_p, C.c_void_p]),
    "ge2e_loss_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ge2e_loss_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                    C.c_void_p, C.c_void_p, C.c_size_t]),
    "ge2e_loss_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ge2e_clip_adamw_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p,
                                       C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64]),
    "ge2e_clip_adamw_step_scaled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p,
                                              C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                              C.c_float, C.c_float, C.c_int]),
    "ge2e_mel_frames": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "ge2e_mel_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ge2e_mel_spectrogram": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ge2e_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "ge2e_profile_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_int64)]),
    "ge2e_debug_tap": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ge2e_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "ge2e_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "ge2e_option_name": (C.c_char_p, [C.c_int]),
    "ge2e_drop_key": (C.c_uint32, [C.c_uint64, C.c_uint64, C.c_int]),
    "ge2e_drop_keep": (C.c_int, [C.c_uint32, C.c_uint32, C.c_float]),
}
SYMBOLS = tuple(_SIG)


def option_names(lib=None):
    lib = lib or load()
    out, i = [], 0
    while True:
        n = lib.ge2e_option_name(i)
        if n is None:
            return out
        out.append(n.decode())
        i += 1


def set_option(name, value):
    """Development / diagnostic options of the library (include/ge2e_hip.h, ge2e_set_option); process-wide."""
    if load().ge2e_set_option(name.encode(), int(value)) != 0:
        raise KeyError(name)


def get_option(name):
    v = C.c_int()
    if load().ge2e_get_option(name.encode(), C.byref(v)) != 0:
        raise KeyError(name)
    return v.value


def _dev_switches(lib):
    """GE2E_DEV_SWITCHES=1 (tools/ab.sh, tools/switch_test.sh, the scheduling tests): GE2E_<NAME>=value -> ge2e_set_option(name, value).
    Without it the environment has no influence on the library: the product's kernel selection is the compiled-in default."""
    for name in option_names(lib):
        v = os.environ.get("GE2E_" + name.upper())
        if v is not None:
            try:
                lib.ge2e_set_option(name.encode(), int(v))
            except ValueError:
                raise RuntimeError(f"GE2E_{name.upper()}={v!r}: integer expected")


def load():
    """dlopen the in-tree library (built by __graft_entry__.build / _build.build)."""
    global _lib
    with _lock:
        if _lib is None:
            dev = os.environ.get("GE2E_DEV_SWITCHES") == "1"
            # a same-box A/B of two BUILDS (tools/ab_lib.sh) loads its variants from tools/abl/ and leaves the in-tree library alone
            override = os.environ.get("GE2E_LIB_OVERRIDE") if dev else None
            if override:
                lib = C.CDLL(override)
                for name, (res, args) in _SIG.items():
                    fn = getattr(lib, name)
                    fn.restype, fn.argtypes = res, args
                import sys
                print(f"[ge2e] development build loaded from {override} (hash {lib.ge2e_source_hash().decode()})", file=sys.stderr)
                _dev_switches(lib)
                _lib = lib
                return _lib
            if not os.path.exists(LIB):
                raise RuntimeError(
                    f"{LIB} is missing: the GE2E HIP extension has not been built "
                    "(run `python -m speaker_embedding_torch_amd._build`); there is no CPU fallback")
            lib = C.CDLL(LIB)
            for name, (res, args) in _SIG.items():
                fn = getattr(lib, name)        # AttributeError if the ABI lost a symbol
                fn.restype, fn.argtypes = res, args
            if lib.ge2e_abi_version() != ABI_VERSION:
                raise RuntimeError("libge2e_hip.so ABI version mismatch")
            built, want = lib.ge2e_source_hash().decode(), _build.source_hash()
            if built != want:
                raise RuntimeError(
                    f"{LIB} is stale: it was compiled from sources with hash {built}, csrc/ now hashes to {want} "
                    "(run `python -m speaker_embedding_torch_amd._build`)")
            if dev:
                _dev_switches(lib)
            _lib = lib
    return _lib


class Handle:
    """Owns one ge2e_handle; raises RuntimeError(ge2e_last_error) on any non-zero return."""

    def __init__(self, mel_dim=80, emb=256, heads=4, layers=3, ffn=1024, max_position=1024,
                 pe_dropout=0.1, tf_dropout=0.1, ln_eps=1e-5, precision=PREC_F32):
        self.lib = load()
        self.cfg = Config(mel_dim, emb, heads, layers, ffn, max_position, pe_dropout, tf_dropout, ln_eps, precision)
        self._h = C.c_void_p()
        rc = self.lib.ge2e_create(C.byref(self.cfg), C.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"ge2e_create failed ({rc}): unsupported configuration -- the kernels are laid out for "
                               f"GE2E.Embedding_Size 256 with 64-wide heads (GE2E.Transformer.Head = 4, the reference's shipped "
                               f"Hyper_Parameters.yaml:11,17), 1..8 layers, Mel_Dim <= 128, ffn a multiple of 128; got emb={emb}, "
                               f"heads={heads}, layers={layers}, mel_dim={mel_dim}, ffn={ffn}")
        n = self.lib.ge2e_param_count(self._h)
        self.param_names = [self.lib.ge2e_param_name(self._h, i).decode() for i in range(n)]
        self.param_numel = [self.lib.ge2e_param_numel(self._h, i) for i in range(n)]
        self.param_offset = [self.lib.ge2e_param_offset(self._h, i) for i in range(n)]
        self.param_total = self.lib.ge2e_param_total(self._h)

    def __del__(self):
        try:
            if self._h:
                self.lib.ge2e_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def check(self, rc, what):
        if rc != 0:
            msg = self.lib.ge2e_last_error(self._h)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def workspace_bytes(self, n, t, train):
        return self.lib.ge2e_workspace_bytes(self._h, n, t, 1 if train else 0)

    def max_frames(self):
        return self.lib.ge2e_max_frames(self._h)

    def ptr_table(self, tensors):
        arr = (C.c_void_p * len(tensors))()
        for i, t in enumerate(tensors):
            arr[i] = t.data_ptr()
        return arr

    def encoder_forward(self, stream, mel, n, t, samples, ptrs, pe, out, ws, train, seed, step, prepared=False):
        """mel: device float32 or float16 [n, mel_dim, t] (the fp16 form is widened by the packing kernel).
        prepared (eval only): the workspace still holds the weight copies of an earlier eval forward (GE2E_FWD_PREPARED)."""
        fn = self.lib.ge2e_encoder_forward_mel16 if str(mel.dtype) == "torch.float16" else self.lib.ge2e_encoder_forward
        self.check(fn(self._h, stream, mel.data_ptr(), n, t, samples, ptrs, pe.data_ptr(),
                      out.data_ptr(), ws.data_ptr(), ws.numel() * ws.element_size(),
                      1 if train else (FWD_PREPARED if prepared else 0), seed, step), "ge2e_encoder_forward")

    def encoder_backward(self, stream, mel, n, t, samples, ptrs, d_emb, grads, ws, seed, step, cb=None):
        if cb is None:
            rc = self.lib.ge2e_encoder_backward(self._h, stream, mel.data_ptr(), n, t, samples, ptrs, d_emb.data_ptr(),
                                                grads.data_ptr(), ws.data_ptr(), ws.numel() * ws.element_size(), seed, step)
        else:
            rc = self.lib.ge2e_encoder_backward_cb(self._h, stream, mel.data_ptr(), n, t, samples, ptrs, d_emb.data_ptr(),
                                                   grads.data_ptr(), ws.data_ptr(), ws.numel() * ws.element_size(),
                                                   seed, step, cb, None)
        self.check(rc, "ge2e_encoder_backward")

    def bucket_stream(self, stream):
        """Raw hipStream_t behind which the buckets of encoder_backward(…, cb) are final (valid inside the callback)."""
        return self.lib.ge2e_bucket_stream(self._h, stream) or 0

    def loss_workspace_bytes(self, speakers, utts):
        return self.lib.ge2e_loss_workspace_bytes(speakers, utts, self.cfg.emb)

    def loss_forward(self, stream, emb, speakers, utts, w, b, loss, ws):
        self.check(self.lib.ge2e_loss_forward(self._h, stream, emb.data_ptr(), speakers, utts, w, b, loss.data_ptr(),
                                              ws.data_ptr(), ws.numel() * ws.element_size()), "ge2e_loss_forward")

    def loss_backward(self, stream, emb, speakers, utts, w, b, d_loss, d_emb, ws, d_wb=None):
        self.check(self.lib.ge2e_loss_backward(self._h, stream, emb.data_ptr(), speakers, utts, w, b, d_loss.data_ptr(),
                                               d_emb.data_ptr(), d_wb.data_ptr() if d_wb is not None else None,
                                               ws.data_ptr(), ws.numel() * ws.element_size()),
                   "ge2e_loss_backward")

    def clip_adamw_step(self, stream, ptrs_p, ptrs_g, ptrs_m, ptrs_v, numel, norm, max_norm, lr, b1, b2, eps, wd, step):
        self.check(self.lib.ge2e_clip_adamw_step(self._h, stream, len(ptrs_p), ptrs_p, ptrs_g, ptrs_m, ptrs_v, numel,
                                                 norm.data_ptr(), max_norm, lr, b1, b2, eps, wd, step), "ge2e_clip_adamw_step")

    def clip_adamw_step_scaled(self, stream, ptrs_p, ptrs_g, ptrs_m, ptrs_v, numel, norm, max_norm, lr, b1, b2, eps, wd,
                               scaler_state, growth, backoff, interval):
        self.check(self.lib.ge2e_clip_adamw_step_scaled(self._h, stream, len(ptrs_p), ptrs_p, ptrs_g, ptrs_m, ptrs_v, numel,
                                                        norm.data_ptr(), max_norm, lr, b1, b2, eps, wd, scaler_state.data_ptr(),
                                                        growth, backoff, interval), "ge2e_clip_adamw_step_scaled")

    def mel_frames(self, samples, n_fft, hop):
        return self.lib.ge2e_mel_frames(samples, n_fft, hop)

    def mel_workspace_bytes(self, batch, samples, n_fft, hop, n_mels):
        return self.lib.ge2e_mel_workspace_bytes(batch, samples, n_fft, hop, n_mels)

    def mel_spectrogram(self, stream, wav, n_fft, hop, n_mels, basis, out, ws):
        b, l = wav.shape
        self.check(self.lib.ge2e_mel_spectrogram(self._h, stream, wav.data_ptr(), b, l, n_fft, hop, n_mels, basis.data_ptr(),
                                                 out.data_ptr(), ws.data_ptr(), ws.numel() * ws.element_size()), "ge2e_mel_spectrogram")

    def profile_enable(self, mask):
        self.check(self.lib.ge2e_profile_enable(self._h, mask), "ge2e_profile_enable")

    def profile_read(self, klass):
        ms, work, nbytes, cnt = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        self.check(self.lib.ge2e_profile_read(self._h, klass, C.byref(ms), C.byref(work), C.byref(nbytes), C.byref(cnt)),
                   "ge2e_profile_read")
        return ms.value, work.value, nbytes.value, cnt.value

    def debug_tap(self, name, n, t, train):
        off, size = C.c_size_t(), C.c_size_t()
        rc = self.lib.ge2e_debug_tap(self._h, name.encode(), n, t, 1 if train else 0, C.byref(off), C.byref(size))
        if rc != 0:
            raise KeyError(name)
        return off.value, size.value
