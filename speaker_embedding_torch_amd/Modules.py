"""`Modules.py`-shaped host side of the MI355X GE2E hot path.

Mirrors the reference's module API for this path (reference Modules.py):
  * `GE2E(hyper_parameters)`, `.forward(features[N, Mel, T], samples=1) -> [N // samples, Emb]`   (Modules.py:5-59)
  * `GE2E_Loss(init_weight=10.0, init_bias=-5.0)`, `.forward(embeddings, pattern_per_speaker)`       (Modules.py:112-156)
with the SAME state_dict keys (43 parameters + the `positional_encoding.pe` buffer), so reference
checkpoints (`state['Model']`, Train.py:285) load strictly.  All arithmetic runs in libge2e_hip.so
(hand-written gfx950 kernels) through `torch.autograd.Function`s; PyTorch only owns the memory and the
stream.  There is no CPU or eager fallback: a non-GPU tensor or a missing library raises.
"""
from argparse import Namespace
import math
import weakref

import torch

from . import _lib

__all__ = ["GE2E", "GE2E_Loss", "Conv1d", "Positional_Encoding"]


# ------------------------------------------------------------------------------------------------
# parameter containers (no compute): names/shapes/initialisation of the reference modules
# ------------------------------------------------------------------------------------------------
class Conv1d(torch.nn.Module):
    """k=1 Conv1d parameters with the reference initialisation (Modules.py:61-72)."""

    def __init__(self, in_channels, out_channels, w_init_gain="relu"):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(out_channels, in_channels, 1))
        self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        if w_init_gain in ("relu", "leaky_relu"):
            torch.nn.init.kaiming_uniform_(self.weight, nonlinearity=w_init_gain)
        else:
            torch.nn.init.xavier_uniform_(self.weight, gain=torch.nn.init.calculate_gain(w_init_gain))


class Positional_Encoding(torch.nn.Module):
    """`pe` buffer [1, D, max_position] and scalar `alpha` exactly as Modules.py:76-96."""

    def __init__(self, max_position, embedding_size, dropout_rate):
        super().__init__()
        self.dropout_rate = dropout_rate
        pe = torch.zeros(max_position, embedding_size)
        position = torch.arange(0, max_position, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, embedding_size, 2).float() * (-math.log(10000.0) / embedding_size))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0).transpose(2, 1).contiguous())
        self.alpha = torch.nn.Parameter(torch.ones(1))


class _Linear(torch.nn.Module):
    def __init__(self, fan_in, fan_out, zero_bias=False):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(fan_out, fan_in))
        self.bias = torch.nn.Parameter(torch.empty(fan_out))
        torch.nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))      # torch.nn.Linear default
        bound = 1.0 / math.sqrt(fan_in)
        if zero_bias:
            torch.nn.init.zeros_(self.bias)
        else:
            torch.nn.init.uniform_(self.bias, -bound, bound)


class _LayerNorm(torch.nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(d))
        self.bias = torch.nn.Parameter(torch.zeros(d))


class _SelfAttention(torch.nn.Module):
    """torch.nn.MultiheadAttention parameter set (packed in_proj, out_proj with zero bias)."""

    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = torch.nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = torch.nn.Parameter(torch.zeros(3 * d))
        torch.nn.init.xavier_uniform_(self.in_proj_weight)
        self.out_proj = _Linear(d, d, zero_bias=True)


class _EncoderLayer(torch.nn.Module):
    def __init__(self, d, ffn):
        super().__init__()
        self.self_attn = _SelfAttention(d)
        self.linear1 = _Linear(d, ffn)
        self.linear2 = _Linear(ffn, d)
        self.norm1 = _LayerNorm(d)
        self.norm2 = _LayerNorm(d)


class _Transformer(torch.nn.Module):
    def __init__(self, d, ffn, layers):
        super().__init__()
        first = _EncoderLayer(d, ffn)
        self.layers = torch.nn.ModuleList([first] + [_EncoderLayer(d, ffn) for _ in range(layers - 1)])
        # torch.nn.TransformerEncoder deep-copies ONE initialised layer (Modules.py:25-36): identical at step 0
        for layer in self.layers[1:]:
            layer.load_state_dict(first.state_dict())
        self.norm = _LayerNorm(d)


# ------------------------------------------------------------------------------------------------
# autograd bridge
# ------------------------------------------------------------------------------------------------
def _require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live on the MI355X (got device {t.device}): the GE2E hot path has no CPU fallback")


class _Token:
    """Lifetime marker of one pending backward (dies with the autograd graph)."""
    done = False


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, features, samples, *params):
        _require_gpu(features, "features")
        if features.dim() != 3 or features.dtype not in (torch.float32, torch.float16):
            raise RuntimeError("features must be float32 (or float16: SURVEY row f2) [Batch * Sample, Mel_dim, Time]")
        features = features.contiguous()
        n, mel, t = features.shape
        hnd = module._handle()
        if mel != hnd.cfg.mel_dim:
            raise RuntimeError(f"Mel_dim {mel} != configured {hnd.cfg.mel_dim}")
        if n % samples != 0:
            raise RuntimeError("batch is not a multiple of samples")
        train = bool(module.training)
        plist = [p.detach() for p in params]
        for p in plist:
            _require_gpu(p, "parameter")
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("parameters must be contiguous float32")
        ptrs = hnd.ptr_table(plist)
        token = _Token() if (train and torch.is_grad_enabled()) else None
        ws = module._workspace(n, t, train, token)
        out = torch.empty(n // samples, hnd.cfg.emb, device=features.device, dtype=torch.float32)
        seed, step = module.seed, module._step
        with torch.cuda.device(features.device):      # the library launches on (and creates its side stream for) the CURRENT device
            stream = torch.cuda.current_stream(features.device).cuda_stream
            # eval: the 16-bit weight copies in the workspace stay valid while nothing they depend on changed (an inference loop
            # over one checkpoint): same workspace, shape, stream and parameter storage / version counters
            prepared = False
            if not train and not module._poison:
                pe = module.positional_encoding.pe
                key = (ws.data_ptr(), n, t, stream, hnd, pe.data_ptr(), pe._version) + tuple((p.data_ptr(), p._version) for p in plist)
                prepared = module._prepared_key == key
                module._prepared_key = key
            hnd.encoder_forward(stream, features, n, t, samples, ptrs, module.positional_encoding.pe, out, ws, train, seed, step,
                                prepared=prepared)
        if train:
            module._step += 1
            module._prepared_key = None     # a training step follows: whatever updates the weights may not bump their version counters
        ctx.module, ctx.ws, ctx.train, ctx.token = module, ws, train, token
        ctx.dims = (n, t, samples, seed, step)
        ctx.save_for_backward(features, *params)
        return out

    @staticmethod
    def backward(ctx, d_emb):
        module = ctx.module
        if not ctx.train:
            raise RuntimeError("GE2E backward needs a forward in train() mode (eval keeps no activations)")
        features, *params = ctx.saved_tensors
        n, t, samples, seed, step = ctx.dims
        hnd = module._handle()
        plist = [p.detach() for p in params]
        ptrs = hnd.ptr_table(plist)
        d_emb = d_emb.contiguous().float()
        grads = torch.empty(hnd.param_total, device=features.device, dtype=torch.float32)
        with torch.cuda.device(features.device):      # backward runs on the autograd thread: make the tensors' device current there too
            stream = torch.cuda.current_stream(features.device).cuda_stream
            sync = module._grad_sync
            cb = sync.bucket_callback(grads, hnd, stream) if sync is not None else None
            hnd.encoder_backward(stream, features, n, t, samples, ptrs, d_emb, grads, ctx.ws, seed, step, cb)
            if sync is not None:
                sync.finish(grads)
        if ctx.token is not None:
            ctx.token.done = True
        out = [grads[o:o + k].view_as(p) for o, k, p in zip(hnd.param_offset, hnd.param_numel, params)]
        return (None, None, None, *out)


def default_precision(hp):
    """Arithmetic mode a hyper-parameter file asks for (reference Train.py:134,145; see GE2E.__doc__)."""
    if not getattr(hp, "Use_Mixed_Precision", False):
        return "fp32"
    return str(getattr(hp, "Mixed_Precision_Dtype", "fp16")).lower().replace("float16", "fp16").replace("bfloat16", "bf16")


class GE2E(torch.nn.Module):
    """Speaker encoder of reference Modules.py:5-59 on the HIP path.

    `precision`: 'fp32' (fp32 MFMA; d-vectors within 1e-4 of the reference CPU path), 'fp32x3' (fp32 storage, the projection
    products as three bf16 MFMAs on split operands: still within 1e-4, a multiple of the fp32 speed), 'bf16' or 'fp16' (16-bit
    storage / fp32 accumulate; fp16 is the reference's own mixed precision and wants the GradScaler of Optim.py).
    Default follows `hp.Use_Mixed_Precision` like Train.py:134,145: false -> 'fp32'; true -> the optional key
    `hp.Mixed_Precision_Dtype` ('bf16' | 'fp16'), 'fp16' when the key is absent (what autocast means in the reference).
    """

    def __init__(self, hyper_parameters: Namespace, precision=None, seed=0):
        super().__init__()
        self.hp = hyper_parameters
        d = self.hp.GE2E.Embedding_Size
        self.prenet = Conv1d(self.hp.Sound.Mel_Dim, d, w_init_gain="relu")
        self.positional_encoding = Positional_Encoding(
            max_position=self.hp.GE2E.Positional_Encoding.Max_Position, embedding_size=d,
            dropout_rate=self.hp.GE2E.Positional_Encoding.Dropout_Rate)
        self.transformer = _Transformer(d, d * 4, self.hp.GE2E.Transformer.Num_Layers)
        self.projection = Conv1d(d, d, w_init_gain="linear")
        if precision is None:
            precision = default_precision(self.hp)
        if precision not in _lib.PRECISIONS:
            raise ValueError("precision must be 'fp32', 'fp32x3', 'bf16' or 'fp16'")
        self.precision = precision
        self.seed = int(seed)
        self._step = 0
        self._handles = {}
        self._ws = {}
        self._prepared_key = None       # what the eval workspace's weight copies were prepared from (_EncoderFn.forward)
        self._ws_owner = {}
        self._grad_sync = None          # set by distributed.apply_gradient_allreduce
        self._poison = False            # tests: fill every handed-out workspace with NaN bit patterns first

    # -- plumbing ------------------------------------------------------------------------------
    def _handle(self):
        # one handle per (arithmetic mode, device): a handle's side stream and fence events belong to one device
        dev = self.prenet.weight.device
        hkey = (self.precision, dev.index if dev.type == "cuda" else -1)
        h = self._handles.get(hkey)
        if h is None:
            g = self.hp.GE2E
            h = _lib.Handle(
                mel_dim=self.hp.Sound.Mel_Dim, emb=g.Embedding_Size, heads=g.Transformer.Head,
                layers=g.Transformer.Num_Layers, ffn=g.Embedding_Size * 4,
                max_position=g.Positional_Encoding.Max_Position,
                pe_dropout=g.Positional_Encoding.Dropout_Rate, tf_dropout=g.Transformer.Dropout_Rate,
                ln_eps=1e-5, precision=_lib.PRECISIONS[self.precision])
            names = [n for n, _ in self.named_parameters()]
            if names != h.param_names:
                raise RuntimeError("parameter table of libge2e_hip.so differs from the module's state_dict keys")
            self._handles[hkey] = h
        return h

    def _workspace(self, n, t, train, token=None):
        """Caller-owned scratch for libge2e_hip.so.  A train-mode workspace holds the activations of a pending
        backward: it is handed out again only after that backward ran (or its graph was dropped)."""
        need = self._handle().workspace_bytes(n, t, train)
        if need == 0:
            raise RuntimeError("ge2e_workspace_bytes returned 0 (bad shape)")
        key = (bool(train), self.precision)
        dev = self.prenet.weight.device
        ws = self._ws.get(key)
        owner = self._ws_owner.get(key)
        busy = owner is not None and owner() is not None and not owner().done
        if ws is None or ws.numel() < need or ws.device != dev or busy:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            self._ws[key] = ws
        self._ws_owner[key] = weakref.ref(token) if token is not None else None
        if self._poison:                # any read of a never-written workspace byte then shows up as NaN
            ws.fill_(0xFF)
        return ws

    def workspace_view(self, name, n, t, train=True, dtype=None):
        """Diagnostics for the parity tests: a typed view of a named intermediate of the last forward
        (`dtype=torch.float32` for the fp32 side tables "rstd1.<l>", "rstd2.<l>", "lse.<l>")."""
        off, size = self._handle().debug_tap(name, n, t, train)
        ws = self._ws[(bool(train), self.precision)]
        dt = dtype or {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32, "fp32x3": torch.float32}[self.precision]
        return ws[off:off + size].view(dt)

    # -- reference API -------------------------------------------------------------------------
    def forward(self, features, samples=1):
        """features: [Batch * Sample, Mel_dim, Time] -> unit-norm embeddings [Batch, Emb] (Modules.py:46-59)."""
        return _EncoderFn.apply(self, features, int(samples), *self.parameters())


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, embeddings, pattern_per_speaker, weight, bias):
        _require_gpu(embeddings, "embeddings")
        emb = embeddings.contiguous().float()
        n, d = emb.shape
        if n % pattern_per_speaker != 0:
            raise RuntimeError("batch is not a multiple of pattern_per_speaker")
        hnd = module._handle(d, emb.device)
        speakers = n // pattern_per_speaker
        ws = torch.empty(hnd.loss_workspace_bytes(speakers, pattern_per_speaker), dtype=torch.uint8, device=emb.device)
        loss = torch.empty(1, device=emb.device, dtype=torch.float32)
        w, b = module._scalars()
        with torch.cuda.device(emb.device):
            stream = torch.cuda.current_stream(emb.device).cuda_stream
            hnd.loss_forward(stream, emb, speakers, pattern_per_speaker, w, b, loss, ws)
        ctx.module, ctx.ws, ctx.dims = module, ws, (speakers, pattern_per_speaker, w, b)
        ctx.save_for_backward(emb)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, d_loss):
        (emb,) = ctx.saved_tensors
        speakers, utts, w, b = ctx.dims
        hnd = ctx.module._handle(emb.shape[1], emb.device)
        d_emb = torch.empty_like(emb)
        # the criterion's own weight / bias (reference Modules.py:115-116): autograd fills their .grad there, so it does here
        want_wb = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        d_wb = torch.empty(2, device=emb.device, dtype=torch.float32) if want_wb else None
        with torch.cuda.device(emb.device):
            stream = torch.cuda.current_stream(emb.device).cuda_stream
            hnd.loss_backward(stream, emb, speakers, utts, w, b, d_loss.reshape(1).contiguous().float(), d_emb, ctx.ws, d_wb)
        return (None, d_emb, None, d_wb[0].reshape(()) if ctx.needs_input_grad[3] else None,
                d_wb[1].reshape(()) if ctx.needs_input_grad[4] else None)


class GE2E_Loss(torch.nn.Module):
    """GE2E softmax loss with self-inclusive centroids and logits w*cos - b (Modules.py:112-156).

    `weight`/`bias` are nn.Parameters as in the reference: never optimised, all-reduced or checkpointed there
    (Train.py:121-127,298-303), but autograd fills their `.grad`, and so does the HIP backward (two scalar sums that ride in the
    row kernel); `requires_grad_(False)` on them skips that.
    """

    def __init__(self, init_weight=10.0, init_bias=-5.0):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.tensor(float(init_weight)))
        self.bias = torch.nn.Parameter(torch.tensor(float(init_bias)))
        self._cache = None
        self._hnd = None
        self._hnd_key = None

    def _scalars(self):
        ver = (self.weight._version, self.bias._version, self.weight.data_ptr(), self.bias.data_ptr())
        if self._cache is None or self._cache[0] != ver:
            self._cache = (ver, float(self.weight.detach()), float(self.bias.detach()))   # host read only when changed
        return self._cache[1], self._cache[2]

    def _handle(self, emb, device):
        # one handle per device: a handle's streams / events belong to the device of its first call
        key = (emb, device.index if device.type == "cuda" else -1)
        if self._hnd is None or self._hnd_key != key:
            self._hnd, self._hnd_key = _lib.Handle(emb=emb, heads=emb // 64), key
        return self._hnd

    def forward(self, embeddings, pattern_per_speaker):
        """embeddings: [Batch, Emb_dim], speaker-major (Datasets.Collater order); returns a 0-d loss."""
        return _LossFn.apply(self, embeddings, int(pattern_per_speaker), self.weight, self.bias)


class _GatherBatchFn(torch.autograd.Function):
    """all-gather of the per-rank d-vectors (rank-major = speaker-major, every rank holds whole speakers).  Every rank then
    computes the SAME global loss, so dL/d(e_local) needs no second collective: it is the local slice of the full
    gradient, times `world` because the parameter gradients are averaged (not summed) across ranks afterwards."""

    @staticmethod
    def forward(ctx, emb, group):
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        emb = emb.contiguous()
        parts = [torch.empty_like(emb) for _ in range(world)]
        dist.all_gather(parts, emb, group=group)
        ctx.n, ctx.rank, ctx.world = emb.shape[0], rank, world
        return torch.cat(parts, dim=0)

    @staticmethod
    def backward(ctx, d_full):
        return d_full[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n] * float(ctx.world), None


class GE2E_Loss_Global(GE2E_Loss):
    """SURVEY row f1 (opt-in, NOT the reference's semantics): one GE2E loss over the speakers of ALL ranks.  The
    reference gives every rank its own speakers and a local loss (Train.py:90-99), so the softmax is only as wide as a
    rank's batch; here the [N_local, Emb] d-vectors are all-gathered (world x 245 KB at 64 x 15) and the similarity
    matrix spans world x Speaker centroids -- e.g. 8 ranks x 8 speakers keep the 64-wide softmax of the single-GPU
    recipe while each rank encodes 1/8 of the utterances (strong scaling).  With the mean gradient all-reduce that
    follows (distributed.GradSync) the parameter gradient equals the one a single process would compute on the
    concatenated batch.  Returns the global loss (identical on every rank)."""

    def __init__(self, init_weight=10.0, init_bias=-5.0, group=None):
        super().__init__(init_weight, init_bias)
        self.group = group

    def forward(self, embeddings, pattern_per_speaker):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            embeddings = _GatherBatchFn.apply(embeddings, self.group)
        return super().forward(embeddings, pattern_per_speaker)
