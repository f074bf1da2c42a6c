"""CPU oracle for the GE2E hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a plain-numpy restatement of what the reference computes on its
`Device: '-1'` CPU path for one training step's hot path:

  * encoder forward      reference Modules.py:46-59  (GE2E.forward), prenet :10-17,
                         Positional_Encoding :76-109, torch TransformerEncoder built at :25-36
                         (post-LN, ReLU, eps 1e-5, no mask), projection :38-44, F.normalize :57
  * GE2E loss forward    reference Modules.py:121-156 (GE2E_Loss.forward)
  * backward of both     what torch autograd derives for Train.py:153
  * clip + AdamW         reference Train.py:122-127 (AdamW, default weight_decay 0.01),
                         Train.py:154-162 (clip_grad_norm_ then step)

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; the product path (speaker_embedding_torch_amd/) never does.

Pinning: the reference ships no tests or golden vectors.  The
oracle is therefore pinned by golden vectors produced by importing the reference's
`Modules.py` in the build container (`tests/golden/make_golden.py`, outputs committed
in `tests/golden/*.npz`) and checked in `tests/test_oracle_golden.py`.

Dropout: the reference's 13 dropout sites draw from torch's RNG stream, which cannot
be reproduced outside torch.  The oracle fixes site placement and the inverted-dropout
scaling exactly as the reference, but draws its keep-masks from the counter-based hash
below (shared bit-for-bit with the HIP kernels), so train-mode parity HIP<->oracle is
exact-mask; reference<->oracle parity is pinned at dropout 0 / eval mode only
("dropout masks: parity unpinned").
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

# --------------------------------------------------------------------------------------
# parameter table: checkpoint key set of reference `GE2E.state_dict()`
# --------------------------------------------------------------------------------------

def param_specs(mel=80, d=256, layers=3, ffn=None):
    """Ordered (name, shape) list == reference parameters() order (Modules.py:6-44)."""
    ffn = ffn or 4 * d
    specs = [("prenet.weight", (d, mel, 1)), ("prenet.bias", (d,)),
             ("positional_encoding.alpha", (1,))]
    for l in range(layers):
        p = f"transformer.layers.{l}."
        specs += [
            (p + "self_attn.in_proj_weight", (3 * d, d)),
            (p + "self_attn.in_proj_bias", (3 * d,)),
            (p + "self_attn.out_proj.weight", (d, d)),
            (p + "self_attn.out_proj.bias", (d,)),
            (p + "linear1.weight", (ffn, d)),
            (p + "linear1.bias", (ffn,)),
            (p + "linear2.weight", (d, ffn)),
            (p + "linear2.bias", (d,)),
            (p + "norm1.weight", (d,)),
            (p + "norm1.bias", (d,)),
            (p + "norm2.weight", (d,)),
            (p + "norm2.bias", (d,)),
        ]
    specs += [("transformer.norm.weight", (d,)), ("transformer.norm.bias", (d,)),
              ("projection.weight", (d, d, 1)), ("projection.bias", (d,))]
    return specs


# --------------------------------------------------------------------------------------
# formula-defined tensors (weights / inputs reproducible without any RNG library)
# --------------------------------------------------------------------------------------

_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x):
    """lowbias32 integer finaliser on uint32 arrays (wraps mod 2^32)."""
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def formula_uniform(tag: int, shape, lo: float, hi: float, dtype=np.float32):
    """u[i] = lo + (hi-lo) * (mix32(i*2654435761 + tag*40503 + 12345) >> 8) / 2^24."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64)
    h = _mix32(((i * np.uint64(2654435761)) + np.uint64(tag) * np.uint64(40503) + np.uint64(12345)) & _M32)
    u = (h >> np.uint32(8)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(dtype).reshape(shape)


def formula_normal(tag: int, shape, dtype=np.float32):
    """Box-Muller on two formula_uniform streams (deterministic, library-free)."""
    u1 = formula_uniform(tag * 2 + 1, shape, 1e-7, 1.0, np.float64)
    u2 = formula_uniform(tag * 2 + 2, shape, 0.0, 1.0, np.float64)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)).astype(dtype)


def formula_params(mel=80, d=256, layers=3, ffn=None, dtype=np.float32):
    """Per-tensor distinct, per-layer distinct, non-zero biases, LN gains around 1
    (SURVEY.md 8c golden-vector plan; section 7 'all 3 layers are initialised identical')."""
    out = OrderedDict()
    for tag, (name, shape) in enumerate(param_specs(mel, d, layers, ffn), start=1):
        if name.endswith("alpha"):
            out[name] = np.full(shape, 0.9, dtype)
        elif "norm" in name and name.endswith("weight"):
            out[name] = formula_uniform(tag, shape, 0.7, 1.3, dtype)
        elif name.endswith("bias"):
            out[name] = formula_uniform(tag, shape, -0.1, 0.1, dtype)
        else:
            fan_in = shape[1]
            a = math.sqrt(3.0 / fan_in)
            out[name] = formula_uniform(tag, shape, -a, a, dtype)
    return out


def formula_mel(tag: int, n: int, mel: int, t: int, logmel: bool = False, dtype=np.float32):
    """Synthetic mel batch [N, Mel, T].  logmel=True mimics log(clamp(.,1e-5)) range
    (x = clamp(-5 + 2 z, -11.5129, 2.0))."""
    z = formula_normal(1000 + tag, (n, mel, t), np.float64)
    if logmel:
        z = np.clip(-5.0 + 2.0 * z, -11.5129, 2.0)
    return z.astype(dtype)


# --------------------------------------------------------------------------------------
# counter-based dropout (bit-identical to csrc/common.cuh: drop_key / drop_keep)
# --------------------------------------------------------------------------------------

def drop_key(seed: int, step: int, site: int) -> int:
    """splitmix64-style scramble of (seed, step, site) -> 32-bit site key (host side)."""
    z = (seed * 0x9E3779B97F4A7C15 + step * 0xBF58476D1CE4E5B9 + (site + 1) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 31
    return int(z & 0xFFFFFFFF)


def drop_threshold(p: float) -> int:
    return int(math.floor(p * 65536.0))


def _xs32(x):
    """one xorshift32 step (13, 17, 5)"""
    x = x.astype(np.uint64)
    x ^= (x << np.uint64(13)) & _M32
    x ^= x >> np.uint64(17)
    x ^= (x << np.uint64(5)) & _M32
    return x.astype(np.uint32)


def drop_keep_at(key: int, idx, p: float):
    """keep[idx]: w0 = mix32((idx >> 2) ^ key), w1 = xs32(w0); word = w1 if idx & 2 else w0;
    field = word >> 16 if idx odd else word & 0xFFFF; keep = field >= floor(p * 2^16).
    One multiplicative hash serves an aligned index quad (see csrc/common.cuh)."""
    idx = np.asarray(idx, dtype=np.uint64)
    w0 = _mix32(((idx >> np.uint64(2)) ^ np.uint64(key)) & _M32)
    w = np.where((idx & np.uint64(2)).astype(bool), _xs32(w0), w0)
    field = np.where((idx & np.uint64(1)).astype(bool), w >> np.uint32(16), w & np.uint32(0xFFFF))
    return field >= np.uint32(drop_threshold(p))


def drop_keep(key: int, count: int, p: float, start: int = 0):
    return drop_keep_at(key, np.arange(start, start + count, dtype=np.uint64), p)


def _dropout(x, key, p, train, index=None):
    """inverted dropout as torch.nn.Dropout (scale kept values by 1/(1-p)); element counter = flat C-order
    index unless `index` (same shape as x) is given."""
    if (not train) or p <= 0.0:
        return x, None
    keep = (drop_keep(key, x.size, p).reshape(x.shape) if index is None else drop_keep_at(key, index, p))
    scale = x.dtype.type(1.0 / (1.0 - p))
    return x * keep * scale, keep


def attn_drop_index(n, heads, t):
    """counter of P[n, h, i, j] = ((n*H + h)*T + i) * T4 + j with T4 = T rounded up to a multiple of 4."""
    t4 = (t + 3) // 4 * 4
    rows = np.arange(n * heads * t, dtype=np.uint64).reshape(n, heads, t, 1)
    return rows * np.uint64(t4) + np.arange(t, dtype=np.uint64).reshape(1, 1, 1, t)


SITE_PE = 0


def site_attn(l):   # dropout on attention probabilities (SDPA dropout_p)
    return 1 + 4 * l


def site_sa(l):     # TransformerEncoderLayer.dropout1
    return 2 + 4 * l


def site_ffh(l):    # TransformerEncoderLayer.dropout (after ReLU)
    return 3 + 4 * l


def site_ff(l):     # TransformerEncoderLayer.dropout2
    return 4 + 4 * l


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------

def sinusoid_pe(max_position: int, d: int, dtype=np.float32):
    """pe[p, 2i] = sin(p * w_i), pe[p, 2i+1] = cos(p * w_i), w_i = exp(-2i ln(1e4)/d)
    computed in float32 like the reference (Modules.py:84-89).  Returns [max_position, d]."""
    pos = np.arange(max_position, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, d, 2, dtype=np.float32) * np.float32(-math.log(10000.0) / d))
    pe = np.zeros((max_position, d), np.float32)
    pe[:, 0::2] = np.sin(pos * div)
    pe[:, 1::2] = np.cos(pos * div)
    return pe.astype(dtype)


def _ln_fwd(x, w, b, eps):
    mean = x.mean(-1, keepdims=True)
    var = ((x - mean) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + x.dtype.type(eps))
    xhat = (x - mean) * rstd
    return xhat * w + b, xhat, rstd


def _ln_bwd(dy, xhat, rstd, w):
    dxhat = dy * w
    m1 = dxhat.mean(-1, keepdims=True)
    m2 = (dxhat * xhat).mean(-1, keepdims=True)
    dx = rstd * (dxhat - m1 - xhat * m2)
    red = tuple(range(dy.ndim - 1))
    return dx, (dy * xhat).sum(red), dy.sum(red)


# --------------------------------------------------------------------------------------
# encoder forward / backward
# --------------------------------------------------------------------------------------

def encoder_forward(params, x, samples=1, heads=4, train=False, seed=0, step=0,
                    p_pe=0.1, p_tf=0.1, eps=1e-5, max_position=1024, taps=None, pe=None):
    """GE2E.forward (Modules.py:46-59).  x: [N, Mel, T].  Returns (emb [N/samples, D], cache)."""
    dt = x.dtype
    n, mel, t = x.shape
    wp = params["prenet.weight"][:, :, 0]
    d = wp.shape[0]
    dh = d // heads
    layers = sum(1 for k in params if k.endswith("self_attn.in_proj_weight"))
    c = {"x": x, "samples": samples, "heads": heads, "layers": layers, "train": train,
         "seed": seed, "step": step, "p_pe": p_pe, "p_tf": p_tf}

    # prenet 1x1 conv + ReLU (Modules.py:50-51)
    z0 = np.einsum("nmt,dm->ntd", x, wp) + params["prenet.bias"]
    a0 = np.maximum(z0, 0)
    # pe: the `positional_encoding.pe` buffer is checkpoint DATA (state_dict key); callers may pass it as
    # [max_position, d] so that oracle and device read the same table.  Default: rebuild it (Modules.py:84-90).
    pe = (sinusoid_pe(max_position, d, dt) if pe is None else np.asarray(pe, dt))[:t]   # Modules.py:98-109
    alpha = params["positional_encoding.alpha"][0]
    h = a0 + alpha * pe[None]
    h, keep_pe = _dropout(h, drop_key(seed, step, SITE_PE), p_pe, train)
    c.update(z0=z0, pe=pe, keep_pe=keep_pe)
    if taps is not None:
        taps["prenet_pe"] = h
    c["lay"] = []
    scale = dt.type(1.0 / math.sqrt(dh))
    for l in range(layers):
        p = f"transformer.layers.{l}."
        lc = {"h_in": h}
        qkv = h @ params[p + "self_attn.in_proj_weight"].T + params[p + "self_attn.in_proj_bias"]
        q, k, v = [qkv[..., i * d:(i + 1) * d].reshape(n, t, heads, dh).transpose(0, 2, 1, 3) for i in range(3)]
        s = (q @ k.transpose(0, 1, 3, 2)) * scale                   # [n, H, T, T]
        s = s - s.max(-1, keepdims=True)
        e = np.exp(s)
        prob = e / e.sum(-1, keepdims=True)
        probd, keep_a = _dropout(prob, drop_key(seed, step, site_attn(l)), p_tf, train,
                                 index=attn_drop_index(n, heads, t) if (train and p_tf > 0) else None)
        o = (probd @ v).transpose(0, 2, 1, 3).reshape(n, t, d)
        a = o @ params[p + "self_attn.out_proj.weight"].T + params[p + "self_attn.out_proj.bias"]
        a, keep_sa = _dropout(a, drop_key(seed, step, site_sa(l)), p_tf, train)
        h1, xhat1, rstd1 = _ln_fwd(h + a, params[p + "norm1.weight"], params[p + "norm1.bias"], eps)
        f_pre = h1 @ params[p + "linear1.weight"].T + params[p + "linear1.bias"]
        f = np.maximum(f_pre, 0)
        f, keep_fh = _dropout(f, drop_key(seed, step, site_ffh(l)), p_tf, train)
        g = f @ params[p + "linear2.weight"].T + params[p + "linear2.bias"]
        g, keep_ff = _dropout(g, drop_key(seed, step, site_ff(l)), p_tf, train)
        h2, xhat2, rstd2 = _ln_fwd(h1 + g, params[p + "norm2.weight"], params[p + "norm2.bias"], eps)
        lc.update(q=q, k=k, v=v, prob=prob, keep_a=keep_a, probd=probd, o=o, keep_sa=keep_sa,
                  xhat1=xhat1, rstd1=rstd1, h1=h1, f_pre=f_pre, f=f, keep_fh=keep_fh,
                  keep_ff=keep_ff, xhat2=xhat2, rstd2=rstd2)
        c["lay"].append(lc)
        h = h2
        if taps is not None:
            taps[f"layer{l}"] = h
            taps[f"qkv{l}"], taps[f"o{l}"], taps[f"h1_{l}"], taps[f"f{l}"] = qkv, o, h1, f
    # final LN applies to every row in torch, but only t = 0 is consumed (Modules.py:54)
    zf, xhatf, rstdf = _ln_fwd(h[:, 0, :], params["transformer.norm.weight"], params["transformer.norm.bias"], eps)
    if taps is not None:
        taps["final_ln_t0"] = zf
    zm = zf.reshape(n // samples, samples, d).mean(1)               # Modules.py:55
    wq = params["projection.weight"][:, :, 0]
    e_raw = zm @ wq.T + params["projection.bias"]                   # Modules.py:56
    nrm = np.maximum(np.sqrt((e_raw * e_raw).sum(-1, keepdims=True)), dt.type(1e-12))
    emb = e_raw / nrm                                               # Modules.py:57
    c.update(xhatf=xhatf, rstdf=rstdf, zm=zm, nrm=nrm, emb=emb, t=t, n=n, d=d)
    return emb, c


def encoder_backward(params, c, d_emb, taps=None):
    """Gradient of every parameter given d(loss)/d(emb); what autograd computes at Train.py:153."""
    dt = d_emb.dtype
    n, t, d, heads, samples = c["n"], c["t"], c["d"], c["heads"], c["samples"]
    dh = d // heads
    p_pe, p_tf, train = c["p_pe"], c["p_tf"], c["train"]
    g = OrderedDict()
    emb, nrm = c["emb"], c["nrm"]
    d_raw = (d_emb - emb * (d_emb * emb).sum(-1, keepdims=True)) / nrm
    wq = params["projection.weight"][:, :, 0]
    g["projection.weight"] = (d_raw.T @ c["zm"])[:, :, None]
    g["projection.bias"] = d_raw.sum(0)
    dzm = d_raw @ wq
    dzf = np.repeat(dzm / dt.type(samples), samples, axis=0)
    dh0, gw, gb = _ln_bwd(dzf, c["xhatf"], c["rstdf"], params["transformer.norm.weight"])
    g["transformer.norm.weight"], g["transformer.norm.bias"] = gw, gb
    dhh = np.zeros((n, t, d), dt)
    dhh[:, 0, :] = dh0
    scale = dt.type(1.0 / math.sqrt(dh))

    def undrop(dy, keep, p):
        if keep is None:
            return dy
        return dy * keep * dt.type(1.0 / (1.0 - p))

    for l in reversed(range(c["layers"])):
        p = f"transformer.layers.{l}."
        lc = c["lay"][l]
        dpre2, gw, gb = _ln_bwd(dhh, lc["xhat2"], lc["rstd2"], params[p + "norm2.weight"])
        g[p + "norm2.weight"], g[p + "norm2.bias"] = gw, gb
        dg = undrop(dpre2, lc["keep_ff"], p_tf)
        g[p + "linear2.weight"] = np.einsum("ntd,ntf->df", dg, lc["f"])
        g[p + "linear2.bias"] = dg.sum((0, 1))
        df = dg @ params[p + "linear2.weight"]
        df = undrop(df, lc["keep_fh"], p_tf) * (lc["f_pre"] > 0)
        g[p + "linear1.weight"] = np.einsum("ntf,ntd->fd", df, lc["h1"])
        g[p + "linear1.bias"] = df.sum((0, 1))
        dh1 = dpre2 + df @ params[p + "linear1.weight"]
        dpre1, gw, gb = _ln_bwd(dh1, lc["xhat1"], lc["rstd1"], params[p + "norm1.weight"])
        g[p + "norm1.weight"], g[p + "norm1.bias"] = gw, gb
        da = undrop(dpre1, lc["keep_sa"], p_tf)
        g[p + "self_attn.out_proj.weight"] = np.einsum("ntd,nte->de", da, lc["o"])
        g[p + "self_attn.out_proj.bias"] = da.sum((0, 1))
        do = (da @ params[p + "self_attn.out_proj.weight"]).reshape(n, t, heads, dh).transpose(0, 2, 1, 3)
        dv = lc["probd"].transpose(0, 1, 3, 2) @ do
        dprob = undrop(do @ lc["v"].transpose(0, 1, 3, 2), lc["keep_a"], p_tf)
        prob = lc["prob"]
        ds = prob * (dprob - (dprob * prob).sum(-1, keepdims=True)) * scale
        dq = ds @ lc["k"]
        dk = ds.transpose(0, 1, 3, 2) @ lc["q"]
        dqkv = np.concatenate([a.transpose(0, 2, 1, 3).reshape(n, t, d) for a in (dq, dk, dv)], -1)
        g[p + "self_attn.in_proj_weight"] = np.einsum("ntf,ntd->fd", dqkv, lc["h_in"])
        g[p + "self_attn.in_proj_bias"] = dqkv.sum((0, 1))
        dhh = dpre1 + dqkv @ params[p + "self_attn.in_proj_weight"]
        if taps is not None:
            taps[f"dF{l}"], taps[f"dHb{l}"], taps[f"dP{l}"], taps[f"dM{l}"] = df, dh1, dpre1, da
            taps[f"dO{l}"] = do.transpose(0, 2, 1, 3).reshape(n, t, d)
            taps[f"dQKV{l}"] = dqkv
    dhp = undrop(dhh, c["keep_pe"], p_pe)
    g["positional_encoding.alpha"] = np.array([(dhp * c["pe"][None]).sum()], dt)
    dz0 = dhp * (c["z0"] > 0)
    if taps is not None:
        taps["dHa"] = dz0
    g["prenet.weight"] = np.einsum("ntd,nmt->dm", dz0, c["x"])[:, :, None]
    g["prenet.bias"] = dz0.sum((0, 1))
    return OrderedDict((name, g[name]) for name, _ in param_specs(c["x"].shape[1], d, c["layers"], params["transformer.layers.0.linear1.weight"].shape[0]))


# --------------------------------------------------------------------------------------
# GE2E loss (Modules.py:121-156) forward / backward
# --------------------------------------------------------------------------------------

def loss_forward(emb, pattern_per_speaker, w=10.0, b=-5.0):
    """loss = mean_i( logsumexp_s sim[i,s] - sim[i,spk(i)] ), sim = w*cos(e_i, c_s) - b with
    self-inclusive centroids and eps 1e-8 norm clamps (Modules.py:132-146)."""
    dt = emb.dtype
    n, d = emb.shape
    P = pattern_per_speaker
    S = n // P
    cent = emb.reshape(S, P, d).mean(1)
    en = np.maximum(np.sqrt((emb * emb).sum(-1, keepdims=True)), dt.type(1e-8))
    cn = np.maximum(np.sqrt((cent * cent).sum(-1, keepdims=True)), dt.type(1e-8))
    cos = (emb @ cent.T) / (en * cn.T)
    sim = dt.type(w) * cos - dt.type(b)
    m = sim.max(-1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(sim - m).sum(-1))
    own = np.arange(n) // P
    loss = (lse - sim[np.arange(n), own]).mean()
    cache = dict(emb=emb, cent=cent, en=en, cn=cn, cos=cos, sim=sim, lse=lse, own=own, P=P, w=w)
    return dt.type(loss), cache


def loss_backward(c, d_loss=1.0, with_wb=False):
    """SURVEY.md Appendix A matrix form (checked there against reference autograd in fp64).  with_wb: also the gradients of the
    criterion's own weight / bias (Modules.py:115-116; sim = w cos - b): dL/dw = sum (softmax - onehot) cos / N, dL/db = -sum(...) = 0."""
    emb, cent, en, cn, cos = c["emb"], c["cent"], c["en"], c["cn"], c["cos"]
    dt = emb.dtype
    n, d = emb.shape
    P = c["P"]
    S = n // P
    soft = np.exp(c["sim"] - c["lse"][:, None])
    G = soft.copy()
    G[np.arange(n), c["own"]] -= 1.0
    G *= dt.type(c["w"] * d_loss / n)                                # dL/dcos
    ehat = emb / en
    chat = cent / cn
    direct = (G @ chat) / en - (G * cos).sum(-1, keepdims=True) * emb / (en * en)
    dC = (G.T @ ehat) / cn - (G * cos).sum(0)[:, None] * cent / (cn * cn)
    d_emb = (direct + np.repeat(dC, P, axis=0) / dt.type(P)).astype(dt)
    if not with_wb:
        return d_emb
    G0 = G.astype(np.float64) / float(c["w"])
    return d_emb, float((G0 * cos).sum()), float(-G0.sum())


# --------------------------------------------------------------------------------------
# optimiser step of Train.py:154-162 (clip_grad_norm_ 1.0 -> AdamW)
# --------------------------------------------------------------------------------------

def clip_grad_norm(grads, max_norm):
    total = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values()))
    coef = min(1.0, max_norm / (total + 1e-6))
    if coef < 1.0:
        for k in grads:
            grads[k] = grads[k] * grads[k].dtype.type(coef)
    return total


def adamw_step(params, grads, state, lr=1e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01):
    """torch.optim.AdamW single step (decoupled decay first, bias-corrected moments)."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    b1, b2 = betas
    for k, p in params.items():
        g = grads[k]
        m = state.setdefault("m", {}).setdefault(k, np.zeros_like(p))
        v = state.setdefault("v", {}).setdefault(k, np.zeros_like(p))
        p *= p.dtype.type(1.0 - lr * weight_decay)
        m *= p.dtype.type(b1); m += p.dtype.type(1 - b1) * g
        v *= p.dtype.type(b2); v += p.dtype.type(1 - b2) * g * g
        bc1 = 1.0 - b1 ** t
        bc2 = 1.0 - b2 ** t
        denom = np.sqrt(v) / p.dtype.type(math.sqrt(bc2)) + p.dtype.type(eps)
        p -= p.dtype.type(lr / bc1) * (m / denom)


def train_step(params, x, pattern_per_speaker, opt_state, seed=0, step=0, p_pe=0.1, p_tf=0.1,
               train=True, max_norm=1.0, lr=1e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01):
    """One Trainer.Train_Step (Train.py:140-168) on CPU: fwd -> loss -> bwd -> clip -> AdamW."""
    emb, c = encoder_forward(params, x, 1, train=train, seed=seed, step=step, p_pe=p_pe, p_tf=p_tf)
    loss, lc = loss_forward(emb, pattern_per_speaker)
    grads = encoder_backward(params, c, loss_backward(lc))
    gnorm = clip_grad_norm(grads, max_norm) if max_norm > 0 else None
    adamw_step(params, grads, opt_state, lr, betas, eps, weight_decay)
    return float(loss), gnorm, grads
