"""Every kernel of the 16-bit arithmetic modes (bf16 / fp16), checked ONE BY ONE against an fp64 evaluation of its own inputs.

The end-to-end bounds of the 16-bit modes (tests/test_gpu_parity.py) are wide because storage rounding accumulates over
~20 kernels; a band that wide could hide a real bug in a kernel that exists only in 16-bit form (the weight-stationary
GEMMs, the stage-stream FFN2 + LayerNorm, LayerNorm-backward fused into a GEMM prologue, the chained FFN kernel).  Here each kernel's
INPUTS are read back from the workspace (ge2e_debug_tap) exactly as the kernel saw them -- already rounded to 16 bits
by their producers -- the kernel's formula is evaluated in fp64 on them (reference maths: SURVEY.md appendix A, i.e.
Modules.py:46-59 and the autograd of it; dropout masks from the shared counter hash), and the kernel's OUTPUT must be
that value rounded to the storage type.  Device output and expectation are BOTH rounded values, so they differ only where
the fp32-vs-fp64 accumulation difference moves a result across a rounding boundary (a small fraction of the elements, by
one ulp): the bound is a relative L2 of 0.05 x the unit roundoff of the type (observed <= 0.02; attention, whose
probabilities go through a fast exp2 and an internal rounding: 0.4 x, observed <= 0.15) and no element further than 3 ulp
of the tensor's scale -- two orders of magnitude below what a wrong term, a missed mask or a mis-addressed ragged row
would produce (the end-to-end bf16 gradient bound of test_gpu_parity.py is ~60 x the unit roundoff).  Weight gradients (fp32 outputs) get
2e-4 relative.
Shapes are ragged on purpose (T = 77: not a multiple of 16; R = 462 rows: not a multiple of any tile)."""
import os

import numpy as np
import pytest
import torch

from oracle import ge2e_oracle as O
from test_gpu_parity import build, mods  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

EPS = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11}            # unit roundoff (half an ulp, relative)
TDT = {"bf16": torch.bfloat16, "fp16": torch.float16}
F64 = torch.float64


def rt(x, prec):
    """round to the storage type and back (what a kernel's final store does)"""
    return x.to(torch.float32).to(TDT[prec]).to(F64)


OBSERVED = []        # (what, relative L2 error in units of the unit roundoff): printed with `pytest -s` to recalibrate the bounds


def close(got, ref, prec, what, l2=0.05, ulps=3.0):
    """`got` (device output, already in the storage type) vs `ref` (fp64, NOT yet rounded)."""
    got, want = got.to(F64), rt(ref, prec)
    err = (got - want).norm().item() / max(want.norm().item(), 1e-30)
    # device value and expectation are both ROUNDED values: they differ where the fp32-vs-fp64 sums fall on the two sides of a rounding
    # boundary, one unit in the last place at a time.  On the last layer's compact tensors (one row per utterance: ~1,500 elements) a
    # single such flip is already ~0.05 roundoffs of relative L2, so the bound allows three flips of the largest elements on top
    bound = EPS[prec] * (l2 ** 2 + 3.0 * 4.0 * (want.abs().max().item() / max(want.norm().item(), 1e-30)) ** 2) ** 0.5
    OBSERVED.append((prec, what, round(err / EPS[prec], 4)))
    assert torch.isfinite(got).all(), what
    assert err < bound, (what, err, bound)
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= ulps * 2.0 * EPS[prec] * scale, (what, (got - want).abs().max().item(), scale)


def keep_rows(key, rows, width, p, row_mul=1):
    """keep mask [rows, width] of a dropout site whose counter is (row * row_mul) * width + col"""
    idx = (np.arange(rows, dtype=np.uint64)[:, None] * np.uint64(row_mul)) * np.uint64(width) + np.arange(width, dtype=np.uint64)[None, :]
    return torch.from_numpy(O.drop_keep_at(key, idx, p)).to(F64)


def ln_fwd(v, gamma, beta, eps=1e-5):
    mean = v.mean(-1, keepdim=True)
    var = ((v - mean) ** 2).mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    return (v - mean) * rstd * gamma + beta, rstd[:, 0]


def ln_bwd(dy, y, rstd, gamma, beta):
    """as the kernels do it: xhat rebuilt from the saved OUTPUT y"""
    xhat = (y - beta) / gamma
    dxh = dy * gamma
    dx = rstd[:, None] * (dxh - dxh.mean(-1, keepdim=True) - xhat * (dxh * xhat).mean(-1, keepdim=True))
    return dx, (dy * xhat).sum(0), dy.sum(0)


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("n,t", [(6, 77), (3, 160)])
def test_every_16bit_kernel_against_fp64_of_its_own_inputs(mods, prec, n, t):  # noqa: F811
    GE2E, _ = mods
    p, seed, heads, d, ffn, L = 0.1, 1234, 4, 256, 1024, 3
    m, params, pe = build(GE2E, prec, p)
    m.train()
    x_np = O.formula_mel(21, n, 80, t, logmel=True)
    x = torch.from_numpy(x_np).cuda()
    with torch.no_grad():
        m(x)                                              # train-mode forward, (seed, step) = (1234, 0); activations stay in the workspace
    R, scale_d = n * t, 1.0 / (1.0 - p)
    W = {k: rt(torch.from_numpy(v).to(F64), prec) for k, v in params.items()}      # prep_weights_kernel rounds the masters
    Pf = {k: torch.from_numpy(v).to(F64) for k, v in params.items()}                # biases / LayerNorm affine stay fp32

    def tap(name, shape, dtype=None):
        return m.workspace_view(name, n, t, True, dtype=dtype).cpu().to(F64).reshape(shape)

    # ---------------------------------------------------------------------------------- forward
    xt = tap("xt", (R, 128))[:, :80]
    assert torch.equal(xt, rt(torch.from_numpy(x_np).to(F64).permute(0, 2, 1).reshape(R, 80), prec))      # mel_pack_kernel
    pe_t = torch.from_numpy(pe[:t]).to(F64)                                                                   # [t, d]
    pre0 = xt @ W["prenet.weight"][:, :, 0].t() + Pf["prenet.bias"]
    h0_ref = (torch.relu(pre0) + Pf["positional_encoding.alpha"] * pe_t.repeat(n, 1)) * keep_rows(O.drop_key(seed, 0, O.SITE_PE), R, d, p) * scale_d
    h_in = tap("h0", (R, d))
    close(h_in, h0_ref, prec, "prenet + PE (gemm_nt EPI_PRENET)")
    saved = []
    for l in range(L):
        last = l == L - 1
        pre = f"transformer.layers.{l}."
        Rl, rmul = (n, t) if last else (R, 1)
        rows0 = torch.arange(n) * t                                   # frame-0 rows
        hin_c = h_in[rows0] if last else h_in                         # rows the layer's compact part works on
        qkv_ref = h_in @ W[pre + "self_attn.in_proj_weight"].t() + Pf[pre + "self_attn.in_proj_bias"]
        if last:     # Q of frame 0 only; K and V are never materialised (attn_last.cuh works on the layer input): fp64 K | V here
            q0 = tap("q0", (n, d))
            close(q0, qkv_ref[rows0, :d], prec, f"in_proj q0 layer {l}")
            qkv = qkv_ref.clone()
            qkv[rows0, :d] = q0
        else:
            qkv = tap(f"qkv.{l}", (R, 3 * d))
            close(qkv, qkv_ref, prec, f"in_proj layer {l} (gemm_ws EPI_BIAS)")
        # attention on the tapped q, k, v
        q, k, v = [qkv[:, i * d:(i + 1) * d].reshape(n, t, heads, 64).permute(0, 2, 1, 3) for i in range(3)]
        s = (q @ k.transpose(-1, -2)) / 8.0
        prob = torch.softmax(s, dim=-1)
        keep_a = torch.from_numpy(O.drop_keep_at(O.drop_key(seed, 0, O.site_attn(l)), O.attn_drop_index(n, heads, t), p)).to(F64)
        if last:
            # attn_last.cuh: one query per (utterance, head), no K / V.  qk_h = Wk_h^T q0_h and ctx_h = sum_t pd_t x_t are the operands of
            # matrix-pipe products, so they are rounded to the storage type; the probabilities stay fp32
            Wi, bi = W[pre + "self_attn.in_proj_weight"], Pf[pre + "self_attn.in_proj_bias"]
            q0h = qkv[rows0, :d].reshape(n, heads, 64)
            qk = rt(torch.einsum("nhj,hjc->nhc", q0h, Wi[d:2 * d].reshape(heads, 64, d)), prec)
            xin = h_in.reshape(n, t, d)
            prob0 = torch.softmax(torch.einsum("nhc,ntc->nht", qk, xin) / 8.0, dim=-1)
            pd0 = prob0 * keep_a[:, :, 0, :] * scale_d
            ctx = rt(torch.einsum("nht,ntc->nhc", pd0, xin), prec)
            o_ref = (torch.einsum("nhc,hjc->nhj", ctx, Wi[2 * d:].reshape(heads, 64, d))
                     + bi[2 * d:].reshape(1, heads, 64) * pd0.sum(-1, keepdim=True)).reshape(n, d)
        else:
            pd = rt(prob * keep_a * scale_d, prec)                                                 # P is packed to the storage type for P.V
            o_ref = (pd @ v).permute(0, 2, 1, 3).reshape(R, d)
        o = tap(f"o.{l}", (Rl, d))
        close(o, o_ref, prec, f"attention forward layer {l}", l2=0.4, ulps=4.0)
        if not last:
            lse = tap(f"lse.{l}", (R, heads), torch.float32)
            lse_ref = torch.logsumexp(s, dim=-1).permute(0, 2, 1).reshape(R, heads)
            assert (lse - lse_ref).abs().max().item() < 2e-5 * max(1.0, lse_ref.abs().max().item())
        a = (o @ W[pre + "self_attn.out_proj.weight"].t() + Pf[pre + "self_attn.out_proj.bias"]) \
            * keep_rows(O.drop_key(seed, 0, O.site_sa(l)), Rl, d, p, rmul) * scale_d
        h1_ref, rstd1_ref = ln_fwd(hin_c + a, Pf[pre + "norm1.weight"], Pf[pre + "norm1.bias"])
        h1 = tap(f"h1.{l}", (Rl, d))
        close(h1, h1_ref, prec, f"out_proj + residual + norm1 layer {l} (gemm_ws EPI_LN)")
        rstd1 = tap(f"rstd1.{l}", (Rl,), torch.float32)
        assert ((rstd1 - rstd1_ref).abs() / rstd1_ref).max().item() < 1e-5
        f_ref = torch.relu(h1 @ W[pre + "linear1.weight"].t() + Pf[pre + "linear1.bias"]) \
            * keep_rows(O.drop_key(seed, 0, O.site_ffh(l)), Rl, ffn, p, rmul) * scale_d
        f = tap(f"f.{l}", (Rl, ffn))
        close(f, f_ref, prec, f"FFN1 + ReLU + dropout layer {l}")
        g2 = (f @ W[pre + "linear2.weight"].t() + Pf[pre + "linear2.bias"]) * keep_rows(O.drop_key(seed, 0, O.site_ff(l)), Rl, d, p, rmul) * scale_d
        h2_ref, rstd2_ref = ln_fwd(h1 + g2, Pf[pre + "norm2.weight"], Pf[pre + "norm2.bias"])
        h2 = tap(f"h2.{l}", (Rl, d))
        close(h2, h2_ref, prec, f"FFN2 + residual + norm2 layer {l}")
        rstd2 = tap(f"rstd2.{l}", (Rl,), torch.float32)
        assert ((rstd2 - rstd2_ref).abs() / rstd2_ref).max().item() < 1e-5
        saved.append(dict(h_in=h_in, qkv=qkv, q=q, k=k, v=v, s=s, prob=prob, keep_a=keep_a, o=o, h1=h1, f=f, h2=h2,
                          rstd1=rstd1, rstd2=rstd2))
        h_in = h2

    # ---------------------------------------------------------------------------------- backward, full layer 1
    # (layer 2 is the frame-0-only layer; its compact kernels are the same instantiations at M = n.)  The backward is run twice
    # through the C ABI on the forward's workspace: stopped after the last layer to capture layer 1's incoming gradient, then
    # stopped after layer 1 to read that layer's scratch.
    hnd = m._handle()
    plist = [q_.detach() for q_ in m.parameters()]
    ptrs = hnd.ptr_table(plist)
    gen = torch.Generator().manual_seed(5)
    gs = 64.0 if prec == "fp16" else 1.0                                # half wants its gradients scaled up (GradScaler)
    d_emb = (torch.randn(n, d, generator=gen) * 0.05 * gs).cuda()
    grads = torch.empty(hnd.param_total, device="cuda")
    ws = m._ws[(True, prec)]
    stream = torch.cuda.current_stream().cuda_stream

    def run_backward(stop):
        from speaker_embedding_torch_amd import _lib
        _lib.set_option("debug_bwd_stop", stop)
        try:
            hnd.encoder_backward(stream, x, n, t, 1, ptrs, d_emb, grads, ws, seed, 0)
            torch.cuda.synchronize()
        finally:
            _lib.set_option("debug_bwd_stop", -1)

    run_backward(1)
    dy = tap("dHa.1", (R, d)).clone()                                     # dL/d(h2 of layer 1), written by the last layer's dgrad
    run_backward(2)
    l, sv = 1, saved[1]
    pre = f"transformer.layers.{l}."
    goff = {nm: (o_, k_) for nm, o_, k_ in zip(hnd.param_names, hnd.param_offset, hnd.param_numel)}

    def grad(nm):
        o_, k_ = goff[nm]
        return grads[o_:o_ + k_].cpu().to(F64)

    def wclose(nm, ref, tol=2e-4):
        got = grad(nm).reshape(ref.shape)
        err = (got - ref).norm().item() / max(ref.norm().item(), 1e-30)
        assert err < tol, (nm, err)

    dx2, dg2, db2 = ln_bwd(dy, sv["h2"], sv["rstd2"], Pf[pre + "norm2.weight"], Pf[pre + "norm2.bias"])
    # norm2 backward rides in the prologue of the chained FFN backward: dM (dropout2' of the input gradient dP) is stored for the weight
    # gradient; dP itself never leaves the chip (it is the start value of the dH1 accumulators, unrounded)
    dM1 = tap(f"dM1.{l}", (R, d))
    close(dM1, dx2 * keep_rows(O.drop_key(seed, 0, O.site_ff(l)), R, d, p) * scale_d, prec, "norm2 backward dmask")
    wclose(pre + "norm2.weight", dg2); wclose(pre + "norm2.bias", db2)
    dF = tap(f"dF.{l}", (R, ffn))
    close(dF, (dM1 @ W[pre + "linear2.weight"]) * (sv["f"] > 0).to(F64) * scale_d, prec, "dF = (dG W2) o mask (chained FFN backward, product 1)")
    wclose(pre + "linear2.weight", dM1.t() @ sv["f"]); wclose(pre + "linear2.bias", dM1.sum(0))
    wclose(pre + "linear1.weight", dF.t() @ sv["h1"]); wclose(pre + "linear1.bias", dF.sum(0))
    dHb = tap("dHb", (R, d))
    close(dHb, dx2 + dF @ W[pre + "linear1.weight"], prec, "dH1 = dPre2 + dF W1 (chained FFN backward, product 2)")
    dx1, dg1, db1 = ln_bwd(dHb, sv["h1"], sv["rstd1"], Pf[pre + "norm1.weight"], Pf[pre + "norm1.bias"])
    dP2, dM2 = tap(f"dP.{l}", (R, d)), tap(f"dM.{l}", (R, d))
    close(dP2, dx1, prec, "norm1 backward in the dO GEMM prologue (gemm_ws_lnbwd) dpre")
    close(dM2, dx1 * keep_rows(O.drop_key(seed, 0, O.site_sa(l)), R, d, p) * scale_d, prec, "norm1 backward dmask")
    wclose(pre + "norm1.weight", dg1); wclose(pre + "norm1.bias", db1)
    dO = tap("dO", (R, d))
    close(dO, dM2 @ W[pre + "self_attn.out_proj.weight"], prec, "dO = dA Wo (gemm_ws_lnbwd GEMM phase)")
    wclose(pre + "self_attn.out_proj.weight", dM2.t() @ sv["o"]); wclose(pre + "self_attn.out_proj.bias", dM2.sum(0))
    # attention backward on the tapped q, k, v, dO, O and the saved lse
    do_h = dO.reshape(n, t, heads, 64).permute(0, 2, 1, 3)
    o_h = sv["o"].reshape(n, t, heads, 64).permute(0, 2, 1, 3)
    prob, keep_a = sv["prob"], sv["keep_a"]
    delta = (do_h * o_h).sum(-1, keepdim=True)                          # = sum_k P dP, formed from the SAVED (rounded) O as the kernel does
    dpd = (do_h @ sv["v"].transpose(-1, -2)) * keep_a * scale_d
    ds = rt(prob * (dpd - delta) / 8.0, prec)                           # dS and the dropped P are packed to the storage type
    pdr = rt(prob * keep_a, prec)                                       # (the kept probabilities are packed UNSCALED; 1 / (1 - p) goes on the finished dV)
    dq, dk, dv = ds @ sv["k"], ds.transpose(-1, -2) @ sv["q"], (pdr.transpose(-1, -2) @ do_h) * scale_d
    dqkv_ref = torch.cat([z.permute(0, 2, 1, 3).reshape(R, d) for z in (dq, dk, dv)], dim=1)
    dQKV = tap(f"dQKV.{l}", (R, 3 * d))
    close(dQKV, dqkv_ref, prec, "attention backward (attn_bwd_kernel)", l2=0.4, ulps=6.0)
    wclose(pre + "self_attn.in_proj_weight", dQKV.t() @ sv["h_in"]); wclose(pre + "self_attn.in_proj_bias", dQKV.sum(0))
    dHa = tap("dHin.1", (R, d))
    close(dHa, dP2 + dQKV @ W[pre + "self_attn.in_proj_weight"], prec, "dH = dPre1 + dQKV Win (gemm_nt EPI_ADD, K = 768)")
    print("\n".join(f"  {a} {c:8.4f} eps  {b}" for a, b, c in OBSERVED[-40:]))


@pytest.mark.parametrize("prec,n,t", [("bf16", 12, 160), ("fp16", 9, 150), ("bf16", 7, 77)])
def test_fused_attention_sublayer_matches_its_three_launches(prec, n, t):
    """attn_sub.cuh (option attn_sub = 1; SURVEY section 7's second fusion: in_proj -> attention -> out_proj + dropout + residual + norm1 in ONE launch per
    utterance) against the three launches it replaces, same inputs, same dropout stream, train mode: q|k|v, the attention output and lse are the SAME
    arithmetic in the same order (bitwise); h1 and rstd1 differ only in the order of the LayerNorm sums (one rounding of the storage type)."""
    from speaker_embedding_torch_amd import _lib
    from speaker_embedding_torch_amd.Modules import GE2E
    import test_gpu_parity as tp
    x = torch.from_numpy(O.formula_mel(13, n, 80, t, logmel=True)).cuda()
    taps = {}
    for mode in (0, 1):
        _lib.set_option("attn_sub", mode)
        try:
            m, _, _ = tp.build(GE2E, prec, 0.1)
            m.train()
            emb = m(x)
            torch.cuda.synchronize()
            taps[mode] = {k: m.workspace_view(k, n, t, True).float().clone() for k in ("qkv.0", "o.0", "h1.0", "qkv.1", "o.1", "h1.1", "h2.1")}
            taps[mode]["lse.0"] = m.workspace_view("lse.0", n, t, True, dtype=torch.float32).clone()
            taps[mode]["rstd1.0"] = m.workspace_view("rstd1.0", n, t, True, dtype=torch.float32).clone()
            taps[mode]["emb"] = emb.detach().clone()
        finally:
            _lib.set_option("attn_sub", 0)
    a, b = taps[0], taps[1]
    for k in ("qkv.0", "o.0", "lse.0"):
        assert torch.equal(a[k], b[k]), k
    ulp = 2.0 ** -8 if prec == "bf16" else 2.0 ** -11
    assert (a["rstd1.0"] - b["rstd1.0"]).abs().max() <= 1e-5 * a["rstd1.0"].abs().max()
    assert (a["h1.0"] - b["h1.0"]).abs().max() <= 2 * ulp * a["h1.0"].abs().max()
    assert ((a["h1.0"] != b["h1.0"]).float().mean()) < 0.02              # a handful of single-ulp flips
    assert tp.rel_l2(b["h2.1"].cpu().numpy(), a["h2.1"].cpu().numpy()) < 5e-3 and tp.rel_l2(b["emb"].cpu().numpy(), a["emb"].cpu().numpy()) < 5e-3


@pytest.mark.parametrize("first", [0, 1])
def test_option_flipped_between_forward_and_backward(first):
    """Options are read at every call; what a forward LEFT in its workspace (the FFN's ReLU / dropout mask as bits, or not) is noted by the forward and
    read back by its backward, so flipping `no_ffn_chain` in between must not make the backward read bits that were never written: the gradients
    are those of the run whose forward had the same setting (same kernels on the forward side; the backward side may differ by the dF kernel's rounding)."""
    from speaker_embedding_torch_amd import _lib
    from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss
    import test_gpu_parity as tp
    n, t = 12, 96
    x = torch.from_numpy(O.formula_mel(17, n, 80, t, logmel=True)).cuda()

    def run(flip):
        _lib.set_option("no_ffn_chain", first)
        try:
            m, _, _ = tp.build(GE2E, "bf16", 0.1)
            m.train()
            m._step = 0
            emb = m(x)
            loss = GE2E_Loss().cuda()(emb, 3)
            torch.cuda.synchronize()
            if flip:
                _lib.set_option("no_ffn_chain", 1 - first)
            loss.backward()
            torch.cuda.synchronize()
            return {k: p.grad.detach().float().cpu().numpy().copy() for k, p in m.named_parameters()}
        finally:
            _lib.set_option("no_ffn_chain", 0)

    a, b = run(False), run(True)
    for k in a:
        if a[k].size > 1:
            assert tp.rel_l2(b[k], a[k]) < 2e-2, (k, tp.rel_l2(b[k], a[k]))


@pytest.mark.parametrize("prec,n,t", [("bf16", 20, 96), ("fp16", 37, 160), ("bf16", 16, 50)])
def test_last_layer_chain_matches_its_separate_launches(prec, n, t):
    """lastc.cuh (default; option no_last_chain = 1 restores the launches it replaces): the last layer below its attention and the tail -- out_proj +
    norm1, FFN + norm2, transformer.norm, projection, F.normalize -- as ONE launch on the compact rows, and the same chain backwards as one launch.
    Same inputs, same dropout stream, train mode, a partial last row tile: the saved tensors agree with the separate launches to a rounding of the
    storage type (the LayerNorm sums run in another order), the d-vectors and every gradient to the error those roundings propagate."""
    from speaker_embedding_torch_amd import _lib
    from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss
    import test_gpu_parity as tp
    x = torch.from_numpy(O.formula_mel(19, n, 80, t, logmel=True)).cuda()
    out = {}
    for mode in (1, 0):
        _lib.set_option("no_last_chain", mode)
        try:
            m, _, _ = tp.build(GE2E, prec, 0.1)
            m.train()
            m._step = 0
            emb = m(x)
            taps = {k: m.workspace_view(k, n, t, True).float().clone() for k in ("o.2", "h1.2", "f.2", "h2.2")}
            scale = 1024.0 if prec == "fp16" else 1.0
            (GE2E_Loss().cuda()(emb, 1 if n % 5 else 5) * scale).backward()
            torch.cuda.synchronize()
            out[mode] = (taps, emb.detach().clone(), {k: p.grad.detach().float().cpu().numpy() / scale for k, p in m.named_parameters()})
        finally:
            _lib.set_option("no_last_chain", 0)
    (ta, ea, ga), (tb, eb, gb) = out[1], out[0]
    assert torch.equal(ta["o.2"], tb["o.2"])                               # the attention above it is the same launch
    ulp = 2.0 ** -8 if prec == "bf16" else 2.0 ** -11
    assert (ta["h1.2"] - tb["h1.2"]).abs().max() <= 2 * ulp * ta["h1.2"].abs().max()
    for k in ("f.2", "h2.2"):
        assert tp.rel_l2(tb[k].cpu().numpy(), ta[k].cpu().numpy()) < 4 * ulp, k
    assert tp.rel_l2(eb.cpu().numpy(), ea.cpu().numpy()) < 4 * ulp
    nb = np.linalg.norm(ga["prenet.bias"])
    for k in ga:
        if ga[k].size == 1:
            assert abs(float(ga[k].ravel()[0] - gb[k].ravel()[0])) < 0.05 * nb, k
        else:
            assert tp.rel_l2(gb[k], ga[k]) < (0.05 if prec == "bf16" else 0.01), (k, tp.rel_l2(gb[k], ga[k]))


@pytest.mark.parametrize("prec,groups,samples,t", [("bf16", 7, 5, 64), ("fp16", 33, 3, 80), ("bf16", 5, 16, 40)])
def test_last_layer_chain_multislice_eval(prec, groups, samples, t):
    """Eval mode with `samples` slices per utterance (Inference.py:157-159; the slice mean of Modules.py:55 sits between transformer.norm and the
    projection): a row tile of the chain kernel holds 16 // samples whole utterances, so the mean stays inside it.  Against the separate launches
    (option no_last_chain) on the same input: the d-vectors agree to the roundings of the storage type that the two LayerNorm orders produce."""
    from speaker_embedding_torch_amd import _lib
    from speaker_embedding_torch_amd.Modules import GE2E
    import test_gpu_parity as tp
    n = groups * samples
    x = torch.from_numpy(O.formula_mel(23, n, 80, t, logmel=True)).cuda()
    out = {}
    for mode in (1, 0):
        _lib.set_option("no_last_chain", mode)
        try:
            m, _, _ = tp.build(GE2E, prec, 0.1)
            m.eval()
            with torch.no_grad():
                out[mode] = m(x, samples).detach().clone()
            torch.cuda.synchronize()
        finally:
            _lib.set_option("no_last_chain", 0)
    assert out[0].shape == (groups, 256)
    assert torch.isfinite(out[0]).all()
    ulp = 2.0 ** -8 if prec == "bf16" else 2.0 ** -11
    assert tp.rel_l2(out[0].cpu().numpy(), out[1].cpu().numpy()) < 4 * ulp
    assert (out[0].norm(dim=1) - 1.0).abs().max() < 1e-5
