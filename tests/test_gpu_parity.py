"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (ctypes -> libge2e_hip.so), against
  * the committed golden vectors produced by the reference's own Modules.py (tests/golden/), and
  * the CPU oracle on the same seeded inputs (same dropout stream),
plus size-independent properties at BASELINE.json's full size (64 x 15 x 160).

Tolerances (north_star: d-vectors within 1e-4 of the CPU reference on the fp32 path):
  fp32 path : d-vector max-abs <= 1e-5 and vector-relative <= 1e-4 (observed ~1e-6);
              gradients per tensor relative L2 <= 2e-3 (observed ~1e-4; fp32 atomics order varies)
  bf16 path : storage is bf16 (8 mantissa bits) -> d-vector max-abs <= 6e-3 on unit-norm vectors,
              relative L2 <= 2e-2; gradients: cosine >= 0.99 per tensor and relative L2 <= 0.15 (observed 0.094 worst + 50 %;
              ReLU masks flip where a pre-activation is within bf16 rounding of zero).
  fp16 path : storage is IEEE half (11 mantissa bits) -> 8x tighter than bf16: d-vector relative L2 <= 3e-3,
              gradients (taken under a loss scale, as the reference's GradScaler does) cosine >= 0.999, relative L2 <= 5e-2.
  16-bit kernels, one by one: tests/test_gpu_kernels_16bit.py checks every kernel of the bf16 / fp16 path against an
              fp64 evaluation of ITS OWN inputs (the tapped tensors), where the only error left is the output rounding.
"""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import ge2e_oracle as O
from conftest import rel_l2

pytestmark = pytest.mark.gpu


def make_hp(p=0.1, layers=3):
    return Namespace(Sound=Namespace(Mel_Dim=80),
                     GE2E=Namespace(Embedding_Size=256,
                                    Positional_Encoding=Namespace(Max_Position=1024, Dropout_Rate=p),
                                    Transformer=Namespace(Num_Layers=layers, Head=4, Dropout_Rate=p)))


@pytest.fixture(scope="module")
def mods():
    from speaker_embedding_torch_amd import _lib
    from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    _lib.load()      # the in-tree .so must be the thing that runs
    return GE2E, GE2E_Loss


def build(GE2E, precision, p, layers=3, seed=1234):
    m = GE2E(make_hp(p, layers), precision=precision, seed=seed).cuda()
    m._poison = True          # workspace starts as NaN patterns: reading anything the kernels did not write fails loudly
    params = O.formula_params(layers=layers)
    sd = m.state_dict()
    for k, v in params.items():
        sd[k].copy_(torch.from_numpy(v))
    pe = m.positional_encoding.pe[0].t().contiguous().cpu().numpy()
    return m, params, pe


# ------------------------------------------------------------------------------------------ goldens
def test_fp32_eval_matches_reference_goldens(mods, golden):
    GE2E, GE2E_Loss = mods
    m, _, _ = build(GE2E, "fp32", 0.0)
    m.eval()
    with torch.no_grad():
        emb = m(torch.from_numpy(O.formula_mel(1, 20, 80, 160)).cuda())
        loss = GE2E_Loss().cuda()(emb, 5)
    e = emb.cpu().numpy()
    assert np.abs(e - golden["G1_emb"]).max() < 1e-5
    assert rel_l2(e, golden["G1_emb"]) < 1e-4                      # north_star bound
    assert abs(loss.item() - float(golden["G2_loss"][0])) < 1e-5
    m2, _, _ = build(GE2E, "fp32", 0.1)
    m2.eval()
    with torch.no_grad():
        e5 = m2(torch.from_numpy(O.formula_mel(2, 20, 80, 64, logmel=True)).cuda(), 5).cpu().numpy()
        e77 = m2(torch.from_numpy(O.formula_mel(3, 6, 80, 77, logmel=True)).cuda())
        l77 = GE2E_Loss().cuda()(e77, 3).item()
    assert e5.shape == (4, 256) and rel_l2(e5, golden["G5_emb_samples5"]) < 1e-4
    assert rel_l2(e77.cpu().numpy(), golden["G5_emb_T77"]) < 1e-4
    assert abs(l77 - float(golden["G5_loss_T77"][0])) < 1e-5


def test_fp32_gradients_match_reference_goldens(mods, golden):
    """G3: dL/dtheta of the reference's autograd at dropout 0 (train mode)."""
    GE2E, GE2E_Loss = mods
    m, _, _ = build(GE2E, "fp32", 0.0)
    m.train()
    loss = GE2E_Loss().cuda()(m(torch.from_numpy(O.formula_mel(1, 20, 80, 160)).cuda()), 5)
    loss.backward()
    assert abs(loss.item() - float(golden["G3_loss_train"][0])) < 1e-5
    for i, (name, p) in enumerate(m.named_parameters()):
        g = p.grad.cpu().numpy()
        ref_norm = golden["G3_grad_norm"][i]
        assert abs(np.linalg.norm(g.astype(np.float64)) - ref_norm) < 2e-3 * ref_norm, name
        k = min(8, g.size)
        assert np.abs(g.reshape(-1)[:k] - golden["G3_grad_head"][i][:k]).max() < 2e-3 * ref_norm, name
    assert rel_l2(m.prenet.weight.grad.cpu().numpy(), golden["G3_grad_prenet_w"]) < 2e-3
    assert rel_l2(m.transformer.layers[1].self_attn.in_proj_bias.grad.cpu().numpy(), golden["G3_grad_l1_inproj_b"]) < 2e-3


def test_fp32x3_matches_reference_goldens_within_the_north_star_bound(mods, golden):
    """precision = 'fp32x3' (round 4): fp32 storage, the projection products and weight gradients as three bf16 MFMAs on operands split into
    bf16 hi + lo halves (common.cuh split_bf16x3: ~2^-17 relative per product, fp32 accumulation), everything else exact fp32.  The goldens of
    the reference's own Modules.py hold it to the SAME bounds as the fp32 mode where north_star states one -- d-vectors within 1e-4 relative
    (G1, G5) -- and to the fp32 gradient bound 2e-3 (G3); the absolute bounds are the observed error x 3 (max-abs 1e-5 -> 5e-5)."""
    GE2E, GE2E_Loss = mods
    m, _, _ = build(GE2E, "fp32x3", 0.0)
    m.eval()
    with torch.no_grad():
        emb = m(torch.from_numpy(O.formula_mel(1, 20, 80, 160)).cuda())
        loss = GE2E_Loss().cuda()(emb, 5)
        e5 = m(torch.from_numpy(O.formula_mel(2, 20, 80, 64, logmel=True)).cuda(), 5).cpu().numpy()
        e77 = m(torch.from_numpy(O.formula_mel(3, 6, 80, 77, logmel=True)).cuda()).cpu().numpy()
    e = emb.cpu().numpy()
    print("fp32x3 d-vector errors: G1 max-abs %.2e rel %.2e | samples5 rel %.2e | T77 rel %.2e | loss %.2e" % (
        np.abs(e - golden["G1_emb"]).max(), rel_l2(e, golden["G1_emb"]), rel_l2(e5, golden["G5_emb_samples5"]), rel_l2(e77, golden["G5_emb_T77"]),
        abs(loss.item() - float(golden["G2_loss"][0]))))
    assert np.abs(e - golden["G1_emb"]).max() < 5e-5
    assert rel_l2(e, golden["G1_emb"]) < 1e-4                      # north_star bound
    assert abs(loss.item() - float(golden["G2_loss"][0])) < 1e-4
    assert rel_l2(e5, golden["G5_emb_samples5"]) < 1e-4 and rel_l2(e77, golden["G5_emb_T77"]) < 1e-4
    m.train()                                                      # G3: dropout 0, train mode
    loss = GE2E_Loss().cuda()(m(torch.from_numpy(O.formula_mel(1, 20, 80, 160)).cuda()), 5)
    loss.backward()
    assert abs(loss.item() - float(golden["G3_loss_train"][0])) < 1e-4
    worst = 0.0
    for i, (name, p) in enumerate(m.named_parameters()):
        g = p.grad.cpu().numpy()
        ref_norm = golden["G3_grad_norm"][i]
        worst = max(worst, abs(np.linalg.norm(g.astype(np.float64)) - ref_norm) / ref_norm)
        assert abs(np.linalg.norm(g.astype(np.float64)) - ref_norm) < 2e-3 * ref_norm, name
        k = min(8, g.size)
        assert np.abs(g.reshape(-1)[:k] - golden["G3_grad_head"][i][:k]).max() < 2e-3 * ref_norm, name
    print("fp32x3 worst gradient-norm error %.2e | prenet.weight rel %.2e | in_proj_bias.1 rel %.2e" % (
        worst, rel_l2(m.prenet.weight.grad.cpu().numpy(), golden["G3_grad_prenet_w"]),
        rel_l2(m.transformer.layers[1].self_attn.in_proj_bias.grad.cpu().numpy(), golden["G3_grad_l1_inproj_b"])))
    assert rel_l2(m.prenet.weight.grad.cpu().numpy(), golden["G3_grad_prenet_w"]) < 2e-3
    assert rel_l2(m.transformer.layers[1].self_attn.in_proj_bias.grad.cpu().numpy(), golden["G3_grad_l1_inproj_b"]) < 2e-3


@pytest.mark.parametrize("n,t,P,p,tag", [(12, 77, 3, 0.1, 3), (20, 160, 5, 0.1, 1)])
def test_fp32x3_train_step_vs_oracle(mods, n, t, P, p, tag):
    """fp32x3, dropout on, ragged and headline frame counts: d-vectors within 1e-4 of the oracle, loss 1e-4, every gradient tensor 2e-3 (the
    fp32 mode's bounds: test_fp32_train_step_vs_oracle)."""
    GE2E, GE2E_Loss = mods
    m, params, pe = build(GE2E, "fp32x3", p)
    m.train()
    x_np = O.formula_mel(tag, n, 80, t, logmel=True)
    emb_ref, c = O.encoder_forward(params, x_np, train=True, seed=1234, step=0, p_pe=p, p_tf=p, pe=pe)
    loss_ref, lc = O.loss_forward(emb_ref, P)
    grads_ref = O.encoder_backward(params, c, O.loss_backward(lc))
    emb = m(torch.from_numpy(x_np).cuda())
    loss = GE2E_Loss().cuda()(emb, P)
    loss.backward()
    e = emb.detach().cpu().numpy()
    assert np.abs(e - emb_ref).max() < 5e-5 and rel_l2(e, emb_ref) < 1e-4
    assert abs(loss.item() - float(loss_ref)) < 1e-4
    tol = 2e-3 if _relu_margin_ok(c, tol=2e-4) else 0.2
    worst = max(rel_l2(prm.grad.cpu().numpy(), grads_ref[name]) for name, prm in m.named_parameters())
    print("fp32x3 train step %dx%d: d-vector rel %.2e, worst gradient rel %.2e" % (n, t, rel_l2(e, emb_ref), worst))
    for name, prm in m.named_parameters():
        assert rel_l2(prm.grad.cpu().numpy(), grads_ref[name]) < tol, name


@pytest.mark.parametrize("tag,s,p", [(0, 4, 5), (1, 64, 15), (2, 256, 10)])
def test_loss_matches_reference_goldens(mods, golden, tag, s, p):
    _, GE2E_Loss = mods
    e = O.formula_normal(50 + tag, (s * p, 256))
    e = e + 2.0 * np.repeat(O.formula_normal(60 + tag, (s, 256)), p, axis=0)
    e = (e / np.linalg.norm(e, axis=1, keepdims=True)).astype(np.float32)
    et = torch.from_numpy(e).cuda().requires_grad_(True)
    crit = GE2E_Loss().cuda()
    loss = crit(et, p)
    loss.backward()
    assert abs(loss.item() - float(golden[f"G6_loss_{s}x{p}"][0])) < 1e-5
    # G8: the criterion's own weight / bias gradients as the reference's autograd fills them (Modules.py:115-116); dL/db is 0 up to rounding
    dw_ref, db_ref = golden[f"G8_dw_db_{s}x{p}"]
    assert abs(crit.weight.grad.item() - dw_ref) < 1e-4 * abs(dw_ref) + 1e-8 and abs(crit.bias.grad.item() - db_ref) < 1e-6
    g = et.grad.cpu().numpy()
    ref_norm = float(golden[f"G6_demb_norm_{s}x{p}"][0])
    assert abs(np.linalg.norm(g.astype(np.float64)) - ref_norm) < 1e-3 * ref_norm
    assert np.abs(g[:4] - golden[f"G6_demb_head_{s}x{p}"]).max() < 5e-3 * np.abs(golden[f"G6_demb_head_{s}x{p}"]).max()


def test_loss_unnormalised_and_upstream_scale(mods, golden):
    _, GE2E_Loss = mods
    e = O.formula_normal(70, (12, 256)).astype(np.float32) * np.float32(0.3)
    et = torch.from_numpy(e).cuda().requires_grad_(True)
    crit = GE2E_Loss().cuda()
    loss = crit(et, 4)
    (loss * 3.0).backward()                     # upstream gradient is read on the device
    assert abs(loss.item() - float(golden["G6_loss_unnorm_3x4"][0])) < 1e-5
    dw_ref, db_ref = golden["G8_dw_db_unnorm_3x4"]
    assert abs(crit.weight.grad.item() / 3.0 - dw_ref) < 1e-4 * abs(dw_ref) and abs(crit.bias.grad.item()) < 1e-6
    frozen = GE2E_Loss().cuda()
    frozen.weight.requires_grad_(False); frozen.bias.requires_grad_(False)
    frozen(et.detach().requires_grad_(True), 4).backward()
    assert frozen.weight.grad is None and frozen.bias.grad is None
    assert rel_l2(et.grad.cpu().numpy() / 3.0, golden["G6_demb_unnorm_3x4"]) < 1e-3


# ------------------------------------------------------------------------------------------ oracle, same dropout stream
def _relu_margin_ok(cache, tol=2e-5):
    """A pre-activation within rounding of 0 can flip its ReLU mask between two fp32 implementations;
    such measure-zero cases are excluded from the tight comparison (they are covered loosely)."""
    return all(np.abs(lc["f_pre"]).min() > tol for lc in cache["lay"]) and np.abs(cache["z0"]).min() > tol


CASES = [  # n, t, P, dropout, tag
    (20, 160, 5, 0.1, 1),      # config 1 of BASELINE.json
    (12, 77, 3, 0.1, 5),       # ragged: T not a multiple of 16, rows not a multiple of the 128-row tile
    (8, 270, 4, 0.1, 4),       # longest trained length of the shipped YAML (Frame_Length.Max)
    (6, 33, 2, 0.25, 6),       # short, heavy dropout
    (10, 180, 5, 0.1, 7),      # frame count of BASELINE.json configs[4]
]


@pytest.mark.parametrize("n,t,P,p,tag", CASES)
def test_fp32_train_step_vs_oracle(mods, n, t, P, p, tag):
    GE2E, GE2E_Loss = mods
    m, params, pe = build(GE2E, "fp32", p)
    m.train()
    x_np = O.formula_mel(tag, n, 80, t, logmel=True)
    taps = {}
    emb_ref, c = O.encoder_forward(params, x_np, train=True, seed=1234, step=0, p_pe=p, p_tf=p, taps=taps, pe=pe)
    loss_ref, lc = O.loss_forward(emb_ref, P)
    grads_ref = O.encoder_backward(params, c, O.loss_backward(lc))
    emb = m(torch.from_numpy(x_np).cuda())
    loss = GE2E_Loss().cuda()(emb, P)
    loss.backward()
    for dev, ora, w in (("h0", "prenet_pe", 256), ("qkv.1", "qkv1", 768), ("o.1", "o1", 256), ("f.1", "f1", 1024), ("h2.1", "layer1", 256)):
        got = m.workspace_view(dev, n, t, True).float().cpu().numpy().reshape(n, t, w)
        assert rel_l2(got, taps[ora]) < 1e-5, dev
    # the last layer is evaluated for frame 0 only (Modules.py:54 consumes nothing else): compact [n, w] taps
    for dev, ora, w in (("o.2", "o2", 256), ("h1.2", "h1_2", 256), ("f.2", "f2", 1024), ("h2.2", "layer2", 256)):
        got = m.workspace_view(dev, n, t, True).float().cpu().numpy().reshape(n, w)
        assert rel_l2(got, taps[ora][:, 0, :]) < 1e-5, dev
    e = emb.detach().cpu().numpy()
    assert np.abs(e - emb_ref).max() < 1e-5 and rel_l2(e, emb_ref) < 1e-4
    assert abs(loss.item() - float(loss_ref)) < 1e-5
    tol = 2e-3 if _relu_margin_ok(c) else 0.2
    for name, prm in m.named_parameters():
        assert rel_l2(prm.grad.cpu().numpy(), grads_ref[name]) < tol, name


# (+ one case with dropout OFF in train mode: the attention kernels are compiled per dropout state -- DROP = false instantiations of the backward)
@pytest.mark.parametrize("n,t,P,p,tag", CASES[:3] + [(20, 160, 5, 0.0, 1)])
def test_bf16_train_step_vs_oracle(mods, n, t, P, p, tag):
    GE2E, GE2E_Loss = mods
    m, params, pe = build(GE2E, "bf16", p)
    m.train()
    x_np = O.formula_mel(tag, n, 80, t, logmel=True)
    emb_ref, c = O.encoder_forward(params, x_np, train=True, seed=1234, step=0, p_pe=p, p_tf=p, pe=pe)
    loss_ref, lc = O.loss_forward(emb_ref, P)
    grads_ref = O.encoder_backward(params, c, O.loss_backward(lc))
    emb = m(torch.from_numpy(x_np).cuda())
    loss = GE2E_Loss().cuda()(emb, P)
    loss.backward()
    e = emb.detach().cpu().numpy()
    assert np.abs(np.linalg.norm(e, axis=1) - 1).max() < 1e-5          # the tail is fp32 in both modes
    assert np.abs(e - emb_ref).max() < 6e-3 and rel_l2(e, emb_ref) < 2e-2
    assert abs(loss.item() - float(loss_ref)) < 2e-2 * max(1.0, abs(float(loss_ref)))
    for name, prm in m.named_parameters():
        g, r = prm.grad.cpu().numpy().ravel().astype(np.float64), grads_ref[name].ravel().astype(np.float64)
        if g.size == 1:      # alpha: one scalar = a sum over R*256 terms of both signs; bound it by the terms' scale (observed over the
            # three cases and two dropout streams: up to 0.205 of that norm -- the error of ONE weighted column sum of the tensor whose
            # 256 plain column sums, prenet.bias, are off by 0.094 of their norm)
            assert abs(g[0] - r[0]) < 0.3 * np.linalg.norm(grads_ref["prenet.bias"]), name
            continue
        cos = float(g @ r / max(np.linalg.norm(g) * np.linalg.norm(r), 1e-30))
        # observed on the MI355X over the three cases: worst relative L2 0.094 (prenet.bias), lowest cosine 0.9957 -> bound = that + 50 %.
        # (Each bf16 kernel on its own is exact to a rounding: tests/test_gpu_kernels_16bit.py; profiles/r02_precision_taps.md
        # shows where the storage rounding accrues.)
        # (the dropout-off case: the last layer's linear1 gradients -- 20 compact rows, a handful of hidden units whose pre-activation rounds
        # across zero in bf16 -- sit at 0.151 / 0.9889; the same case in fp16 passes at 0.05 / 0.999)
        lim, cmin = (0.2, 0.985) if p == 0.0 else (0.15, 0.99)
        assert cos > cmin and rel_l2(g, r) < lim, (name, cos, rel_l2(g, r))


@pytest.mark.parametrize("n,t,P,p,tag", CASES[:3] + [(20, 160, 5, 0.0, 1)])
def test_f16_train_step_vs_oracle(mods, n, t, P, p, tag):
    """BASELINE.json configs[4]'s arithmetic: float16 storage (the reference's autocast dtype, Train.py:145).  Gradients are
    taken under a loss scale like GradScaler's (Train.py:153) and compared after unscaling.  Half has 3 more mantissa
    bits than bf16, so every bound is tighter than the bf16 test's."""
    GE2E, GE2E_Loss = mods
    m, params, pe = build(GE2E, "fp16", p)
    m.train()
    x_np = O.formula_mel(tag, n, 80, t, logmel=True)
    emb_ref, c = O.encoder_forward(params, x_np, train=True, seed=1234, step=0, p_pe=p, p_tf=p, pe=pe)
    loss_ref, lc = O.loss_forward(emb_ref, P)
    grads_ref = O.encoder_backward(params, c, O.loss_backward(lc))
    emb = m(torch.from_numpy(x_np).cuda())
    loss = GE2E_Loss().cuda()(emb, P)
    scale = 4096.0
    (loss * scale).backward()
    e = emb.detach().cpu().numpy()
    assert np.abs(np.linalg.norm(e, axis=1) - 1).max() < 1e-5
    assert np.abs(e - emb_ref).max() < 1e-3 and rel_l2(e, emb_ref) < 3e-3
    assert abs(loss.item() - float(loss_ref)) < 3e-3 * max(1.0, abs(float(loss_ref)))
    for name, prm in m.named_parameters():
        g, r = (prm.grad.cpu().numpy().ravel() / scale).astype(np.float64), grads_ref[name].ravel().astype(np.float64)
        assert np.isfinite(g).all(), name
        if g.size == 1:      # alpha: a signed sum over R x 256 terms; its error scale is that of the prenet tensors next to it (observed 0.033-0.034
            # of |d prenet.bias| where prenet.weight / prenet.bias themselves sit at 0.029 / 0.030 relative L2): the same 5e-2 as below
            assert abs(g[0] - r[0]) < 0.05 * np.linalg.norm(grads_ref["prenet.bias"]), name
            continue
        cos = float(g @ r / max(np.linalg.norm(g) * np.linalg.norm(r), 1e-30))
        assert cos > 0.999 and rel_l2(g, r) < 5e-2, (name, cos, rel_l2(g, r))


@pytest.mark.parametrize("prec,n,t,P,tag", [("bf16", 20, 160, 5, 1), ("bf16", 12, 77, 3, 3), ("fp16", 20, 160, 5, 1), ("fp16", 12, 77, 3, 3)])
def test_16bit_gradient_error_pinned_without_dropout(mods, prec, n, t, P, tag):
    """the end-to-end 16-bit bounds of the dropout-on tests were widened to cover the variance of one mask sample (alpha: bf16 0.15 ->
    0.3 of |d prenet.bias|, fp16 0.03 -> 0.05); a precision regression in a packed epilogue could hide under that.  With dropout OFF nothing varies:
    the bounds here are round 2's TIGHT ones (alpha 0.15 / 0.03) halved, and per tensor the observed error + 30 % (profiles/r04_grad_err.txt: bf16
    worst 0.162 = the last layer's linear1 on 20 compact rows, 0.117 elsewhere; fp16 0.040) -- so the widened limits above cover mask-sample
    variance only."""
    GE2E, GE2E_Loss = mods
    m, params, pe = build(GE2E, prec, 0.0)
    m.train()
    x_np = O.formula_mel(tag, n, 80, t, logmel=True)
    emb_ref, c = O.encoder_forward(params, x_np, train=True, seed=1234, step=0, p_pe=0.0, p_tf=0.0, pe=pe)
    _, lc = O.loss_forward(emb_ref, P)
    grads_ref = O.encoder_backward(params, c, O.loss_backward(lc))
    scale = 4096.0 if prec == "fp16" else 1.0
    (GE2E_Loss().cuda()(m(torch.from_numpy(x_np).cuda()), P) * scale).backward()
    nb = np.linalg.norm(grads_ref["prenet.bias"])
    a_lim, r_lim, r_last, c_min = (0.075, 0.155, 0.21, 0.985) if prec == "bf16" else (0.015, 0.052, 0.052, 0.9985)
    for name, prm in m.named_parameters():
        g, r = (prm.grad.cpu().numpy().ravel() / scale).astype(np.float64), grads_ref[name].ravel().astype(np.float64)
        if g.size == 1:
            assert abs(g[0] - r[0]) < a_lim * nb, (name, abs(g[0] - r[0]) / nb)
            continue
        cos = float(g @ r / max(np.linalg.norm(g) * np.linalg.norm(r), 1e-30))
        lim = r_last if name.startswith("transformer.layers.2.linear1") else r_lim
        assert rel_l2(g, r) < lim and cos > c_min, (name, rel_l2(g, r), cos)


@pytest.mark.parametrize("prec,n,t,P", [("fp32", 4, 512, 2), ("fp32", 3, 300, 3), ("bf16", 4, 512, 2), ("fp16", 4, 1024, 2)])
def test_long_sequences_up_to_max_position(mods, prec, n, t, P):
    """reference Modules.py:107-109 slices the positional table for any T <= Max_Position (1024).  Beyond 288 frames the
    attention streams K / V through LDS in chunks (online softmax forward, two chunked backward kernels): a full train step
    against the oracle, same dropout masks, at T = 512 (8 chunks), 300 (just past the resident kernels, ragged last chunk)
    and the maximum 1024."""
    GE2E, GE2E_Loss = mods
    p = 0.1
    m, params, pe = build(GE2E, prec, p)
    m.train()
    x_np = O.formula_mel(31, n, 80, t, logmel=True)
    taps = {}
    emb_ref, c = O.encoder_forward(params, x_np, train=True, seed=1234, step=0, p_pe=p, p_tf=p, taps=taps, pe=pe)
    loss_ref, lc = O.loss_forward(emb_ref, P)
    grads_ref = O.encoder_backward(params, c, O.loss_backward(lc))
    emb = m(torch.from_numpy(x_np).cuda())
    loss = GE2E_Loss().cuda()(emb, P)
    scale = 1024.0 if prec == "fp16" else 1.0
    (loss * scale).backward()
    e = emb.detach().cpu().numpy()
    o1 = m.workspace_view("o.1", n, t, True).float().cpu().numpy().reshape(n, t, 256)
    if prec == "fp32":
        assert rel_l2(o1, taps["o1"]) < 1e-5                           # the chunked attention itself
        assert np.abs(e - emb_ref).max() < 1e-5 and rel_l2(e, emb_ref) < 1e-4
        assert abs(loss.item() - float(loss_ref)) < 1e-5
        tol = 2e-3 if _relu_margin_ok(c) else 0.2
        for name, prm in m.named_parameters():
            assert rel_l2(prm.grad.cpu().numpy(), grads_ref[name]) < tol, name
    else:
        # (fp16 at 1,024 frames: worst tensor 0.068 / 0.9977 under the quad-hash dropout stream, 0.05 / 0.9985 under round 2's pair hash:
        # a few hidden units whose pre-activation rounds across zero; bounds = observed + ~30 %)
        etol, gtol, ctol = (2e-2, 0.3, 0.97) if prec == "bf16" else (3e-3, 9e-2, 0.996)
        assert rel_l2(o1, taps["o1"]) < etol and rel_l2(e, emb_ref) < etol
        for name, prm in m.named_parameters():
            g, r = (prm.grad.cpu().numpy().ravel() / scale).astype(np.float64), grads_ref[name].ravel().astype(np.float64)
            if g.size == 1:
                continue
            assert np.isfinite(g).all(), name
            cos = float(g @ r / max(np.linalg.norm(g) * np.linalg.norm(r), 1e-30))
            assert cos > ctol and rel_l2(g, r) < gtol, (name, cos, rel_l2(g, r))
    # eval mode (multi-slice inference shape family) at the same length
    m.eval()
    ref, _ = O.encoder_forward(params, x_np, pe=pe)
    with torch.no_grad():
        ev = m(torch.from_numpy(x_np).cuda()).cpu().numpy()
    assert rel_l2(ev, ref) < {"fp32": 1e-4, "bf16": 2e-2, "fp16": 3e-3}[prec]


@pytest.mark.parametrize("layers", [1, 2])
def test_fp32_other_layer_counts(mods, layers):
    """Num_Layers is a hyper-parameter: the frame-0-only treatment of the LAST layer must hold for any depth."""
    GE2E, GE2E_Loss = mods
    m, params, pe = build(GE2E, "fp32", 0.1, layers=layers)
    m.train()
    x_np = O.formula_mel(11, 8, 80, 96, logmel=True)
    emb_ref, c = O.encoder_forward(params, x_np, train=True, seed=1234, step=0, pe=pe)
    loss_ref, lc = O.loss_forward(emb_ref, 4)
    grads_ref = O.encoder_backward(params, c, O.loss_backward(lc))
    emb = m(torch.from_numpy(x_np).cuda())
    loss = GE2E_Loss().cuda()(emb, 4)
    loss.backward()
    assert rel_l2(emb.detach().cpu().numpy(), emb_ref) < 1e-4 and abs(loss.item() - float(loss_ref)) < 1e-5
    tol = 2e-3 if _relu_margin_ok(c) else 0.2
    for name, prm in m.named_parameters():
        assert rel_l2(prm.grad.cpu().numpy(), grads_ref[name]) < tol, name


_DEPTH_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, {repo!r}); sys.path.insert(0, {tests!r})
import test_gpu_parity as tp
from oracle import ge2e_oracle as O
from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss
m, params, pe = tp.build(GE2E, {prec!r}, 0.1, layers={layers})          # NaN-poisoned workspace
m.train()
x = torch.from_numpy(O.formula_mel(41, {n}, 80, {t}, logmel=True)).cuda()
out = {{}}
for rep in range(2):                                                     # twice: the second backward recycles the first one's event set
    m.zero_grad(); m._step = 0
    emb = m(x); loss = GE2E_Loss().cuda()(emb, 4); (loss * {scale}).backward()
    torch.cuda.synchronize()
    out["emb%d" % rep] = emb.detach().cpu().numpy(); out["loss%d" % rep] = np.float32(loss.item())
    for k, p in m.named_parameters():
        out["g%d_" % rep + k] = p.grad.cpu().numpy() / {scale}
np.savez({out!r}, **out)
"""


@pytest.mark.parametrize("prec,layers", [("fp32", 4), ("bf16", 5), ("fp16", 4)])
def test_deep_stacks_reuse_backward_scratch_behind_waits(mods, tmp_path, prec, layers):
    """Num_Layers (reference Hyper_Parameters.yaml:16) above 3: the backward's scratch sets alias between layers (Layout: set
    l % 2, dQKV l % 3) and the main chain must WAIT for the weight-gradient stream's reader of a buffer before overwriting it
    (`sc.wait(g_set1 / g_dF / g_set2 / g_dQKV)` in backward_body).  A full train step against the oracle on a NaN-poisoned
    workspace, three ways: as shipped; with the weight-gradient stream held back 400 us after every fork
    (option debug_side_delay_us: a missing wait then overwrites an operand the weight gradient has not read yet) and one
    wgrad_ks block per CU (wgrad_ks_blocks = 256); and serialised on one stream (no_overlap).  All three must agree
    with each other to fp32 summation order and with the oracle within the mode's bound."""
    import subprocess, sys as _sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n, t = 24, 128
    scale = 1024.0 if prec == "fp16" else 1.0
    res = {}
    dev = {"GE2E_DEV_SWITCHES": "1"}        # the loader forwards GE2E_<OPTION> to ge2e_set_option only with this set
    for tag, extra in (("shipped", {}), ("delayed", dict(dev, GE2E_DEBUG_SIDE_DELAY_US="400", GE2E_WGRAD_KS_BLOCKS="256")),
                       ("serial", dict(dev, GE2E_NO_OVERLAP="1"))):
        out = str(tmp_path / f"{tag}.npz")
        code = _DEPTH_SCRIPT.format(repo=repo, tests=os.path.join(repo, "tests"), out=out, prec=prec, layers=layers, n=n, t=t, scale=scale)
        r = subprocess.run([_sys.executable, "-c", code], env=dict(os.environ, **extra), cwd=repo, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = dict(np.load(out))
    params = O.formula_params(layers=layers)
    from speaker_embedding_torch_amd.Modules import Positional_Encoding
    pe = Positional_Encoding(1024, 256, 0.1).pe[0].t().contiguous().numpy()
    x_np = O.formula_mel(41, n, 80, t, logmel=True)
    emb_ref, c = O.encoder_forward(params, x_np, train=True, seed=1234, step=0, p_pe=0.1, p_tf=0.1, pe=pe)
    loss_ref, lc = O.loss_forward(emb_ref, 4)
    grads_ref = O.encoder_backward(params, c, O.loss_backward(lc))
    base = res["serial"]
    for tag in ("shipped", "delayed"):
        a = res[tag]
        for rep in (0, 1):
            assert np.array_equal(a[f"emb{rep}"], base["emb0"]), tag                # forward: no atomics, bitwise
            for k in grads_ref:
                g, r = a[f"g{rep}_{k}"], base["g0_" + k]
                assert np.isfinite(g).all(), (tag, k)
                assert rel_l2(g, r) < 1e-4, (tag, rep, k, rel_l2(g, r))               # same kernels, fp32 sums in another order
    e = base["emb0"]
    if prec == "fp32":
        assert rel_l2(e, emb_ref) < 1e-4 and abs(float(base["loss0"]) - float(loss_ref)) < 1e-5
        tol = 2e-3 if _relu_margin_ok(c) else 0.2
        for k in grads_ref:
            assert rel_l2(base["g0_" + k], grads_ref[k]) < tol, k
    else:
        # 16-bit storage error grows with depth (profiles/r02_precision_taps.md: ~0.4 roundoffs per stored tensor): bounds = the
        # 3-layer bounds x 2 for bf16; float16 keeps its own
        etol, gtol, ctol = (4e-2, 0.3, 0.97) if prec == "bf16" else (5e-3, 8e-2, 0.997)
        assert rel_l2(e, emb_ref) < etol
        for k in grads_ref:
            g, r = base["g0_" + k].ravel().astype(np.float64), grads_ref[k].ravel().astype(np.float64)
            if g.size == 1:
                continue
            cos = float(g @ r / max(np.linalg.norm(g) * np.linalg.norm(r), 1e-30))
            assert cos > ctol and rel_l2(g, r) < gtol, (k, cos, rel_l2(g, r))


def test_multislice_inference_fp32_and_bf16(mods):
    """config 4 shape family: `samples` overlapping slices averaged BEFORE projection (Modules.py:55)."""
    GE2E, _ = mods
    x_np = O.formula_mel(8, 5 * 8, 80, 64, logmel=True)
    for prec, tol in (("fp32", 1e-4), ("bf16", 2e-2), ("fp16", 3e-3)):
        m, params, pe = build(GE2E, prec, 0.1)
        m.eval()
        ref, _ = O.encoder_forward(params, x_np, samples=5, pe=pe)
        with torch.no_grad():
            e = m(torch.from_numpy(x_np).cuda(), 5).cpu().numpy()
        assert e.shape == (8, 256) and rel_l2(e, ref) < tol


def test_eval_forward_reuses_prepared_weights_only_while_they_are_valid(mods):
    """Inference loop over one checkpoint: the second eval forward skips the weight preparation (GE2E_FWD_PREPARED) and returns
    the same bits; an in-place parameter update (or another shape) invalidates the prepared copies."""
    GE2E, _ = mods
    m, _, _ = build(GE2E, "bf16", 0.1)
    m._poison = False                       # (the NaN-poisoned test workspace would wipe the prepared copies on every call)
    m.eval()
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.randn(40, 80, 64, device="cuda", generator=g) * 2 - 5).clamp_(-11.5129, 2.0)
    calls = []
    hnd = m._handle()
    orig = hnd.encoder_forward
    hnd.encoder_forward = lambda *a, **k: (calls.append(k.get("prepared", False)), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            e1 = m(x, 5); e2 = m(x, 5)
            assert calls == [False, True] and torch.equal(e1, e2)
            m.prenet.weight.mul_(1.5)                                   # in place: same storage, new version
            e3 = m(x, 5); e4 = m(x, 5)
            assert calls == [False, True, False, True] and torch.equal(e3, e4) and (e3 - e1).abs().max() > 1e-3
            m(x[:20].contiguous(), 5)                                   # another shape: another layout
            assert calls[-1] is False
    finally:
        hnd.encoder_forward = orig
    m2, _, _ = build(GE2E, "bf16", 0.1)
    m2.eval()
    with torch.no_grad():
        m2.prenet.weight.mul_(1.5)
        assert torch.equal(m2(x, 5), e3)


def test_eval_after_fused_optimizer_step_sees_the_new_weights(mods, tmp_path):
    """The fused optimizer updates the parameters through raw pointers inside libge2e_hip.so.  An eval forward after Train_Steps
    (Trainer.Evaluation_Step on a fixed validation batch: same shape, workspace, stream) must not reuse weight copies prepared
    before the update: eval(x) -> Train_Step x 2 -> eval(x) == a fresh model loaded with the updated state_dict."""
    from speaker_embedding_torch_amd.Optim import FusedClipAdamW
    GE2E, GE2E_Loss = mods
    m, _, _ = build(GE2E, "bf16", 0.1)
    m._poison = False
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.randn(40, 80, 64, device="cuda", generator=g) * 2 - 5).clamp_(-11.5129, 2.0)
    opt = FusedClipAdamW(m.parameters(), lr=1e-2, max_norm=1.0)
    crit = GE2E_Loss().cuda()
    calls = []
    hnd = m._handle()
    orig = hnd.encoder_forward
    hnd.encoder_forward = lambda *a, **k: (calls.append(k.get("prepared", False)), orig(*a, **k))[1]
    try:
        m.eval()
        with torch.no_grad():
            e_before = m(x); m(x)
        assert calls == [False, True]
        versions = [p._version for p in m.parameters()]
        m.train()
        for _ in range(2):
            loss = crit(m(x), 5); opt.zero_grad(); loss.backward(); opt.step()
        assert all(p._version > v for p, v in zip(m.parameters(), versions))    # the raw-pointer update is visible to autograd
        m.eval()
        with torch.no_grad():
            e_after = m(x)
        assert calls[-1] is False                                               # re-prepared
    finally:
        hnd.encoder_forward = orig
    assert (e_after - e_before).abs().max() > 1e-3
    fresh, _, _ = build(GE2E, "bf16", 0.1)
    fresh.load_state_dict(m.state_dict())
    fresh.eval()
    with torch.no_grad():
        assert torch.equal(fresh(x), e_after)


def test_full_size_multislice_inference(mods, golden):
    """BASELINE.json configs[3] at full size: Inference.py's multi-slice d-vector extraction, 256 utterances x 5 slices x 64
    frames, embed only (reference Inference.py:157-159: model(mels, Samples) in eval / no_grad; Modules.py:55 averages the slices
    BEFORE the projection).  Size-independent properties + an oracle check of the first 16 utterances (80 slices), bf16 / fp16 /
    fp32; the golden G5 vector (the reference's own output for 4 x 5 slices) sits in the same batch family and is checked in
    test_fp32_eval_matches_reference_goldens."""
    GE2E, _ = mods
    U, S5, T = 256, 5, 64
    g = torch.Generator(device="cuda").manual_seed(17)
    x = (torch.randn(U * S5, 80, T, device="cuda", generator=g) * 2 - 5).clamp_(-11.5129, 2.0)
    for prec, tol in (("bf16", 2e-2), ("fp16", 3e-3), ("fp32", 1e-4)):
        m, params, pe = build(GE2E, prec, 0.1)
        m.eval()
        with torch.no_grad():
            e = m(x, S5)
            e2 = m(x, S5)
            perm = torch.randperm(U, device="cuda", generator=g)
            xp = x.view(U, S5, 80, T)[perm].reshape(U * S5, 80, T).contiguous()
            ep = m(xp, S5)
            e_sub = m(x[: 16 * S5].contiguous(), S5)
        assert e.shape == (U, 256) and torch.isfinite(e).all()
        assert (e.norm(dim=1) - 1).abs().max() < 1e-5
        assert torch.equal(e, e2)                                              # no atomics in the forward: bitwise
        assert (ep - e[perm]).abs().max() < 1e-6                               # an utterance's d-vector does not depend on its place
        assert (e_sub - e[:16]).abs().max() < 1e-6                             # ... nor on the batch size
        ref, _ = O.encoder_forward(params, x[: 16 * S5].cpu().numpy(), samples=S5, pe=pe)
        assert rel_l2(e[:16].cpu().numpy(), ref) < tol, prec


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_ffn_block_shapes_give_the_same_rows(mods, prec):
    """The chained FFN kernel has two block shapes (ffn.cuh: one 8-wave block or two 4-wave blocks per CU) and the launcher picks one
    by the number of rows; both accumulate in the same order, so an utterance's d-vector must not change BITWISE with the batch it
    sits in: 400 x 160 rows (250 passes of 256 rows fill 256 CUs: the 8-wave shape) against the first 40 alone (the 4-wave shape)."""
    GE2E, _ = mods
    m, _, _ = build(GE2E, prec, 0.1)
    m.eval()
    g = torch.Generator(device="cuda").manual_seed(11)
    x = (torch.randn(400, 80, 160, device="cuda", generator=g) * 2 - 5).clamp_(-11.5129, 2.0)
    with torch.no_grad():
        e_big = m(x)
        e_small = m(x[:40].contiguous())
    assert torch.equal(e_big[:40], e_small)


def test_profile_classes_and_serialised_backward(mods):
    """ge2e_profile_enable / ge2e_profile_read (bench.py's roofline leg): a class returns its launches, time, algorithmic FLOPs and
    bytes and is reset by the read; with GE2E_K_SERIAL the backward keeps the weight gradients on the caller's stream -- the same
    gradients (fp32 sums reorder only) from kernels timed alone."""
    from speaker_embedding_torch_amd import _lib
    GE2E, GE2E_Loss = mods
    m, _, _ = build(GE2E, "bf16", 0.1)
    m.train()
    crit = GE2E_Loss().cuda()
    g = torch.Generator(device="cuda").manual_seed(3)
    x = (torch.randn(40, 80, 160, device="cuda", generator=g) * 2 - 5).clamp_(-11.5129, 2.0)
    hnd = m._handle()
    grads = []
    for mask in (_lib.K_WGRAD, _lib.K_WGRAD | _lib.K_SERIAL):
        m.zero_grad(); m._step = 0
        hnd.profile_enable(mask)
        crit(m(x), 5).backward()
        torch.cuda.synchronize()
        hnd.profile_enable(0)
        ms, flops, nbytes, launches = hnd.profile_read(_lib.K_WGRAD)
        assert launches >= 12 and ms > 0 and flops > 0 and nbytes > 0      # 4 per full layer, k|v + q + 3 compact of the last, prenet
        assert hnd.profile_read(_lib.K_WGRAD)[3] == 0                      # the read reset the class
        grads.append(torch.cat([p_.grad.flatten() for p_ in m.parameters()]).clone())
    assert ((grads[0] - grads[1]).norm() / grads[0].norm()).item() < 1e-3


# ------------------------------------------------------------------------------------------ full size properties
@pytest.mark.parametrize("prec,S,P,T", [("bf16", 64, 15, 160), ("fp32", 64, 15, 160),
                                        ("fp16", 256, 10, 180), ("bf16", 256, 10, 180)])
def test_full_size_properties(mods, prec, S, P, T):
    """BASELINE.json's full sizes -- configs[1] 64 spk x 15 utt x 160 frames and configs[4] 256 spk x 10 utt x 180 frames
    (per GPU; float16 as stated there, and bf16) -- through size-independent properties: unit norm, finiteness, determinism,
    batch-independence in eval mode (an utterance's d-vector must not depend on its neighbours), an oracle spot check of
    8 utterances, a full step with finite gradients, and the dropout stream following the step counter."""
    GE2E, GE2E_Loss = mods
    m, params, pe = build(GE2E, prec, 0.1)
    N = S * P
    g = torch.Generator(device="cuda").manual_seed(7)
    x = (torch.randn(N, 80, T, device="cuda", generator=g) * 2 - 5).clamp_(-11.5129, 2.0)
    m.eval()
    with torch.no_grad():
        e_full = m(x)
        e_again = m(x)
        perm = torch.randperm(N, device="cuda", generator=g)
        e_perm = m(x[perm].contiguous())
        e_small = m(x[100:120].contiguous())
    assert torch.isfinite(e_full).all()
    assert (e_full.norm(dim=1) - 1).abs().max() < 1e-5
    assert torch.equal(e_full, e_again)                                # forward has no atomics: bitwise
    assert (e_perm - e_full[perm]).abs().max() < 1e-6
    assert (e_small - e_full[100:120]).abs().max() < 1e-6
    # spot-check 8 utterances of the big batch against the oracle
    ref, _ = O.encoder_forward(params, x[:8].cpu().numpy(), pe=pe)
    assert rel_l2(e_full[:8].cpu().numpy(), ref) < {"fp32": 1e-4, "bf16": 2e-2, "fp16": 3e-3}[prec]
    # one full training step: finite gradients, loss near the value of the eval embeddings' loss
    m.train()
    crit = GE2E_Loss().cuda()
    ls = 1024.0 if prec == "fp16" else 1.0                             # float16 gradients are taken under a loss scale
    emb1 = m(x); loss1 = crit(emb1, P); (loss1 * ls).backward()
    g1 = torch.cat([p.grad.flatten() for p in m.parameters()])
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    # the S-wide softmax loss of the step (S = 256 for configs[4]) against the oracle's loss on the same d-vectors
    loss_ref, _ = O.loss_forward(emb1.detach().cpu().numpy(), P)
    assert abs(loss1.item() - float(loss_ref)) < 2e-5 * max(1.0, float(loss_ref))
    emb2 = m(x)                                                        # next step counter -> other masks
    assert (emb1 - emb2).abs().max() > 1e-4
    m.zero_grad()
    m._step = 0
    emb3 = m(x); (crit(emb3, P) * ls).backward()                       # same (seed, step) -> same masks
    assert torch.equal(emb3, emb1)
    g3 = torch.cat([p.grad.flatten() for p in m.parameters()])
    assert ((g3 - g1).norm() / g1.norm()).item() < 1e-3                # atomics reorder fp32 sums only


# ------------------------------------------------------------------------------------------ error behaviour
def test_error_behaviour(mods):
    GE2E, GE2E_Loss = mods
    m, _, _ = build(GE2E, "fp32", 0.1)
    with pytest.raises(RuntimeError, match="frames > 1024"):
        m(torch.zeros(2, 80, 1030, device="cuda"))
    with pytest.raises(RuntimeError, match="Mel_dim"):
        m(torch.zeros(2, 64, 32, device="cuda"))
    with pytest.raises(RuntimeError, match="multiple of samples"):
        m(torch.zeros(3, 80, 32, device="cuda"), 2)
    with pytest.raises(RuntimeError, match="pattern_per_speaker"):
        GE2E_Loss().cuda()(torch.zeros(5, 256, device="cuda"), 2)
    m.eval()
    e = m(torch.zeros(2, 80, 32, device="cuda").requires_grad_(False))
    with pytest.raises(RuntimeError):
        e.sum().backward()      # eval forward keeps no activations / params need grad -> explicit error


# ------------------------------------------------------------------------------------------ fp16 mel input (row f2)
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_fp16_mel_input_is_bitwise_the_widened_input(mods, prec):
    """ge2e_encoder_forward_mel16: patterns are fp16 on disk (Pattern_Generator.py:191-198); feeding the fp16 batch and
    feeding its float32 widening (what Datasets.py:84 does on the host) must give the same bits, forward and backward -- AND both must
    be what the oracle computes from the widened batch (same dropout stream): d-vectors within the mode's bound (fp32: the
    north_star 1e-4), every parameter gradient within the mode's gradient bound."""
    GE2E, GE2E_Loss = mods
    p, P = 0.1, 3
    x16 = torch.from_numpy(O.formula_mel(3, 12, 80, 77, logmel=True)).half().cuda()
    outs = []
    for x in (x16, x16.float()):
        m, params, pe = build(GE2E, prec, p)
        m.train()
        emb = m(x)
        GE2E_Loss().cuda()(emb, P).backward()
        outs.append((emb.detach().clone(), {n_: q.grad.clone() for n_, q in m.named_parameters()}))
    assert torch.equal(outs[0][0], outs[1][0])
    # weight gradients use fp32 atomics (summation order varies run to run): equal to rounding, not bitwise
    for k in outs[0][1]:
        assert rel_l2(outs[0][1][k].cpu().numpy(), outs[1][1][k].cpu().numpy()) < 1e-5, k
    # against the oracle on the widened values (reference Datasets.py:84: the collater's FloatTensor of the fp16 patterns)
    x_np = x16.float().cpu().numpy()
    emb_ref, c = O.encoder_forward(params, x_np, train=True, seed=1234, step=0, p_pe=p, p_tf=p, pe=pe)
    _, lc = O.loss_forward(emb_ref, P)
    grads_ref = O.encoder_backward(params, c, O.loss_backward(lc))
    e = outs[0][0].cpu().numpy()
    if prec == "fp32":
        assert np.abs(e - emb_ref).max() < 1e-5 and rel_l2(e, emb_ref) < 1e-4
        tol = 2e-3 if _relu_margin_ok(c) else 0.2
        for k, g in outs[0][1].items():
            assert rel_l2(g.cpu().numpy(), grads_ref[k]) < tol, k
    else:
        assert np.abs(e - emb_ref).max() < 6e-3 and rel_l2(e, emb_ref) < 2e-2
        for k, g in outs[0][1].items():
            g, r = g.cpu().numpy().ravel().astype(np.float64), grads_ref[k].ravel().astype(np.float64)
            if g.size == 1:
                assert abs(g[0] - r[0]) < 0.3 * np.linalg.norm(grads_ref["prenet.bias"]), k
                continue
            cos = float(g @ r / max(np.linalg.norm(g) * np.linalg.norm(r), 1e-30))
            assert cos > 0.99 and rel_l2(g, r) < 0.15, (k, cos, rel_l2(g, r))


def test_device_prefetcher_order_and_values(mods):
    from speaker_embedding_torch_amd.Datasets import DevicePrefetcher
    batches = [(torch.full((4, 80, 32), float(i), dtype=torch.float16).pin_memory(), [f"s{i}"]) for i in range(5)]
    got = list(DevicePrefetcher(batches, "cuda"))
    assert len(got) == 5
    for i, (feat, names) in enumerate(got):
        assert feat.is_cuda and feat.dtype == torch.float16 and names == [f"s{i}"]
        assert torch.equal(feat.cpu(), batches[i][0])


# ------------------------------------------------------------------------------------------ kernel-path agreement
_PATH_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, {repo!r}); sys.path.insert(0, {tests!r})
import test_gpu_parity as tp
from oracle import ge2e_oracle as O
from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss
m, params, pe = tp.build(GE2E, "bf16", 0.1)
m.train()
x = torch.from_numpy(O.formula_mel(11, 12, 80, 77, logmel=True)).cuda()
emb = m(x); loss = GE2E_Loss().cuda()(emb, 3); loss.backward()
out = {{"emb": emb.detach().cpu().numpy(), "loss": np.float32(loss.item())}}
for tap, w in (("qkv.1", 768), ("f.1", 1024), ("h2.1", 256), ("h1.0", 256)):
    out["tap_" + tap] = m.workspace_view(tap, 12, 77, True).float().cpu().numpy()
for k, p in m.named_parameters():
    out["g_" + k] = p.grad.cpu().numpy()
np.savez({out!r}, **out)
"""


def test_bf16_streaming_and_tiled_kernels_agree(mods, tmp_path):
    """The bf16 mode runs the K=256 products on gemm_ws_kernel and FFN2+LN on gemm_kl_kernel; the options no_ws_gemm /
    no_kl_gemm put the same step on the tiled gemm_nt_kernel (the kernels the fp32 parity mode uses).  Same
    dropout stream, same inputs, ragged rows (12 x 77 = 924: not a multiple of the 16- and 128-row tiles): the
    projections are bit-identical between the two, the LayerNorm statistics are summed in a different order."""
    import subprocess, sys as _sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, extra in (("stream", {}), ("tiled", {"GE2E_DEV_SWITCHES": "1", "GE2E_NO_WS_GEMM": "1", "GE2E_NO_KL_GEMM": "1"})):
        out = str(tmp_path / f"{tag}.npz")
        env = dict(os.environ, **extra)
        code = _PATH_SCRIPT.format(repo=repo, tests=os.path.join(repo, "tests"), out=out)
        r = subprocess.run([_sys.executable, "-c", code], env=env, cwd=repo, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = dict(np.load(out))
    a, b = res["stream"], res["tiled"]
    # first product of the path: in_proj of layer 1 sees inputs that already went through one LayerNorm pair
    assert np.array_equal(res["stream"]["tap_h1.0"].shape, res["tiled"]["tap_h1.0"].shape)
    assert rel_l2(a["tap_h1.0"], b["tap_h1.0"]) < 2e-3        # first LayerNorm output: bf16 rounding flips only
    for k in ("tap_qkv.1", "tap_f.1", "tap_h2.1"):
        assert rel_l2(a[k], b[k]) < 6e-3, k
    assert np.abs(a["emb"] - b["emb"]).max() < 3e-3 and abs(float(a["loss"]) - float(b["loss"])) < 5e-3
    for k in a:
        if k.startswith("g_") and a[k].size > 1:
            g, r = a[k].ravel().astype(np.float64), b[k].ravel().astype(np.float64)
            cos = float(g @ r / max(np.linalg.norm(g) * np.linalg.norm(r), 1e-30))
            assert cos > 0.995, (k, cos)
