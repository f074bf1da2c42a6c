"""Pin the CPU oracle (oracle/ge2e_oracle.py) against outputs of the reference's own Modules.py.

The reference ships no tests/fixtures; the goldens were produced by importing
/root/reference/Modules.py in the build container (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle import ge2e_oracle as O
from conftest import rel_l2

S, P, T = 4, 5, 160


@pytest.fixture(scope="module")
def cfg1():
    params = O.formula_params()
    x = O.formula_mel(1, S * P, 80, T)
    return params, x


def test_param_table_matches_reference_count():
    specs = O.param_specs()
    assert len(specs) == 43
    assert sum(int(np.prod(s)) for _, s in specs) == 2456321


def test_G1_G2_eval_embeddings_and_loss(cfg1, golden):
    params, x = cfg1
    taps = {}
    emb, _ = O.encoder_forward(params, x, train=False, taps=taps)
    assert np.abs(emb - golden["G1_emb"]).max() < 2e-6
    assert rel_l2(emb, golden["G1_emb"]) < 1e-5
    loss, _ = O.loss_forward(emb, P)
    assert abs(float(loss) - float(golden["G2_loss"][0])) < 2e-6
    # G7 taps localise bugs layer by layer
    sel_n, sel_t = [0, 7, 19], [0, 1, 79, 159]
    for k in ("prenet_pe", "layer0", "layer1", "layer2"):
        got = taps[k]
        assert np.abs(got[np.ix_(sel_n, sel_t)] - golden[f"G7_{k}_slice"]).max() < 2e-5, k
        rn = np.sqrt((got.astype(np.float64) ** 2).sum(-1))
        assert np.abs(rn - golden[f"G7_{k}_rownorm"]).max() / golden[f"G7_{k}_rownorm"].max() < 1e-5, k
    assert np.abs(taps["final_ln_t0"] - golden["G7_final_ln_t0"]).max() < 2e-5


def test_G3_gradients_dropout0(cfg1, golden):
    params, x = cfg1
    emb, c = O.encoder_forward(params, x, train=True, p_pe=0.0, p_tf=0.0)
    loss, lc = O.loss_forward(emb, P)
    assert abs(float(loss) - float(golden["G3_loss_train"][0])) < 2e-6
    grads = O.encoder_backward(params, c, O.loss_backward(lc))
    names = [n for n, _ in O.param_specs()]
    for i, n in enumerate(names):
        g = grads[n]
        ref_norm = golden["G3_grad_norm"][i]
        got_norm = float(np.linalg.norm(g.astype(np.float64)))
        assert abs(got_norm - ref_norm) <= 2e-4 * ref_norm + 1e-9, (n, got_norm, ref_norm)
        k = min(8, g.size)
        assert np.abs(g.reshape(-1)[:k] - golden["G3_grad_head"][i][:k]).max() <= 2e-4 * max(ref_norm, 1e-6), n
    assert rel_l2(grads["prenet.weight"], golden["G3_grad_prenet_w"]) < 1e-4
    assert rel_l2(grads["transformer.layers.1.self_attn.in_proj_bias"], golden["G3_grad_l1_inproj_b"]) < 1e-4
    assert rel_l2(grads["transformer.layers.2.norm2.weight"], golden["G3_grad_l2_norm2_w"]) < 1e-4


def test_G4_clip_adamw_two_steps(cfg1, golden):
    params, x = cfg1
    params = {k: v.copy() for k, v in params.items()}
    st = {}
    loss, gnorm, _ = O.train_step(params, x, P, st, p_pe=0.0, p_tf=0.0)
    assert abs(gnorm - golden["G4_total_grad_norm"][0]) < 2e-4 * golden["G4_total_grad_norm"][0]
    names = [n for n, _ in O.param_specs()]
    for i, n in enumerate(names):
        s = float(params[n].astype(np.float64).sum())
        assert abs(s - golden["G4_param_sum"][i]) < 1e-5 * max(1.0, abs(golden["G4_param_sum"][i])) + 2e-4, n
        k = min(8, params[n].size)
        assert np.abs(params[n].reshape(-1)[:k] - golden["G4_param_head"][i][:k]).max() < 2e-6, n
    loss2, _, _ = O.train_step(params, x, P, st, p_pe=0.0, p_tf=0.0)
    assert abs(loss2 - float(golden["G4_loss_step2"][0])) < 5e-6
    for i, n in enumerate(names):
        s = float(params[n].astype(np.float64).sum())
        assert abs(s - golden["G4_param_sum_step2"][i]) < 1e-5 * max(1.0, abs(golden["G4_param_sum_step2"][i])) + 4e-4, n


def test_G5_multislice_and_odd_T(golden):
    params = O.formula_params()
    xs = O.formula_mel(2, 20, 80, 64, logmel=True)
    emb, _ = O.encoder_forward(params, xs, samples=5)
    assert emb.shape == (4, 256)
    assert np.abs(emb - golden["G5_emb_samples5"]).max() < 2e-6
    xo = O.formula_mel(3, 6, 80, 77, logmel=True)
    emb, _ = O.encoder_forward(params, xo)
    assert np.abs(emb - golden["G5_emb_T77"]).max() < 2e-6
    loss, _ = O.loss_forward(emb, 3)
    assert abs(float(loss) - float(golden["G5_loss_T77"][0])) < 2e-6


@pytest.mark.parametrize("tag,s,p", [(0, 4, 5), (1, 64, 15), (2, 256, 10)])
def test_G6_loss_only(golden, tag, s, p):
    e = O.formula_normal(50 + tag, (s * p, 256))
    e = e + 2.0 * np.repeat(O.formula_normal(60 + tag, (s, 256)), p, axis=0)
    e = (e / np.linalg.norm(e, axis=1, keepdims=True)).astype(np.float32)
    loss, lc = O.loss_forward(e, p)
    assert abs(float(loss) - float(golden[f"G6_loss_{s}x{p}"][0])) < 5e-6
    g = O.loss_backward(lc)
    ref_norm = float(golden[f"G6_demb_norm_{s}x{p}"][0])
    assert abs(np.linalg.norm(g.astype(np.float64)) - ref_norm) < 1e-4 * ref_norm
    # unit-norm inputs make d_emb a difference of nearly equal terms: fp32 cancellation in the
    # reference itself limits agreement to ~1e-3 of the largest component
    assert np.abs(g[:4] - golden[f"G6_demb_head_{s}x{p}"]).max() < 2e-3 * np.abs(golden[f"G6_demb_head_{s}x{p}"]).max()
    # G8: gradients of the criterion's own weight / bias (reference Modules.py:115-116, filled by its autograd)
    _, dw, db = O.loss_backward(lc, with_wb=True)
    dw_ref, db_ref = golden[f"G8_dw_db_{s}x{p}"]
    assert abs(dw - dw_ref) < 1e-4 * abs(dw_ref) + 1e-8 and abs(db - db_ref) < 1e-6


def test_G6_loss_unnormalised(golden):
    e = O.formula_normal(70, (12, 256)).astype(np.float32) * np.float32(0.3)
    loss, lc = O.loss_forward(e, 4)
    assert abs(float(loss) - float(golden["G6_loss_unnorm_3x4"][0])) < 5e-6
    assert rel_l2(O.loss_backward(lc), golden["G6_demb_unnorm_3x4"]) < 1e-4


def test_dropout_hash_statistics_and_determinism():
    key = O.drop_key(1234, 7, O.site_attn(1))
    k1 = O.drop_keep(key, 1 << 20, 0.1)
    k2 = O.drop_keep(key, 1 << 20, 0.1)
    assert (k1 == k2).all()
    assert abs(k1.mean() - 0.9) < 2e-3
    other = O.drop_keep(O.drop_key(1234, 8, O.site_attn(1)), 1 << 20, 0.1)
    assert abs((k1 == other).mean() - 0.82) < 5e-3     # independent masks agree with prob .81+.01
    # windows of the index space agree with the full stream (kernels hash arbitrary offsets)
    assert (O.drop_keep(key, 1000, 0.1, start=5000) == k1[5000:6000]).all()


def test_train_mode_dropout_expectation():
    """inverted dropout keeps E[h]: averaged over seeds the train-mode tap approaches eval."""
    params = O.formula_params()
    x = O.formula_mel(5, 2, 80, 32)
    ev = {}
    O.encoder_forward(params, x, train=False, taps=ev)
    acc = np.zeros_like(ev["prenet_pe"], dtype=np.float64)
    K = 64
    for s in range(K):
        tp = {}
        O.encoder_forward(params, x, train=True, seed=s, taps=tp)
        acc += tp["prenet_pe"]
    err = np.abs(acc / K - ev["prenet_pe"]).mean() / np.abs(ev["prenet_pe"]).mean()
    assert err < 0.05


def test_backward_matches_finite_difference_fp64():
    """independent check of encoder_backward incl. dropout masks (float64 central differences)."""
    params = {k: v.astype(np.float64) for k, v in O.formula_params(mel=8, d=16, layers=2, ffn=32).items()}
    x = O.formula_mel(9, 6, 8, 12).astype(np.float64)
    kw = dict(samples=1, heads=2, train=True, seed=3, step=2, p_pe=0.1, p_tf=0.2)

    def f():
        emb, c = O.encoder_forward(params, x, **kw)
        loss, lc = O.loss_forward(emb, 3)
        return float(loss), c, lc

    loss, c, lc = f()
    grads = O.encoder_backward(params, c, O.loss_backward(lc))
    rng = np.random.default_rng(0)
    for name in params:
        p = params[name]
        for _ in range(3):
            idx = tuple(rng.integers(0, s) for s in p.shape)
            old = p[idx]
            h = 1e-6
            p[idx] = old + h; lp = f()[0]
            p[idx] = old - h; lm = f()[0]
            p[idx] = old
            fd = (lp - lm) / (2 * h)
            assert abs(fd - grads[name][idx]) < 1e-6 + 1e-4 * abs(fd), (name, idx, fd, grads[name][idx])


def test_torch_cpu_restatement_is_the_reference_computation(cfg1, golden):
    """bench.py's cpu_baseline times oracle/torch_restatement.py; pin it to the reference's goldens (eval d-vectors, loss)
    and to the numpy oracle's gradients, so the baseline number belongs to the right computation."""
    import torch
    from oracle import torch_restatement as TR
    params, x = cfg1
    m = TR.EncoderCPU(p_pe=0.0, p_tf=0.0)
    m.load_named(params)
    m.eval()
    with torch.no_grad():
        emb = m(torch.from_numpy(x))
        loss = TR.ge2e_loss_cpu(emb, P)
    assert np.abs(emb.numpy() - golden["G1_emb"]).max() < 2e-6
    assert abs(float(loss) - float(golden["G2_loss"][0])) < 2e-6
    m.train()                                    # dropout rates are 0: train mode differs only by being differentiable
    TR.ge2e_loss_cpu(m(torch.from_numpy(x)), P).backward()
    e_np, c = O.encoder_forward(params, x, train=False)
    _, lc = O.loss_forward(e_np, P)
    g_ref = O.encoder_backward(params, c, O.loss_backward(lc))
    named = dict(m.named_parameters())
    for ours, theirs in (("transformer.layers.1.linear1.weight", "transformer.layers.1.linear1.weight"),
                         ("transformer.layers.0.self_attn.in_proj_weight", "transformer.layers.0.self_attn.in_proj_weight"),
                         ("alpha", "positional_encoding.alpha")):
        assert rel_l2(named[ours].grad.numpy().reshape(-1), g_ref[theirs].reshape(-1)) < 2e-4, ours
    assert rel_l2(named["prenet.weight"].grad.numpy(), g_ref["prenet.weight"][:, :, 0]) < 2e-4
    ms = m(torch.from_numpy(O.formula_mel(3, 4 * 5, 80, 64)), samples=5)          # slice mean before the projection
    assert ms.shape == (4, 256)
