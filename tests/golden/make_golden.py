"""Generate golden vectors by running the REFERENCE's own Modules.py on CPU (Device '-1' path).

Run in the build container only (the reference tree does not travel to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

Weights and inputs are formula-defined (oracle.ge2e_oracle.formula_*), so the fixtures hold
OUTPUTS of the reference only: embeddings, loss, gradients (norm + head slice), post-AdamW
parameter checksums, intermediate taps, multi-slice inference embeddings and loss-only vectors.
Nothing from the reference's source text is stored.
"""
import os
import sys
from argparse import Namespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.environ.get("GE2E_REFERENCE", "/root/reference"))

from oracle import ge2e_oracle as O  # noqa: E402
import Modules as R  # noqa: E402  (the reference)

OUT = os.path.join(REPO, "tests", "golden")


def make_hp(dropout):
    return Namespace(
        Sound=Namespace(Mel_Dim=80),
        GE2E=Namespace(
            Embedding_Size=256,
            Positional_Encoding=Namespace(Max_Position=1024, Dropout_Rate=dropout),
            Transformer=Namespace(Num_Layers=3, Head=4, Dropout_Rate=dropout)))


def load_formula(model):
    params = O.formula_params()
    sd = model.state_dict()
    for k, v in params.items():
        assert tuple(sd[k].shape) == v.shape, k
        sd[k].copy_(torch.from_numpy(v))
    return params


def taps_of(model, x):
    """final output of prenet+PE, each encoder layer and final LN(t=0) via forward hooks."""
    taps = {}
    hs = [model.positional_encoding.register_forward_hook(
        lambda m, i, o: taps.__setitem__("prenet_pe", o.detach().permute(0, 2, 1).numpy().copy()))]
    for l, layer in enumerate(model.transformer.layers):
        hs.append(layer.register_forward_hook(
            lambda m, i, o, l=l: taps.__setitem__(f"layer{l}", o.detach().permute(1, 0, 2).numpy().copy())))
    hs.append(model.transformer.norm.register_forward_hook(
        lambda m, i, o: taps.__setitem__("final_ln_t0", o.detach()[0].numpy().copy())))
    with torch.no_grad():
        model(x)
    for h in hs:
        h.remove()
    return taps


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    # ---------------- cfg 1: 4 spk x 5 utt x 160 fr, eval & dropout-0 train ----------------
    S, P, T = 4, 5, 160
    x_np = O.formula_mel(1, S * P, 80, T)
    x = torch.from_numpy(x_np)
    model = R.GE2E(make_hp(0.0))
    load_formula(model)
    crit = R.GE2E_Loss()
    model.eval()
    with torch.no_grad():
        emb = model(x)
        loss = crit(emb, P)
    out["G1_emb"] = emb.numpy()
    out["G2_loss"] = np.array([loss.item()], np.float32)
    taps = taps_of(model, x)
    sel_n, sel_t = [0, 7, 19], [0, 1, 79, 159]
    for k, v in taps.items():
        if v.ndim == 3:
            out[f"G7_{k}_slice"] = v[np.ix_(sel_n, sel_t)].copy()
            out[f"G7_{k}_rownorm"] = np.sqrt((v.astype(np.float64) ** 2).sum(-1)).astype(np.float32)
        else:
            out[f"G7_{k}"] = v
    # G3: gradients with dropout 0 in train mode
    model.train()
    emb = model(x)
    loss = crit(emb, P)
    model.zero_grad()
    loss.backward()
    out["G3_loss_train"] = np.array([loss.item()], np.float32)
    names = [n for n, _ in model.named_parameters()]
    assert names == [n for n, _ in O.param_specs()], "parameter order differs from the oracle's table"
    out["G3_grad_norm"] = np.array([p.grad.double().norm().item() for p in model.parameters()], np.float64)
    out["G3_grad_head"] = np.stack([
        np.pad(p.grad.reshape(-1)[:8].numpy(), (0, max(0, 8 - p.numel()))) for p in model.parameters()])
    out["G3_grad_prenet_w"] = model.prenet.weight.grad.numpy().copy()
    out["G3_grad_l1_inproj_b"] = model.transformer.layers[1].self_attn.in_proj_bias.grad.numpy().copy()
    out["G3_grad_l2_norm2_w"] = model.transformer.layers[2].norm2.weight.grad.numpy().copy()
    # G4: clip 1.0 + AdamW exactly as Train.py:122-127,154-162
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-6)
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    out["G4_total_grad_norm"] = np.array([gn.item()], np.float64)
    out["G4_param_sum"] = np.array([p.detach().double().sum().item() for p in model.parameters()], np.float64)
    out["G4_param_head"] = np.stack([
        np.pad(p.detach().reshape(-1)[:8].numpy(), (0, max(0, 8 - p.numel()))) for p in model.parameters()])
    # second step to pin the Adam state update
    emb = model(x); loss = crit(emb, P); opt.zero_grad(); loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0); opt.step()
    out["G4_loss_step2"] = np.array([loss.item()], np.float32)
    out["G4_param_sum_step2"] = np.array([p.detach().double().sum().item() for p in model.parameters()], np.float64)

    # ---------------- G5: multi-slice inference [4*5, 80, 64], samples = 5 ----------------
    model = R.GE2E(make_hp(0.1)).eval()
    load_formula(model)
    xs = torch.from_numpy(O.formula_mel(2, 20, 80, 64, logmel=True))
    with torch.no_grad():
        out["G5_emb_samples5"] = model(xs, 5).numpy()
        # odd T (not a multiple of 16/32) in eval mode to pin padding/masking logic
        xo = torch.from_numpy(O.formula_mel(3, 6, 80, 77, logmel=True))
        out["G5_emb_T77"] = model(xo).numpy()
        out["G5_loss_T77"] = np.array([crit(model(xo), 3).item()], np.float32)

    # ---------------- G6: loss-only vectors on seeded unit-norm embeddings ----------------
    for tag, (s, p) in enumerate([(4, 5), (64, 15), (256, 10)]):
        e = O.formula_normal(50 + tag, (s * p, 256))
        # speaker structure so that the softmax is not uniform
        e = e + 2.0 * np.repeat(O.formula_normal(60 + tag, (s, 256)), p, axis=0)
        e = (e / np.linalg.norm(e, axis=1, keepdims=True)).astype(np.float32)
        et = torch.from_numpy(e).requires_grad_(True)
        crit.zero_grad()
        l = crit(et, p)
        l.backward()
        out[f"G6_loss_{s}x{p}"] = np.array([l.item()], np.float32)
        gr = et.grad.numpy()
        out[f"G6_demb_norm_{s}x{p}"] = np.array([np.linalg.norm(gr.astype(np.float64))])
        out[f"G6_demb_head_{s}x{p}"] = gr[:4].copy()
        # G8: the criterion's own parameters (Modules.py:115-116): autograd fills weight.grad / bias.grad although nothing optimises them
        out[f"G8_dw_db_{s}x{p}"] = np.array([crit.weight.grad.item(), crit.bias.grad.item()], np.float64)
    # un-normalised embeddings exercise the norm clamps / general cosine path
    e = O.formula_normal(70, (12, 256)).astype(np.float32) * 0.3
    et = torch.from_numpy(e).requires_grad_(True)
    crit.zero_grad()
    l = crit(et, 4); l.backward()
    out["G6_loss_unnorm_3x4"] = np.array([l.item()], np.float32)
    out["G6_demb_unnorm_3x4"] = et.grad.numpy().copy()
    out["G8_dw_db_unnorm_3x4"] = np.array([crit.weight.grad.item(), crit.bias.grad.item()], np.float64)

    np.savez_compressed(os.path.join(OUT, "ge2e_golden.npz"), **out)
    sz = os.path.getsize(os.path.join(OUT, "ge2e_golden.npz"))
    print("wrote", len(out), "arrays,", sz, "bytes; torch", torch.__version__)
    for k in sorted(out):
        print(f"  {k:32s} {out[k].shape}")


if __name__ == "__main__":
    main()
