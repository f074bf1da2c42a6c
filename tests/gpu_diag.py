"""Stage-by-stage HIP-vs-oracle report (diagnostic aid; the pass/fail gates are the pytest files).

    python tests/gpu_diag.py [out.txt]
"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from argparse import Namespace  # noqa: E402

from oracle import ge2e_oracle as O  # noqa: E402
from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss  # noqa: E402

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout


def log(*a):
    print(*a, file=out, flush=True)
    if out is not sys.stdout:
        print(*a, flush=True)


def make_hp(p=0.1):
    return Namespace(Sound=Namespace(Mel_Dim=80),
                     GE2E=Namespace(Embedding_Size=256,
                                    Positional_Encoding=Namespace(Max_Position=1024, Dropout_Rate=p),
                                    Transformer=Namespace(Num_Layers=3, Head=4, Dropout_Rate=p)))


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def build(precision, p):
    m = GE2E(make_hp(p), precision=precision, seed=1234).cuda()
    params = O.formula_params()
    sd = m.state_dict()
    for k, v in params.items():
        sd[k].copy_(torch.from_numpy(v))
    return m, params


def run_case(precision, n, t, P, p, train, samples=1, tag=0):
    log(f"--- {precision} n={n} t={t} P={P} dropout={p} train={train} samples={samples}")
    m, params = build(precision, p)
    m.train(train)
    x_np = O.formula_mel(tag, n, 80, t, logmel=(tag % 2 == 0))
    x = torch.from_numpy(x_np).cuda()
    taps = {}
    pe = m.positional_encoding.pe[0].t().contiguous().cpu().numpy()
    emb_ref, c = O.encoder_forward(params, x_np, samples=samples, train=train, seed=1234, step=0, p_pe=p, p_tf=p, taps=taps, pe=pe)
    emb = m(x, samples)
    torch.cuda.synchronize()
    names = [("h0", "prenet_pe", 256)]
    for l in range(3):
        names += [(f"qkv.{l}", f"qkv{l}", 768), (f"o.{l}", f"o{l}", 256), (f"h1.{l}", f"h1_{l}", 256),
                  (f"f.{l}", f"f{l}", 1024), (f"h2.{l}", f"layer{l}", 256)]
    if train:       # eval mode aliases buffers, only train keeps every tap
        for dev_name, ora_name, width in names:
            got = m.workspace_view(dev_name, n, t, True).float().cpu().numpy()
            ref = taps[ora_name]
            if dev_name.endswith(".2") and not dev_name.startswith("qkv"):
                got, ref = got.reshape(n, width), ref[:, 0, :]      # last layer: frame 0 only (compact rows)
            elif dev_name == "qkv.2":
                got, ref = got.reshape(n, t, width)[:, :, 256:], ref[:, :, 256:]   # q exists for frame 0 only
            else:
                got = got.reshape(n, t, width)
            log(f"  tap {dev_name:7s} rel={rel(got, ref):.3e} maxabs={np.abs(got - ref).max():.3e} nan={np.isnan(got).sum()}")
    e = emb.detach().cpu().numpy()
    log(f"  emb rel={rel(e, emb_ref):.3e} maxabs={np.abs(e - emb_ref).max():.3e} nan={np.isnan(e).sum()}")
    if not train:
        return
    crit = GE2E_Loss().cuda()
    loss = crit(emb, P)
    loss_ref, lc = O.loss_forward(emb_ref, P)
    log(f"  loss hip={loss.item():.6f} oracle={float(loss_ref):.6f}")
    loss.backward()
    torch.cuda.synchronize()
    grads_ref = O.encoder_backward(params, c, O.loss_backward(lc))
    worst = 0.0
    for name, prm in m.named_parameters():
        g = prm.grad.detach().cpu().numpy()
        r = rel(g, grads_ref[name])
        worst = max(worst, r)
        log(f"  grad {name:50s} rel={r:.3e} |ref|={np.linalg.norm(grads_ref[name]):.3e} nan={np.isnan(g).sum()}")
    log(f"  worst grad rel = {worst:.3e}")


def loss_only():
    log("--- loss only")
    crit = GE2E_Loss().cuda()
    for tag, (s, p) in enumerate([(4, 5), (64, 15), (256, 10)]):
        e = O.formula_normal(50 + tag, (s * p, 256))
        e = e + 2.0 * np.repeat(O.formula_normal(60 + tag, (s, 256)), p, axis=0)
        e = (e / np.linalg.norm(e, axis=1, keepdims=True)).astype(np.float32)
        l_ref, lc = O.loss_forward(e, p)
        g_ref = O.loss_backward(lc)
        et = torch.from_numpy(e).cuda().requires_grad_(True)
        l = crit(et, p)
        l.backward()
        log(f"  {s}x{p}: loss hip={l.item():.6f} oracle={float(l_ref):.6f} d_emb rel={rel(et.grad.cpu().numpy(), g_ref):.3e}")


if __name__ == "__main__":
    t0 = time.time()
    log("device", torch.cuda.get_device_name(0))
    try:
        loss_only()
    except Exception as ex:  # keep going: the report is the point
        log("LOSS FAILED", repr(ex))
    cases = [("fp32", 20, 160, 5, 0.0, True, 1, 1), ("fp32", 20, 160, 5, 0.1, True, 1, 1),
             ("fp32", 6, 77, 3, 0.1, True, 1, 3), ("fp32", 20, 64, 2, 0.0, False, 5, 2),
             ("bf16", 20, 160, 5, 0.0, True, 1, 1), ("bf16", 20, 160, 5, 0.1, True, 1, 1),
             ("bf16", 6, 77, 3, 0.1, True, 1, 3), ("bf16", 20, 64, 2, 0.0, False, 5, 2),
             ("fp32", 8, 270, 4, 0.1, True, 1, 4), ("bf16", 8, 270, 4, 0.1, True, 1, 4)]
    for cs in cases:
        try:
            run_case(*cs[:6], samples=cs[6], tag=cs[7])
        except Exception as ex:
            log("CASE FAILED", cs, repr(ex))
    log("done in %.1fs" % (time.time() - t0))
