"""1-layer backward-scratch taps vs oracle (diagnostic)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from argparse import Namespace
from oracle import ge2e_oracle as O
from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss

def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

def run(precision, n, t, P, p, layers=1, tag=3, stop=None):
    if stop is not None:
        os.environ["GE2E_DEBUG_BWD_STOP"] = str(stop)
    else:
        os.environ.pop("GE2E_DEBUG_BWD_STOP", None)
    lt = layers - 1 if stop is None else layers - stop
    print(f"--- {precision} n={n} t={t} P={P} p={p} layers={layers}")
    hp = Namespace(Sound=Namespace(Mel_Dim=80), GE2E=Namespace(Embedding_Size=256,
         Positional_Encoding=Namespace(Max_Position=1024, Dropout_Rate=p),
         Transformer=Namespace(Num_Layers=layers, Head=4, Dropout_Rate=p)))
    m = GE2E(hp, precision=precision, seed=1234).cuda()
    params = O.formula_params(layers=layers)
    sd = m.state_dict()
    for k, v in params.items():
        sd[k].copy_(torch.from_numpy(v))
    m.train()
    x_np = O.formula_mel(tag, n, 80, t)
    emb_ref, c = O.encoder_forward(params, x_np, train=True, seed=1234, step=0, p_pe=p, p_tf=p)
    emb = m(torch.from_numpy(x_np).cuda())
    loss = GE2E_Loss().cuda()(emb, P)
    loss.backward(); torch.cuda.synchronize()
    loss_ref, lc = O.loss_forward(emb_ref, P)
    bt = {}
    gref = O.encoder_backward(params, c, O.loss_backward(lc), taps=bt)
    for dev, ora, w in [("dF", f"dF{lt}", 1024), ("dHb", f"dHb{lt}", 256), ("dP", f"dP{lt}", 256), ("dM", f"dM{lt}", 256),
                        ("dO", f"dO{lt}", 256), ("dQKV", f"dQKV{lt}", 768)]:
        got = m.workspace_view(dev, n, t, True).float().cpu().numpy().reshape(n, t, w)
        ref = bt[ora]
        err = np.abs(got - ref)
        bad = np.argwhere(err > 1e-3 * np.abs(ref).max() + 1e-30)
        print(f"  {dev:5s} rel={rel(got, ref):.3e} maxabs={err.max():.3e} |ref|max={np.abs(ref).max():.3e} nbad={len(bad)}")
        if len(bad):
            rows = sorted(set((int(b[0]), int(b[1])) for b in bad))
            print("     bad (n,t):", rows[:20], "cols:", sorted(set(int(b[2]) for b in bad))[:24])
    for name, prm in m.named_parameters():
        print(f"  grad {name:45s} rel={rel(prm.grad.cpu().numpy(), gref[name]):.3e}")

if __name__ == "__main__":
    run("fp32", 6, 77, 3, 0.1, layers=3, stop=1)
    run("fp32", 6, 77, 3, 0.1, layers=2, stop=1)
    run("fp32", 6, 77, 3, 0.0, layers=3, stop=1)
