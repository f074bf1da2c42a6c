"""Per-tensor gradient error of one train step against the oracle (development aid; profiles/rNN_grad_err.txt):
    python tools/grad_err.py <prec> <n> <t> <P> <tag> [dropout p = 0.1]"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as tp
from oracle import ge2e_oracle as O
from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss
prec, n, t, P, tag = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
p = float(sys.argv[6]) if len(sys.argv) > 6 else 0.1
m, params, pe = tp.build(GE2E, prec, p)
m.train()
x_np = O.formula_mel(tag, n, 80, t, logmel=True)
emb_ref, c = O.encoder_forward(params, x_np, train=True, seed=1234, step=0, p_pe=p, p_tf=p, pe=pe)
loss_ref, lc = O.loss_forward(emb_ref, P)
g_ref = O.encoder_backward(params, c, O.loss_backward(lc))
emb = m(torch.from_numpy(x_np).cuda()); loss = GE2E_Loss().cuda()(emb, P)
scale = 4096.0 if prec == "fp16" else 1.0
(loss * scale).backward()
print(f"== {prec} {n} x {t}, P {P}, tag {tag}, dropout {p}: d-vector rel {tp.rel_l2(emb.detach().cpu().numpy(), emb_ref):.3e}")
nb = np.linalg.norm(g_ref["prenet.bias"])
worst, lowcos = (0.0, ""), (1.0, "")
for name, prm in m.named_parameters():
    g, r = (prm.grad.cpu().numpy().ravel() / scale).astype(np.float64), g_ref[name].ravel().astype(np.float64)
    if g.size == 1:
        print(f"  {name:50s} abs {abs(g[0]-r[0]):.3e}  = {abs(g[0]-r[0])/nb:.4f} of |d prenet.bias|")
    else:
        rel = tp.rel_l2(g, r); cos = float(g @ r / max(np.linalg.norm(g) * np.linalg.norm(r), 1e-30))
        print(f"  {name:50s} rel {rel:.4f}  cos {cos:.5f}")
        if rel > worst[0]: worst = (rel, name)
        if cos < lowcos[0]: lowcos = (cos, name)
print(f"  worst rel {worst[0]:.4f} ({worst[1]}), lowest cos {lowcos[0]:.5f} ({lowcos[1]})")
