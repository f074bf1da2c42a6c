#!/usr/bin/env python3
"""Soak test (development tool): many training steps on changing shapes, looking for hangs, faults and non-finite values.
    timeout -k 10 600 python tools/soak.py [--steps 1500] [--shapes 150]
Phase 1: the benchmark shape, `--steps` consecutive Train_Steps (dropout stream advances every step).
Phase 2: `--shapes` random (speakers, utterances, frames) shapes, bf16, fp32, fp16 and fp32x3 in turn (every 5th with the fused attention sub-layer on), fresh NaN-poisoned workspace,
one forward + backward + optimizer step each; every gradient must be finite; every 3rd shape also embeds a batch in eval mode with 1-6 slices per utterance."""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss
from speaker_embedding_torch_amd import _lib
from speaker_embedding_torch_amd.Optim import FusedClipAdamW


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=1500); ap.add_argument("--shapes", type=int, default=150)
    args = ap.parse_args()
    hp = bench.Load_Hyper_Parameters(os.path.join(bench.REPO, "speaker_embedding_torch_amd", "Hyper_Parameters.yaml"))
    dev = torch.device("cuda")
    model = GE2E(hp, precision="bf16", seed=1).to(dev); crit = GE2E_Loss().to(dev)
    opt = FusedClipAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-6, max_norm=1.0)
    model.train()
    xs = [bench.synth_mel(960, 80, 160, 10 + i, dev) for i in range(4)]
    t0 = time.time()
    for i in range(args.steps):
        loss = crit(model(xs[i & 3]), 15); opt.zero_grad(); loss.backward(); opt.step()
        if i % 250 == 249:
            print(f"phase 1 step {i + 1}: loss {loss.item():.4f} ({time.time() - t0:.1f} s)", flush=True)
            assert np.isfinite(loss.item())
    rng = np.random.default_rng(0)
    for k in range(args.shapes):
        S = int(rng.integers(2, 9)); P = int(rng.integers(2, 7)); prec = ("bf16", "fp32", "fp16", "fp32x3")[k % 4]
        _lib.set_option("attn_sub", 1 if k % 5 == 4 else 0)          # every 5th shape: the opt-in fused attention sub-layer kernel (16-bit modes, 49-160 frames)
        T = int(rng.integers(289, 640)) if k % 12 == 11 else int(rng.integers(17, 289))       # every 12th shape: the chunked long-sequence attention
        m = GE2E(hp, precision=prec, seed=k).to(dev); m._poison = True; m.train()
        o = FusedClipAdamW(m.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-6, max_norm=1.0)
        x = bench.synth_mel(S * P, 80, T, 100 + k, dev)
        loss = crit(m(x), P); o.zero_grad(); loss.backward(); o.step()
        ok = np.isfinite(loss.item()) and all(torch.isfinite(p.grad).all().item() for p in m.parameters())
        if k % 3 == 2:                                               # every 3rd shape also embeds: eval mode, 1-6 slices per utterance
            smp = int(rng.integers(1, 7)); m.eval()
            with torch.no_grad():
                e = m(bench.synth_mel(S * smp, 80, T, 200 + k, dev), smp)
            ok = ok and e.shape[0] == S and torch.isfinite(e).all().item() and (e.norm(dim=1) - 1.0).abs().max().item() < 1e-3
        if not ok or k % 25 == 24:
            print(f"phase 2 shape {k + 1}: S={S} P={P} T={T} {prec} loss {loss.item():.4f} finite={ok}", flush=True)
        assert ok, (S, P, T, prec)
    _lib.set_option("attn_sub", 0)
    print("soak ok")


if __name__ == "__main__":
    main()
