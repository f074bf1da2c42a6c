"""PyTorch-CPU restatement of one reference Train_Step  --  TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT CODE.

What `bench.py`'s `cpu_baseline` leg times on the GPU box's host cores: the same hot path the
HIP library runs, expressed with stock torch.nn operators on the CPU, i.e. what the reference's `Device: '-1'` path
executes (reference Modules.py:46-59 encoder, :121-156 loss; Train.py:122-127 AdamW, :152-162 step order).  Written
here from the maths of SURVEY.md appendix A, not from the reference's files: linear layers instead of k=1 convolutions,
the loss in matrix form (one [N,S] cosine matrix, log-softmax against the own-speaker column).
`tests/test_oracle_golden.py` pins it against the golden vectors of the reference (eval d-vectors, loss), so the
baseline number belongs to the right computation.

Only `tests/` and `bench.py`'s `cpu_baseline` leg import this module.
"""
import math
import os
import time

import torch


class EncoderCPU(torch.nn.Module):
    """prenet -> +alpha*pe -> dropout -> 3 post-LN transformer layers + final LN -> frame 0 -> slice mean -> projection
    -> L2 normalise.  Parameter names follow the checkpoint key set so formula weights load by name."""

    def __init__(self, mel=80, d=256, heads=4, layers=3, max_position=1024, p_pe=0.1, p_tf=0.1):
        super().__init__()
        self.prenet = torch.nn.Linear(mel, d)
        self.alpha = torch.nn.Parameter(torch.ones(1))
        pos = torch.arange(max_position, dtype=torch.float32)[:, None]
        freq = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * (-math.log(10000.0) / d))[None, :]
        table = torch.stack([torch.sin(pos * freq), torch.cos(pos * freq)], dim=-1).reshape(max_position, d)
        self.register_buffer("pe", table)                     # [max_position, d]
        self.pe_drop = torch.nn.Dropout(p_pe)
        layer = torch.nn.TransformerEncoderLayer(d_model=d, nhead=heads, dim_feedforward=4 * d, dropout=p_tf)
        self.transformer = torch.nn.TransformerEncoder(layer, num_layers=layers, norm=torch.nn.LayerNorm(d),
                                                       enable_nested_tensor=False)
        self.projection = torch.nn.Linear(d, d)

    def load_named(self, params):
        """params: {checkpoint key: numpy array} (oracle.formula_params / a reference state_dict)."""
        sd = self.state_dict()
        for k, v in params.items():
            t = torch.as_tensor(v)
            if k in ("prenet.weight", "projection.weight"):
                sd[k].copy_(t[:, :, 0])
            elif k == "positional_encoding.alpha":
                sd["alpha"].copy_(t)
            elif k in sd:
                sd[k].copy_(t)
            else:
                raise KeyError(k)

    def forward(self, x, samples=1):
        """x: [N, mel, T] -> [N // samples, d] unit-norm."""
        n, _, t = x.shape
        h = torch.relu(self.prenet(x.transpose(1, 2)))        # [N, T, d]
        h = self.pe_drop(h + self.alpha * self.pe[:t])
        h = self.transformer(h.transpose(0, 1))               # [T, N, d]
        z = h[0].reshape(n // samples, samples, -1).mean(1)
        return torch.nn.functional.normalize(self.projection(z), dim=-1)


def ge2e_loss_cpu(emb, utts, w=10.0, b=-5.0):
    """mean_i( logsumexp_s(w cos(e_i, c_s) - b) - (w cos(e_i, c_own) - b) ), self-inclusive centroids."""
    n, d = emb.shape
    cent = emb.reshape(n // utts, utts, d).mean(1)
    en = emb.norm(dim=1, keepdim=True).clamp_min(1e-8)
    cn = cent.norm(dim=1, keepdim=True).clamp_min(1e-8)
    sim = w * (emb @ cent.t()) / (en * cn.t()) - b
    own = torch.arange(n) // utts
    return torch.nn.functional.cross_entropy(sim, own)


def host_cores():
    """Cores this process may run on (the GPU box gives a job a share of the host), at most the physical count."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    phys = avail
    try:
        seen = set()
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                seen.add((pid, line.split(":")[1].strip()))
        if seen:
            phys = min(avail, len(seen))
    except Exception:
        pass
    return max(1, phys)


def time_train_steps(speakers, utts, frames, mel, params=None, steps=3, budget_s=90.0, seed=1234, threads=None,
                     lr=1e-4, betas=(0.9, 0.999), eps=1e-6, max_norm=1.0):
    """1 warm-up + up to `steps` full Train_Steps (fwd -> loss -> zero_grad -> bwd -> clip -> AdamW, dropout on) on the
    full batch; best-of.  Stops early once `budget_s` of timed work is spent.  Returns (best seconds, steps timed, threads)."""
    threads = threads or host_cores()
    torch.set_num_threads(threads)
    torch.manual_seed(seed)
    model = EncoderCPU(mel=mel)
    if params is not None:
        model.load_named(params)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=lr, betas=betas, eps=eps)      # default weight_decay 0.01, as Train.py:122-127
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(speakers * utts, mel, frames, generator=g) * 2.0 - 5.0).clamp_(-11.5129, 2.0)

    def step():
        loss = ge2e_loss_cpu(model(x), utts)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        opt.step()
        return float(loss.detach())

    step()                                                    # warm-up: thread pool, allocator, page-in
    best, done, spent = float("inf"), 0, 0.0
    while done < steps and (done == 0 or spent < budget_s):
        t0 = time.perf_counter()
        loss = step()
        dt = time.perf_counter() - t0
        if not math.isfinite(loss):
            raise RuntimeError("non-finite loss in the CPU baseline")
        best, done, spent = min(best, dt), done + 1, spent + dt
    return best, done, threads
