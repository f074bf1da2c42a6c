#!/usr/bin/env python3
"""bench.py -- utterances/sec of one GE2E training step on the HIP path (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W
N > 1 works both ways: under a launcher (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 ... bench.py --gpus N ...`, RANK / WORLD_SIZE in the environment) or plain -- then this process, BEFORE it
touches the GPU, starts that launcher itself as a fresh child (one rank per GPU over RCCL, as reference multi_gpu.sh:2
does with torch.distributed.launch), relays rank 0's JSON line and exits with the child's code.

A "step" is one full Trainer.Train_Step of the reference (Train.py:140-168) on one synthetic batch per
rank: forward -> GE2E loss -> backward (bucketed RCCL gradient mean overlapped with it when N > 1)
-> clip_grad_norm_(1.0) -> AdamW.  Workload = BASELINE.json configs[1]: 64 speakers x 15 utterances x
160 frames x 80 mel per rank (weak scaling; the reference's DistributedSampler semantics), dropout 0.1
on, bf16 storage / fp32 accumulate.  Inputs are generated on the device before the timed region.

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline":     live hipEvent timing of the dominant kernel class (--roofline-kernel, default the weight gradients: the top
                  rocprofv3 row) inside the timed region, and the same launches again with the backward's two streams serialised
  "cpu_baseline": a PyTorch-CPU restatement of the same Train_Step (oracle/torch_restatement.py: what the reference's
                  Device '-1' path executes) timed on this box's host cores on the FULL batch, 1 warm-up + 3 steps,
                  best-of (N = 1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from speaker_embedding_torch_amd import _lib  # noqa: E402
from speaker_embedding_torch_amd.Arg_Parser import Load_Hyper_Parameters  # noqa: E402
from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss  # noqa: E402
from speaker_embedding_torch_amd.Optim import FusedClipAdamW, GradScaler  # noqa: E402

PEAK = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3, "fp32x3": 2500.0 / 3.0}            # dense MFMA TFLOP/s, MI355X_MICROARCH.md chip table
DTYPE_NAME = {"bf16": "bf16", "fp16": "f16", "fp32": "f32", "fp32x3": "f32 storage, split-bf16 (bf16x3) products"}
HBM_PEAK_GBS = 8000.0                             # HBM3E spec peak (6.3 TB/s is what a copy kernel reaches)
ROOFLINE_CLASSES = {"gemm": _lib.K_GEMM, "gemm_ln": _lib.K_GEMM_LN, "wgrad": _lib.K_WGRAD,
                    "attn_fwd": _lib.K_ATTN_FWD, "attn_bwd": _lib.K_ATTN_BWD, "ffn": _lib.K_FFN}


def synth_mel(n, mel, t, seed, device):
    """x = clamp(-5 + 2 z, log(1e-5), 2): the log-mel range of meldataset.py:51-52,93-94."""
    g = torch.Generator(device=device).manual_seed(seed)
    return (torch.randn(n, mel, t, device=device, generator=g) * 2.0 - 5.0).clamp_(-11.5129, 2.0)


def load_pmc_traffic(kernel):
    """HBM bytes per launch of the class from the COMMITTED rocprofv3 --pmc passes (profiles/*_pmc_traffic.json: FETCH_SIZE doubled
    as MI355X_MICROARCH.md prescribes for gfx950, plus WRITE_SIZE) and the file they came from -- the counters cannot be collected
    inside this run (two separate --pmc passes under the profiler), so the line says where the figure was measured; (None, None)
    until measured."""
    import glob
    best, src = None, None
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc_traffic.json"))):
        try:
            v = json.load(open(path)).get(kernel)
        except Exception:
            v = None
        if v is not None:
            best, src = v, os.path.relpath(path, REPO)
    return best, src


def cpu_baseline(speakers, utts, frames, mel, steps=3):
    """the reference's CPU path restated with stock torch.nn operators (oracle/torch_restatement.py,
    pinned against the reference's golden vectors), full batch, fp32, dropout on, clip + AdamW, threads = the physical
    cores this job may use, 1 warm-up + `steps` steps, best-of."""
    from oracle import torch_restatement as TR      # checker / baseline only -- never on the product path
    before = torch.get_num_threads()
    try:
        best, done, threads = TR.time_train_steps(speakers, utts, frames, mel, steps=steps)
    finally:
        torch.set_num_threads(before)
    return {"value": round(speakers * utts / best, 2), "unit": "utterances/sec", "cores": int(threads), "kind": "port",
            "sample": f"full Train_Step of {speakers} spk x {utts} utt x {frames} fr x {mel} mel ({speakers * utts} utterances), "
                      f"torch-CPU restatement of the reference's Device '-1' path, fp32, dropout on, clip + AdamW; "
                      f"1 warm-up + {done} steps, best {best:.2f} s/step"}


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv):
    """--gpus N > 1 without a launcher: start `torch.distributed.run` with N ranks as a FRESH child process (never an
    exec, and before this process has made any GPU call), relay the ranks' single JSON line, return the child's code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.strip()]
    js = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if ln not in js:
            print(ln, file=sys.stderr)
    if js:
        print(js[-1], flush=True)
    elif proc.returncode == 0:
        print("bench.py: the ranks printed no JSON line", file=sys.stderr)
        return 1
    return proc.returncode


def dry_launch(rank, world, emit):
    """--dry-launch: rendezvous of the ranks over gloo on the CPU (what the N > 1 path needs from the launcher, minus the
    GPUs): every rank contributes its rank to an all-reduce; rank 0 prints the JSON line."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        ok = int(t.item()) == world * (world - 1) // 2
        dist.barrier()
        dist.destroy_process_group()
    else:
        ok = True
    print(f"dry-launch rank {rank} / world {world}", file=sys.stderr)
    if rank == 0:
        emit({"dry_launch": True, "ok": bool(ok), "n_gpus": world, "config": {"parallelism": f"dp{world}" if world > 1 else "single"}})
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32", "fp32x3"])
    ap.add_argument("--speakers", type=int, default=None)
    ap.add_argument("--utts", type=int, default=None)
    ap.add_argument("--frames", type=int, default=160)
    ap.add_argument("--roofline-kernel", default="wgrad", choices=sorted(ROOFLINE_CLASSES))
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="infer: BASELINE.json configs[3] style embed-only run (eval forward, --samples slices per utterance)")
    ap.add_argument("--samples", type=int, default=5)
    ap.add_argument("--input", default="resident", choices=["resident", "host32", "host16"],
                    help="resident: batches already in HBM (the metric's definition). host32 / host16: every step first brings "
                         "its batch from pinned host memory as float32 / float16 -- the PCIe-inclusive rate quoted in DESIGN.md, "
                         "never the headline value")
    ap.add_argument("--prefetch", action="store_true", help="with --input host*: copy batch i+1 on a side stream under step i")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--roofline-every", type=int, default=4, help="profile the roofline kernel class on every N-th timed step")
    ap.add_argument("--dry-launch", action="store_true",
                    help="only start the ranks, rendezvous over gloo on the CPU and print rank / world (tests the N > 1 launch path)")
    ap.add_argument("-hp", "--hyper_parameters", default=os.path.join(REPO, "speaker_embedding_torch_amd", "Hyper_Parameters.yaml"))
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # ---- N > 1 without a launcher: become the launcher (no GPU call has been made in this process)
    if args.gpus > 1 and "RANK" not in os.environ:
        if not args.dry_launch and torch.cuda.device_count() < args.gpus:      # device_count() does not initialise the GPU
            raise SystemExit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible on this node")
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    # Contract: rank 0 prints exactly ONE line (the JSON) on stdout.  Libraries chat on fd 1 too (RCCL prints a version
    # banner at communicator creation), so fd 1 is pointed at stderr for the whole run and the JSON line is written
    # to the saved, real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.dry_launch:
        sys.exit(dry_launch(rank, world, emit))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the GE2E hot path has no CPU fallback")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dev = torch.device("cuda", torch.cuda.current_device())
    use_dist = world > 1 or os.environ.get("GE2E_BENCH_FORCE_DIST") == "1"   # the override exercises the RCCL path on one GPU
    if use_dist:
        from speaker_embedding_torch_amd.distributed import apply_gradient_allreduce
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group(backend="nccl", rank=rank, world_size=world)

    hp = Load_Hyper_Parameters(args.hyper_parameters)
    S = args.speakers or hp.Train.Batch.Train.Speaker
    P = args.utts or hp.Train.Batch.Train.Pattern_per_Speaker
    T, mel = args.frames, hp.Sound.Mel_Dim

    torch.manual_seed(0)                       # same initial weights on every rank (then broadcast anyway)
    model = GE2E(hp, precision=args.precision, seed=1234 + rank).to(dev)
    criterion = GE2E_Loss().to(dev)
    for p_ in criterion.parameters():          # as Trainer.Model_Generate: never optimised, their .grad would only be accumulated
        p_.requires_grad_(False)
    if use_dist:
        model = apply_gradient_allreduce(model)
    optimizer = FusedClipAdamW(model.parameters(), lr=hp.Train.Learning_Rate.Initial,
                               betas=(hp.Train.ADAM.Beta1, hp.Train.ADAM.Beta2), eps=hp.Train.ADAM.Epsilon,
                               max_norm=hp.Train.Gradient_Norm)      # clip_grad_norm_ + AdamW, as Train.py:154-162
    scaler = GradScaler(enabled=args.precision == "fp16")        # Train.py:134: float16 runs under dynamic loss scaling
    model.train()
    batches = [synth_mel(S * P, mel, T, 1234 + rank + 1000 * i, dev) for i in range(2)]   # resident in HBM
    if args.mode == "infer":      # secondary figure (not BASELINE.json's metric): multi-slice d-vector extraction
        model.eval()
        n_utt = S * P
        xs = [synth_mel(n_utt * args.samples, mel, T, 99 + i, dev) for i in range(2)]
        with torch.no_grad():
            for i in range(args.warmup):
                model(xs[i & 1], args.samples)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                e = model(xs[i & 1], args.samples)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        emit({"metric": "utterances/sec, embed-only (eval forward)", "value": round(n_utt * args.steps / dt, 1),
                          "unit": "utterances/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 3), "dtype": DTYPE_NAME[args.precision],
                          "data": "synthetic", "config": {"workload": f"{n_utt} utt x {args.samples} slices x {T} fr x {mel} mel, "
                                                                      f"d-vectors [{n_utt}, 256]"},
              "unit_norm_err": float((e.norm(dim=1) - 1).abs().max())})
        return

    host = None
    if args.input != "resident":
        hdt = torch.float16 if args.input == "host16" else torch.float32
        host = [b.to(hdt).cpu().pin_memory() for b in batches]
        copy_stream = torch.cuda.Stream(dev) if args.prefetch else None
        staged = {}

        def fetch(i):
            """H2D of step i's batch: on the compute stream (no prefetch) or on the copy stream, fenced by an event."""
            if copy_stream is None:
                return host[i & 1].to(dev, non_blocking=True)
            if i not in staged:
                with torch.cuda.stream(copy_stream):
                    t = host[i & 1].to(dev, non_blocking=True)
                    e = torch.cuda.Event(); e.record(copy_stream)
                staged[i] = (t, e)
            t, e = staged.pop(i)
            with torch.cuda.stream(copy_stream):      # start the next one before this step's kernels are enqueued
                tn = host[(i + 1) & 1].to(dev, non_blocking=True)
                en = torch.cuda.Event(); en.record(copy_stream)
            staged[i + 1] = (tn, en)
            torch.cuda.current_stream(dev).wait_event(e)
            t.record_stream(torch.cuda.current_stream(dev))
            return t

    def train_step(i):
        """Trainer.Train_Step (Train.py:140-168) without the logging-only loss.item() host sync."""
        emb = model(batches[i & 1] if host is None else fetch(i))
        loss = criterion(emb, P)
        optimizer.zero_grad()
        scaler.backward(loss)                # = scaler.scale(loss).backward(), as Trainer.Train_Step issues it
        scaler.unscale_(optimizer)
        scaler.step(optimizer)               # (un)scale + inf check + clip + AdamW (+ scale update), fused: Trainer.Train_Step's calls
        scaler.update()
        return loss

    def fence():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        train_step(i)
    hnd = model._handle()
    klass = ROOFLINE_CLASSES[args.roofline_kernel]
    # The per-launch hipEvents of the roofline leg break the back-to-back queueing of ~44 GEMM launches per step
    # (measured: +0.25 ms on a 4.5 ms step), so only every `--roofline-every`-th step of the timed region carries them:
    # still live launches of the timed region, a few percent of perturbation instead of 5.5 %.
    every = max(1, args.roofline_every)
    profiled_steps = 0
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        sampled = (not args.no_roofline) and (i % every == 0)
        hnd.profile_enable(klass if sampled else 0)
        profiled_steps += int(sampled)
        loss = train_step(i)
    fence()
    dt = time.perf_counter() - t0
    hnd.profile_enable(0)
    if use_dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = tt.item()
    final_loss = loss.item()
    if not (final_loss == final_loss):
        raise SystemExit("non-finite loss in the timed region")

    roofline = None
    if not args.no_roofline:
        ms, flops, nbytes, launches = hnd.profile_read(klass)
        # the same class with the backward's two streams serialised (outside the timed region): in the step the weight gradients
        # share the chip with the main chain, so `achieved` above is a kernel on part of the machine; this is the kernel alone
        alone = None
        if world == 1:
            for i in range(3):
                hnd.profile_enable(klass | _lib.K_SERIAL)
                train_step(args.steps + i)
            torch.cuda.synchronize()
            hnd.profile_enable(0)
            ams, aflops, abytes_, alaunches = hnd.profile_read(klass)
            if alaunches and ams > 0:
                alone = (aflops / (ams * 1e-3) / 1e12, abytes_ / (ams * 1e-3) / 1e9, ams * 1e3 / alaunches)
        if launches and ms > 0:
            tf = flops / (ms * 1e-3) / 1e12
            gbs = nbytes / (ms * 1e-3) / 1e9
            ai = flops / max(nbytes, 1.0)                       # algorithmic FLOP per HBM byte of the class
            ridge = PEAK[args.precision] * 1e12 / (HBM_PEAK_GBS * 1e9)
            hbm_bound = ai < ridge                              # which roof is lower at this arithmetic intensity
            roofline = {
                "bound": "hbm" if hbm_bound else "mfma",
                "kernel": {"gemm": "projection GEMM class: gemm_ws_kernel / gemm_ws_lnbwd_kernel (K=256) + gemm_nt_kernel<128x128>", "gemm_ln": "LayerNorm GEMMs: gemm_ws_kernel<LN> + gemm_kl_kernel<LN>", "wgrad": "wgrad_ks_kernel + wgrad_ks_reduce_kernel (256x256 split-K tiles; narrow products: wgrad_kernel)",
                           "attn_fwd": "attn_fwd_kernel", "attn_bwd": "attn_bwd_kernel",
                           "ffn": "ffn_chain_kernel (FFN1 + ReLU + dropout + FFN2 + residual + LayerNorm)"}[args.roofline_kernel],
                "achieved": round(gbs if hbm_bound else tf, 2), "peak": HBM_PEAK_GBS if hbm_bound else PEAK[args.precision],
                "unit": "GB/s" if hbm_bound else "TFLOP/s",
                "frac": round((gbs / HBM_PEAK_GBS) if hbm_bound else (tf / PEAK[args.precision]), 4),
                "traffic": load_pmc_traffic(args.roofline_kernel)[0],
                "traffic_source": f"committed rocprofv3 --pmc passes of this kernel class ({load_pmc_traffic(args.roofline_kernel)[1]}; tools/final_meas.sh), not collected in this run",
                "launches": launches, "avg_launch_us": round(ms * 1e3 / launches, 2),
                "class_ms_per_step": round(ms / max(profiled_steps, 1), 3), "profiled_steps": profiled_steps,
                "algorithmic_flop_per_byte": round(ai, 1), "ridge_flop_per_byte": round(ridge, 1),
                "mfma_tflops": round(tf, 2), "mfma_frac": round(tf / PEAK[args.precision], 4),
                "hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_launch": round(nbytes / launches), "algorithmic_flops_per_launch": round(flops / launches),
            }
            if alone:
                roofline["alone"] = {"note": "same launches with the backward's weight-gradient stream serialised (after the timed region)",
                                     "achieved": round(alone[1] if hbm_bound else alone[0], 2),
                                     "frac": round((alone[1] / HBM_PEAK_GBS) if hbm_bound else (alone[0] / PEAK[args.precision]), 4),
                                     "avg_launch_us": round(alone[2], 2)}
    if rank == 0:
        out = {
            "metric": f"utterances/sec ({S}spk x {P}utt, T={T}, {mel}-mel) training step",
            "value": round(S * P * world * args.steps / dt, 1), "unit": "utterances/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE_NAME[args.precision], "data": "synthetic",
            "input": args.input + ("+prefetch" if (args.prefetch and args.input != "resident") else ""),
            "config": {"workload": f"{S} spk x {P} utt x {T} fr x {mel} mel per GPU, full Train_Step "
                                   f"(fwd+GE2E loss+bwd+clip+AdamW{'+loss scaling' if scaler.is_enabled() else ''}), "
                                   f"dropout {hp.GE2E.Transformer.Dropout_Rate:g}, random-init weights",
                       "per_gpu_batch": S * P, "global_batch": S * P * world,
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "final_loss": round(final_loss, 5),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(S, P, T, mel)
        emit(out)
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
