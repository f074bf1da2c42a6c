"""GPU tests of the Train.py / Inference.py shaped glue: the Train_Step order (forward -> loss -> backward ->
clip -> AdamW) pinned against the reference's own numbers (golden G4), checkpoints with the reference key set,
multi-slice inference from a saved checkpoint, and 2-rank data parallelism with real HIP backward kernels."""
import os
import sys

import numpy as np
import pytest
import torch
import yaml

from oracle import ge2e_oracle as O
from conftest import rel_l2
from test_host_glue import make_patterns

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HP = os.path.join(REPO, "speaker_embedding_torch_amd", "Hyper_Parameters.yaml")


def write_hp(tmp_path, **over):
    hp = yaml.safe_load(open(HP))
    hp["Checkpoint_Path"] = str(tmp_path / "ckpt")
    hp["Log_Path"] = str(tmp_path / "log")
    hp["Use_Mixed_Precision"] = over.pop("bf16", False)         # with the shipped optional key Mixed_Precision_Dtype: 'bf16'
    mp = over.pop("mp_dtype", None)                              # 'fp16': the reference's autocast dtype (+ GradScaler)
    if mp is not None:
        hp["Use_Mixed_Precision"], hp["Mixed_Precision_Dtype"] = True, mp
    p = over.pop("dropout", 0.1)
    hp["GE2E"]["Positional_Encoding"]["Dropout_Rate"] = p
    hp["GE2E"]["Transformer"]["Dropout_Rate"] = p
    for k, v in over.items():
        node = hp
        *path, last = k.split(".")
        for q in path:
            node = node[q]
        node[last] = v
    path = tmp_path / "hp.yaml"
    yaml.safe_dump(hp, open(path, "w"))
    return str(path)


def load_formula(model):
    sd = model.state_dict()
    for k, v in O.formula_params().items():
        sd[k].copy_(torch.from_numpy(v))


def test_train_step_order_matches_reference_G4(tmp_path, golden):
    """Two Trainer.Train_Steps at dropout 0 (fp32 kernels) reproduce the reference's post-AdamW parameters."""
    from speaker_embedding_torch_amd.Train import Trainer
    hp_path = write_hp(tmp_path, dropout=0.0, **{"Train.Batch.Train.Speaker": 4, "Train.Batch.Train.Pattern_per_Speaker": 5})
    tr = Trainer(hp_path, datasets={})
    load_formula(tr.model)
    x = torch.from_numpy(O.formula_mel(1, 20, 80, 160))
    tr.Train_Step(x)
    names = [n for n, _ in O.param_specs()]
    params = dict(tr.model.named_parameters())
    for i, n in enumerate(names):
        s = params[n].detach().double().sum().item()
        assert abs(s - golden["G4_param_sum"][i]) < 1e-5 * max(1.0, abs(golden["G4_param_sum"][i])) + 5e-4, n
        k = min(8, params[n].numel())
        assert np.abs(params[n].detach().reshape(-1)[:k].cpu().numpy() - golden["G4_param_head"][i][:k]).max() < 5e-6, n
    loss2 = tr.Train_Step(x)
    assert abs(loss2.item() - float(golden["G4_loss_step2"][0])) < 2e-5
    for i, n in enumerate(names):
        s = params[n].detach().double().sum().item()
        assert abs(s - golden["G4_param_sum_step2"][i]) < 1e-5 * max(1.0, abs(golden["G4_param_sum_step2"][i])) + 1e-3, n
    assert tr.steps == 2


def test_trainer_epoch_checkpoint_resume_and_inferencer(tmp_path):
    from speaker_embedding_torch_amd.Inference import Inferencer
    from speaker_embedding_torch_amd.Train import Trainer
    pat = tmp_path / "patterns"
    make_patterns(str(pat), speakers=6, files=5)
    hp_path = write_hp(tmp_path, bf16=True, **{
        "Train.Train_Pattern.Path": str(pat), "Train.Eval_Pattern.Path": str(pat),
        "Train.Batch.Train.Speaker": 3, "Train.Batch.Train.Pattern_per_Speaker": 4,
        "Train.Batch.Eval.Speaker": 3, "Train.Batch.Eval.Pattern_per_Speaker": 4,
        "Train.Frame_Length.Min": 60, "Train.Frame_Length.Max": 90, "Train.Max_Step": 4,
        "Train.Checkpoint_Save_Interval": 2, "Train.Logging_Interval": 2, "Train.Evaluation_Interval": 4,
        "Train.Inference_Interval": 4})
    tr = Trainer(hp_path)
    tr.Train()
    assert tr.steps == 4
    ck = os.path.join(tr.hp.Checkpoint_Path, "S_4.pt")
    assert os.path.exists(ck) and os.path.exists(os.path.join(tr.hp.Checkpoint_Path, "S_2.pt"))
    state = torch.load(ck, map_location="cpu", weights_only=True)
    assert set(state) == {"Model", "Optimizer", "Scheduler", "Steps"} and state["Steps"] == 4
    assert len(state["Model"]) == 44 and "positional_encoding.pe" in state["Model"]
    assert os.path.exists(os.path.join(tr.hp.Checkpoint_Path, "Hyper_Parameters.yaml"))
    ev = tr.Evaluation_Epoch()
    assert np.isfinite(ev["Loss/Embedding"])
    emb, speakers = tr.Inference_Epoch()
    assert emb.shape == (len(speakers), 256) and np.allclose(np.linalg.norm(emb, axis=1), 1, atol=1e-4)
    # auto-resume from the newest checkpoint (Train.py:270-282)
    tr2 = Trainer(hp_path)
    assert tr2.steps == 4
    for (n1, p1), (n2, p2) in zip(tr.model.state_dict().items(), tr2.model.state_dict().items()):
        assert n1 == n2 and torch.equal(p1.cpu(), p2.cpu())
    # Inferencer: multi-slice d-vectors from .npy mels, equal to model(mels, Samples) on the same windows
    inf = Inferencer(hp_path, ck, batch_size=3)
    rng = np.random.default_rng(1)
    paths = []
    for i in range(5):
        p = str(tmp_path / f"utt{i}.npy")
        np.save(p, rng.standard_normal((80, 200 + 7 * i)).astype(np.float32))
        paths.append(p)
    np.random.seed(3)
    emb, labels = inf.Inference(paths + [str(tmp_path / "missing.npy")], list("abcde") + ["x"])
    assert emb.shape == (5, 256) and labels == list("abcde")
    assert torch.allclose(emb.norm(dim=1), torch.ones(5, device=emb.device), atol=1e-4)
    with pytest.raises(NotImplementedError):
        inf.Inference([__file__], ["wav"])


def _dp_gpu_worker(rank, world, port, hp_path, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    sys.path.insert(0, REPO)
    from speaker_embedding_torch_amd import distributed as D
    from speaker_embedding_torch_amd.Arg_Parser import Load_Hyper_Parameters
    from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)   # one GPU box: gloo carries the CUDA buffers
    torch.cuda.set_device(0)
    hp = Load_Hyper_Parameters(hp_path)
    torch.manual_seed(100 + rank)                                # different init per rank: broadcast must fix it
    model = GE2E(hp, precision="fp32", seed=7).cuda()
    model = D.apply_gradient_allreduce(model)
    model.train()
    crit = GE2E_Loss().cuda()
    x = torch.from_numpy(O.formula_mel(20 + rank, 8, 80, 48, logmel=True)).cuda()
    crit(model(x), 4).backward()
    torch.cuda.synchronize()
    flat = torch.cat([p.grad.flatten() for p in model.parameters()]).cpu()
    w = torch.cat([p.detach().flatten() for p in model.parameters()]).cpu()
    out[rank] = (flat, w, list(model._grad_sync.buckets_seen))
    torch.distributed.destroy_process_group()


def test_two_rank_data_parallel_gradient_mean(tmp_path):
    """Each rank: own batch, local loss (Train.py:90-99); gradients = mean over ranks, bucket by bucket."""
    import torch.multiprocessing as mp
    from speaker_embedding_torch_amd.Arg_Parser import Load_Hyper_Parameters
    from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss
    hp_path = write_hp(tmp_path, dropout=0.0)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_gpu_worker, args=(2, 29600 + os.getpid() % 2000, hp_path, out), nprocs=2, join=True)
    g0, w0, b0 = out[0]
    g1, w1, b1 = out[1]
    assert torch.equal(w0, w1) and torch.allclose(g0, g1, atol=0, rtol=0)
    # single-process reference: same weights, both batches, average of the two gradients
    hp = Load_Hyper_Parameters(hp_path)
    model = GE2E(hp, precision="fp32", seed=7).cuda()
    torch.nn.utils.vector_to_parameters(w0.cuda(), model.parameters())
    model.train()
    crit = GE2E_Loss().cuda()
    acc = 0
    for r in range(2):
        model.zero_grad()
        crit(model(torch.from_numpy(O.formula_mel(20 + r, 8, 80, 48, logmel=True)).cuda()), 4).backward()
        acc = acc + torch.cat([p.grad.flatten() for p in model.parameters()]).cpu()
    assert rel_l2(g0.numpy(), (acc / 2).numpy()) < 1e-4
    # ... and the oracle (reference distributed.py:88-112: gradients summed over ranks / world): mean of O.encoder_backward over
    # the two rank batches with the broadcast weights
    names = [k for k, _ in model.named_parameters()]
    params = {k: v.detach().cpu().numpy() for k, v in model.named_parameters()}
    pe = model.positional_encoding.pe[0].t().contiguous().cpu().numpy()
    mean_ref = None
    for r in range(2):
        emb_ref, c = O.encoder_forward(params, O.formula_mel(20 + r, 8, 80, 48, logmel=True), train=True, seed=7, step=0, p_pe=0.0, p_tf=0.0, pe=pe)
        _, lc = O.loss_forward(emb_ref, 4)
        gr = O.encoder_backward(params, c, O.loss_backward(lc))
        flat = np.concatenate([gr[k].ravel() for k in names])
        mean_ref = flat / 2 if mean_ref is None else mean_ref + flat / 2
    assert rel_l2(g0.numpy(), mean_ref) < 2e-3
    total = g0.numel()
    assert len(b0) == 5 and sum(c for _, c in b0) == total         # tail, 3 layers, prenet: disjoint cover
    assert b0[0][0] + b0[0][1] == total and b0[-1][0] == 0


def _global_loss_worker(rank, world, port, hp_path, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    sys.path.insert(0, REPO)
    from speaker_embedding_torch_amd import distributed as D
    from speaker_embedding_torch_amd.Arg_Parser import Load_Hyper_Parameters
    from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss_Global
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    hp = Load_Hyper_Parameters(hp_path)
    torch.manual_seed(0)
    model = GE2E(hp, precision="fp32", seed=7).cuda()
    load_formula(model)
    model = D.apply_gradient_allreduce(model)
    model.train()
    x_all = torch.from_numpy(O.formula_mel(31, 12, 80, 48, logmel=True))        # 4 speakers x 3 utterances, speaker-major
    x = x_all[rank * 6:(rank + 1) * 6].cuda()                                     # this rank: 2 whole speakers
    loss = GE2E_Loss_Global().cuda()(model(x), 3)
    loss.backward()
    torch.cuda.synchronize()
    out[rank] = (loss.item(), torch.cat([p.grad.flatten() for p in model.parameters()]).cpu())
    torch.distributed.destroy_process_group()


def test_global_batch_loss_equals_single_process_on_the_concatenated_batch(tmp_path):
    """2 ranks x 2 speakers with GE2E_Loss_Global + the mean gradient all-reduce == one process on
    all 4 speakers (dropout 0, fp32 kernels): same loss, same parameter gradient."""
    import torch.multiprocessing as mp
    from speaker_embedding_torch_amd.Arg_Parser import Load_Hyper_Parameters
    from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss
    hp_path = write_hp(tmp_path, dropout=0.0)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_global_loss_worker, args=(2, 31600 + os.getpid() % 2000, hp_path, out), nprocs=2, join=True)
        res = dict(out)
    hp = Load_Hyper_Parameters(hp_path)
    model = GE2E(hp, precision="fp32", seed=7).cuda()
    load_formula(model)
    model.train()
    x_all = torch.from_numpy(O.formula_mel(31, 12, 80, 48, logmel=True)).cuda()
    loss = GE2E_Loss().cuda()(model(x_all), 3)
    loss.backward()
    ref = torch.cat([p.grad.flatten() for p in model.parameters()]).cpu()
    # the oracle on the concatenated 4-speaker batch
    names = [k for k, _ in model.named_parameters()]
    params = {k: v.detach().cpu().numpy() for k, v in model.named_parameters()}
    pe = model.positional_encoding.pe[0].t().contiguous().cpu().numpy()
    emb_ref, c = O.encoder_forward(params, O.formula_mel(31, 12, 80, 48, logmel=True), train=True, seed=7, step=0, p_pe=0.0, p_tf=0.0, pe=pe)
    loss_ref, lc = O.loss_forward(emb_ref, 3)
    gr = O.encoder_backward(params, c, O.loss_backward(lc))
    ora = np.concatenate([gr[k].ravel() for k in names])
    for rank in (0, 1):
        l, g = res[rank]
        assert abs(l - loss.item()) < 1e-5 and abs(l - float(loss_ref)) < 1e-5
        assert rel_l2(g.numpy(), ref.numpy()) < 1e-4
        assert rel_l2(g.numpy(), ora) < 2e-3
    assert torch.equal(res[0][1], res[1][1])


def test_fused_clip_adamw_matches_torch():
    """FusedClipAdamW == clip_grad_norm_ + torch.optim.AdamW (reference Train.py:154-162), three steps, incl. state_dict."""
    from speaker_embedding_torch_amd.Optim import FusedClipAdamW
    torch.manual_seed(0)
    shapes = [(256, 80, 1), (256,), (1,), (768, 256), (1024, 256), (5000,)]
    pa = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-6)
    oa = FusedClipAdamW(pa, max_norm=1.0, **kw)
    ob = torch.optim.AdamW(pb, **kw)
    for step in range(3):
        gs = [torch.randn_like(p) * (0.5 if step != 1 else 1e-3) for p in pa]      # step 1: norm below max_norm -> no clipping
        for p, q, g in zip(pa, pb, gs):
            p.grad, q.grad = g.clone(), g.clone()
        ref_norm = torch.nn.utils.clip_grad_norm_(pb, 1.0)
        ob.step()
        oa.step()
        assert abs(oa.total_grad_norm().item() - ref_norm.item()) < 1e-4 * ref_norm.item()
        for p, q in zip(pa, pb):
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-6)
            assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7)           # gradients left clipped in place
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["state"].keys() == sb["state"].keys()
    for k in sa["state"]:
        assert set(sa["state"][k]) == {"step", "exp_avg", "exp_avg_sq"}
        assert float(sa["state"][k]["step"]) == 3.0
        assert torch.allclose(sa["state"][k]["exp_avg_sq"], sb["state"][k]["exp_avg_sq"], rtol=1e-5, atol=1e-8)
    ob2 = torch.optim.AdamW(pb, **kw)
    ob2.load_state_dict(sa)                                                       # torch loads the fused optimizer's state


@pytest.mark.parametrize("bf16", [True, False, "fp16"])
def test_training_learns_synthetic_speakers(tmp_path, bf16):
    """End to end through Trainer.Train_Step (forward, GE2E loss, backward, clip, AdamW; dropout on): on mel batches whose
    only structure is a per-speaker spectral envelope the loss must fall well below its initial value within 40 steps and
    held-out utterances of a speaker must end up closer to their own centroid than to any other."""
    from speaker_embedding_torch_amd.Train import Trainer
    S, P, T = 8, 6, 64
    kw = {"mp_dtype": "fp16"} if bf16 == "fp16" else {"bf16": bf16}
    hp_path = write_hp(tmp_path, **kw, **{"Train.Batch.Train.Speaker": S, "Train.Batch.Train.Pattern_per_Speaker": P,
                                          "Train.Learning_Rate.Initial": 5e-4})
    tr = Trainer(hp_path, datasets={})
    assert tr.model.precision == {True: "bf16", False: "fp32", "fp16": "fp16"}[bf16]
    assert tr.scaler.is_enabled() == (bf16 == "fp16")              # GradScaler(enabled=...) as Train.py:134, float16 only
    g = torch.Generator().manual_seed(0)
    envelopes = torch.randn(S, 80, 1, generator=g) * 1.5                        # what identifies a speaker

    def batch(seed):
        gg = torch.Generator().manual_seed(seed)
        x = envelopes[:, None].expand(S, P, 80, 1) + torch.randn(S, P, 80, T, generator=gg)
        return x.reshape(S * P, 80, T).contiguous()

    losses = [tr.Train_Step(batch(100 + i)).item() for i in range(40)]
    assert all(np.isfinite(losses))
    first, last = np.mean(losses[:3]), np.mean(losses[-5:])
    assert last < 0.5 * first, (first, last)
    tr.model.eval()
    with torch.no_grad():
        e = tr.model(batch(999).cuda()).float().cpu().reshape(S, P, 256)
    cent = torch.nn.functional.normalize(e.mean(1), dim=1)
    sim = torch.einsum("spd,cd->spc", e, cent)                                   # [S, P, S]
    assert (sim.argmax(-1) == torch.arange(S)[:, None]).float().mean() > 0.9
    if bf16 == "fp16":        # the dynamic loss scale settled somewhere finite and most steps were taken
        assert 1.0 <= tr.scaler.get_scale() <= 65536.0 * 4 and tr.scaler.steps_taken() >= 30


def test_loss_scaled_step_matches_torch_gradscaler_semantics():
    """ge2e_clip_adamw_step_scaled == GradScaler.unscale_ + clip_grad_norm_ + AdamW + GradScaler.update (reference
    Train.py:153-162) with the state on the device: clean steps match the unscaled optimizer fed the true gradients,
    an inf / nan gradient skips the step (parameters, moments, step count untouched) and halves the scale, and
    `growth_interval` clean steps in a row double it."""
    from speaker_embedding_torch_amd.Optim import FusedClipAdamW, GradScaler
    torch.manual_seed(1)
    shapes = [(256, 80, 1), (1,), (768, 256), (4099,)]
    pa = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-6)
    oa, ob = FusedClipAdamW(pa, max_norm=1.0, **kw), FusedClipAdamW(pb, max_norm=1.0, **kw)
    sc = GradScaler(init_scale=1024.0, growth_interval=3)
    scale, tracker, taken = 1024.0, 0, 0
    for step in range(9):
        bad = step in (2, 6)
        gs = [torch.randn_like(p) * (0.5 if step % 2 == 0 else 1e-3) for p in pa]
        before = [p.detach().clone() for p in pa]
        for p, q, g in zip(pa, pb, gs):
            p.grad, q.grad = g * scale, g.clone()
        if bad:
            pa[2].grad[5, 7] = float("inf") if step == 2 else float("nan")
        sc.step(oa)
        sc.update()
        if bad:
            for p, b in zip(pa, before):
                assert torch.equal(p, b)                                    # skipped: nothing moved
            scale, tracker = scale * 0.5, 0
        else:
            ob.step()
            taken += 1
            tracker += 1
            if tracker == 3:
                scale, tracker = scale * 2.0, 0
            for p, q in zip(pa, pb):
                assert torch.allclose(p, q, rtol=2e-5, atol=2e-6), step
                assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7)   # left unscaled and clipped in place
        assert sc.get_scale() == scale and sc.steps_taken() == taken, (step, sc.get_scale(), scale)
        assert float(sc.state(pa[0].device)[2]) == (1.0 if bad else 0.0)
    sd = oa.state_dict()
    assert all(float(st["step"]) == float(taken) for st in sd["state"].values())    # skipped steps do not count
    assert set(sc.state_dict()) == {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"}
    assert GradScaler(enabled=False).scale(torch.ones((), device="cuda")).item() == 1.0


def test_loss_scaling_refuses_param_groups_and_reseeds_steps_after_load():
    """One scale / growth-tracker update per step: several param_groups under a GradScaler are refused (the fused step ends with the
    scaler update, once per call); and `load_state_dict` after scaled steps re-seeds the device-side step counter from the loaded 'step'."""
    from speaker_embedding_torch_amd.Optim import FusedClipAdamW, GradScaler
    torch.manual_seed(2)
    a, b = torch.nn.Parameter(torch.randn(300, device="cuda")), torch.nn.Parameter(torch.randn(70, device="cuda"))
    two = FusedClipAdamW([{"params": [a]}, {"params": [b], "lr": 1e-3}], lr=1e-2, max_norm=1.0)
    a.grad, b.grad = torch.randn_like(a), torch.randn_like(b)
    with pytest.raises(RuntimeError, match="one param_group"):
        GradScaler(init_scale=8.0).step(two)
    two.step()                                                            # without a scaler several groups are fine
    one = FusedClipAdamW([a, b], lr=1e-2, max_norm=1.0)
    sc = GradScaler(init_scale=8.0)
    for _ in range(3):
        a.grad, b.grad = torch.randn_like(a) * 8, torch.randn_like(b) * 8
        sc.step(one)
    sd = one.state_dict()
    assert all(float(st["step"]) == 3.0 for st in sd["state"].values())
    for st in sd["state"].values():
        st["step"] = torch.tensor(10.0)
    one.load_state_dict(sd)
    a.grad, b.grad = torch.randn_like(a) * 8, torch.randn_like(b) * 8
    sc.step(one)
    assert sc.steps_taken() == 11                                         # continued from the loaded count, not from the device's 3


@pytest.mark.timeout(300)
def test_many_steps_enqueued_without_a_host_sync(tmp_path):
    """Regression for the round-1 stream deadlock (a fence event re-recorded while the side stream still had to wait on
    its previous record): the host runs many Train_Steps ahead of the GPU, never synchronising, and everything must
    drain.  Each in-flight backward now owns its event set, recycled only after that backward completed on the device."""
    from speaker_embedding_torch_amd.Train import Trainer
    hp_path = write_hp(tmp_path, bf16=True, **{"Train.Batch.Train.Speaker": 16, "Train.Batch.Train.Pattern_per_Speaker": 8})
    tr = Trainer(hp_path, datasets={})
    xs = [torch.randn(128, 80, 96, device="cuda") for _ in range(2)]
    losses = [tr.Train_Step(xs[i & 1]) for i in range(60)]      # no .item(), no synchronize inside the loop
    torch.cuda.synchronize()
    vals = torch.stack([l.detach() for l in losses]).cpu()
    assert torch.isfinite(vals).all()
    hnd = tr.model._handle()
    assert hnd is not None and tr.steps == 60


# ------------------------------------------------------------------------------------------ the RCCL bucket schedule on one GPU
_NCCL_SCRIPT = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="{port}", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
from oracle import ge2e_oracle as O
from speaker_embedding_torch_amd import distributed as D
from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss
import test_gpu_parity as tp
D.init_distributed(0, 1, "nccl")                              # reference distributed.py:28-36; backend "nccl" IS RCCL on ROCm
out = {{}}
for tag, n, t, P in (("full", 960, 160, 15), ("small", 12, 77, 3)):
    m, params, pe = tp.build(GE2E, "bf16", 0.1)
    m.train()
    x = torch.from_numpy(O.formula_mel(5, n, 80, t, logmel=True)).cuda()
    crit = GE2E_Loss().cuda()
    grads = {{}}
    for mode in ("plain", "nccl"):
        if mode == "nccl":
            D.apply_gradient_allreduce(m)                    # broadcast + GradSync: every backward from here on reports buckets
        for q in m.parameters():
            q.grad = None
        m._step = 0                                          # the dropout stream follows the step counter: both runs are step 0
        emb = m(x)
        crit(emb, P).backward()
        torch.cuda.synchronize()
        grads[mode] = torch.cat([q.grad.flatten() for q in m.parameters()]).cpu().numpy()
        out[tag + "_emb_" + mode] = emb.detach().cpu().numpy()
    sync = m._grad_sync
    assert sync is not None and torch.distributed.get_backend() == "nccl" and sync._avg
    out[tag + "_buckets"] = np.array(sync.buckets_seen, dtype=np.int64)
    hnd = m._handle()
    st = torch.cuda.current_stream().cuda_stream
    out[tag + "_side_stream_differs"] = np.array([int(hnd.bucket_stream(st) not in (0, st))])
    out[tag + "_g_plain"], out[tag + "_g_nccl"] = grads["plain"], grads["nccl"]
torch.distributed.destroy_process_group()
np.savez({out!r}, **out)
"""


def test_bucket_schedule_through_a_one_rank_rccl_group(tmp_path):
    """VERDICT r3 #3 / next-round 4b.  With a bucket callback the backward runs a DIFFERENT schedule from the benchmarked one (the norm2
    column sums stay on the weight-gradient stream, every bucket forks) and `GradSync` issues `all_reduce(AVG, async_op=True)` with the
    library's weight-gradient stream current (`ExternalStream(ge2e_bucket_stream)`, speaker_embedding_torch_amd/distributed.py).  This pool has
    one GPU per box, so the only RCCL execution possible is a 1-rank `nccl` group: a fresh process runs configs[1] (64 x 15 x 160, bf16,
    dropout on) plainly and through `apply_gradient_allreduce` (reference distributed.py:73-125) and the two must agree to fp32 summation
    order; the same at 12 x 77, where the oracle (same dropout stream) finishes in seconds, against the oracle within the bf16 bounds."""
    import subprocess
    out = str(tmp_path / "nccl.npz")
    code = _NCCL_SCRIPT.format(repo=REPO, port=29700 + os.getpid() % 2000, out=out)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = dict(np.load(out))
    for tag in ("full", "small"):
        assert np.array_equal(res[tag + "_emb_plain"], res[tag + "_emb_nccl"]), tag            # forward: no atomics, bitwise
        b = res[tag + "_buckets"]
        total = res[tag + "_g_plain"].size
        assert len(b) == 5 and int(b[:, 1].sum()) == total and b[0, 0] + b[0, 1] == total and b[-1, 0] == 0
        assert int(res[tag + "_side_stream_differs"][0]) == 1, "the all-reduce must run behind the weight-gradient stream, not the caller's"
        assert np.isfinite(res[tag + "_g_nccl"]).all()
        assert rel_l2(res[tag + "_g_nccl"], res[tag + "_g_plain"]) < 1e-4, tag              # mean over ONE rank == the rank's own gradient
    # the oracle on the small case: the bf16 bounds of test_bf16_train_step_vs_oracle
    from speaker_embedding_torch_amd.Modules import GE2E
    import test_gpu_parity as tp
    m, params, pe = tp.build(GE2E, "bf16", 0.1)
    x_np = O.formula_mel(5, 12, 80, 77, logmel=True)
    emb_ref, c = O.encoder_forward(params, x_np, train=True, seed=1234, step=0, p_pe=0.1, p_tf=0.1, pe=pe)
    _, lc = O.loss_forward(emb_ref, 3)
    grads_ref = O.encoder_backward(params, c, O.loss_backward(lc))
    assert rel_l2(res["small_emb_nccl"], emb_ref) < 2e-2
    flat, off = res["small_g_nccl"].astype(np.float64), 0
    for name, prm in m.named_parameters():
        k = prm.numel()
        g, rr = flat[off:off + k], grads_ref[name].ravel().astype(np.float64)
        off += k
        if k == 1:
            assert abs(g[0] - rr[0]) < 0.3 * np.linalg.norm(grads_ref["prenet.bias"]), name
            continue
        cos = float(g @ rr / max(np.linalg.norm(g) * np.linalg.norm(rr), 1e-30))
        assert cos > 0.99 and rel_l2(g, rr) < 0.15, (name, cos, rel_l2(g, rr))


@pytest.mark.parametrize("where,value", [("transformer.layers.0.linear1.weight", float("inf")), ("transformer.layers.0.linear1.bias", float("nan")),
                                          ("transformer.layers.1.linear2.weight", float("-inf"))])
def test_non_finite_hidden_activations_still_skip_the_scaled_step(where, value):
    """the chained FFN applies ReLU and dropout as INTEGER operations on packed 16-bit pairs (common.cuh pk_relu16 / pk_keep16): a NaN
    whose sign bit is set becomes +0 and a dropped Inf / NaN becomes 0, where torch.relu / dropout propagate the NaN.  The loss scaler's skip
    decision (reference Train.py:153-162: GradScaler.step skips on inf / nan gradients) must survive that: a non-finite FFN weight or bias makes
    whole hidden COLUMNS non-finite (+Inf and positive NaN pass the packed ReLU, kept elements pass the dropout), so the layer output, the loss
    and every gradient are non-finite, found_inf is set, parameters and moments stay untouched and the scale backs off.
    (What the packed forms can hide is a lone NEGATIVE-signed NaN in a hidden activation whose inputs are all finite -- which finite
    weights and finite LayerNorm outputs cannot produce.)"""
    from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss
    from speaker_embedding_torch_amd.Optim import FusedClipAdamW, GradScaler
    import test_gpu_parity as tp
    m, _, _ = tp.build(GE2E, "fp16", 0.1)
    m.train()
    opt = FusedClipAdamW(m.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-6, max_norm=1.0)
    sc = GradScaler(init_scale=1024.0)
    x = torch.from_numpy(O.formula_mel(9, 12, 80, 77, logmel=True)).cuda()
    crit = GE2E_Loss().cuda()
    for q in crit.parameters():
        q.requires_grad_(False)

    def one_step():
        opt.zero_grad()
        loss = crit(m(x), 3)
        sc.backward(loss)
        sc.step(opt)
        sc.update()
        return loss

    loss = one_step()                                                       # a clean step first: moments exist, scale unchanged
    assert torch.isfinite(loss) and float(sc.state(x.device)[2]) == 0.0 and sc.get_scale() == 1024.0
    prm = dict(m.named_parameters())[where]
    with torch.no_grad():
        prm.view(-1)[7] = value
    m._prepared_key = None
    before = [q.detach().clone() for q in m.parameters()]
    state_before = {k: {kk: vv.clone() if torch.is_tensor(vv) else vv for kk, vv in st.items()} for k, st in opt.state_dict()["state"].items()}
    loss = one_step()
    assert not torch.isfinite(loss), "a non-finite FFN parameter must reach the loss"
    assert float(sc.state(x.device)[2]) == 1.0, "found_inf not set: the step was NOT skipped"
    assert sc.get_scale() == 512.0 and sc.steps_taken() == 1
    for q, b in zip(m.parameters(), before):
        assert torch.equal(torch.nan_to_num(q.detach(), nan=123.0, posinf=456.0, neginf=-456.0), torch.nan_to_num(b, nan=123.0, posinf=456.0, neginf=-456.0))
    for k, st in opt.state_dict()["state"].items():
        for kk, vv in st.items():
            if torch.is_tensor(vv):
                assert torch.equal(vv, state_before[k][kk]), (k, kk)
