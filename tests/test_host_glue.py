"""CPU tests of the host-side glue: pattern datasets, collaters, the bucketed gradient mean (gloo, world 2),
and the loud failure of Trainer / Inferencer without a GPU."""
import os
import pickle

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from speaker_embedding_torch_amd import Datasets

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HP = os.path.join(REPO, "speaker_embedding_torch_amd", "Hyper_Parameters.yaml")


def make_patterns(root, speakers=5, files=4, mel=80):
    """Tiny pattern directory in the reference's on-disk format (Pattern_Generator.py:191-198)."""
    rng = np.random.default_rng(0)
    table = {}
    for s in range(speakers):
        spk = f"SPK{s:02d}"
        table[spk] = []
        for f in range(files + (s % 2)):
            name = f"{spk}/{f:03d}.PICKLE"
            os.makedirs(os.path.join(root, spk), exist_ok=True)
            t = int(rng.integers(40, 300))
            with open(os.path.join(root, name), "wb") as fh:
                pickle.dump({"Mel": rng.standard_normal((mel, t)).astype(np.float16), "Speaker": spk, "Dataset": "SYN"}, fh)
            table[spk].append(name)
    with open(os.path.join(root, "METADATA.PICKLE"), "wb") as fh:
        pickle.dump({"File_List_by_Speaker_Dict": table}, fh)
    return table


def test_correction_crop_and_reflect_pad():
    np.random.seed(0)
    x = np.arange(80 * 50, dtype=np.float32).reshape(80, 50)
    c = Datasets.Correction(x, 20)
    assert c.shape == (80, 20)
    start = int(c[0, 0])
    assert np.array_equal(c, x[:, start:start + 20])
    p = Datasets.Correction(x, 57)           # deficit 7 -> 3 left, 4 right, reflect (no edge repeat)
    assert p.shape == (80, 57)
    assert np.array_equal(p[:, 3:53], x)
    assert np.array_equal(p[:, :3], x[:, 3:0:-1]) and np.array_equal(p[:, 53:], x[:, -2:-6:-1])
    assert np.array_equal(Datasets.Correction(x, 50), x)


def test_dataset_and_collaters(tmp_path):
    make_patterns(str(tmp_path))
    ds = Datasets.Dataset(str(tmp_path), "METADATA.PICKLE", pattern_per_speaker=5)
    assert len(ds) == 2                       # only speakers with >= 5 files survive
    ds = Datasets.Dataset(str(tmp_path), "METADATA.PICKLE", pattern_per_speaker=3)
    assert len(ds) == 5
    item = ds[1]
    assert len(item) == 3 and item[0][0].dtype == np.float16 and item[0][1] == ds.speakers[1]
    col = Datasets.Collater(60, 70)
    batch = col([ds[0], ds[3]])
    assert batch.dtype == torch.float32 and batch.shape[:2] == (6, 80) and 60 <= batch.shape[2] <= 70
    assert batch.is_contiguous()
    inf = Datasets.Inference_Collater(samples=5, frame_length=64, overlap_length=32)
    assert inf.required_length == 192
    feats, speakers = inf([ds[0], ds[2]])
    assert feats.shape == (6 * 5, 80, 64) and len(speakers) == 6
    # consecutive windows overlap by 32 frames
    assert torch.equal(feats[0][:, 32:], feats[1][:, :32])
    sub = Datasets.Dataset(str(tmp_path), "METADATA.PICKLE", pattern_per_speaker=3, num_speakers=2)
    assert len(sub) == 2


def test_half_collater_and_prefetcher_cpu(tmp_path):
    """with half=True the batch keeps the on-disk float16 (same values as the float32 batch), and the
    prefetcher is a pass-through without a GPU."""
    make_patterns(str(tmp_path))
    ds = Datasets.Dataset(str(tmp_path), "METADATA.PICKLE", pattern_per_speaker=3)
    items = [ds[0], ds[3]]
    np.random.seed(5); b32 = Datasets.Collater(60, 70)(items)
    np.random.seed(5); b16 = Datasets.Collater(60, 70, half=True)(items)
    assert b16.dtype == torch.float16 and b16.shape == b32.shape and torch.equal(b16.float(), b32)
    np.random.seed(6); f32, s32 = Datasets.Inference_Collater(5, 64, 32)(items)
    np.random.seed(6); f16, s16 = Datasets.Inference_Collater(5, 64, 32, half=True)(items)
    assert f16.dtype == torch.float16 and torch.equal(f16.float(), f32) and s16 == s32
    batches = [torch.full((2, 3), float(i)) for i in range(4)]
    pre = Datasets.DevicePrefetcher(batches, "cpu")
    assert len(pre) == 4 and all(torch.equal(a, b) for a, b in zip(pre, batches))


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from speaker_embedding_torch_amd import distributed as D
    D.init_distributed(rank, world, dist_backend="gloo")
    torch.manual_seed(rank)
    lin = torch.nn.Linear(4, 3)
    lin = D.apply_gradient_allreduce(lin)           # broadcast rank 0's weights
    w = lin.weight.detach().clone()
    sync = lin._grad_sync
    grads = torch.arange(10, dtype=torch.float32) * (rank + 1)
    cb = sync.bucket_callback(grads)
    for off, cnt in ((7, 3), (3, 4), (0, 3)):        # the order the HIP backward reports buckets in
        cb(None, off, cnt)
    sync.finish(grads)
    red = D.reduce_tensor(torch.tensor([float(rank + 1)]), world)
    out[rank] = (w, grads, red, list(sync.buckets_seen))
    dist.destroy_process_group()


def test_bucketed_gradient_mean_gloo_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    w0, g0, r0, b0 = out[0]
    w1, g1, r1, b1 = out[1]
    assert torch.equal(w0, w1)                                   # rank 0's parameters everywhere
    expect = torch.arange(10, dtype=torch.float32) * 1.5         # mean of x1 and x2
    assert torch.allclose(g0, expect) and torch.allclose(g1, expect)
    assert float(r0) == float(r1) == 1.5
    assert b0 == [(7, 3), (3, 4), (0, 3)]


def test_trainer_and_inferencer_need_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from speaker_embedding_torch_amd.Inference import Inferencer
    from speaker_embedding_torch_amd.Train import Trainer
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Trainer(HP, datasets={})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Inferencer(HP)


def test_hyper_parameters_yaml_has_reference_schema():
    from speaker_embedding_torch_amd.Arg_Parser import Load_Hyper_Parameters
    hp = Load_Hyper_Parameters(HP)
    assert hp.Sound.Mel_Dim == 80 and hp.GE2E.Embedding_Size == 256 and hp.GE2E.Transformer.Head == 4
    assert hp.Train.Batch.Train.Speaker == 64 and hp.Train.Batch.Train.Pattern_per_Speaker == 15
    assert hp.Train.Inference.Samples == 5 and hp.Train.ADAM.Epsilon == 1e-6
    for key in ("Checkpoint_Path", "Log_Path", "Use_Mixed_Precision", "Use_Multi_GPU", "Device"):
        assert hasattr(hp, key)


def test_bench_starts_its_own_ranks_for_gpus_gt_1():
    """`python bench.py --gpus N` (no launcher, as the driver calls it) must start N ranks itself.
    --dry-launch rendezvous them over gloo on the CPU and prints the one JSON line."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-launch"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                   # exactly ONE line on stdout
    out = json.loads(lines[0])
    assert out["dry_launch"] and out["ok"] and out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2"
    err = r.stderr.decode()
    assert "dry-launch rank 0 / world 2" in err and "dry-launch rank 1 / world 2" in err


def test_trainer_launcher_command_line():
    """`Use_Multi_GPU: true` + a Device list makes Train.main start one rank per GPU through torch.distributed.run
    (the reference does it with multi_gpu.sh:2); here only the command it would run is checked."""
    import subprocess
    from speaker_embedding_torch_amd import Train
    seen = {}

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 0
        return R()
    orig = subprocess.run
    subprocess.run = fake_run
    try:
        assert Train.launch_ranks(4, ["-hp", "x.yaml"]) == 0
    finally:
        subprocess.run = orig
    cmd = seen["cmd"]
    assert "--nproc-per-node=4" in cmd and "torch.distributed.run" in cmd and cmd[-2:] == ["-hp", "x.yaml"]
    assert "127.0.0.1" in cmd and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert os.access(os.path.join(REPO, "multi_gpu.sh"), os.X_OK)


def test_dropout_stream_keep_rate_and_independence_inside_a_quad():
    """The dropout stream hashes an aligned index QUAD once (mix32) and takes the quad's second word from one xorshift32 step
    (csrc/common.cuh, oracle drop_keep_at).  Over 2^20 quads: keep rate of each of the four positions, the 16 joint keep patterns of
    a quad against independent draws, and the correlation between neighbouring quads."""
    from oracle import ge2e_oracle as O
    nq = 1 << 20
    for key, p in ((0x12345678, 0.1), (O.drop_key(1234, 7, 3), 0.5)):
        keep = O.drop_keep(key, 4 * nq, p).reshape(nq, 4)
        pk = 1.0 - O.drop_threshold(p) / 65536.0
        assert np.abs(keep.mean(0) - pk).max() < 4.5 * np.sqrt(pk * (1 - pk) / nq)
        code = (keep * np.array([1, 2, 4, 8])).sum(1)
        freq = np.bincount(code, minlength=16) / nq
        exp = np.array([np.prod([pk if (c >> b) & 1 else 1 - pk for b in range(4)]) for c in range(16)])
        assert (np.abs(freq - exp) / np.sqrt(exp * (1 - exp) / nq)).max() < 4.5
        for a, b in ((3, 0), (0, 0), (2, 2)):          # position a of quad q against position b of quad q + 1
            assert abs(np.corrcoef(keep[:-1, a], keep[1:, b])[0, 1]) < 4.5 / np.sqrt(nq)
