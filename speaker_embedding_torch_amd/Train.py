"""`Train.py`-shaped trainer on the MI355X hot path.

Keeps the reference's surface (reference Train.py): `Trainer(hp_path, steps)`, `.Train()`, `.Train_Step(features)`,
`.Evaluation_Step`, `.Inference_Step`, `S_<steps>.pt` checkpoints holding {'Model','Optimizer','Scheduler','Steps'}
with the reference's state_dict keys, auto-resume from the newest checkpoint, the same Hyper_Parameters.yaml, and
the CLI `python -m speaker_embedding_torch_amd.Train -hp <yaml> [-s steps]`.

What differs, deliberately:
  * model + loss run in libge2e_hip.so (no torch fallback); `Use_Mixed_Precision: false` selects the fp32 MFMA kernels,
    true the 16-bit-storage kernels: float16 with dynamic loss scaling as in the reference (Train.py:134,145,153-162;
    `Optim.GradScaler`, device-side) or -- optional key `Mixed_Precision_Dtype: 'bf16'` -- bfloat16, which needs no scaler;
  * multi-GPU = one process per GPU; gradients are averaged bucket by bucket over RCCL while backward is still running
    (distributed.py) instead of one all-reduce after it.  `python -m speaker_embedding_torch_amd.Train -hp <yaml>` with
    `Use_Multi_GPU: true` starts the ranks itself (one per entry of `Device`), as multi_gpu.sh does for the reference;
  * the training loss is accumulated ON DEVICE and read back only every `Logging_Interval` steps (the
    reference's per-step `.item()` is a host sync, Train.py:167).
The step order is the reference's: forward -> loss -> zero_grad -> backward(+all-reduce) -> clip -> AdamW
(Train.py:140-162), with AdamW's default weight_decay 0.01 (the YAML's Weight_Decay is never read, Train.py:122-127).
"""
import argparse
import logging
import math
import os
import sys
from collections import defaultdict

import torch
import yaml

from .Arg_Parser import Load_Hyper_Parameters
from .Datasets import Collater, Dataset, DevicePrefetcher, Inference_Collater
from .distributed import apply_gradient_allreduce, init_distributed, reduce_tensor
from .Logger import Logger
from .Modules import GE2E, GE2E_Loss, GE2E_Loss_Global
from .Optim import FusedClipAdamW, GradScaler

logging.basicConfig(level=logging.INFO, stream=sys.stdout,
                    format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s")


class Trainer:
    def __init__(self, hp_path, steps=0, datasets=None):
        """datasets: optional {'Train','Dev','Inference'} DataLoader dict (synthetic benchmarks / tests);
        default builds them from the pattern directories named in the YAML like Train.py:59-117."""
        self.hp_path = hp_path
        self.gpu_id = int(os.getenv("RANK", "0"))
        self.num_gpus = int(os.getenv("WORLD_SIZE", "1"))
        self.hp = Load_Hyper_Parameters(hp_path)
        if not torch.cuda.is_available():
            raise RuntimeError("Trainer needs an MI355X: the GE2E hot path has no CPU fallback "
                               "(the reference's Device: '-1' CPU path is what oracle/ restates for testing)")
        local = int(os.getenv("LOCAL_RANK", str(self.gpu_id))) % torch.cuda.device_count()
        self.device = torch.device("cuda:{}".format(local))
        torch.cuda.set_device(local)
        self.steps = steps
        self.dataloader_dict = datasets if datasets is not None else self.Dataset_Generate()
        self.Model_Generate()
        self.Load_Checkpoint()
        self._Set_Distribution()
        self.scalar_dict = {"Train": defaultdict(float), "Evaluation": defaultdict(float)}
        self._loss_accum = torch.zeros((), device=self.device)
        self._loss_count = 0
        self.writer_dict = None
        self.tqdm = None

    # -------------------------------------------------------------------------------------- data
    def Dataset_Generate(self):
        tr, ev = self.hp.Train.Train_Pattern, self.hp.Train.Eval_Pattern
        batch = self.hp.Train.Batch
        train_dataset = Dataset(tr.Path, tr.Metadata_File, batch.Train.Pattern_per_Speaker)
        dev_dataset = Dataset(ev.Path, ev.Metadata_File, batch.Eval.Pattern_per_Speaker)
        inference_dataset = Dataset(ev.Path, ev.Metadata_File, batch.Eval.Pattern_per_Speaker, num_speakers=50)
        logging.info("The number of train speakers = {}.".format(len(train_dataset)))
        logging.info("The number of development speakers = {}.".format(len(dev_dataset)))
        # patterns are fp16 on disk; with Half_Mel_Transfer (default on, optional key) a batch stays fp16
        # through pinned memory and PCIe and is widened by the HIP packing kernel -- same numbers, half the bytes
        half = bool(getattr(self.hp.Train, "Half_Mel_Transfer", True))
        collater = Collater(self.hp.Train.Frame_Length.Min, self.hp.Train.Frame_Length.Max, half=half)
        inference_collater = Inference_Collater(self.hp.Train.Inference.Samples, self.hp.Train.Inference.Frame_Length,
                                                self.hp.Train.Inference.Overlap_Length, half=half)

        def sampler(ds, distributed):
            if distributed:        # each rank draws its OWN speakers; loss stays local (Train.py:90-99)
                return torch.utils.data.DistributedSampler(ds, shuffle=True)
            return torch.utils.data.RandomSampler(ds)

        common = dict(num_workers=self.hp.Train.Num_Workers, pin_memory=True)
        return {
            "Train": torch.utils.data.DataLoader(train_dataset, sampler=sampler(train_dataset, self.hp.Use_Multi_GPU),
                                                 collate_fn=collater, batch_size=batch.Train.Speaker, **common),
            "Dev": torch.utils.data.DataLoader(dev_dataset, sampler=sampler(dev_dataset, self.num_gpus > 1),
                                               collate_fn=collater, batch_size=batch.Eval.Speaker, **common),
            "Inference": torch.utils.data.DataLoader(inference_dataset, shuffle=True, collate_fn=inference_collater,
                                                     batch_size=batch.Eval.Speaker, **common),
        }

    # -------------------------------------------------------------------------------------- model
    def Model_Generate(self):
        self.model = GE2E(self.hp, seed=1234 + self.gpu_id).to(self.device)
        # optional key Train.Global_Batch_Loss (default off = the reference's per-rank loss)
        global_loss = bool(getattr(self.hp.Train, "Global_Batch_Loss", False)) and self.num_gpus > 1
        self.criterion = (GE2E_Loss_Global() if global_loss else GE2E_Loss()).to(self.device)
        # The criterion's weight / bias are nn.Parameters nothing optimises, reduces or saves (Train.py:121-127,298-303); with requires_grad
        # their .grad would be accumulated every step (three small launches on the step's critical path) and never read: switched off HERE,
        # not in the module (GE2E_Loss itself produces them like the reference's autograd does)
        for p_ in self.criterion.parameters():
            p_.requires_grad_(False)
        # torch.optim.AdamW semantics and state_dict (incl. the default weight_decay 0.01 the reference ends up with,
        # Train.py:122-127), with clip_grad_norm_(Gradient_Norm) fused into the same two device launches
        self.optimizer = FusedClipAdamW(
            params=self.model.parameters(), lr=self.hp.Train.Learning_Rate.Initial,
            betas=(self.hp.Train.ADAM.Beta1, self.hp.Train.ADAM.Beta2), eps=self.hp.Train.ADAM.Epsilon,
            max_norm=self.hp.Train.Gradient_Norm)
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(
            optimizer=self.optimizer, gamma=self.hp.Train.Learning_Rate.Decay, last_epoch=-1)
        # Train.py:134: GradScaler(enabled=Use_Mixed_Precision).  Only float16 needs it (bfloat16 keeps fp32's exponent range)
        self.scaler = GradScaler(enabled=self.model.precision == "fp16")
        # `Use_Mixed_Precision: true` without the optional key `Mixed_Precision_Dtype` means float16 + dynamic loss scaling (what
        # autocast is in the reference); say which mode runs, so a resumed run that changed mode is visible in the log
        logging.info("Arithmetic mode: {}{}.".format(self.model.precision, " + dynamic loss scaling" if self.scaler.is_enabled() else ""))

    # -------------------------------------------------------------------------------------- steps
    def Train_Step(self, features):
        features = features.to(self.device, non_blocking=True)
        embeddings = self.model(features)
        loss = self.criterion(embeddings, self.hp.Train.Batch.Train.Pattern_per_Speaker)
        self.optimizer.zero_grad()
        self.scaler.backward(loss)           # = scaler.scale(loss).backward() (Train.py:153); HIP backward, buckets all-reduced as they complete
        self.scaler.unscale_(self.optimizer)
        self.scaler.step(self.optimizer)     # unscale + inf check + clip_grad_norm_(Gradient_Norm) + AdamW, fused (Train.py:154-162)
        self.scaler.update()
        self.steps += 1
        if self.tqdm is not None:
            self.tqdm.update(1)
        self._loss_accum += loss.detach()    # stays on the device until the next logging point
        self._loss_count += 1
        return loss

    def _flush_train_loss(self):
        if self._loss_count == 0:
            return
        mean = self._loss_accum / self._loss_count
        if self.num_gpus > 1:
            mean = reduce_tensor(mean, self.num_gpus)
        self.scalar_dict["Train"]["Loss/Embedding"] = mean.item()
        self._loss_accum.zero_()
        self._loss_count = 0

    def Train_Epoch(self):
        steps_per_epoch = math.ceil(len(self.dataloader_dict["Train"].dataset) / self.hp.Train.Batch.Train.Speaker)
        for features in DevicePrefetcher(self.dataloader_dict["Train"], self.device):   # batch i+1 crosses PCIe under step i
            self.Train_Step(features)
            if self.steps % steps_per_epoch == 0:
                self.scheduler.step()
            if self.steps % self.hp.Train.Checkpoint_Save_Interval == 0:
                self.Save_Checkpoint()
            if self.steps % self.hp.Train.Logging_Interval == 0:
                self._flush_train_loss()
                if self.gpu_id == 0 and self.writer_dict is not None:
                    self.scalar_dict["Train"]["Learning_Rate"] = self.scheduler.get_last_lr()[0]
                    self.writer_dict["Train"].add_scalar_dict(self.scalar_dict["Train"], self.steps)
                self.scalar_dict["Train"] = defaultdict(float)
            if self.steps % self.hp.Train.Evaluation_Interval == 0:
                self.Evaluation_Epoch()
            if self.steps % self.hp.Train.Inference_Interval == 0:
                self.Inference_Epoch()
            if self.steps >= self.hp.Train.Max_Step:
                return

    @torch.no_grad()
    def Evaluation_Step(self, features):
        features = features.to(self.device, non_blocking=True)
        embeddings = self.model(features)
        loss = self.criterion(embeddings, self.hp.Train.Batch.Eval.Pattern_per_Speaker)
        if self.num_gpus > 1:
            loss = reduce_tensor(loss, self.num_gpus)
        self.scalar_dict["Evaluation"]["Loss/Embedding"] += loss.item()
        return loss

    def Evaluation_Epoch(self):
        logging.info("(Steps: {}) Start evaluation in GPU {}.".format(self.steps, self.gpu_id))
        self.model.eval()
        step = 0
        for step, features in enumerate(DevicePrefetcher(self.dataloader_dict["Dev"], self.device), 1):
            self.Evaluation_Step(features)
        self.scalar_dict["Evaluation"] = {tag: loss / max(step, 1) for tag, loss in self.scalar_dict["Evaluation"].items()}
        if self.writer_dict is not None:
            self.writer_dict["Evaluation"].add_scalar_dict(self.scalar_dict["Evaluation"], self.steps)
            self.writer_dict["Evaluation"].add_histogram_model(self.model, "GE2E", self.steps,
                                                               delete_keywords=["layer_Dict", "layer"])
        result = dict(self.scalar_dict["Evaluation"])
        self.scalar_dict["Evaluation"] = defaultdict(float)
        self.model.train()
        return result

    @torch.no_grad()
    def Inference_Step(self, features):
        return self.model(features=features.to(self.device, non_blocking=True), samples=self.hp.Train.Inference.Samples)

    def Inference_Epoch(self):
        if self.gpu_id != 0:
            return None
        logging.info("(Steps: {}) Start inference.".format(self.steps))
        self.model.eval()
        embeddings, speakers = [], []
        for features, names in self.dataloader_dict["Inference"]:
            embeddings.append(self.Inference_Step(features))
            speakers.extend(names)
        embeddings = torch.cat(embeddings, dim=0).cpu().numpy()
        if self.writer_dict is not None:
            self.writer_dict["Evaluation"].add_embedding(embeddings, metadata=speakers, global_step=self.steps, tag="Embeddings")
        self.model.train()
        return embeddings, speakers

    # -------------------------------------------------------------------------------------- checkpoints
    def Load_Checkpoint(self):
        if self.steps == 0:
            paths = [os.path.join(root, file).replace("\\", "/")
                     for root, _, files in os.walk(self.hp.Checkpoint_Path) for file in files
                     if os.path.splitext(file)[1] == ".pt"]
            if not paths:
                return      # initial training
            path = max(paths, key=os.path.getctime)
        else:
            path = os.path.join(self.hp.Checkpoint_Path, "S_{}.pt".format(self.steps)).replace("\\", "/")
        # weights_only=True: tensors / plain containers only; nothing from the file is executed
        state_dict = torch.load(path, map_location="cpu", weights_only=True)
        self.model.load_state_dict(state_dict["Model"])           # strict: the reference's 44 keys
        self.optimizer.load_state_dict(state_dict["Optimizer"])
        self.scheduler.load_state_dict(state_dict["Scheduler"])
        self.steps = state_dict["Steps"]
        self.model._step = int(self.steps)                        # dropout counter follows the step count
        logging.info("Checkpoint loaded at {} steps.".format(self.steps))

    def Save_Checkpoint(self):
        if self.gpu_id != 0:
            return None
        os.makedirs(self.hp.Checkpoint_Path, exist_ok=True)
        path = os.path.join(self.hp.Checkpoint_Path, "S_{}.pt".format(self.steps)).replace("\\", "/")
        torch.save({"Model": self.model.state_dict(), "Optimizer": self.optimizer.state_dict(),
                    "Scheduler": self.scheduler.state_dict(), "Steps": self.steps}, path)
        logging.info("Checkpoint saved at {} steps.".format(self.steps))
        return path

    def _Set_Distribution(self):
        if self.num_gpus > 1:
            self.model = apply_gradient_allreduce(self.model)

    # -------------------------------------------------------------------------------------- main loop
    def Train(self):
        hp_copy = os.path.join(self.hp.Checkpoint_Path, "Hyper_Parameters.yaml").replace("\\", "/")
        if self.gpu_id == 0 and not os.path.exists(hp_copy):
            os.makedirs(self.hp.Checkpoint_Path, exist_ok=True)
            if getattr(self.hp, "Use_Mixed_Precision", False) and not hasattr(self.hp, "Mixed_Precision_Dtype"):
                self.hp.Mixed_Precision_Dtype = self.model.precision      # the copy records the RESOLVED 16-bit mode (a resume keeps it)
            with open(hp_copy, "w") as f:
                yaml.dump(self.hp, f)
        if self.gpu_id == 0:
            self.writer_dict = {"Train": Logger(os.path.join(self.hp.Log_Path, "Train")),
                                "Evaluation": Logger(os.path.join(self.hp.Log_Path, "Evaluation"))}
        if self.steps == 0:
            self.Evaluation_Epoch()
        if self.hp.Train.Initial_Inference:
            self.Inference_Epoch()
        from tqdm import tqdm
        self.tqdm = tqdm(initial=self.steps, total=self.hp.Train.Max_Step, desc="[Training]", disable=self.gpu_id != 0)
        while self.steps < self.hp.Train.Max_Step:
            try:
                self.Train_Epoch()
            except KeyboardInterrupt:
                self.Save_Checkpoint()
                sys.exit(1)
        self.tqdm.close()
        logging.info("Finished training.")


def launch_ranks(n, argv):
    """`Use_Multi_GPU: true` without a launcher: start one rank per GPU (reference multi_gpu.sh:2 does it with
    torch.distributed.launch) as a fresh child -- before this process makes any GPU call -- and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "speaker_embedding_torch_amd.Train", *argv]
    return subprocess.run(cmd, env=env).returncode


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-hp", "--hyper_parameters", required=True, type=str)
    parser.add_argument("-s", "--steps", default=0, type=int)
    parser.add_argument("-p", "--port", default=54321, type=int)          # accepted, unused (as in the reference)
    parser.add_argument("-r", "--local_rank", default=0, type=int)        # accepted, unused (as in the reference)
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parser.parse_args(argv)
    hp = Load_Hyper_Parameters(args.hyper_parameters)
    devices = [d for d in str(hp.Device).split(",") if d.strip() != ""] if hp.Device is not None else []
    if hp.Use_Multi_GPU and "RANK" not in os.environ and len(devices) > 1:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", ",".join(devices))   # Train.py:356 sets CUDA_VISIBLE_DEVICES from hp.Device
        sys.exit(launch_ranks(len(devices), argv))
    if int(os.getenv("WORLD_SIZE", "1")) == 1 and hp.Device is not None:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", str(hp.Device))      # Train.py:356 sets CUDA_VISIBLE_DEVICES
    if hp.Use_Multi_GPU:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        init_distributed(rank=int(os.getenv("RANK", "0")), num_gpus=int(os.getenv("WORLD_SIZE", "1")), dist_backend="nccl")
    Trainer(hp_path=args.hyper_parameters, steps=args.steps).Train()


if __name__ == "__main__":
    main()
