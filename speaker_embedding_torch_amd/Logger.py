"""Scalar logging with the reference's Logger interface (reference Logger.py:6-81).  TensorBoard is used when
it is importable; otherwise scalars go to `<log_dir>/scalars.jsonl` (observability only, not the hot path)."""
import json
import os

try:
    from torch.utils.tensorboard import SummaryWriter as _Base
except Exception:  # tensorboard is not installed in the build image
    _Base = None


class Logger:
    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self._tb = _Base(log_dir) if _Base is not None else None
        self._f = None if self._tb else open(os.path.join(log_dir, "scalars.jsonl"), "a")

    def add_scalar_dict(self, scalar_dict, global_step=None, walltime=None):
        for tag, scalar in scalar_dict.items():
            if self._tb:
                self._tb.add_scalar(tag, scalar, global_step, walltime)
            else:
                self._f.write(json.dumps({"tag": tag, "value": float(scalar), "step": global_step}) + "\n")
        if self._f:
            self._f.flush()

    def add_histogram_model(self, model, model_label=None, global_step=None, delete_keywords=(), **_):
        if not self._tb:
            return
        for tag, parameter in model.named_parameters():
            tag = "/".join(x for x in tag.split(".") if x not in delete_keywords)
            if model_label is not None:
                tag = "{}/{}".format(model_label, tag)
            self._tb.add_histogram(tag, parameter.detach().cpu().numpy(), global_step)
            if parameter.grad is not None:
                self._tb.add_histogram(tag + "/gradient", parameter.grad.detach().cpu().numpy(), global_step)

    def add_embedding(self, embeddings, metadata=None, global_step=None, tag="Embeddings"):
        if self._tb:
            self._tb.add_embedding(embeddings, metadata=metadata, global_step=global_step, tag=tag)

    def close(self):
        if self._tb:
            self._tb.close()
        elif self._f:
            self._f.close()
