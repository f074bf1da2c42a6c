"""Speaker-batch data parallelism over the GPUs of one node: RCCL (torch.distributed backend "nccl") over xGMI.

Same entry points as reference distributed.py (`init_distributed` :28, `apply_gradient_allreduce` :73,
`reduce_tensor` :22), different mechanism: the reference flattens all gradients into one bucket and
all-reduces it once AFTER the whole backward (`queue_callback`, distributed.py:88-118).  Here the HIP
backward reports each gradient bucket of the flat fp32 gradient buffer as soon as its last kernel is
enqueued ([final norm + projection], [layer L-1], ..., [layer 0], [prenet]); the bucket's all-reduce is
issued immediately with async_op=True, so RCCL runs it on its own stream behind exactly the kernels that
produced it and overlaps the rest of backward.  One process per GPU; semantics per rank are the
reference's (each rank draws its own speakers and computes a local GE2E loss, Train.py:90-99).
"""
import os

import torch
import torch.distributed as dist

from . import _lib


def reduce_tensor(tensor, num_gpus):
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= num_gpus
    return rt


def init_distributed(rank, num_gpus, dist_backend="nccl"):
    """env:// rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the launcher), one GPU per rank."""
    if dist_backend in (None, "nccl"):
        assert torch.cuda.is_available(), "Distributed mode requires the GPUs (backend nccl = RCCL)."
        torch.cuda.set_device(rank % torch.cuda.device_count())
    print("> initializing distributed for rank {} out of {}".format(rank, num_gpus))
    dist.init_process_group(backend=dist_backend or "nccl", rank=rank, world_size=num_gpus)


class GradSync:
    """Bucketed, backward-overlapped mean all-reduce of the flat gradient buffer."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        # RCCL averages in the collective (ncclAvg); gloo has no AVG: sum, then one scaling pass
        self._avg = dist.get_backend(group) == "nccl" and not os.environ.get("GE2E_ALLREDUCE_SUM")
        self._works = []
        self._error = None
        self.buckets_seen = []          # (offset, count) of the last backward, for tests / logging

    def bucket_callback(self, grads_flat, handle=None, stream=0):
        """`handle`/`stream`: the library handle and the raw stream the backward runs on.  A bucket is final behind
        handle.bucket_stream(stream) (the weight-gradient stream), so its all-reduce is issued with THAT stream
        current: RCCL's stream then waits for exactly the bucket's producers and the backward chain is not held."""
        self._works, self._error, self.buckets_seen = [], None, []

        def on_bucket(_user, offset, count):
            try:        # exceptions must not unwind through the C frame
                self.buckets_seen.append((int(offset), int(count)))
                view = grads_flat[offset:offset + count]
                ptr = handle.bucket_stream(stream) if (handle is not None and grads_flat.is_cuda) else 0
                op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
                if ptr and ptr != stream:
                    with torch.cuda.stream(torch.cuda.ExternalStream(ptr, device=grads_flat.device)):
                        w = dist.all_reduce(view, op=op, group=self.group, async_op=True)
                else:
                    w = dist.all_reduce(view, op=op, group=self.group, async_op=True)
                self._works.append(w)
            except BaseException as ex:  # noqa: BLE001
                self._error = ex
        return _lib.BUCKET_CB(on_bucket)

    def finish(self, grads_flat):
        """The compute stream waits for every bucket (no host block with RCCL), then sum -> mean."""
        if self._error is not None:
            raise RuntimeError("gradient all-reduce failed") from self._error
        for w in self._works:
            if w is not None:
                w.wait()
        self._works = []
        if not self._avg:
            grads_flat.mul_(1.0 / self.world)


def apply_gradient_allreduce(module, group=None):
    """Broadcast rank 0's state (every state_dict tensor, as distributed.py:83-86) and arm the module's
    backward with the overlapped gradient mean.  Returns the same module (no wrapper class)."""
    for p in module.state_dict().values():
        if torch.is_tensor(p):
            dist.broadcast(p, 0, group=group)
    module._grad_sync = GradSync(group)
    return module
