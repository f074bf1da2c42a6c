"""Fused gradient clipping + AdamW on the device.

`FusedClipAdamW` is a `torch.optim.AdamW` whose `step()` runs `clip_grad_norm_(max_norm)` and the AdamW
update of every parameter in two kernels of libge2e_hip.so (reference Train.py:154-162 runs them as ~10
multi-tensor launches).  State layout and `state_dict()` are torch.optim.AdamW's ('step', 'exp_avg',
'exp_avg_sq' per parameter), so reference checkpoints' 'Optimizer' entries load and save unchanged.
Arithmetic is torch's (decoupled decay, bias-corrected moments, eps added after the sqrt)."""
import ctypes as C

import torch

from . import _lib


class FusedClipAdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.max_norm = float(max_norm)
        self._hnd = None
        self._norm = None
        self._tables = None          # cached pointer tables (rebuilt when a tensor moves)

    def _state_for(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise RuntimeError("FusedClipAdamW does not take a closure")
        if self._hnd is None:
            self._hnd = _lib.Handle()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedClipAdamW needs contiguous fp32 parameters on the GPU (no CPU fallback)")
            states = [self._state_for(p) for p in ps]
            for st in states:        # a freshly loaded state_dict keeps moments on the parameter's device already
                if st["exp_avg"].device != ps[0].device:
                    st["exp_avg"] = st["exp_avg"].to(ps[0].device)
                    st["exp_avg_sq"] = st["exp_avg_sq"].to(ps[0].device)
            step = int(float(states[0]["step"])) + 1
            key = tuple((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr())
                        for p, st in zip(ps, states))
            if self._tables is None or self._tables[0] != key:
                n = len(ps)
                mk = lambda vals: (C.c_void_p * n)(*vals)
                self._tables = (key, mk([k[0] for k in key]), mk([k[1] for k in key]), mk([k[2] for k in key]),
                                mk([k[3] for k in key]), (C.c_int64 * n)(*[p.numel() for p in ps]))
            if self._norm is None or self._norm.device != ps[0].device:
                self._norm = torch.zeros(1, device=ps[0].device, dtype=torch.float32)
            _, tp, tg, tm, tv, tn = self._tables
            b1, b2 = group["betas"]
            stream = torch.cuda.current_stream(ps[0].device).cuda_stream
            self._hnd.clip_adamw_step(stream, tp, tg, tm, tv, tn, self._norm, self.max_norm, float(group["lr"]),
                                      float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), step)
            for st in states:
                st["step"] = torch.tensor(float(step))
        return None

    def total_grad_norm(self):
        """L2 norm of the (unclipped) gradients of the last step, as a device tensor (no host sync)."""
        return None if self._norm is None else self._norm.sqrt()
