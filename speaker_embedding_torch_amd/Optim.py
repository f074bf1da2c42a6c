"""Fused gradient clipping + AdamW on the device, with the reference's loss scaling.

`FusedClipAdamW` is a `torch.optim.AdamW` whose `step()` runs `clip_grad_norm_(max_norm)` and the AdamW
update of every parameter in two kernels of libge2e_hip.so (reference Train.py:154-162 runs them as ~10
multi-tensor launches).  State layout and `state_dict()` are torch.optim.AdamW's ('step', 'exp_avg',
'exp_avg_sq' per parameter), so reference checkpoints' 'Optimizer' entries load and save unchanged.
Arithmetic is torch's (decoupled decay, bias-corrected moments, eps added after the sqrt).

`GradScaler` has the call pattern of `torch.cuda.amp.GradScaler` as the reference uses it (Train.py:134,153-162:
`scaler.scale(loss).backward(); scaler.unscale_(opt); clip; scaler.step(opt); scaler.update()`), but its state lives
on the device and unscale / inf-check / clip / step / update all happen inside `ge2e_clip_adamw_step_scaled`: a
skipped step costs no host synchronisation (torch's GradScaler.step reads found_inf back to the host)."""
import ctypes as C

import torch

from . import _lib


class GradScaler:
    """Dynamic loss scaling for the fp16 arithmetic mode.  state = [scale, growth_tracker, found_inf, steps_taken] (device)."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self._enabled = bool(enabled)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self._init = (float(init_scale), 0.0, 0.0, 0.0)
        self._state = None

    def is_enabled(self):
        return self._enabled

    def state(self, device):
        if self._state is None or self._state.device != device:
            init = self._init if self._state is None else tuple(self._state.tolist())
            self._state = torch.tensor(init, dtype=torch.float32, device=device)
        return self._state

    def scale(self, loss):
        """loss x scale, on the device (no host read): backward then produces scale x the gradients."""
        if not self._enabled:
            return loss
        return loss * self.state(loss.device)[0]

    def backward(self, loss):
        """`scale(loss).backward()` without the two elementwise launches it costs (the multiply and autograd's ones_like root): the
        loss scale -- or a cached 1 -- is handed to backward as the upstream gradient, a device scalar the HIP loss backward reads."""
        if self._enabled:
            g = self.state(loss.device)[0]
        else:
            if getattr(self, "_one", None) is None or self._one.device != loss.device:
                self._one = torch.ones((), device=loss.device, dtype=torch.float32)
            g = self._one
        loss.backward(gradient=g.reshape(loss.shape))

    def unscale_(self, optimizer):
        """Accepted for call-pattern parity: the unscale is fused into step()."""
        return None

    def step(self, optimizer):
        if not self._enabled:
            return optimizer.step()
        return optimizer.step(scaler=self)

    def update(self):
        """Accepted for call-pattern parity: the scale is updated on the device at the end of step()."""
        return None

    def get_scale(self):
        """Host read (synchronises): for logging / tests only."""
        return float(self._init[0]) if self._state is None else float(self._state[0])

    def steps_taken(self):
        return 0 if self._state is None else int(self._state[3])

    def state_dict(self):
        if not self._enabled:
            return {}
        st = self._init if self._state is None else tuple(self._state.tolist())
        return {"scale": st[0], "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(st[1])}

    def load_state_dict(self, sd):
        if not self._enabled or not sd:
            return
        steps = 0.0 if self._state is None else float(self._state[3])
        self.growth_factor, self.backoff_factor = float(sd["growth_factor"]), float(sd["backoff_factor"])
        self.growth_interval = int(sd["growth_interval"])
        self._init = (float(sd["scale"]), float(sd.get("_growth_tracker", 0)), 0.0, steps)
        if self._state is not None:
            self._state.copy_(torch.tensor(self._init, dtype=torch.float32))


class FusedClipAdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.max_norm = float(max_norm)
        self._hnd = None
        self._hnd_dev = None
        self._norm = None
        self._tables = None          # cached pointer tables (rebuilt when a tensor moves)
        self._scaled_steps = None    # device step counter owner (a GradScaler) when steps may be skipped on the device

    def _state_for(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None, scaler=None):
        if closure is not None:
            raise RuntimeError("FusedClipAdamW does not take a closure")
        if scaler is not None and scaler.is_enabled() and sum(1 for g in self.param_groups if any(p.grad is not None for p in g["params"])) > 1:
            # the scaled step ends with ONE scale / growth-tracker / steps-taken update: with several groups that update would run once
            # per group, and an inf in a later group could not undo an earlier group's update (torch's GradScaler skips the whole step)
            raise RuntimeError("FusedClipAdamW: loss scaling supports one param_group (the reference's optimizer has one, Train.py:122-127)")
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedClipAdamW needs contiguous fp32 parameters on the GPU (no CPU fallback)")
            dev = ps[0].device
            if self._hnd is None or self._hnd_dev != dev:     # a handle belongs to the device of its first call
                self._hnd, self._hnd_dev = _lib.Handle(), dev
            states = [self._state_for(p) for p in ps]
            for st in states:        # a freshly loaded state_dict keeps moments on the parameter's device already
                if st["exp_avg"].device != dev:
                    st["exp_avg"] = st["exp_avg"].to(dev)
                    st["exp_avg_sq"] = st["exp_avg_sq"].to(dev)
            key = tuple((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr())
                        for p, st in zip(ps, states))
            if self._tables is None or self._tables[0] != key:
                n = len(ps)
                mk = lambda vals: (C.c_void_p * n)(*vals)
                self._tables = (key, mk([k[0] for k in key]), mk([k[1] for k in key]), mk([k[2] for k in key]),
                                mk([k[3] for k in key]), (C.c_int64 * n)(*[p.numel() for p in ps]))
            if self._norm is None or self._norm.device != dev:
                self._norm = torch.zeros(1, device=dev, dtype=torch.float32)
            _, tp, tg, tm, tv, tn = self._tables
            b1, b2 = group["betas"]
            with torch.cuda.device(dev):          # the library launches on the CURRENT device's stream
                stream = torch.cuda.current_stream(dev).cuda_stream
                if scaler is not None and scaler.is_enabled():
                    state = scaler.state(dev)
                    if self._scaled_steps is not scaler:      # continue from the host-side step count (e.g. a loaded checkpoint)
                        state[3] = float(states[0]["step"])
                        self._scaled_steps = scaler
                    self._hnd.clip_adamw_step_scaled(stream, tp, tg, tm, tv, tn, self._norm, self.max_norm, float(group["lr"]),
                                                     float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                                     state, scaler.growth_factor, scaler.backoff_factor, scaler.growth_interval)
                else:
                    step = int(float(states[0]["step"])) + 1
                    self._hnd.clip_adamw_step(stream, tp, tg, tm, tv, tn, self._norm, self.max_norm, float(group["lr"]),
                                              float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), step)
                    for st in states:
                        st["step"] = torch.tensor(float(step))
            # The kernels wrote the parameters through raw pointers: autograd's version counters did not see it.  Bump them so that
            # anything keyed on `p._version` (Modules._EncoderFn's prepared-weights key, saved-tensor checks) notices the update.
            torch._C._increment_version(ps)
        return None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._scaled_steps = None    # re-seed the device-side step counter from the loaded 'step' at the next scaled step
        self._tables = None

    def state_dict(self):
        """torch.optim.AdamW's layout.  Under device-side loss scaling the number of steps actually taken lives on the device
        (skipped steps do not count): it is read back here, the one place that needs it on the host."""
        if self._scaled_steps is not None:
            taken = float(self._scaled_steps.steps_taken())
            for st in self.state.values():
                if "step" in st:
                    st["step"] = torch.tensor(taken)
        return super().state_dict()

    def total_grad_norm(self):
        """L2 norm of the (unclipped; under loss scaling: still scaled) gradients of the last step, as a device tensor."""
        return None if self._norm is None else self._norm.sqrt()
