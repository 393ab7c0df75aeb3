"""Fused Adam / AdamW: the optimizer update that closes the reference's training step
(ACM-Geometric/train.py:113-119 construct torch.optim.Adam / AdamW, :137 ``optimizer.step()``;
ACM-Pytorch/train.py:70-84, utils.py:572) as ONE kernel launch per 40 parameter tensors
(``acm_adam_step``) instead of torch's ~80 launches for the 26 parameters of the two-layer model.

Same constructor arguments, update formulas, ``state`` keys (``step`` / ``exp_avg`` /
``exp_avg_sq``, with ``step`` a device fp32 scalar as in torch's ``capturable=True`` mode) and
``state_dict`` layout as torch.optim.Adam / AdamW, so checkpoints interchange.  The launch reads
no host memory at run time, so a step that uses it can be captured in a hipGraph.  No CPU path.
"""
import ctypes as C

import torch

from . import _lib
from .graph import _device_ctx, _require_cuda, _stream


def _timed(name):
    from .functional import _Timed          # (functional imports nothing from here; late to keep import order free)
    return _Timed(name)


class _FusedAdamBase(torch.optim.Optimizer):
    _decoupled = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False,
                 maximize=False, **ignored):
        if amsgrad or maximize:
            raise NotImplementedError("FusedAdam: amsgrad / maximize are not implemented (the reference never uses them)")
        if isinstance(lr, torch.Tensor):
            raise TypeError("FusedAdam: lr must be a Python number (it is a kernel argument)")
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("FusedAdam: invalid hyper-parameter")
        # torch-only switches (foreach / fused / capturable / differentiable) are accepted and meaningless here
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        # optional device int64 counter incremented by the last launch of every step() (the DropoutState.step
        # of the model being trained: acm_adam_config_t.also_advance)
        self.also_advance = None
        self._tables = {}
        self._arrive = {}                     # per device: the zeroed int32 arrival counter of acm_adam_step

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}                     # the cached pointer tables refer to the replaced state tensors
        for p, st in self.state.items():
            self._normalize_state(p, st)

    @staticmethod
    def _normalize_state(p, st):
        """The kernel dereferences these on the device: after loading a torch.optim.Adam / AdamW checkpoint (`step` is
        a CPU tensor or a Python number there unless capturable=True), or one mapped to the CPU, move `step` to a
        0-dim fp32 tensor on the parameter's device and the moments to contiguous fp32 on it."""
        if not st:
            return
        step = st.get("step", 0.0)
        if not torch.is_tensor(step):
            st["step"] = torch.tensor(float(step), dtype=torch.float32, device=p.device)
        elif step.device != p.device or step.dtype != torch.float32 or step.dim() != 0:
            st["step"] = step.detach().to(device=p.device, dtype=torch.float32).reshape(()).clone()
        for key in ("exp_avg", "exp_avg_sq"):
            v = st.get(key)
            if v is None:
                st[key] = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif v.device != p.device or v.dtype != torch.float32 or not v.is_contiguous() or v.shape != p.shape:
                if v.numel() != p.numel():
                    raise RuntimeError(f"FusedAdam: state '{key}' has {v.numel()} elements, the parameter {p.numel()}")
                st[key] = v.detach().to(device=p.device, dtype=torch.float32).reshape(p.shape).contiguous()

    @torch.no_grad()
    def step(self, closure=None, pending=None):
        """``pending``: a functional.DeferredReductions the training loop has NOT flushed -- the update launch flushes it
        itself (acm_adam_config_t.pending: the reducing blocks lead the grid and apply the update to the gradient elements
        they produce; one launch less per step).  Only train.TrainStep passes it, and only when nothing else (a gradient
        all-reduce, a reader of .grad) has to come between the flush and the update."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        groups = [(g, [p for p in g["params"] if p.grad is not None]) for g in self.param_groups]
        groups = [(g, live) for g, live in groups if live]
        if pending is not None and len(groups) != 1:          # several launches (or none): the flush on its own, first
            pending.flush()
            pending = None
        if not groups and self.also_advance is not None:
            self.also_advance.add_(1)
        for gi, (group, live) in enumerate(groups):
            # the table of (param, moments, step) pointers is cached per set of live parameters: only the gradient
            # pointers change from step to step (the host side of an eager step is what bounds it)
            key = tuple(id(p) for p in live)
            cached = self._tables.get(gi)
            if cached is None or cached[0] != key:
                entries = (_lib.AdamTensor * len(live))()
                for e, p in zip(entries, live):
                    _require_cuda(p, "parameter")
                    if p.dtype != torch.float32 or not p.is_contiguous():
                        raise TypeError("FusedAdam: parameters must be contiguous fp32")
                    st = self.state[p]
                    if len(st) == 0:
                        st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    else:
                        self._normalize_state(p, st)
                    e.param = p.data_ptr()
                    e.exp_avg, e.exp_avg_sq, e.step = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr()
                    e.numel = p.numel()
                dev = live[0].device
                if any(p.device != dev for p in live):
                    raise RuntimeError("FusedAdam: one param group must live on one device")
                cached = (key, entries, [(p, self.state[p]) for p in live], dev)
                self._tables[gi] = cached
            _, entries, pairs, dev = cached
            keep = []
            for e, (p, st) in zip(entries, pairs):
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous() or g.is_sparse:
                    if g.is_sparse:
                        raise RuntimeError("FusedAdam does not support sparse gradients")
                    if pending is not None:               # a copy reads the gradient: its sums must have landed
                        pending.flush()
                        pending = None
                    g = g.to(torch.float32).contiguous()
                    keep.append(g)
                e.grad = g.data_ptr()
                # load_state_dict / .to() may have replaced the tensors behind the cached pointers
                if e.param != p.data_ptr() or e.exp_avg != st["exp_avg"].data_ptr():
                    e.param, e.exp_avg = p.data_ptr(), st["exp_avg"].data_ptr()
                    e.exp_avg_sq, e.step = st["exp_avg_sq"].data_ptr(), st["step"].data_ptr()
            arrive = self._arrive.get(dev)
            if arrive is None:
                arrive = self._arrive[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
            cfg = _lib.AdamConfig(float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]),
                                  float(group["eps"]), float(group["weight_decay"]), int(self._decoupled),
                                  self.also_advance.data_ptr() if (self.also_advance is not None and
                                                                   gi == len(groups) - 1) else None,
                                  arrive.data_ptr(), pending.pointer() if pending is not None else None)
            with _device_ctx(dev), _timed("adam" + ("+flush" if pending is not None and pending.pending else "")):
                status = lib.acm_adam_step(len(live), C.cast(entries, C.c_void_p), C.byref(cfg), _stream())
            _lib.check(status, "acm_adam_step")
            if pending is not None:
                pending.flushed()
            del keep
        return loss


class FusedAdam(_FusedAdamBase):
    """torch.optim.Adam (weight decay added to the gradient)."""
    _decoupled = False


class FusedAdamW(_FusedAdamBase):
    """torch.optim.AdamW (decoupled weight decay; default 1e-2 as in torch)."""
    _decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kw)
