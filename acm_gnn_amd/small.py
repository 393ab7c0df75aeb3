"""The training step / evaluation pass of the two-layer ACM model on a SMALL graph as six (three) launches behind ONE
C-ABI call (``acm_small_step``, include/acm_hip.h; kernels: csrc/acm_small.hip).

Caller side of ACM-Pytorch/train.py:95-139 + utils.py:547-574 (and ACM-Geometric/train.py:119-140 on small data sets):
``train.TrainStep`` / ``train.EvalStep`` build a :class:`SmallPlan` when :func:`SmallPlan.why_not` finds nothing against it
-- Cora, Chameleon, Squirrel, Film ... with bag-of-words features (CSR) -- and then run a step as one ctypes call: no
autograd graph, no per-kernel Python dispatch, parameters updated in place by the kernels that finish their gradients
(the optimizer's own ``state`` tensors are used, so ``state_dict`` / checkpoints / a later switch to the general path see
exactly what ``optimizer.step()`` would have left).

Envelope: ``GCN`` of model_type acmgcn | acmgcnp, two layers, hidden width 64, <= 8 classes, <= 16384 nodes, pattern-only
operators on one device, CSR features (``graph.SparseFeatures``, or dense features the model's ``auto_csr`` turns into one),
counter-based dropout (or none), this package's FusedAdam / FusedAdamW with one parameter group.  Anything else stays on
the general path (``why_not`` says why).
"""
import ctypes as C

import torch

from . import _lib, tuning
from .graph import FilterOperators, SparseFeatures, _device_ctx, _stream

_F32 = torch.float32
_ON_DEVICE = lambda t: t.is_cuda          # noqa: E731  (the CPU test double of the library lifts this guard)
MAX_ROWS = 16384
HIDDEN = 64
MAX_CLASSES = 8

_ROLE_NAMES = {
    _lib.SR_W_LOW: "weight_low", _lib.SR_W_HIGH: "weight_high", _lib.SR_W_MLP: "weight_mlp",
    _lib.SR_V_LOW: "att_vec_low", _lib.SR_V_HIGH: "att_vec_high", _lib.SR_V_MLP: "att_vec_mlp", _lib.SR_V_STRUC: "att_struc_low",
    _lib.SR_LNW_LOW: "layer_norm_low.weight", _lib.SR_LNW_HIGH: "layer_norm_high.weight", _lib.SR_LNW_MLP: "layer_norm_mlp.weight",
    _lib.SR_LNW_STRUC: "layer_norm_struc_low.weight",
    _lib.SR_LNB_LOW: "layer_norm_low.bias", _lib.SR_LNB_HIGH: "layer_norm_high.bias", _lib.SR_LNB_MLP: "layer_norm_mlp.bias",
    _lib.SR_LNB_STRUC: "layer_norm_struc_low.bias",
    _lib.SR_MIX: "att_vec", _lib.SR_STRUC: "struc_low",
}


def _roles(cfg):
    """The roles of a layer that take part in the forward (and so take a gradient) under ``cfg``."""
    four = cfg.n_channels == 4
    roles = [_lib.SR_W_LOW, _lib.SR_W_HIGH, _lib.SR_W_MLP, _lib.SR_V_LOW, _lib.SR_V_HIGH, _lib.SR_V_MLP, _lib.SR_MIX]
    if four:
        roles += [_lib.SR_V_STRUC, _lib.SR_STRUC]
    if cfg.layernorm:
        roles += [_lib.SR_LNW_LOW, _lib.SR_LNW_HIGH, _lib.SR_LNW_MLP, _lib.SR_LNB_LOW, _lib.SR_LNB_HIGH, _lib.SR_LNB_MLP]
        if four:
            roles += [_lib.SR_LNW_STRUC, _lib.SR_LNB_STRUC]
    return roles


def _param(layer, role):
    obj = layer
    for part in _ROLE_NAMES[role].split("."):
        obj = getattr(obj, part)
    return obj


def _has_hooks(model, optimizer=None):
    """Forward / backward hooks on any module, gradient hooks on any parameter, step hooks on the optimizer."""
    for m in model.modules():
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None):
            return True
    for p in model.parameters():
        if getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None):
            return True
    if optimizer is not None:
        if getattr(optimizer, "_optimizer_step_pre_hooks", None) or getattr(optimizer, "_optimizer_step_post_hooks", None):
            return True
    return False


class SmallPlan:
    """Everything ``acm_small_step`` needs, bound once to (model, optimizer, features, operators, labels, row weights).

    What a caller of TrainStep / EvalStep does NOT get on this path (by construction: the step is six kernels, not a forward,
    a backward and an optimizer call): ``p.grad`` stays None (``keep_grads=True`` writes the gradients to ``plan.grads``
    instead), ``model.forward`` and ``optimizer.step`` are not called -- hooks would not fire, so ``why_not`` refuses a
    hooked model / parameter / optimizer -- and hyper-parameters are read on every call (eager) or baked in at capture
    (``use_graph``: like the general path, re-capture after a scheduler changed ``lr``).  Parameters and the optimizer's
    own state tensors are updated in place: ``state_dict()`` of both is what a stock loop would have left."""

    @staticmethod
    def why_not(model, x, ops, optimizer=None, need_dropout_state=True):
        """None if the fused small-graph step applies, else the first reason it does not (a string)."""
        from .optim import _FusedAdamBase
        if tuning.HOST.small_step <= 0:
            return "switched off (tuning small_step=0)"
        if getattr(model, "model_type", None) not in ("acmgcn", "acmgcnp"):
            return f"model_type {getattr(model, 'model_type', None)!r} (acmgcn | acmgcnp)"
        gcns = getattr(model, "gcns", None)
        if gcns is None or len(gcns) != 2:
            return "not a two-layer model"
        l0, l1 = gcns
        if l0.out_features != HIDDEN or l1.in_features != HIDDEN:
            return f"hidden width {l0.out_features} (64)"
        if not 1 <= l1.out_features <= MAX_CLASSES:
            return f"{l1.out_features} classes (<= {MAX_CLASSES})"
        c0, c1 = l0._config(), l1._config()
        if (c0.n_channels, c0.relu_before, c0.layernorm, c0.gather_bf16) != (c1.n_channels, c1.relu_before, c1.layernorm, c1.gather_bf16):
            return "the two layers are configured differently"
        if c0.gather_bf16:
            return "bf16 gather tables"
        if not isinstance(ops, FilterOperators) or ops.sharded or getattr(ops, "general", False) or not ops.implicit:
            return "operators: one device, pattern-only form"
        if ops.perm is not None or int(getattr(ops, "hops", 1)) != 1:
            return "relabelled / k-hop operators"
        n = ops.low.n_rows
        if not 1 <= n <= min(MAX_ROWS, tuning.HOST.small_step) or ops.low.n_cols != n:
            return f"{n} rows (<= {min(MAX_ROWS, tuning.HOST.small_step)})"
        if c0.n_channels == 4 and ops.deg is None:
            return "structure channel without degrees"
        if not isinstance(x, SparseFeatures):
            return "dense features (CSR features only)"
        if x.shape != (n, l0.in_features) or x.values.dtype != _F32 or not _ON_DEVICE(x.values):
            return "feature matrix shape / dtype / device"
        dev = x.values.device
        if any(p.device != dev or p.dtype != _F32 or not p.is_contiguous() for p in model.parameters()):
            return "parameters: contiguous fp32 on the features' device"
        if l0.struc_low.shape[0] != n and c0.n_channels == 4:
            return "struc_low rows"
        p_drop = float(getattr(model, "dropout", 0.0))
        if p_drop > 0 and need_dropout_state and not getattr(model, "fused_dropout", False):
            return "F.dropout masks (counter-based dropout only)"
        if _has_hooks(model, optimizer):
            # the fused step never calls model.forward, never materialises a .grad and never calls optimizer.step():
            # nothing a hook is attached to happens, so a hooked model / parameter / optimizer keeps the path
            # on which its hooks fire
            return "hooks registered on the model, a parameter or the optimizer"
        if optimizer is not None:
            if not isinstance(optimizer, _FusedAdamBase) or len(optimizer.param_groups) != 1:
                return "optimizer: FusedAdam / FusedAdamW with one parameter group"
            have = {id(p) for p in optimizer.param_groups[0]["params"]}
            for layer, cfg in ((l0, c0), (l1, c1)):
                if any(id(_param(layer, r)) not in have for r in _roles(cfg)):
                    return "a parameter of the model is not in the optimizer's group"
        return None

    def __init__(self, model, x, ops, labels=None, weights=None, optimizer=None, update=True, keep_grads=False):
        """``optimizer`` None: an evaluation plan (forward only).  ``update=False``: the step stops at the gradients
        (``self.grads[(layer, role name)]``; nothing is updated, no optimizer state is touched); ``keep_grads``: an
        updating plan that ALSO writes every gradient there (tests; the kernels store them on the way)."""
        lib = _lib.load()
        self.model, self.x, self.ops, self.opt = model, x, ops, optimizer
        l0, l1 = model.gcns
        cfg = l0._config()
        self.cfg = cfg
        n, dev = ops.low.n_rows, x.values.device
        self.n, self.C = n, l1.out_features
        self.train = labels is not None                  # labels + row weights: a training plan; neither: evaluation
        self._xt = x.csr_t if self.train else None
        self.low = self._operator(ops)
        nbytes = C.c_size_t()
        _lib.check(lib.acm_small_step_workspace_bytes(self.low.handle, x.csr.handle, self._xt.handle if self._xt is not None else None,
                                                      C.byref(nbytes)), "acm_small_step_workspace_bytes")
        self.workspace = torch.zeros(nbytes.value // 4, dtype=_F32, device=dev)     # zero once: counters, pad columns
        self.logits = torch.empty(n, self.C, dtype=_F32, device=dev)
        self.att1 = torch.zeros(n, 4, dtype=_F32, device=dev)
        self.att2 = torch.zeros(n, 4, dtype=_F32, device=dev)
        self.loss = torch.zeros((), dtype=_F32, device=dev)
        self.arrive = torch.zeros(1, dtype=torch.int32, device=dev)
        self.labels = labels.to(torch.int64).contiguous() if labels is not None else None
        self.weights = weights.to(_F32).contiguous() if weights is not None else None
        self._keep = []
        p = self.p = _lib.SmallStep()
        p.n_classes, p.n_channels, p.relu_before, p.layernorm = self.C, cfg.n_channels, int(cfg.relu_before), int(cfg.layernorm)
        self.update = bool(update) and self.train and optimizer is not None
        self.grads = {}
        if self.train and (keep_grads or not self.update):
            for li, layer in enumerate(model.gcns):
                for role in _roles(cfg):
                    self.grads[(li, _ROLE_NAMES[role])] = torch.zeros_like(_param(layer, role))
        p.scale, p.train, p.update = cfg.scale, int(self.train), int(self.update)
        p.f_in = l0.in_features
        p.x_vals = x.values.data_ptr()
        if self._xt is not None:
            p.xt_src_pos = self._xt._src_pos_ptr
            self.xt_vals = x.values.index_select(0, self._xt.src_pos)      # the (static) values in the transposed order
            p.xt_vals = self.xt_vals.data_ptr()
        p.row_scale = ops.row_scale.data_ptr()
        p.logits, p.att1, p.att2, p.loss = self.logits.data_ptr(), self.att1.data_ptr(), self.att2.data_ptr(), self.loss.data_ptr()
        p.arrive = self.arrive.data_ptr()
        p.workspace, p.workspace_bytes = self.workspace.data_ptr(), self.workspace.numel() * 4
        if self.train:
            p.labels, p.row_weight = self.labels.data_ptr(), self.weights.data_ptr()
        self._bind_parameters()
        self.refresh_hyper()

    @staticmethod
    def _operator(ops):
        """The operator handle the six launches walk: ops.low itself, or a copy of its pattern cut into work items of the
        length that suits one-item-per-wave kernels (made once per operator set, kept on it).  Measured (captured ms per step,
        chunk 128 / 256 / 512): Squirrel 0.192 / 0.168 / 0.179, Chameleon 0.102 / 0.114 / 0.130, Cora (longest row 168) 0.086 /
        0.079 / 0.078 -- short rows want no pieces at all, long rows on a dense graph want pieces of ~3 x the mean degree."""
        low = ops.low
        n, nnz = low.n_rows, low.nnz
        if low.max_degree <= 256:
            want = 128 if low.max_degree <= 128 else 256
        else:
            want = 128 if nnz < 48 * n else 256
        if int(low.chunk) == want or low.n_long_rows == 0 and low.max_degree <= want:
            return low
        cache = ops.__dict__.setdefault("_small_low", {})
        if want not in cache:
            from .graph import CsrGraph
            ip, ix, _ = low.arrays()
            cache[want] = CsrGraph.from_csr(ip, ix, None, low.n_cols, chunk=want)
        return cache[want]

    def _bind_parameters(self):
        p = self.p
        for li, layer in enumerate(self.model.gcns):
            for role in _roles(self.cfg):
                t = _param(layer, role)
                e = p.t[li][role]
                e.param, e.numel = t.data_ptr(), t.numel()
                g = self.grads.get((li, _ROLE_NAMES[role]))
                e.grad = g.data_ptr() if g is not None else None
                if self.update:
                    st = self.opt.state[t]
                    if len(st) == 0:
                        st["step"] = torch.zeros((), dtype=_F32, device=t.device)
                        st["exp_avg"] = torch.zeros_like(t, memory_format=torch.preserve_format)
                        st["exp_avg_sq"] = torch.zeros_like(t, memory_format=torch.preserve_format)
                    else:
                        self.opt._normalize_state(t, st)
                    e.exp_avg, e.exp_avg_sq, e.step = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr()
        self._bound = self._binding_key()

    def _binding_key(self):
        key = []
        for layer in self.model.gcns:
            for role in _roles(self.cfg):
                t = _param(layer, role)
                st = self.opt.state.get(t, {}) if self.update else {}
                key.append((t.data_ptr(), st["exp_avg"].data_ptr() if st else 0, st["step"].data_ptr() if st else 0))
        return tuple(key)

    def refresh_hyper(self):
        """Hyper-parameters and dropout state as they are NOW (a scheduler may have changed lr; a loop sets the dropout
        state after constructing the model)."""
        p, model = self.p, self.model
        if self.update:
            g = self.opt.param_groups[0]
            p.lr, p.beta1, p.beta2, p.eps = float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])
            p.weight_decay, p.decoupled = float(g["weight_decay"]), int(self.opt._decoupled)
            adv = getattr(self.opt, "also_advance", None)
            p.also_advance = adv.data_ptr() if adv is not None else None
        pd = float(getattr(model, "dropout", 0.0))
        st = getattr(model, "dropout_state", None)
        training = self.train
        if training and pd > 0:
            if st is None:
                raise RuntimeError("SmallPlan: the model has dropout but no dropout_state (counter-based dropout only)")
            p.drop_in, p.drop_hidden = st.spec(pd, 0, 0), st.spec(pd, 1, self.ops.row_offset)
        else:
            p.drop_in, p.drop_hidden = _lib.Dropout(), _lib.Dropout()

    def stale(self):
        """A parameter or an optimizer state tensor was replaced (load_state_dict, .to()): bind again."""
        return self._binding_key() != self._bound

    def run(self):
        """One call = the step (training plan) or the forward (evaluation plan).  Returns the loss tensor / the logits."""
        if self.stale():
            self._bind_parameters()
        self.refresh_hyper()
        from .functional import _Timed
        dev = self.logits.device
        with _device_ctx(dev), _Timed("small_step" if self.train else "small_forward"):
            st = _lib.load().acm_small_step(self.low.handle, self.x.csr.handle, self._xt.handle if self._xt is not None else None,
                                            C.byref(self.p), _stream())
        _lib.check(st, "acm_small_step")
        l0, l1 = self.model.gcns
        for layer, att in ((l0, self.att1), (l1, self.att2)):
            layer._att_raw, layer._att_inv, layer._att_k = att, None, self.cfg.n_channels
        return self.loss if self.train else self.logits
