"""Operator-level host code: thin wrappers over the C ABI and the autograd
Functions that make the HIP kernels differentiable.

Every function here launches HIP kernels from libacm_hip.so on the current
torch stream.  There is deliberately no eager/torch fallback: a CPU tensor or a
missing library is an error.
"""
import ctypes as C
import threading

import torch

from . import _lib, tuning
from .graph import CsrGraph, FilterOperators, SparseFeatures, _device_ctx, _require_cuda, _stream

_F32 = torch.float32


class KernelTimer:
    """Per-launch HIP-event timing (torch.cuda.Event on the stream the kernels are launched
    on).  ``only`` restricts timing to labels that start with it, so that the events around the
    one kernel under study do not perturb the rest of a timed region.

    ``external=True``: launches made while the stream is being CAPTURED are bracketed by *external* events (event-record
    nodes of the hipGraph, added through the HIP graph API: _HipEvent): every replay re-records them, so after a replay and a synchronize
    ``captured_ms(label)`` is that kernel's duration INSIDE the replayed graph -- the time that belongs next to a
    hipGraph-replay ``ms_per_step`` (bench.py's roofline).  Without it, launches under capture are not
    timed at all (a plain event recorded in a capture cannot be read)."""

    def __init__(self, only=None, external=False):
        self.only, self.events = only, {}
        self.external, self.captured = bool(external), {}
        self.bytes = {}                 # label -> bytes this rank SENT into collectives under that label

    def wants(self, label):
        return self.only is None or label.startswith(self.only)

    def summary(self):
        """label -> (launches, total ms); synchronises."""
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in self.events.items()}

    def captured_ms(self, label):
        """ms of every captured launch under ``label`` in the LAST replay of the graph(s) that hold them; synchronises."""
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in self.captured.get(label, [])]


class _HipEvent:
    """A timing hipEvent_t recorded by an EVENT-RECORD NODE of the hipGraph the current stream is being captured into:
    hipStreamGetCaptureInfo_v2 -> hipGraphAddEventRecordNode behind the capture's current dependencies ->
    hipStreamUpdateCaptureDependencies (ROCm 7.2 rejects hipEventRecordWithFlags(hipEventRecordExternal), hipError 1,
    and torch refuses Event(external=True) on ROCm; the explicit node is what both stand for)."""
    _hip = None

    def __init__(self):
        if _HipEvent._hip is None:
            hip = C.CDLL("libamdhip64.so")
            hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
            hip.hipStreamGetCaptureInfo_v2.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong),
                                                       C.POINTER(C.c_void_p), C.POINTER(C.POINTER(C.c_void_p)),
                                                       C.POINTER(C.c_size_t)]
            hip.hipGraphAddEventRecordNode.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t,
                                                       C.c_void_p]
            hip.hipStreamUpdateCaptureDependencies.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
            hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
            hip.hipEventDestroy.argtypes = [C.c_void_p]
            _HipEvent._hip = hip
        self.handle = C.c_void_p()
        self._check(self._hip.hipEventCreate(C.byref(self.handle)), "hipEventCreate")

    @staticmethod
    def _check(status, what):
        if status != 0:
            raise RuntimeError(f"{what} failed with hipError_t {status}")

    def record(self):
        hip, stream = self._hip, _stream()
        status, cap_id, graph = C.c_int(), C.c_ulonglong(), C.c_void_p()
        deps, n_deps = C.POINTER(C.c_void_p)(), C.c_size_t()
        self._check(hip.hipStreamGetCaptureInfo_v2(stream, C.byref(status), C.byref(cap_id), C.byref(graph), C.byref(deps),
                                                   C.byref(n_deps)), "hipStreamGetCaptureInfo_v2")
        if status.value != 1:                          # hipStreamCaptureStatusActive
            raise RuntimeError("_HipEvent.record: the current stream is not being captured")
        node = C.c_void_p()
        self._check(hip.hipGraphAddEventRecordNode(C.byref(node), graph, deps, n_deps.value, self.handle),
                    "hipGraphAddEventRecordNode")
        self._check(hip.hipStreamUpdateCaptureDependencies(stream, C.byref(node), 1, 1),     # 1 = set (replace) dependencies
                    "hipStreamUpdateCaptureDependencies")

    def elapsed_time(self, other):
        ms = C.c_float()
        self._check(self._hip.hipEventElapsedTime(C.byref(ms), self.handle, other.handle), "hipEventElapsedTime")
        return ms.value

    def __del__(self):
        hip = getattr(type(self), "_hip", None)            # (module globals are gone at interpreter shutdown)
        if self.handle and hip is not None:
            hip.hipEventDestroy(self.handle)


_TIMER = None


def set_kernel_timer(timer):
    global _TIMER
    _TIMER = timer


class _Timed:
    def __init__(self, label, nbytes=0):
        self.label = label
        self.on = _TIMER is not None and _TIMER.wants(label)
        self.capturing = False
        if self.on and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            self.capturing = True
            self.on = _TIMER.external
        if self.on and nbytes:
            _TIMER.bytes[label] = _TIMER.bytes.get(label, 0) + int(nbytes)

    def __enter__(self):
        if self.on:
            if self.capturing:
                self.a, self.b = _HipEvent(), _HipEvent()
            else:
                self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.b.record()
            (_TIMER.captured if self.capturing else _TIMER.events).setdefault(self.label, []).append((self.a, self.b))
        return False


class DeferredReductions:
    """The pending second phases (acm_reduce_list_t) of the calls made while this object is current: the loss sum and
    the parameter-gradient sums of K3 / acm_proj_bwd / acm_conv_agg_bwd.  Until :meth:`flush` their outputs (the
    loss, every ``.grad`` those kernels produce) are UNDEFINED, so only a loop that owns the whole step may defer
    (``train.TrainStep`` does: nothing reads a gradient between ``backward`` and the flush it issues before the
    optimizer / the gradient all-reduce).  One launch instead of four per step (~5 us each in a replayed graph)."""
    CAP = 96

    def __init__(self):
        self._segs = (_lib.ReduceSeg * self.CAP)()
        self._list = _lib.ReduceList(0, self.CAP, C.cast(self._segs, C.POINTER(_lib.ReduceSeg)))
        self._keep = []                 # workspaces the pending segments read
        self.outputs = []               # (data_ptr, nbytes) of tensors whose content arrives with the flush
        self._allreduce = []            # (tensor, group): row-shard gradient sums to all-reduce right after the flush

    def pointer(self):
        return C.addressof(self._list)

    def hold(self, workspace, outputs, keep=None):
        """A deferred call's workspace, the tensors its second phase will write (`outputs`: address ranges for
        all_adopted) and what must stay allocated for that write (`keep`, default the outputs themselves; pass the
        base buffer when the outputs are views autograd should still be able to adopt -- a held view has one
        reference too many for that)."""
        self._keep.append(workspace)
        self._keep.extend(outputs if keep is None else keep)
        self.outputs.extend((t.data_ptr(), t.numel() * t.element_size()) for t in outputs if t is not None)

    @property
    def pending(self):
        return int(self._list.n)

    def all_adopted(self, tensors):
        """True if every output of a deferred call is (the storage of) at least one of `tensors` -- e.g. the loss and
        the ``.grad`` of every parameter: autograd adopted what the kernels wrote instead of copying or accumulating
        it before the flush."""
        ptrs = sorted(t.data_ptr() for t in tensors if t is not None)
        import bisect
        for a, nb in self.outputs:
            i = bisect.bisect_left(ptrs, a)
            if i == len(ptrs) or ptrs[i] >= a + nb:
                return False
        return True

    def allreduce(self, tensor, group):
        """Row-sharded runs: `tensor` holds this rank's partial sums of replicated-parameter gradients, complete only
        after the flush.  Instead of flushing and all-reducing at every layer's backward, the tensors are collected and
        summed over the ranks right after the ONE flush of the step -- as one coalesced collective where the backend
        offers it (RCCL), one after the other otherwise."""
        self._allreduce.append((tensor, group))

    def flush(self):
        if self._list.n:
            dev = self._keep[0].device
            with _device_ctx(dev), _Timed("reduce_flush"):
                st = _lib.load().acm_reduce_flush(C.byref(self._list), _stream())
            _lib.check(st, "acm_reduce_flush")
        if self._allreduce:
            import torch.distributed as dist
            pending, self._allreduce = self._allreduce, []
            group = pending[0][1]
            coalesce = (len(pending) > 1 and all(g is group for _, g in pending) and hasattr(dist, "_coalescing_manager")
                        and dist.get_backend(group) == "nccl")
            with _Timed("all_reduce/gradients", sum(t.numel() * t.element_size() for t, _ in pending)):
                if coalesce:
                    with dist._coalescing_manager(group=group, device=pending[0][0].device, async_ops=False):
                        for t, _ in pending:
                            dist.all_reduce(t, group=group)
                else:
                    for t, g in pending:
                        dist.all_reduce(t, group=g)
        self._keep.clear()

    @property
    def collectives_pending(self):
        return bool(self._allreduce)

    def flushed(self):
        """The list was flushed by someone else (acm_adam_step with acm_adam_config_t.pending): release what it held."""
        assert self._list.n == 0 and not self._allreduce
        self._keep.clear()

    def discard(self):
        self._list.n = 0
        self._keep.clear()
        self._allreduce = []


class CallContext:
    """What a caller hands DOWN one forward (and, captured by the autograd Functions, its backward) beside the tensors:

    * ``defer``     a DeferredReductions the second phases of the call's kernels are appended to (or None: reduce at once)
    * ``tail``      a pending fused_loss_tail request for the model's output layer
    * ``pipe``      the training loop's InputPipeline
    * ``next_proj`` / ``pre_proj``  the narrow-projection hand-off between a hidden layer and the layer that follows it
    * ``hidden_private``  the hidden activation tensor only the following layer consumes (backward-side hand-off)

    A context belongs to ONE model call: models.GCN / layers.GraphConvolution take it as ``call=`` (train.TrainStep passes
    its own) or derive a fresh one from the thread's ambient context -- what ``with deferred_reductions()`` /
    ``with fused_loss_tail()`` blocks of the calling thread have set.  Nothing lives in module globals: two models, two
    threads or an exception between two layers cannot see each other's hand-offs, and a backward (which autograd may run on
    another thread) uses the context its forward captured."""
    __slots__ = ("defer", "tail", "pipe", "next_proj", "pre_proj", "hidden_private")

    def __init__(self, defer=None, tail=None, pipe=None):
        self.defer, self.tail, self.pipe = defer, tail, pipe
        self.next_proj = self.pre_proj = None
        # set by a model around the call of a layer whose input tensor is the previous layer's output and is used NOWHERE
        # else (models.GCN: the hidden activations of the two-layer models): that layer's backward may then leave its input
        # gradient to the previous layer's backward kernel (acm_conv_agg_bwd_t.proj_*) instead of materialising it
        self.hidden_private = None

    @classmethod
    def from_ambient(cls):
        a = _ambient()
        return cls(a.defer, a.tail, a.pipe)

    def defer_ptr(self):
        return self.defer.pointer() if self.defer is not None else None


_TLS = threading.local()


def _ambient():
    """The calling thread's ambient context (created on first use)."""
    c = getattr(_TLS, "call", None)
    if c is None:
        c = _TLS.call = CallContext()
    return c


def _call_or_ambient(call):
    return call if call is not None else CallContext.from_ambient()


class TapeBroken(RuntimeError):
    """A step cannot run on a Tape (see there): something other than this package's Functions sits between them."""


class _TapeCtx:
    """What a Function's forward / backward see in place of autograd's ctx when the call runs on a Tape."""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = needs_input_grad
        self.saved_tensors = ()
        self.non_differentiable = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        # (the mixing weights ``att``: Tape.run neither tags them nor gives them requires_grad -- a layer that keeps its
        # ``att`` around must not keep the step's saved activations alive with it)
        self.non_differentiable = tuple(id(t) for t in tensors)

    def release(self):
        """Drop everything the record holds (saved activations, hand-offs parked on the ctx by the Functions)."""
        self.__dict__.clear()
        self.saved_tensors = ()

    def set_materialize_grads(self, value):
        pass                                  # (Tape.backward hands None to outputs without a gradient, as the Functions ask)


class Tape:
    """The training step's own record of the Functions it runs, replayed backwards by ``backward`` -- instead of an autograd
    graph (train.TrainStep, eager steps).  A two-layer step is two or three of this package's Functions in a row with 26
    parameters: autograd spends more host time on them (``Function.apply``, 26 AccumulateGrad nodes, gradient validation,
    the engine's thread hand-off: ~0.3 ms) than the kernels take.  The tape calls ``forward(ctx, *args)`` under no_grad,
    remembers (Function, ctx, args, outputs), and ``backward(out, grad)`` walks the records in reverse: output gradients
    by tensor identity, parameter gradients assigned (added, when a parameter is used twice) to ``.grad``.

    It is only sound while every tensor that carries a gradient between two records is the very object a record returned:
    outputs are marked ``requires_grad`` so that a torch op in between leaves a ``grad_fn`` on its result, and such an argument
    (or a final output the tape did not produce) raises TapeBroken -- the step then runs on autograd, for good."""

    def __init__(self):
        self.records = []
        self.produced = {}                    # id(tensor) -> tensor (kept alive: an id must not be handed to another object)
        self.ctx_of = {}                      # id(tensor) -> the record's ctx: the stand-in for ``tensor.grad_fn`` of the
                                              # lazy-gradient hand-off.  On the TAPE, not on the tensor: a tensor that outlives
                                              # the step (a layer's ``att``, an output a caller keeps) must not pin the step's
                                              # saved activations, and ``out -> ctx -> saved out`` must not be a cycle only the
                                              # cyclic collector frees (live bytes grew 1.5 -> 8.9 MB in ten steps)

    def run(self, fn, args):
        needs = []
        for a in args:
            if not isinstance(a, torch.Tensor):
                needs.append(False)
            elif id(a) in self.produced:
                needs.append(True)
            elif a.grad_fn is not None:
                raise TapeBroken(f"{fn.__name__}: an argument was computed by torch operations from tensors that need gradients")
            else:
                needs.append(bool(a.requires_grad))          # a parameter, or a constant (features, a mask tensor)
        ctx = _TapeCtx(tuple(needs))
        with torch.no_grad():
            out = fn.forward(ctx, *args)
        outs = out if isinstance(out, tuple) else (out,)
        if not any(needs):                        # as under autograd: nothing to differentiate, the outputs are constants
            return out                            # (the dropped copy of the input features)
        for o in outs:
            if isinstance(o, torch.Tensor) and o.is_floating_point() and id(o) not in ctx.non_differentiable:
                o.requires_grad_(True)
                self.ctx_of[id(o)] = ctx
                self.produced[id(o)] = o
        self.records.append((fn, ctx, args, outs))
        return out

    def producer(self, t):
        """The ctx of the record that produced ``t`` on this tape (None: not a differentiable output of one)."""
        return self.ctx_of.get(id(t)) if self.produced.get(id(t)) is t else None

    def release(self):
        """End of the step (or of the attempt): every record lets go of what it saved."""
        for _, ctx, _, _ in self.records:
            ctx.release()
        self.records, self.produced, self.ctx_of = [], {}, {}

    def backward(self, out, grad):
        if id(out) not in self.produced:
            raise TapeBroken("the model's output is not the output of one of this package's Functions")
        grads = {id(out): grad}
        try:
            self._backward(grads)
        finally:
            self.release()

    def _backward(self, grads):
        with torch.no_grad():
            for fn, ctx, args, outs in reversed(self.records):
                gouts = tuple(grads.pop(id(o), None) if isinstance(o, torch.Tensor) else None for o in outs)
                if all(g is None for g in gouts):
                    continue
                gins = fn.backward(ctx, *gouts)
                if not isinstance(gins, tuple):
                    gins = (gins,)
                for a, g in zip(args, gins):
                    if g is None or not isinstance(a, torch.Tensor):
                        continue
                    if id(a) in self.produced:
                        grads[id(a)] = g if id(a) not in grads else grads[id(a)] + g
                    elif a.requires_grad:
                        if g.stride() != a.stride():          # AccumulateGrad's layout contract: a strided view is copied (a
                            g = g.contiguous()                # deferred output copied too early fails the step's all_adopted check)
                        if a.grad is None:
                            a.grad = g
                        else:
                            a.grad.add_(g)                    # (in place, as AccumulateGrad does without a graph)


class on_tape:
    """``with on_tape(tape): out = model(...)`` -- the Functions the calling thread runs inside go to ``tape``."""

    def __init__(self, tape):
        self.tape = tape

    def __enter__(self):
        self.prev = getattr(_TLS, "tape", None)
        _TLS.tape = self.tape
        return self.tape

    def __exit__(self, *exc):
        _TLS.tape = self.prev
        return False


def _run(fn, *args):
    """``fn.apply(*args)``, or the same call on the calling thread's Tape."""
    tape = getattr(_TLS, "tape", None)
    return fn.apply(*args) if tape is None else tape.run(fn, args)


class deferred_reductions:
    """``with deferred_reductions() as d: ...; d.flush()`` -- the calls the thread makes inside (and the backward of what
    it ran forward inside) append their second phases to ``d``.  Leaving the block flushes whatever is still pending (or
    drops it when an exception is propagating)."""

    def __enter__(self):
        a = _ambient()
        self.prev, self.d = a.defer, DeferredReductions()
        a.defer = self.d
        return self.d

    def __exit__(self, exc_type, *rest):
        _ambient().defer = self.prev
        if exc_type is None:
            self.d.flush()
        else:
            self.d.discard()
        return False


class deferred_reductions_as:
    """``with deferred_reductions_as(d): ...`` -- make an existing DeferredReductions (or None) the thread's ambient one;
    the caller keeps the responsibility to flush or discard it."""

    def __init__(self, d):
        self.d = d

    def __enter__(self):
        a = _ambient()
        self.prev, a.defer = a.defer, self.d
        return self.d

    def __exit__(self, *exc):
        _ambient().defer = self.prev
        return False


class fused_loss_tail:
    """``with fused_loss_tail(labels, row_weight) as tail: out = model(...)`` -- a request to the model's OUTPUT layer
    (a narrow three-channel ACM layer called with ``output_layer`` set and no post-op) to run its row phase, the
    masked NLL of its logits and its own row-local backward (K3) as ONE kernel (acm_conv_fwd_tail).  Afterwards
    ``tail.matches(out)`` says whether ``out`` is that layer's output; if so ``tail.loss`` / ``tail.dz`` are what
    nll_loss_and_grad(out, labels, row_weight) returns, and ``out.backward(tail.dz)`` skips K3.  Any other gradient
    handed to that layer's backward makes it run K3 as usual, so a request that is not honoured costs nothing but time."""

    def __init__(self, labels, row_weight):
        self.labels, self.row_weight = labels, row_weight
        self.loss = self.dz = self.out = None

    def __enter__(self):
        a = _ambient()
        self.prev = a.tail
        a.tail = self
        return self

    def __exit__(self, *exc):
        _ambient().tail = self.prev
        return False

    def matches(self, out):
        return self.out is not None and out is not None and out.data_ptr() == self.out.data_ptr() \
            and out.shape == self.out.shape


class input_pipeline:
    """``with input_pipeline(pipe): ...`` -- the thread's model calls inside see the training loop's InputPipeline."""

    def __init__(self, pipe):
        self.pipe = pipe

    def __enter__(self):
        a = _ambient()
        self.prev = a.pipe
        a.pipe = self.pipe
        return self.pipe

    def __exit__(self, *exc):
        _ambient().pipe = self.prev
        return False


class InputPipeline:
    """The first layer's input aggregation P_t = A_low dropout_t(x), computed one optimizer step AHEAD inside the layer's
    own backward.  x is constant and the mask of step t + 1 is a function of the step counter (acm_dropout_t.step_offset),
    so P_{t+1} depends on nothing step t computes; the row-local backward kernel of the aggregate-first layer is bound
    by its vector instructions and leaves the memory system idle, the narrow gather is bound by memory latency -- in one
    kernel (two extra waves per workgroup) they overlap, as two kernels they do not.

    Buffers (fixed addresses: a captured graph keeps them): ``filled`` = [dropout(x) padded to 8 columns | P] for the step
    about to run, ``saved`` = the copy of both that the forward's row-local kernel leaves for the backward
    (acm_conv_agg_fwd_t.agg_copy / xs_copy: it reads those rows anyway).  A step runs
    forward (reads filled, writes saved) -> make_next() (filled table <- dropout_{t+1}(x)) -> backward (reads saved;
    its gather waves read the filled table and write the filled P) -> optimizer -> end_step().
    prime() fills ``filled`` for the current counter value; it must be called again whenever the counter is set from
    outside (train.TrainStep does after its warm-up) or x is modified in place (``stale()`` notices the latter between
    eager steps; a captured graph replays without host code, so there the caller has to)."""

    def __init__(self, ops, x, p, state, tag=0):
        # Row-sharded with equal blocks and the replicated input registered (ops.x_full): every rank draws the dropped
        # input of ALL nodes itself (the mask is a function of the global position), so the table has every node's row,
        # P and the saved copies this rank's rows only, and the carried gather needs no exchange -- N ranks run the same
        # step as one.
        self.x_rows = x                                   # the rows the caller hands the model
        src = ops.x_full if getattr(ops, "sharded", False) else x
        n_tab, n = src.shape[0], x.shape[0]
        self.row_offset = int(getattr(ops, "row_offset", 0)) if getattr(ops, "sharded", False) else 0
        self.ops, self.x, self.p, self.state, self.tag = ops, src, float(p), state, int(tag)
        if n_tab == n:
            both = torch.zeros(2, n, 8, dtype=_F32, device=x.device)
            self.filled = (both[0], both[1])
        else:
            self.filled = (torch.zeros(n_tab, 8, dtype=_F32, device=x.device), torch.zeros(n, 8, dtype=_F32, device=x.device))
        self.saved = torch.zeros(2, n, 8, dtype=_F32, device=x.device)
        self.primed = False
        self._x_version = (src._version, x._version)
        self._host_steps = state.host_steps
        self.next_table_ready = False
        self.next_agg_ready = False
        self.adopted = False             # set by the forward that took P from this pipeline (and left its copies in ``saved``)

    @staticmethod
    def eligible(model, ops, x):
        """Conditions under which train.TrainStep pipelines: the twitch-class configuration -- a dense input of 5..8
        features into a three-channel ACM layer of width 64, one device, pattern-only operator, counter-based dropout."""
        from .graph import FilterOperators
        if tuning.HOST.pipeline <= 0 or not isinstance(ops, FilterOperators) or not isinstance(x, torch.Tensor):
            return False
        if getattr(model, "model_type", None) not in ("acmgcn", "acmgcnp", "acmgcnpp"):
            return False
        if model.model_type == "acmgcnpp":       # the residual Linear reads the table too: its fused form only (it then takes
            lins = getattr(getattr(model, "mlpX", None), "lins", None)          # this step's rows from ``saved`` in its backward)
            if lins is None or len(lins) != 1 or lins[0].out_features > 256:
                return False
        gcns = getattr(model, "gcns", None)
        if not gcns or len(gcns) != 2 or not getattr(model, "fused_dropout", False) or getattr(model, "dropout", 0) <= 0:
            return False
        l0 = gcns[0]
        try:
            cfg = l0._config()
            f_in, f = l0.weight_low.shape
        except AttributeError:
            return False
        if cfg.relu_before or f != 64 or not 4 < f_in <= 8 or x.dim() != 2 or x.shape[1] != f_in:
            return False
        if not ops.implicit or getattr(ops, "general", False) or int(getattr(ops, "hops", 1)) != 1:
            return False
        if ops.sharded:                          # equal blocks + the replicated input: the table is drawn locally (no halo)
            xf = ops.x_full
            if (not ops.uniform or xf is None or xf.dtype != _F32 or xf.dim() != 2 or xf.shape != (ops.low.n_cols, f_in)
                    or xf.device != x.device or xf.requires_grad):
                return False
        if x.dtype != _F32 or x.device != l0.weight_low.device or x.requires_grad:
            return False
        n, nnz = ops.low.n_rows, ops.low.nnz
        min_rows = tuning.HOST.pipeline                                             # below (8192): launch-bound, nothing to hide
        if n != x.shape[0] or n < max(min_rows, 32) or not 12.0 * n < nnz <= 160.0 * n:    # the regime of the fused forward
            return False
        if x.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            return False
        # four gather waves per workgroup of the backward, one workgroup per 16 rows at most (its workspace), 256 at most
        gw = 4                                                                      # gather waves per workgroup
        if not ops.low.build_streams(n_waves=max(gw, min(256 * gw, gw * ((n + 15) // 16)))):
            return False
        return ops.low.stream_waves % gw == 0 and gw <= ops.low.stream_waves <= min(256 * gw, gw * ((n + 15) // 16))

    def table(self):
        """dropout_t(x) of every node the operator's columns name, padded to 8 columns."""
        return self.filled[0]

    def local_table(self):
        """This rank's rows of table() (all of it on one device)."""
        return self.filled[0][self.row_offset:self.row_offset + self.x_rows.shape[0]]

    def agg(self):
        return self.filled[1]

    def _drop_into(self, dst, step_offset):
        d = self.state.spec(self.p, self.tag, 0)
        d.step_offset = int(step_offset)
        n, c = self.x.shape
        with _device_ctx(self.x.device), _Timed(f"dropout/{n}x{c}"):
            st = _lib.load().acm_dropout(n, c, _vp(self.x), self.x.stride(0), _vp(dst), dst.stride(0), 8, C.byref(d), _stream())
        _lib.check(st, "acm_dropout")

    def stale(self):
        """The buffers do not belong to the step about to run: never filled, the features edited in place, a step that
        did not finish (exception between make_next and end_step), or the counter advanced by someone else."""
        return (not self.primed or (self.x._version, self.x_rows._version) != self._x_version or self.next_table_ready
                or self.state.host_steps != self._host_steps)

    def prime(self):
        self._drop_into(self.filled[0], 0)
        spmm(self.ops.low, self.filled[0], out=self.filled[1], row_scale=self.ops.row_scale)
        self.primed = True
        self._x_version = (self.x._version, self.x_rows._version)
        self._host_steps = self.state.host_steps
        self.next_table_ready = self.next_agg_ready = self.adopted = False

    def refill_spec(self, p):
        """Ask the forward's row-local kernel (acm_conv_agg_fwd_t.next_x / next_drop) to write the NEXT step's dropped input
        over the table rows it reads -- possible where the table holds exactly the rows the layer works on (one device)."""
        if self.filled[0].shape[0] != self.x_rows.shape[0] or self.next_table_ready or self.x.stride(1) != 1:
            return False
        d = self.state.spec(self.p, self.tag, 0)
        d.step_offset = 1
        p.next_x, p.ld_next_x, p.next_drop = self.x.data_ptr(), self.x.stride(0), d
        return True

    def make_next(self):
        """Between the forward and the backward: the next step's dropped input replaces this step's -- only if the forward
        took this step's from the pipeline and left its copies in ``saved`` (``adopted``); a forward that went another way
        (a tuning switch, another path of the layer) may have saved the table itself for its backward.  (Drawing the
        table on a side stream right behind the first layer's forward was built and measured slower on one device:
        DESIGN.md section 9b.)"""
        if not self.adopted:
            return False
        if not self.next_table_ready:
            self._drop_into(self.filled[0], 1)
            self.next_table_ready = True
        return True

    def end_step(self):
        """After the optimizer step (which advanced the counter).  If the layer's backward did not carry the gather (it
        fell back to another path), ``filled`` is stale: prime() again before the next forward."""
        if not (self.next_table_ready and self.next_agg_ready):
            self.primed = False
        self.next_table_ready = self.next_agg_ready = self.adopted = False
        self._host_steps = self.state.host_steps


def _next_proj_request(call, f, dev, row_local_only=False):
    """(weights, relu_before, F') of the layer named in ``call.next_proj`` when its projection can ride this layer's epilogue.
    Default: only where the row-local stage runs as its own kernel (``row_local_only``: P = A_low X is given -- the input
    pipeline, evaluation passes -- and the sixteen-rows-per-wave kernel of acm_conv_agg16.hip carries the projection for
    ~6 us against the ~20 us of a separate acm_proj_fwd launch); inside the fused gather kernel it costs what it saves
    (built, measured, removed)."""
    nxt, call.next_proj = call.next_proj, None
    if nxt is None or not row_local_only or not (tuning.kernel()["rows16"] & tuning.ROWS16_EPI):
        return None
    try:
        w3 = (nxt.weight_low, nxt.weight_high, nxt.weight_mlp)
        cfg = nxt._config()
    except AttributeError:
        return None
    f2 = w3[0].shape[1]
    ok = (f2 <= 2 and all(w.dtype == _F32 and w.is_contiguous() and w.device == dev and
                                                   tuple(w.shape) == (f, f2) for w in w3))
    # (a four-channel two-column consumer gathers [Z_L | Z_H | struc_low] from packed 32-byte rows: _narrow_tables)
    return (w3, bool(cfg.relu_before), f2, cfg.n_channels == 4 and f2 == 2) if ok else None


def _take_pre_proj(call, x, w3, relu):
    """The projection a preceding layer of the same model call left for (x, w3, relu), or None."""
    pre, call.pre_proj = call.pre_proj, None
    if pre is None or not isinstance(x, torch.Tensor):
        return None
    out, zlh, zi, ptrs, prelu = pre
    if (x.data_ptr() == out.data_ptr() and x.shape == out.shape and x.stride() == out.stride() and prelu == bool(relu)
            and tuple(w.data_ptr() for w in w3) == ptrs):
        return zlh, zi
    return None


def _capturing(dev):
    return dev.type == "cuda" and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _as_f32c(t, name):
    _require_cuda(t, name)
    if t.dtype != _F32:
        t = t.to(_F32)
    return t.contiguous()


def _as_f32_rows(t, name):
    """fp32 with unit column stride: a column slice of a wider matrix is handed to the C ABI as (pointer, ld) instead of
    being copied."""
    _require_cuda(t, name)
    if t.dtype != _F32:
        t = t.to(_F32)
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1] and t.shape[1] > 0:
        return t
    return t.contiguous()


# --------------------------------------------------------------------------
# raw (non-differentiable) launches
# --------------------------------------------------------------------------
def gemm_drop_supported(n_rows, f_in, n_out):
    """Whether BOTH products of a dense projection -- Z = drop(X) W (NN) and dW = drop(X)^T dZ (TN) -- can take the input
    dropout in their tile loads (acm_gemm_drop: the row-panel kernels of acm_gemm_rows.hip)."""
    return (n_rows >= 8192 and 16 <= f_in <= 128 and 1 <= n_out <= 192
            and (tuning.gemm_forms() & (tuning.GEMM_ROWS | tuning.GEMM_ROWS_ALWAYS)) != 0)


def gemm(a, b, trans_a=False, trans_b=False, relu=False, out=None, col_blocks=0, a_drop=None):
    """out = op(a) @ op(b) on the fp32 MFMA pipe (acm_gemm).  ``col_blocks=j`` returns the product as a
    contiguous [j, m, n / j] tensor of column blocks (acm_gemm_blocks).  ``a_drop``: an acm_dropout_t applied to the stored
    matrix ``a`` while its tiles are staged (acm_gemm_drop; see gemm_drop_supported)."""
    a, b = _as_f32_rows(a, "a"), _as_f32_rows(b, "b")          # column slices of wider matrices: (pointer, ld), no copy
    m, k = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    k2, n = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    if k != k2:
        raise ValueError(f"gemm: inner dimensions differ ({k} vs {k2})")
    lib = _lib.load()
    if col_blocks:
        if n % col_blocks:
            raise ValueError("gemm: col_blocks needs n divisible by the block count")
        nb = n // col_blocks
        if out is None:
            out = torch.empty(col_blocks, m, nb, dtype=_F32, device=a.device)
        elif tuple(out.shape) != (col_blocks, m, nb) or not out.is_contiguous() or out.dtype != _F32:
            raise ValueError("gemm: out must be a contiguous fp32 [col_blocks, m, n / col_blocks] tensor")
        nbytes = C.c_size_t()
        _lib.check(lib.acm_gemm_workspace_bytes(int(trans_a), int(trans_b), m, n, k, C.byref(nbytes)))
        ws = torch.empty(max(nbytes.value // 4, 1), dtype=_F32, device=a.device) if nbytes.value else None
        with _device_ctx(a.device), _Timed(f"gemm_{'T' if trans_a else 'N'}{'T' if trans_b else 'N'}/{m}x{n}x{k}"):
            if a_drop is not None:
                st = lib.acm_gemm_drop(int(trans_a), int(trans_b), m, n, k, _vp(a), a.stride(0), _vp(b), b.stride(0),
                                       _vp(out), nb, nb, m * nb, int(relu), C.byref(a_drop), _vp(ws), nbytes.value, _stream())
            else:
                st = lib.acm_gemm_blocks(int(trans_a), int(trans_b), m, n, k, _vp(a), a.stride(0), _vp(b), b.stride(0),
                                         _vp(out), nb, nb, m * nb, int(relu), _vp(ws), nbytes.value, _stream())
        _lib.check(st, "acm_gemm_blocks")
        return out
    if out is None:
        out = torch.empty(m, n, dtype=_F32, device=a.device)
    nbytes = C.c_size_t()
    _lib.check(lib.acm_gemm_workspace_bytes(int(trans_a), int(trans_b), m, n, k, C.byref(nbytes)))
    ws = torch.empty(max(nbytes.value // 4, 1), dtype=_F32, device=a.device) if nbytes.value else None
    with _device_ctx(a.device), _Timed(f"gemm_{'T' if trans_a else 'N'}{'T' if trans_b else 'N'}/{m}x{n}x{k}"):
        if a_drop is not None:
            st = lib.acm_gemm_drop(int(trans_a), int(trans_b), m, n, k, _vp(a), a.stride(0), _vp(b), b.stride(0),
                                   _vp(out), out.stride(0), 0, 0, int(relu), C.byref(a_drop), _vp(ws), nbytes.value, _stream())
        else:
            st = lib.acm_gemm(int(trans_a), int(trans_b), m, n, k, _vp(a), a.stride(0), _vp(b), b.stride(0),
                              _vp(out), out.stride(0), int(relu), _vp(ws), nbytes.value, _stream())
    _lib.check(st, "acm_gemm")
    return out


def gemm_split(a, b, out1, out2, relu=False):
    """[out1 | out2] = a @ b: the first out1.shape[1] columns go to out1, the rest to out2 (acm_gemm_split)."""
    a, b = _as_f32c(a, "a"), _as_f32c(b, "b")
    m, k = a.shape
    n = b.shape[1]
    split = out1.shape[1]
    if b.shape[0] != k or out2.shape[1] != n - split or out1.shape[0] != m or out2.shape[0] != m:
        raise ValueError("gemm_split: shape mismatch")
    lib = _lib.load()
    nbytes = C.c_size_t()
    _lib.check(lib.acm_gemm_workspace_bytes(0, 0, m, n, k, C.byref(nbytes)))
    ws = torch.empty(max(nbytes.value // 4, 1), dtype=_F32, device=a.device) if nbytes.value else None
    with _device_ctx(a.device), _Timed(f"gemm_NN/{m}x{n}x{k}"):
        st = lib.acm_gemm_split(0, 0, m, n, k, _vp(a), a.stride(0), _vp(b), b.stride(0), _vp(out1), out1.stride(0), split,
                                _vp(out2), out2.stride(0), int(relu), _vp(ws), nbytes.value, _stream())
    _lib.check(st, "acm_gemm_split")


def proj3(x, weights, f_block, out, out2=None, relu=False, x_drop=None):
    """[Z_L 0 | Z_H 0 | Z_I] = relu?(drop?(x) @ [W_L 0 | W_H 0 | W_I]) with the three weight matrices read in place
    (acm_proj3: no packed copy of the weights, the input dropout drawn in the operand load): the first two channels in
    blocks of ``f_block`` columns; all 2 f_block + F columns go to ``out``, or the first out.shape[1] to ``out`` and the rest
    to ``out2``.  Returns False -- nothing launched -- outside the kernel's envelope (tall dense x of 32..128 features)."""
    if not isinstance(x, torch.Tensor) or x.dtype != _F32 or x.dim() != 2 or not (tuning.gemm_forms() & tuning.GEMM_BX3):
        return False
    n, k = x.shape
    ws3 = list(weights)
    f = ws3[0].shape[1]
    ncols = 2 * int(f_block) + f
    if (n < 8192 or not 32 <= k <= 128 or k % 4 or x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16 or ncols > 192
            or any(w.dtype != _F32 or tuple(w.shape) != (k, f) or w.stride(1) != 1 or w.stride(0) != ws3[0].stride(0)
                   or w.device != x.device for w in ws3)):
        return False
    split = 0 if out2 is None else out.shape[1]
    if out.shape[0] != n or (out2 is None and out.shape[1] != ncols) or (out2 is not None and tuple(out2.shape) != (n, ncols - split)):
        raise ValueError("proj3: shape mismatch")
    with _device_ctx(x.device), _Timed(f"proj3/{n}x{ncols}x{k}"):
        st = _lib.load().acm_proj3(n, k, _vp(x), x.stride(0), _vp(ws3[0]), _vp(ws3[1]), _vp(ws3[2]), ws3[0].stride(0), f,
                                   int(f_block), _vp(out), out.stride(0), split, _vp(out2), out2.stride(0) if out2 is not None else 0,
                                   int(relu), C.byref(x_drop) if x_drop is not None else None, _stream())
    if st == 4:                                   # ACM_EUNSUPPORTED after all
        return False
    _lib.check(st, "acm_proj3")
    return True


def proj_fwd(x, weights, out_lh, out_i, relu=False, h_col=None):
    """[out_lh | out_i] = relu?(x @ [W_L | W_H | W_I]) for a narrow layer (F <= 8), straight from the three weight
    matrices (acm_proj_fwd): out_lh [n, 2F] is the gathered block, out_i [n, F].  ``h_col`` (acm_proj_fwd_at): Z_H starts
    at that column of out_lh ([n, h_col + F] at least) instead of column F -- channel blocks of 4 / 8 columns."""
    x = _as_f32c(x, "x")
    ws3 = [_as_f32c(w, "weight") for w in weights]
    n, f_in = x.shape
    f = ws3[0].shape[1]
    h_col = f if h_col is None else int(h_col)
    if (any(tuple(w.shape) != (f_in, f) for w in ws3) or out_lh.shape[0] != n or out_lh.shape[1] < h_col + f or h_col < f
            or out_i.shape != (n, f)):
        raise ValueError("proj_fwd: shape mismatch")
    with _device_ctx(x.device), _Timed(f"proj_fwd/{n}x{f_in}x{3 * f}"):
        st = _lib.load().acm_proj_fwd_at(n, f_in, f, _vp(x), x.stride(0), _vp(ws3[0]), _vp(ws3[1]), _vp(ws3[2]),
                                         ws3[0].stride(0), int(relu), _vp(out_lh), out_lh.stride(0), h_col, _vp(out_i),
                                         out_i.stride(0), _stream())
    _lib.check(st, "acm_proj_fwd_at")


def proj_bwd(x, dz, weights, d_w_out, defer=None, dx_out=None):
    """Backward of the skinny projection Z = x @ [W_L | W_H | W_I] in one pass over x (acm_proj_bwd): returns
    dX = dz @ Wcat.T and fills ``d_w_out`` ([3, f_in, F], contiguous) with x.T @ dz.  ``weights``: the three
    [f_in, F] matrices.  ``defer``: a DeferredReductions the second phase is appended to (default: the thread's)."""
    if defer is None:
        defer = _ambient().defer
    x, dz = _as_f32c(x, "x"), _as_f32c(dz, "dz")
    ws3 = [_as_f32c(w, "weight") for w in weights]
    n, f_in = x.shape
    q = 3 * ws3[0].shape[1]
    blocks = d_w_out.shape[0]
    nb = q // blocks
    if (tuple(d_w_out.shape) != (blocks, f_in, nb) or not d_w_out.is_contiguous() or dz.shape != (n, q)
            or any(tuple(w.shape) != (f_in, q // 3) or w.stride(0) != ws3[0].stride(0) for w in ws3)):
        raise ValueError("proj_bwd: shape mismatch")
    lib = _lib.load()
    dx = torch.empty(n, f_in, dtype=_F32, device=x.device) if dx_out is None else dx_out[:, :f_in]
    nbytes = C.c_size_t()
    _lib.check(lib.acm_proj_bwd_workspace_bytes(n, f_in, q, C.byref(nbytes)), "acm_proj_bwd_workspace_bytes")
    ws = torch.empty(max(nbytes.value // 4, 1), dtype=_F32, device=x.device)
    with _device_ctx(x.device), _Timed(f"proj_bwd/{n}x{f_in}x{q}"):
        st = lib.acm_proj_bwd(n, f_in, q, _vp(x), x.stride(0), _vp(dz), dz.stride(0), _vp(ws3[0]), _vp(ws3[1]), _vp(ws3[2]),
                              ws3[0].stride(0), _vp(dx), dx.stride(0), _vp(d_w_out), nb, nb, f_in * nb, _vp(ws),
                              nbytes.value, defer.pointer() if defer is not None else None, _stream())
    _lib.check(st, "acm_proj_bwd")
    if defer is not None:
        defer.hold(ws, [d_w_out])
    return dx


def proj_bwd_supported(q):
    return q in (3, 6, 9, 12, 15)


def spmm(graph, dense, out=None, row_scale=None, bf16=False):
    """out = A @ dense for a CsrGraph A (acm_spmm); with ``row_scale``: diag(row_scale) (A @ dense) (acm_spmm_ex).
    ``bf16``: gather a bf16 copy of ``dense`` (even 8 < width <= 64; fp32 sums)."""
    dense = _as_f32_rows(dense, "dense")
    if dense.shape[0] != graph.n_cols:
        raise ValueError(f"spmm: dense has {dense.shape[0]} rows, operator has {graph.n_cols} columns")
    width = dense.shape[1]
    if out is None:
        out = torch.empty(graph.n_rows, width, dtype=_F32, device=dense.device)
    if width == 0 or graph.n_rows == 0:
        return out
    ws = graph.workspace(min(width, 256))
    bf16 = bool(bf16) and 8 < width <= 64 and width % 2 == 0
    if bf16:
        dense = cast_bf16(dense)
    with _device_ctx(dense.device), _Timed(f"spmm/W{width}{'b' if bf16 else ''}"):
        if row_scale is None and not bf16:
            st = _lib.load().acm_spmm(graph.handle, _vp(dense), dense.stride(0), width, _vp(out), out.stride(0),
                                      _vp(ws), ws.numel() * 4, _stream())
        else:
            o = _lib.SpmmOpts()
            o.g_bf16 = int(bf16)
            o.row_scale = _as_f32c(row_scale, "row_scale").data_ptr() if row_scale is not None else None
            st = _lib.load().acm_spmm_ex(graph.handle, _vp(dense), dense.stride(0), width, _vp(out), out.stride(0),
                                         C.byref(o), _vp(ws), ws.numel() * 4, _stream())
    _lib.check(st, "acm_spmm")
    return out


def spmm_v(graph, vals, dense, relu=False, out=None):
    """out = A(vals) @ dense: the operator's structure with per-call values (acm_spmm_v)."""
    dense = _as_f32c(dense, "dense")
    vals = _as_f32c(vals, "vals")
    if dense.shape[0] != graph.n_cols or vals.numel() != graph.nnz:
        raise ValueError("spmm_v: shape mismatch")
    width = dense.shape[1]
    if out is None:
        out = torch.empty(graph.n_rows, width, dtype=_F32, device=dense.device)
    if width == 0 or graph.n_rows == 0:
        return out
    ws = graph.workspace(min(width, 256))
    with _device_ctx(dense.device), _Timed(f"spmm_v/{graph.n_rows}x{graph.n_cols}W{width}"):
        st = _lib.load().acm_spmm_v(graph.handle, _vp(vals), _vp(dense), dense.stride(0), width, _vp(out),
                                    out.stride(0), int(relu), _vp(ws), ws.numel() * 4, _stream())
    _lib.check(st, "acm_spmm_v")
    return out


class _Mm(torch.autograd.Function):
    """Differentiable dense product on acm_gemm (used by the trivial layer branches)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return gemm(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        ga = gemm(g, b, trans_b=True) if ctx.needs_input_grad[0] else None
        gb = gemm(a, g, trans_a=True) if ctx.needs_input_grad[1] else None
        return ga, gb


def mm(a, b):
    return _run(_Mm, a, b)


class _MaskedNll(torch.autograd.Function):
    """loss = sum_i w_i * (logsumexp(z_i) - z_i[y_i]) with its gradient from the same pass
    (acm_nll_loss)."""

    @staticmethod
    def forward(ctx, logits, labels, row_weight, defer=None):
        lib = _lib.load()
        z = _as_f32c(logits, "logits")
        w = _as_f32c(row_weight, "row_weight")
        _require_cuda(labels, "labels")
        y = labels.to(torch.int64).contiguous().reshape(-1)
        n, c = z.shape
        if y.numel() != n or w.numel() != n:
            raise ValueError("masked_nll: labels / row_weight must have one entry per row")
        loss = torch.empty((), dtype=_F32, device=z.device)
        dz = torch.empty_like(z)
        nbytes = C.c_size_t()
        _lib.check(lib.acm_nll_loss_workspace_bytes(n, C.byref(nbytes)))
        ws = torch.empty(max(nbytes.value // 4, 1), dtype=_F32, device=z.device)
        with _device_ctx(z.device), _Timed(f"nll_loss/{n}x{c}"):
            st = lib.acm_nll_loss(n, c, _vp(z), z.stride(0), _vp(y), _vp(w), _vp(loss), _vp(dz), dz.stride(0),
                                  _vp(ws), ws.numel() * 4, defer.pointer() if defer is not None else None, _stream())
        _lib.check(st, "acm_nll_loss")
        if defer is not None:
            defer.hold(ws, [loss])
        ctx.save_for_backward(dz)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        (dz,) = ctx.saved_tensors
        return dz * grad_loss, None, None, None


def nll_loss_and_grad(logits, labels, row_weight, defer=None):
    """(loss, dloss/dlogits) of the masked NLL in one launch, outside autograd: a training loop can call
    ``logits.backward(gradient=dz)`` directly instead of ``loss.backward()`` (which costs a ones-fill and a
    scalar multiply of dz on top).  ``defer``: see proj_bwd."""
    if defer is None:
        defer = _ambient().defer
    with torch.no_grad():
        ctx = _NoCtx()
        loss = _MaskedNll.forward(ctx, logits.detach(), labels, row_weight, defer)
    return loss, ctx.saved[0]


class _NoCtx:
    def save_for_backward(self, *t):
        self.saved = t


def masked_nll(logits, labels, row_weight):
    """Fused log-softmax + NLL over the rows with non-zero weight (weights = 1/|train| on the
    training rows reproduces F.log_softmax + NLLLoss(out[train_idx], y[train_idx]),
    ACM-Geometric/train.py:133-134)."""
    return _run(_MaskedNll, logits, labels, row_weight, _ambient().defer)


# --------------------------------------------------------------------------
# counter-based dropout (acm_dropout_t)
# --------------------------------------------------------------------------
def eval_metrics_buffers(n_rows, n_sets, device):
    """(result [n_sets + 1], workspace) for :func:`eval_metrics`; the workspace is zeroed ONCE (its arrival counter resets
    itself after every launch)."""
    nbytes = C.c_size_t()
    _lib.check(_lib.load().acm_eval_metrics_workspace_bytes(int(n_rows), int(n_sets), C.byref(nbytes)), "acm_eval_metrics_workspace_bytes")
    return (torch.empty(n_sets + 1, dtype=_F32, device=device),
            torch.zeros(max(nbytes.value // 4, 1), dtype=_F32, device=device))


def eval_metrics(logits, labels, weights, loss_set, buffers=None):
    """Accuracy on every index set and the NLL on set ``loss_set`` from eval-mode logits, as one launch (acm_eval_metrics:
    the evaluation of ACM-Geometric/train.py:138-140 + data_utils.py:153-168 and ACM-Pytorch/train.py:112-139).
    ``weights`` [k, n]: 1 / |set| on the set's rows, 0 elsewhere (rows of weight 0 may carry the label -1).  Returns the
    fp32 tensor [acc_0 .. acc_{k-1}, nll]."""
    _require_cuda(logits, "logits")
    n, c = logits.shape
    k = weights.shape[0]
    if weights.shape[1] != n or weights.dtype != _F32 or weights.stride(1) != 1 or labels.shape[0] != n:
        raise ValueError("eval_metrics: weights must be fp32 [k, n] with contiguous rows, labels [n]")
    res, ws = buffers if buffers is not None else eval_metrics_buffers(n, k, logits.device)
    with _device_ctx(logits.device), _Timed(f"eval_metrics/{n}x{c}k{k}"):
        st = _lib.load().acm_eval_metrics(n, c, _vp(logits), logits.stride(0), _vp(labels), _vp(weights), weights.stride(0), k,
                                          int(loss_set), _vp(res), _vp(ws), ws.numel() * 4, _stream())
    _lib.check(st, "acm_eval_metrics")
    return res


class DropoutState:
    """Seed + device step counter of the counter-based dropout.  The mask of element (row, col) is a pure
    function of (seed, step, tag, row, col), so forward and backward kernels regenerate it instead of storing
    it.  ``advance()`` (or FusedAdam's ``also_advance`` hook) must run once per optimizer step; every forward /
    backward between two advances sees the same masks (distinguished by ``tag``)."""

    def __init__(self, device, seed=None):
        self.seed = int(torch.initial_seed() if seed is None else seed) & 0xFFFFFFFFFFFFFFFF
        self.step = torch.zeros(1, dtype=torch.int64, device=device)
        self.host_steps = 0          # host-side count of advances (whoever advances `step` on the device bumps it too):
                                     # lets a consumer that works ahead (InputPipeline) notice that someone else stepped

    def advance(self):
        self.step.add_(1)
        self.host_steps += 1

    def spec(self, p, tag, row_offset=0):
        d = _lib.Dropout()
        d.p, d.tag, d.seed, d.step, d.row_offset = float(p), int(tag), self.seed, self.step.data_ptr(), int(row_offset)
        return d


def _drop_spec(post_drop, row_offset):
    if post_drop is None:
        return None
    p, tag, state = post_drop
    if not 0.0 <= p < 1.0:
        raise ValueError("dropout probability must be in [0, 1)")
    return state.spec(p, tag, row_offset) if p > 0 else None


class _FusedDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, tag, state, pad_to, row_offset):
        x = _as_f32c(x, "input")
        n, c = x.shape
        width = max(c, int(pad_to or 0))
        buf = torch.empty(n, width, dtype=_F32, device=x.device)
        d = state.spec(p, tag, row_offset)
        with _device_ctx(x.device), _Timed(f"dropout/{n}x{c}"):
            st = _lib.load().acm_dropout(n, c, _vp(x), x.stride(0), _vp(buf), buf.stride(0), width, C.byref(d), _stream())
        _lib.check(st, "acm_dropout")
        ctx.args = (p, tag, state, row_offset, c)
        return buf

    @staticmethod
    def backward(ctx, g):
        p, tag, state, row_offset, c = ctx.args
        g = _as_f32c(g, "grad")                   # [n, width]; the pad columns carry no gradient
        out = torch.empty(g.shape[0], c, dtype=_F32, device=g.device)
        d = state.spec(p, tag, row_offset)
        with _device_ctx(g.device):
            st = _lib.load().acm_dropout(g.shape[0], c, _vp(g), g.stride(0), _vp(out), out.stride(0), c, C.byref(d), _stream())
        _lib.check(st, "acm_dropout")
        return out, None, None, None, None, None


def _drop_now(x, spec):
    """x * keep / (1 - p) for an acm_dropout_t ``spec`` (non-differentiable launch; same mask as the fused forms)."""
    n, c = x.shape
    buf = torch.empty(n, c, dtype=_F32, device=x.device)
    with _device_ctx(x.device), _Timed(f"dropout/{n}x{c}"):
        st = _lib.load().acm_dropout(n, c, _vp(x), x.stride(0), _vp(buf), buf.stride(0), c, C.byref(spec), _stream())
    _lib.check(st, "acm_dropout")
    return buf


def dropout(x, p, state, tag=0, pad_to=None, row_offset=0):
    """x * keep / (1 - p) with the counter-based mask (acm_dropout).  ``pad_to`` > x.shape[1] returns an
    [n, pad_to] tensor whose extra columns are zero -- the row layout the aggregate-first gather wants (pass
    it to the layer with ``input_zero_padded=True``), saving the pad fill + copy."""
    if p <= 0 and not (pad_to and pad_to > x.shape[1]):
        return x
    return _run(_FusedDropout, x, float(p), int(tag), state, pad_to, int(row_offset))


def agg_pad_width(f_in):
    """Row length (floats) of the gathered operand of the aggregate-first path, or f_in when it does not apply."""
    return 4 if f_in <= 4 else (8 if f_in <= 8 else (16 if f_in <= 16 else f_in))


# --------------------------------------------------------------------------
# the ACM-GCN++ residual branch
# --------------------------------------------------------------------------
class _ResidualLinear(torch.autograd.Function):
    """y = dropout(relu(x W^T + b)) (ACM-Geometric/models.py:26-27,55-56): dense x through acm_linear_fwd (bias / ReLU /
    counter-based dropout in the GEMM epilogue), CSR features through acm_spmm_v + acm_bias_act.  Backward:
    acm_bias_act_bwd (masks read off y), then dW = G^T x; row-sharded runs sum [dW | db] over the ranks."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, drop, group, call=None, pipe=None):
        lib = _lib.load()
        ctx.defer = call.defer if call is not None else None
        # ``pipe``: x is the input pipeline's table (InputPipeline.local_table()), which the first layer's forward refills
        # with the NEXT step's dropped input once it has adopted it -- this step's rows are then in pipe.saved[0]
        ctx.pipe = pipe
        sparse_x = isinstance(x, SparseFeatures)
        w = _as_f32c(weight, "weight")
        b = _as_f32c(bias, "bias") if bias is not None else None
        f_out, f_in = w.shape
        n = x.shape[0]
        dev = w.device
        y = torch.empty(n, f_out, dtype=_F32, device=dev)
        spec = _drop_spec(drop[:3], drop[3]) if drop is not None else None
        if sparse_x:
            spmm_v(x.csr, x.values, w.t().contiguous(), out=y)
            with _device_ctx(dev), _Timed(f"bias_act/{n}x{f_out}"):
                st = lib.acm_bias_act(n, f_out, _vp(y), y.stride(0), _vp(b), int(relu),
                                      C.byref(spec) if spec is not None else None, _stream())
            _lib.check(st, "acm_bias_act")
        else:
            x = _as_f32c(x, "input")
            if x.shape[1] < f_in:
                raise ValueError(f"input has {x.shape[1]} columns but the Linear has {f_in} input features")
            nbytes = C.c_size_t()
            _lib.check(lib.acm_gemm_workspace_bytes(0, 1, n, f_out, f_in, C.byref(nbytes)))
            ws = torch.empty(max(nbytes.value // 4, 1), dtype=_F32, device=dev) if nbytes.value else None
            with _device_ctx(dev), _Timed(f"linear_fwd/{n}x{f_out}x{f_in}"):
                st = lib.acm_linear_fwd(n, f_in, f_out, _vp(x), x.stride(0), _vp(w), w.stride(0), _vp(b), int(relu),
                                        C.byref(spec) if spec is not None else None, _vp(y), y.stride(0), _vp(ws),
                                        nbytes.value, _stream())
            _lib.check(st, "acm_linear_fwd")
        ctx.relu, ctx.group, ctx.has_bias = bool(relu), group, b is not None
        ctx.keep_scale = 1.0 / (1.0 - drop[0]) if (drop is not None and drop[0] > 0) else 1.0
        ctx.sparse_x = x if sparse_x else None
        ctx.save_for_backward(w if sparse_x else x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, y = ctx.saved_tensors
        if ctx.pipe is not None and ctx.pipe.adopted:
            x = ctx.pipe.saved[0]                 # (the table itself holds step t + 1's rows by now)
        dy = _as_f32c(dy, "grad")
        n, f_out = y.shape
        f_in = w.shape[1]
        dev = y.device
        flat = torch.empty(f_out * f_in + f_out, dtype=_F32, device=dev)      # [dW | db]: one all-reduce when sharded
        d_w, d_b = flat[: f_out * f_in].view(f_out, f_in), flat[f_out * f_in:]
        if ctx.sparse_x is None and not ctx.needs_input_grad[0] and f_in <= 16 and f_out <= 256:
            # a narrow dense input that takes no gradient (the raw features): dW and db in ONE pass over (y, dy, x), the
            # [n, f_out] matrix G = dL/d(pre-activation) never stored (acm_linear_bwd)
            nbytes = C.c_size_t()
            _lib.check(lib.acm_linear_bwd_workspace_bytes(n, f_in, f_out, C.byref(nbytes)), "acm_linear_bwd_workspace_bytes")
            ws = torch.empty(max(nbytes.value // 4, 1), dtype=_F32, device=dev)
            with _device_ctx(dev), _Timed(f"linear_bwd/{n}x{f_out}x{f_in}"):
                st = lib.acm_linear_bwd(n, f_in, f_out, _vp(x), x.stride(0), _vp(y), y.stride(0), _vp(dy), dy.stride(0),
                                        float(ctx.keep_scale), int(ctx.relu), _vp(d_w), f_in, _vp(d_b), _vp(ws), nbytes.value,
                                        ctx.defer.pointer() if ctx.defer is not None else None, _stream())
            _lib.check(st, "acm_linear_bwd")
            if ctx.defer is not None:
                ctx.defer.hold(ws, [d_w, d_b], keep=[flat, x, y, dy])
            if ctx.group is not None:
                import torch.distributed as dist
                if ctx.defer is not None:
                    ctx.defer.allreduce(flat, ctx.group)
                else:
                    dist.all_reduce(flat, group=ctx.group)
            return None, d_w, (d_b if ctx.has_bias else None), None, None, None, None, None
        g = torch.empty(n, f_out, dtype=_F32, device=dev)
        nbytes = C.c_size_t()
        _lib.check(lib.acm_bias_act_bwd_workspace_bytes(n, f_out, C.byref(nbytes)), "acm_bias_act_bwd_workspace_bytes")
        ws = torch.empty(max(nbytes.value // 4, 1), dtype=_F32, device=dev)
        with _device_ctx(dev), _Timed(f"bias_act_bwd/{n}x{f_out}"):
            st = lib.acm_bias_act_bwd(n, f_out, _vp(y), y.stride(0), _vp(dy), dy.stride(0), float(ctx.keep_scale),
                                      int(ctx.relu), _vp(g), g.stride(0), _vp(d_b), _vp(ws), nbytes.value,
                                      ctx.defer.pointer() if ctx.defer is not None else None, _stream())
        _lib.check(st, "acm_bias_act_bwd")
        if ctx.defer is not None:
            ctx.defer.hold(ws, [d_b], keep=[flat])
        d_x = None
        if ctx.sparse_x is not None:                          # dW^T = X_csr^T G
            xs = ctx.sparse_x
            xt = xs.csr_t
            d_w.copy_(spmm_v(xt, xs.values.index_select(0, xt.src_pos), g).t())
        else:
            if x.shape[1] == f_in:
                gemm(g, x, trans_a=True, out=d_w)
            else:                                             # zero-padded input rows (dropout(..., pad_to=...))
                d_w.copy_(gemm(g, x, trans_a=True)[:, :f_in])
            if ctx.needs_input_grad[0]:
                d_x = gemm(g, w)
                if d_x.shape[1] != x.shape[1]:
                    d_x = torch.nn.functional.pad(d_x, (0, x.shape[1] - d_x.shape[1]))
        if ctx.group is not None:
            import torch.distributed as dist
            if ctx.defer is not None:
                ctx.defer.allreduce(flat, ctx.group)
            else:
                dist.all_reduce(flat, group=ctx.group)
        return d_x, d_w, (d_b if ctx.has_bias else None), None, None, None, None, None


class _ResidualAddLinear(torch.autograd.Function):
    """out = fea + dropout(relu(x W^T + b)) in ONE launch (acm_linear_fwd_add: the ACM-GCN++ hidden activations fea1 + xX,
    ACM-Geometric/models.py:55-56,73) for a narrow dense x that takes no gradient; the backward recomputes both masks
    (acm_linear_bwd_recompute: reads dY and x only) and hands dY through to fea."""

    @staticmethod
    def forward(ctx, fea, x, weight, bias, relu, drop, group, call):
        lib = _lib.load()
        ctx.defer = call.defer if call is not None else None
        fea, x = _as_f32_rows(fea, "fea"), _as_f32c(x, "input")
        w = _as_f32c(weight, "weight")
        b = _as_f32c(bias, "bias") if bias is not None else None
        f_out, f_in = w.shape
        n, dev = x.shape[0], w.device
        out = torch.empty(n, f_out, dtype=_F32, device=dev)
        spec = _drop_spec(drop[:3], drop[3]) if drop is not None else None
        with _device_ctx(dev), _Timed(f"linear_fwd_add/{n}x{f_out}x{f_in}"):
            st = lib.acm_linear_fwd_add(n, f_in, f_out, _vp(x), x.stride(0), _vp(w), w.stride(0), _vp(b), int(relu),
                                        C.byref(spec) if spec is not None else None, _vp(fea), fea.stride(0), _vp(out),
                                        out.stride(0), _stream())
        _lib.check(st, "acm_linear_fwd_add")
        ctx.relu, ctx.group, ctx.has_bias, ctx.spec = bool(relu), group, b is not None, spec
        ctx.save_for_backward(x, w, b if b is not None else w.new_zeros(0))
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, b = ctx.saved_tensors
        dy = _as_f32_rows(dy, "grad")
        f_out, f_in = w.shape
        n, dev = x.shape[0], w.device
        flat = torch.empty(f_out * f_in + f_out, dtype=_F32, device=dev)      # [dW | db]: one all-reduce when sharded
        d_w, d_b = flat[: f_out * f_in].view(f_out, f_in), flat[f_out * f_in:]
        nbytes = C.c_size_t()
        _lib.check(lib.acm_linear_bwd_workspace_bytes(n, f_in, f_out, C.byref(nbytes)), "acm_linear_bwd_workspace_bytes")
        ws = torch.empty(max(nbytes.value // 4, 1), dtype=_F32, device=dev)
        with _device_ctx(dev), _Timed(f"linear_bwd_recompute/{n}x{f_out}x{f_in}"):
            st = lib.acm_linear_bwd_recompute(n, f_in, f_out, _vp(x), x.stride(0), _vp(w), w.stride(0),
                                              _vp(b) if ctx.has_bias else None, int(ctx.relu),
                                              C.byref(ctx.spec) if ctx.spec is not None else None, _vp(dy), dy.stride(0),
                                              _vp(d_w), f_in, _vp(d_b), _vp(ws), nbytes.value,
                                              ctx.defer.pointer() if ctx.defer is not None else None, _stream())
        _lib.check(st, "acm_linear_bwd_recompute")
        if ctx.defer is not None:
            ctx.defer.hold(ws, [d_w, d_b], keep=[flat, x, dy])
        if ctx.group is not None:
            import torch.distributed as dist
            if ctx.defer is not None:
                ctx.defer.allreduce(flat, ctx.group)
            else:
                dist.all_reduce(flat, group=ctx.group)
        return dy, None, d_w, (d_b if ctx.has_bias else None), None, None, None, None


def residual_add_supported(x, weight):
    """Shapes acm_linear_fwd_add / acm_linear_bwd_recompute take: a dense input of <= 16 columns that needs no gradient,
    <= 256 outputs."""
    return (isinstance(x, torch.Tensor) and not x.requires_grad and x.dim() == 2 and weight.shape[1] <= 16
            and weight.shape[1] <= x.shape[1] and weight.shape[0] <= 256)


def residual_add_linear(fea, x, weight, bias, relu=True, drop=None, group=None, call=None):
    """fea + dropout(relu(x @ weight.T + bias)) as one launch, masks recomputed in the backward (see _ResidualAddLinear);
    arguments as residual_linear."""
    if drop is not None and not drop[0] > 0:
        drop = None
    return _run(_ResidualAddLinear, fea, x, weight, bias, bool(relu), drop, group, _call_or_ambient(call))


def residual_linear(x, weight, bias, relu=True, drop=None, group=None, call=None, pipe=None):
    """dropout(relu(x @ weight.T + bias)) on the HIP kernels.  ``drop = (p, tag, DropoutState, row_offset)`` draws the
    counter-based mask in the epilogue; ``group``: row-sharded run (the parameter gradients are summed over it);
    ``call``: the model call's CallContext (its deferral list; default: the thread's ambient one); ``pipe``: x is the
    table of that InputPipeline (the backward then reads this step's rows from its saved copy once the first layer's forward
    has refilled the table)."""
    if drop is not None and not drop[0] > 0:
        drop = None
    return _run(_ResidualLinear, x, weight, bias, bool(relu), drop, group, _call_or_ambient(call), pipe)


# --------------------------------------------------------------------------
# the fused ACM layer
# --------------------------------------------------------------------------
class AcmConfig:
    """Static description of one layer variant (what the reference selects with
    model_type / variant / structure_info, ACM-Geometric/layers.py:78-116)."""

    __slots__ = ("n_channels", "relu_before", "relu_after", "relu_mlp", "layernorm", "scale", "gather_bf16")

    def __init__(self, model_type, variant, structure_info, attn_layernorm, gather_dtype="fp32"):
        if gather_dtype not in ("fp32", "bf16"):
            raise ValueError("gather_dtype must be 'fp32' or 'bf16'")
        self.gather_bf16 = gather_dtype == "bf16"
        plus = model_type in ("acmgcnp", "acmgcnpp", "acmgcn+", "acmgcn++")
        if model_type == "acmsgc":
            self.n_channels, self.relu_before, self.relu_after, self.relu_mlp = 3, False, False, False
            self.layernorm = False
        elif model_type == "acmsnowball":                 # the layer's generic branch: three channels, no LayerNorm, the
            self.n_channels = 3                           # structure channel never (layers.py:59,106-108)
            self.relu_before, self.relu_after, self.relu_mlp = bool(variant), not bool(variant), True
            self.layernorm = False
        else:
            self.n_channels = 4 if (plus and structure_info) else 3
            self.relu_before = bool(variant)          # ACMII: ReLU between projection and filter
            self.relu_after = not bool(variant)       # ACM: ReLU after the filter
            self.relu_mlp = True
            self.layernorm = bool(attn_layernorm) and plus
        self.scale = 1.0 if self.n_channels == 4 else 3.0


def _ptr_array(tensors):
    n = len(tensors)
    return _VP4(*[tensors[i].data_ptr() if i < n and tensors[i] is not None else None for i in range(4)])


_VP4 = C.c_void_p * 4


def cast_bf16(src):
    """fp32 [n, c] (any row pitch) -> new contiguous bf16 [n, c] (acm_cast_bf16, round to nearest even)."""
    _require_cuda(src, "src")
    n, c = src.shape
    dst = torch.empty(n, c, dtype=torch.bfloat16, device=src.device)
    with _device_ctx(src.device), _Timed(f"cast_bf16/{n}x{c}"):
        st = _lib.load().acm_cast_bf16(n, c, _vp(src), src.stride(0), _vp(dst), dst.stride(0), _stream())
    _lib.check(st, "acm_cast_bf16")
    return dst


def _gather_rows(ops, local):
    """All-gather row blocks of a row-sharded dense matrix (halo exchange) into the layout the local operators'
    column ids refer to: rank r's rows start at r * n_max, shorter blocks are zero-padded (distributed.ShardPlan).
    Single process: identity."""
    if not ops.sharded:
        return local
    import torch.distributed as dist
    world = dist.get_world_size(ops.group)
    n_max = ops.n_gathered // world
    local = local.contiguous()
    if local.shape[0] != n_max:                       # work-balanced blocks differ in length
        buf = local.new_zeros(n_max, local.shape[1])
        buf[: local.shape[0]] = local
        local = buf
    full = torch.empty(world * n_max, local.shape[1], dtype=local.dtype, device=local.device)
    with _Timed(f"all_gather/{n_max}x{local.shape[1]}", local.numel() * local.element_size()):
        dist.all_gather_into_tensor(full, local, group=ops.group)
    return full


def _hop_buffer(t_like, width):
    """[n, width] view of a buffer whose rows are padded to the next of 4 / 8 columns: the narrow gather fetches such rows
    as aligned 16-byte blocks (see _chan_block)."""
    pitch = width if width > 8 else (8 if width > 4 else (4 if width > 2 else width))
    return torch.empty(t_like.shape[0], pitch, dtype=_F32, device=t_like.device)[:, :width]


def _low_product(ops, t_local, transpose=False, out=None):
    """A_low @ t (or A_low^T @ t) for a row-local t [n_local, w]: one hop of the ACM-SGC k-hop chain, with the
    halo all-gather when row-sharded.  Pattern-only operators: D^-1 (P t) and P (D^-1 t).  ``out``: where the product
    goes (default: a row-padded buffer, _hop_buffer)."""
    if out is None:
        out = _hop_buffer(t_local, t_local.shape[1])
    if transpose:
        if ops.implicit:
            scaled = _hop_buffer(t_local, t_local.shape[1])
            torch.mul(t_local, ops.row_scale[:, None], out=scaled)
            return spmm(ops.low_t, _gather_rows(ops, scaled), out=out)
        return spmm(ops.low_t, _gather_rows(ops, t_local), out=out)
    return spmm(ops.low, _gather_rows(ops, t_local), out=out, row_scale=ops.row_scale if ops.implicit else None)


def _flat_views(flat, nw, k, f, layernorm):
    """The head-parameter gradients as views of the layer's flat gradient buffer (fresh tensor objects on every call:
    autograd adopts a returned gradient only while nobody else holds that tensor object)."""
    # (one split call: 3 k + 1 views; slicing them one by one costs a layer 30 us of host time)
    if layernorm:
        parts = flat[nw:].split([f] * (3 * k) + [k * k])
        d_vec = [t.view(f, 1) for t in parts[:k]]
        d_lnw, d_lnb = list(parts[k:2 * k]), list(parts[2 * k:3 * k])
    else:
        parts = flat[nw:].split([f] * k + [k * k])
        d_vec, d_lnw, d_lnb = [t.view(f, 1) for t in parts[:k]], [], []
    d_mix = parts[-1].view(k, k)
    return d_vec, d_lnw, d_lnb, d_mix


def _chan_block(f):
    """Column distance of the two gathered channels inside [Z_L | Z_H] / [G_L | G_H].  F in {3, 5, 6, 7} pads each channel
    to a block of 4 / 8 columns: the narrow gather then fetches a neighbour's row with aligned 16-byte loads (the
    merged path of spmm_narrow_kernel) instead of 2 F scalar ones -- on the arXiv-year-shaped graph (5 classes) the
    output layer's gathers are 3-4x faster.  The pad columns are never read into a result."""
    if f in (3, 5, 6, 7):
        return 4 if f == 3 else 8
    return f


def _narrow_tables(n, fb, f, dev, packed):
    """The gathered tables of a narrow layer: ([c0 | c1] block rows, third-channel rows or None).  ``packed`` (four channels
    of two columns: the output layer of a two-class model with structure_info): views of ONE table of 32-byte rows
    [c0 c0 c1 c1 | c2 c2 - -], which the pair-lane gather (acm_conv.hip: spmm_narrow_pair3_kernel) walks with one line per
    neighbour instead of two."""
    if packed:
        base = torch.empty(n, 8, dtype=_F32, device=dev)
        return base[:, :4], base[:, 4:6]
    return torch.empty(n, 2 * fb, dtype=_F32, device=dev), None


def _k3_setup(cfg, ops, k, f, n, dev, f_in_w, pre, zi, vecs, lnw, lnb, mix, grad_out, post_relu, post_scale, post_drop,
              fb=None):
    """Buffers and acm_conv_bwd_local_t of the row-local backward of one layer: G tables, dZ, the flat buffer every
    replicated-parameter gradient is a view of."""
    four = k == 4
    fb = f if fb is None else fb
    g, gs = _narrow_tables(n, fb, f, dev, packed=four and f == 2 and fb == 2)      # [G_L | G_H] (channel blocks of fb columns)
    dz = torch.empty(n, 3 * f, dtype=_F32, device=dev)           # [dZ_L | dZ_H | dZ_I]
    if four and gs is None:
        gs = torch.empty(n, f, dtype=_F32, device=dev)
    # every replicated-parameter gradient is a view of one flat buffer: a row-sharded run sums the partials
    # with a single all-reduce and no pack / unpack launches
    nw, nln = 3 * f_in_w * f, (k * f if cfg.layernorm else 0)
    flat = torch.empty(nw + k * f + 2 * nln + k * k, dtype=_F32, device=dev)
    d_vec, d_lnw, d_lnb, d_mix = _flat_views(flat, nw, k, f, cfg.layernorm)

    q = _lib.ConvBwdLocal()
    q.f_out, q.n_channels = f, k
    q.relu_after, q.relu_mlp, q.layernorm, q.scale = int(cfg.relu_after), int(cfg.relu_mlp), int(cfg.layernorm), cfg.scale
    q.grad_out, q.ld_grad_out = grad_out.data_ptr(), grad_out.stride(0)
    q.pre, q.ld_pre = pre.data_ptr(), pre.stride(0)
    q.s_mlp, q.ld_s_mlp = zi.data_ptr(), zi.stride(0)
    general = bool(getattr(ops, "general", False))
    ones = ops.zeros(n, 1).new_ones(n) if (four and general) else None
    # pattern-only backward: A_low^T G = P (D^-1 G), so G_L / G_H are written pre-scaled and G_S unscaled
    # (A_low^T (D G_S) = P G_S)
    q.deg = None if (not four or ops.implicit) else (ones if general else ops.deg).data_ptr()
    if ops.implicit:
        q.g_scale = ops.row_scale.data_ptr()
    q.att_vec, q.ln_weight, q.ln_bias = _ptr_array(vecs), _ptr_array(lnw), _ptr_array(lnb)
    q.att_mix = mix.data_ptr()
    q.g_low, q.ld_g_low = g.data_ptr(), g.stride(0)
    q.g_high, q.ld_g_high = g.data_ptr() + 4 * fb, g.stride(0)
    q.g_mlp, q.ld_g_mlp = dz.data_ptr() + 8 * f, dz.stride(0)
    if four:
        q.g_struc, q.ld_g_struc = gs.data_ptr(), gs.stride(0)
    q.d_att_vec, q.d_ln_weight, q.d_ln_bias = _ptr_array(d_vec), _ptr_array(d_lnw), _ptr_array(d_lnb)
    q.d_att_mix = d_mix.data_ptr()
    q.post_relu = int(post_relu)
    if post_scale is not None:
        q.post_scale, q.ld_post_scale = post_scale.data_ptr(), post_scale.stride(0)
    spec = _drop_spec(post_drop, ops.row_offset)
    if spec is not None:
        q.post_drop = spec
    return dict(q=q, g=g, dz=dz, gs=gs, flat=flat, nw=nw, d_vec=d_vec, d_lnw=d_lnw, d_lnb=d_lnb, d_mix=d_mix,
                general=general, ones=ones, grad_out=grad_out)


class AcmConvFunction(torch.autograd.Function):
    """out, att = ACM layer(x; parameters) over the operators in ``ops``.

    forward : K1 acm_gemm (X [W_L|W_H|W_I]) -> K2 acm_conv_fwd
    backward: K3 acm_conv_bwd_local -> K4 acm_conv_bwd_spmm -> K5 acm_gemm (X^T dZ, dZ Wcat^T)
    """

    @staticmethod
    def forward(ctx, x, w_low, w_high, w_mlp, v_low, v_high, v_mlp, v_struc, struc_low, att_mix,
                lnw_low, lnw_high, lnw_mlp, lnw_struc, lnb_low, lnb_high, lnb_mlp, lnb_struc, ops, cfg,
                post_relu=False, post_scale=None, post_drop=None, call=None, tail_layer=False, agg_holder=None,
                in_drop=None):
        lib = _lib.load()
        ctx.set_materialize_grads(False)          # no zero-filled gradient for the (non-differentiable) att output
        # the model call's context (deferral list, loss-tail request, input pipeline, projection hand-off); ``tail_layer``:
        # the caller is an output layer without post-op working in the operator's numbering (it may take call.tail);
        # ``agg_holder``: layers.GraphConvolution's {"agg": P-or-None} of an evaluation pass over a static input
        # ``in_drop = (p, tag, DropoutState)``: the caller's INPUT dropout (models.py:54), left to this layer: the dense
        # projection applies it while staging X (acm_gemm_drop), forward and backward, and X itself is saved un-dropped
        call = ctx.call = _call_or_ambient(call)
        ctx.in_drop = in_drop if (in_drop is not None and in_drop[0] > 0) else None
        sparse_x = isinstance(x, SparseFeatures)
        if not sparse_x:
            x = _as_f32c(x, "input")
        dev = x.device
        n, f = x.shape[0], w_low.shape[1]
        k = cfg.n_channels
        if post_scale is not None:
            post_scale = _as_f32c(post_scale, "post_scale")
            if tuple(post_scale.shape) != (n, f):
                raise ValueError(f"post_scale must be [{n}, {f}]")
        ctx.post_relu, ctx.post_scale = bool(post_relu), post_scale
        ctx.post_drop = post_drop if (post_drop is not None and post_drop[0] > 0) else None

        def set_post(st):
            st.post_relu = int(ctx.post_relu)
            if post_scale is not None:
                st.post_scale, st.ld_post_scale = post_scale.data_ptr(), post_scale.stride(0)
            spec = _drop_spec(ctx.post_drop, ops.row_offset)
            if spec is not None:
                st.post_drop = spec
        if n != ops.n_local:
            raise ValueError(f"input has {n} rows but the graph operator has {ops.n_local}")
        pregathered = getattr(ops, "_pregathered", None)      # one-shot hand-over from the caller (models.GCN)
        ops._pregathered = None
        f_in = w_low.shape[0]
        ctx.x_width = x.shape[1]
        zero_padded = x.shape[1] != f_in          # dropout(..., pad_to=...) output: extra columns are zero
        if zero_padded and (sparse_x or x.shape[1] < f_in):
            raise ValueError(f"input has {x.shape[1]} columns but the weights have {f_in} rows")
        # Aggregate-first (A (X W) = (A X) W): legal without a ReLU between projection and
        # filter, worth it when F_in < F, and free of any backward SpMM when x needs no gradient.
        ctx.agg_first = (not cfg.relu_before and f_in <= 16 and f_in < f and f <= 64 and not sparse_x
                         and not ctx.needs_input_grad[0] and (tuning.HOST.rewrites & tuning.REWRITE_AGG_FIRST) != 0)
        four = k == 4
        general = bool(getattr(ops, "general", False))
        # k-hop low-pass channel (ACM-SGC, ACM-Pytorch/utils.py:631-637 materialises the dense A_low^k): here the
        # chain A_low (A_low (... Z_L)) with the 1-hop operator, adj_high stays 1-hop like the reference's
        hops = int(getattr(ops, "hops", 1))
        ctx.hops = hops
        ctx.fb = f                               # set by the literal path below (_chan_block)
        if hops > 1:
            if cfg.relu_before or cfg.relu_after or four or general:
                raise NotImplementedError("hops > 1 is the ACM-SGC chain: model_type 'acmsgc' only")
            ctx.agg_first = False
        if general:
            if ops.sharded:
                raise NotImplementedError("general operator pairs are not row-sharded")
            ctx.agg_first = False
        # ACMII first layer with a narrow input: gather the input rows and recompute relu(x_j [W_L | W_H]) per edge on
        # the matrix pipe instead of gathering the 2F-wide projected rows (acm_conv_acmii_fwd = K1 + K2)
        ctx.recompute = (cfg.relu_before and not cfg.relu_after and cfg.relu_mlp and f == 64 and f_in <= 8
                         and not sparse_x and not general and hops == 1
                         and (tuning.HOST.rewrites & tuning.REWRITE_ACMII_RECOMPUTE) != 0)
        fb = f                                   # column distance of the two gathered channels (see _chan_block)
        if ctx.agg_first or ctx.recompute:
            if ctx.in_drop is not None:               # these forms gather the input itself: they need the dropped rows
                x = _drop_now(x, _drop_spec(ctx.in_drop, ops.row_offset))
                ctx.in_drop_materialised = True
            fp = 4 if f_in <= 4 else (8 if f_in <= 8 else 16)
            if ctx.recompute:
                fp = 8
            # (the pipeline's table is refilled in place through raw pointers -- no version bump: never through the holder;
            #  ACMII recomputes per edge from the gathered rows: there is no P to keep)
            if agg_holder is not None and (ops.sharded or call.pipe is not None or ctx.recompute or ctx.in_drop is not None):
                agg_holder = None
            if x.shape[1] == fp:
                xpad = x
            else:                                     # the zero-padded copy of a static input is kept with its P
                xpad = agg_holder.get("xpad") if agg_holder is not None else None
                if xpad is None or tuple(xpad.shape) != (n, fp):
                    xpad = torch.nn.functional.pad(x[:, :f_in], (0, fp - f_in))
                    if agg_holder is not None:
                        agg_holder["xpad"] = xpad
            agg_given = agg_holder.get("agg") if agg_holder is not None else None
            if agg_given is not None and tuple(agg_given.shape) != (n, fp):
                agg_given = None
            # a training loop's input pipeline (InputPipeline): P for this step came out of the previous step's backward
            pipe = call.pipe
            ctx.pipe = None
            if (pipe is not None and pipe.primed and ctx.agg_first and fp == 8 and f == 64 and ops is pipe.ops
                    and xpad.data_ptr() == pipe.local_table().data_ptr() and agg_holder is None):   # (only a training step carries a pipe)
                agg_given = pipe.agg()
                ctx.pipe = pipe
                pipe.adopted = True               # the loop may refill the table: this forward leaves its copies in ``saved``
            if agg_given is not None:
                xg = xpad                             # not read: P = A_low X comes from the holder
            elif (pregathered is not None and pregathered[0].data_ptr() == xpad.data_ptr()
                    and pregathered[1].shape[1] == fp and pregathered[1].shape[0] == ops.n_gathered):
                xg = pregathered[1]                   # the caller already holds every node's (dropped) input
            else:
                xg = _gather_rows(ops, xpad)
            wl, wh, wm = (_as_f32c(t, "weight") for t in (w_low, w_high, w_mlp))
            if ctx.recompute:
                w3 = (wl, wh, wm)
                x = x[:, :f_in].contiguous() if zero_padded else x      # saved for K5 (dWcat = X^T dZ)
        else:
            if zero_padded:
                x = x[:, :f_in].contiguous()
            w3 = tuple(_as_f32c(t, "weight") for t in (w_low, w_high, w_mlp))
            # narrow dense layers (F <= 5) project with the streaming kernel straight from the three weights; everything
            # else packs [W_L | W_H | W_I] for the MFMA GEMM / the CSR-feature product
            use_proj = (not sparse_x and f <= 5 and f_in <= 64     # wider inputs: the MFMA GEMM is the faster stream
                        and w3[0].stride(0) == w3[1].stride(0) == w3[2].stride(0))
            fb = ctx.fb = _chan_block(f)
            # Row pitch of Z: for narrow layers the gathered block [Z_L | Z_H] (2F floats) must be
            # one aligned vector fetch, so rows are padded to a multiple of that block.
            ldz = 3 * f
            if f in (2, 4, 8):
                ldz = -(-3 * f // (2 * f)) * (2 * f)
            elif fb != f:
                ldz = -(-(2 * fb + f) // 4) * 4
            pre = _take_pre_proj(call, x, w3, cfg.relu_before) if use_proj else None
            drop_spec = _drop_spec(ctx.in_drop, ops.row_offset) if ctx.in_drop is not None else None
            # [Z_L | Z_H] as a compact table of its own (what a narrow gather / the k-hop chain walks: 16-byte-block rows at
            # their own pitch instead of [Z_L | Z_H | Z_I | pad] rows), Z_I next to it
            two_tables = (use_proj or f in (2, 4, 8) or hops > 1) and not sparse_x
            # four channels of two columns: [Z_L | Z_H] and the gathered struc_low rows share 32-byte rows (_narrow_tables)
            pack4 = four and f == 2 and fb == 2 and two_tables and not ops.sharded and not general
            done3 = False
            if pre is None and not sparse_x:
                # tall dense inputs of 32..128 features: the three weight matrices in place on the split-bf16 kernel (no cat)
                if two_tables:
                    zlh, _ = _narrow_tables(n, fb, f, dev, pack4)
                    zi = torch.empty(n, f, dtype=_F32, device=dev)
                    done3 = proj3(x, w3, fb, zlh, zi, relu=cfg.relu_before, x_drop=drop_spec)
                else:
                    z = torch.empty(n, ldz, dtype=_F32, device=dev)[:, : 2 * fb + f]
                    done3 = proj3(x, w3, fb, z, relu=cfg.relu_before, x_drop=drop_spec)
                    zlh, zi = z[:, : 2 * fb], z[:, 2 * fb:]
                if done3:
                    ctx.in_drop_used = drop_spec is not None
            if (not done3 and drop_spec is not None
                    and (pre is not None or use_proj or two_tables or sparse_x or not gemm_drop_supported(n, f_in, 2 * fb + f))):
                # the caller left its input dropout to this layer (in_drop) but the projection about to run cannot draw
                # the mask in its operand load (the two-table / k-hop GEMM, the narrow streaming projection, shapes outside
                # acm_gemm_drop): apply it here, same counter-based mask, and save the DROPPED input for the backward
                if sparse_x:
                    raise NotImplementedError("in_drop with CSR features: the caller applies the dropout to the values")
                x, pre = _drop_now(x, drop_spec), None
                ctx.in_drop_materialised = True
                drop_spec = None
            if done3:
                pass                                       # [Z_L | Z_H], Z_I are written
            elif pre is not None:
                zlh, zi = pre                              # computed in the preceding layer's epilogue
            elif use_proj:
                zlh, _ = _narrow_tables(n, fb, f, dev, pack4)
                zi = torch.empty(n, f, dtype=_F32, device=dev)
                proj_fwd(x, w3, zlh, zi, relu=cfg.relu_before, h_col=fb)
            else:
                if fb != f:                        # [W_L 0 | W_H 0 | W_I]: the product lands in channel blocks of fb columns
                    zpad = w3[0].new_zeros(w3[0].shape[0], fb - f)
                    wcat = torch.cat((w3[0], zpad, w3[1], zpad, w3[2]), dim=1).contiguous()
                else:
                    wcat = torch.cat(w3, dim=1).contiguous()                            # [F_in, 3F]
                if two_tables:                            # one GEMM with a two-matrix output
                    zlh, _ = _narrow_tables(n, fb, f, dev, pack4)
                    zi = torch.empty(n, f, dtype=_F32, device=dev)
                    gemm_split(x, wcat, zlh, zi, relu=cfg.relu_before)
                else:
                    z = torch.empty(n, ldz, dtype=_F32, device=dev)[:, : 2 * fb + f]
                    if sparse_x:                              # Z = X_csr Wcat: nnz(X) * 3F FMAs
                        spmm_v(x.csr, x.values, wcat, relu=cfg.relu_before, out=z)
                    else:
                        gemm(x, wcat, relu=cfg.relu_before, out=z, a_drop=drop_spec)        # [n, 3F] view
                        ctx.in_drop_used = drop_spec is not None
                    zlh, zi = z[:, : 2 * fb], z[:, 2 * fb:]
            if hops > 2:
                # [A_low^(k-1) Z_L | Z_H] in place: the last of the k - 1 >= 2 products reads a hop buffer and writes over
                # Z_L (which nothing reads again: no ReLU mask in the k-hop layer) -- no copy of Z_H into a second table
                t = zlh[:, :f]
                for hop in range(hops - 1):
                    t = _low_product(ops, t, out=zlh[:, :f] if hop == hops - 2 else None)
                zg = _gather_rows(ops, zlh) if ops.sharded else zlh
            elif hops > 1:
                zc = torch.empty(n, 2 * fb, dtype=_F32, device=dev)              # [A_low Z_L | Z_H]
                _low_product(ops, zlh[:, :f], out=zc[:, :f])
                zc[:, fb:fb + f] = zlh[:, fb:fb + f]
                zg = _gather_rows(ops, zc)
            else:
                zg = _gather_rows(ops, zlh) if ops.sharded else zlh                 # gathered [Z_L|Z_H]
        if four and general:
            if ops.un is None:
                raise RuntimeError("structure_info=1 needs adj_low_unnormalized")
            s_local = _as_f32c(struc_low, "struc_low")
        elif four:
            if ops.deg is None:
                raise RuntimeError("structure_info=1 needs adj_low_unnormalized")
            s_local = _as_f32c(struc_low, "struc_low")
            if s_local.shape[0] != n:
                raise ValueError("struc_low rows != local nodes")
            s_gath = _gather_rows(ops, s_local)
        vecs = [_as_f32c(t, "att_vec") for t in ((v_low, v_high, v_mlp, v_struc) if four else (v_low, v_high, v_mlp))]
        lnw = [_as_f32c(t, "ln") for t in (lnw_low, lnw_high, lnw_mlp, lnw_struc)[:k]] if cfg.layernorm else []
        lnb = [_as_f32c(t, "ln") for t in (lnb_low, lnb_high, lnb_mlp, lnb_struc)[:k]] if cfg.layernorm else []
        mix = _as_f32c(att_mix, "att_vec")
        if tuple(mix.shape) != (k, k):
            raise RuntimeError(f"att_vec is {tuple(mix.shape)} but the layer mixes {k} channels "
                               "(structure_info is only valid with acmgcnp/acmgcnpp)")
        out = torch.empty(n, f, dtype=_F32, device=dev)
        att = torch.empty(n, 4, dtype=_F32, device=dev)
        if ctx.recompute:
            zlh = torch.empty(n, 2 * f, dtype=_F32, device=dev)
            zi = torch.empty(n, f, dtype=_F32, device=dev)
            pre = torch.empty(n, (k - 1) * f, dtype=_F32, device=dev)
            p = _lib.ConvAcmiiFwd()
            p.f_in, p.f_pad, p.f_out, p.layernorm, p.scale = f_in, fp, f, int(cfg.layernorm), cfg.scale
            p.n_channels = k
            if four:                                  # ps = A_low S: one F-wide single-channel gather of the parameter
                ps = spmm(ops.low, s_gath, row_scale=ops.row_scale if ops.implicit else None, bf16=cfg.gather_bf16)
                p.ps, p.ld_ps = ps.data_ptr(), ps.stride(0)
                p.ss, p.ld_ss = s_local.data_ptr(), s_local.stride(0)
                p.deg = ops.deg.data_ptr()
            p.xg, p.ld_xg = xg.data_ptr(), xg.stride(0)
            p.xs, p.ld_xs = xpad.data_ptr(), xpad.stride(0)
            p.w_low, p.w_high, p.w_mlp, p.ld_w = wl.data_ptr(), wh.data_ptr(), wm.data_ptr(), f
            p.att_vec, p.ln_weight, p.ln_bias = _ptr_array(vecs), _ptr_array(lnw), _ptr_array(lnb)
            p.att_mix = mix.data_ptr()
            p.out, p.ld_out = out.data_ptr(), out.stride(0)
            p.pre, p.ld_pre = pre.data_ptr(), pre.stride(0)
            p.att = att.data_ptr()
            p.zlh, p.ld_zlh = zlh.data_ptr(), zlh.stride(0)
            p.zi, p.ld_zi = zi.data_ptr(), zi.stride(0)
            if ops.implicit:
                p.row_scale = ops.row_scale.data_ptr()
            set_post(p)
            nbytes = C.c_size_t()
            _lib.check(lib.acm_conv_acmii_fwd_workspace_bytes(ops.low.handle, C.byref(nbytes)), "acm_conv_acmii_fwd_workspace_bytes")
            ws = torch.empty(max(nbytes.value // 4, 1), dtype=_F32, device=dev)
            # The mask form (acm_conv_acmii_v.hip): relu(x_j W) = m_j * (x_j W), so the aggregate is W contracted with
            # V_i = sum_j m_j (x) x_j -- a product over the neighbour index on the bf16 matrix pipe, exact operands -- and the
            # weight gradients are the same V contracted with dH: no transposed product for them (the structure channel's
            # parameter keeps its one F-wide transposed product).  A pattern-only operator over this process's own rows, no
            # gradient into x.
            ctx.mask_table = None
            st = 4
            if (ops.implicit and not ctx.needs_input_grad[0] and n > 0 and xg.shape[0] == ops.low.n_cols
                    and (tuning.HOST.rewrites & tuning.REWRITE_ACMII_MASK) != 0
                    and (getattr(ops.low, "item_stream_waves", 0) > 0         # one-off per operator (synchronises: never
                         or (not _capturing(dev) and ops.low.build_item_streams()))):     # inside a capture, whose warm-up built them)
                # the table covers every column of the operator: this process's rows, or (row-sharded) the all-gathered input --
                # each rank evaluates the masks of its halo itself, and its backward then needs NO all-gather of gradients
                ng = xg.shape[0]
                tb = C.c_size_t()
                _lib.check(lib.acm_acmii_table_bytes(ng, C.byref(tb)), "acm_acmii_table_bytes")
                table = torch.empty(tb.value // 4, dtype=torch.int32, device=dev)
                with _device_ctx(dev), _Timed(f"acmii_table/{ng}x{f_in}"):
                    st = lib.acm_acmii_table(ng, f_in, xg.data_ptr(), xg.stride(0), wl.data_ptr(), wh.data_ptr(), f,
                                             table.data_ptr(), tb.value, _stream())
                if st == 0:
                    if ops.low.n_long_rows == 0:          # only the fix-up of long rows reads zlh (its high-pass half)
                        p.zlh, p.ld_zlh = None, 0
                    with _device_ctx(dev), _Timed(f"conv_acmii_v_fwd/F{f}i{f_in}"):
                        st = lib.acm_conv_acmii_v_fwd(ops.low.handle, C.byref(p), table.data_ptr(), _vp(ws), ws.numel() * 4, _stream())
                    if st == 0:
                        ctx.mask_table, ctx.mask_x = table, xpad
                        ctx.mask_self_offset = 0
                        if ops.sharded:                   # this rank's rows inside the gathered numbering (_gather_rows)
                            import torch.distributed as dist
                            ctx.mask_self_offset = dist.get_rank(ops.group) * (ops.n_gathered // dist.get_world_size(ops.group))
                    else:
                        p.zlh, p.ld_zlh = zlh.data_ptr(), zlh.stride(0)
                if st not in (0, 4):                      # 4 = ACM_EUNSUPPORTED: the fp32 kernel below
                    _lib.check(st, "acm_conv_acmii_v_fwd")
            if ctx.mask_table is None:
                with _device_ctx(dev), _Timed(f"conv_acmii_fwd/F{f}i{f_in}"):
                    st = lib.acm_conv_acmii_fwd(ops.low.handle, C.byref(p), _vp(ws), ws.numel() * 4, _stream())
                _lib.check(st, "acm_conv_acmii_fwd")
            ctx.agg_first, ctx.tail, ctx.sparse_x = False, None, None
            ctx.ops, ctx.cfg = ops, cfg
            ctx.save_for_backward(x, *w3, zlh, zi, pre, mix, *vecs, *lnw, *lnb)
            ctx.mark_non_differentiable(att)
            return out, att
        if ctx.agg_first:
            p = _lib.ConvAggFwd()
            p.f_in, p.f_pad, p.f_out = f_in, fp, f
            p.relu_after, p.relu_mlp, p.layernorm, p.scale = int(cfg.relu_after), int(cfg.relu_mlp), int(cfg.layernorm), cfg.scale
            p.xg, p.ld_xg = xg.data_ptr(), xg.stride(0)
            p.xs, p.ld_xs = xpad.data_ptr(), xpad.stride(0)
            p.w_low, p.w_high, p.w_mlp, p.ld_w = wl.data_ptr(), wh.data_ptr(), wm.data_ptr(), f
            p.att_vec, p.ln_weight, p.ln_bias = _ptr_array(vecs), _ptr_array(lnw), _ptr_array(lnb)
            p.att_mix = mix.data_ptr()
            agg = agg_given if agg_given is not None else torch.empty(n, fp, dtype=_F32, device=dev)
            p.agg_given = int(agg_given is not None)
            refill = False
            if ctx.pipe is not None:                  # the backward's operands: copies the row-local kernel leaves
                p.agg_copy, p.ld_agg_copy = ctx.pipe.saved[1].data_ptr(), ctx.pipe.saved[1].stride(0)
                p.xs_copy, p.ld_xs_copy = ctx.pipe.saved[0].data_ptr(), ctx.pipe.saved[0].stride(0)
                # ... and (one device: the table IS this rank's rows) it refills the table with dropout_{t+1}(x) over the rows
                # it has just copied: make_next()'s acm_dropout launch is gone
                refill = ctx.pipe.refill_spec(p)
            p.out, p.ld_out = out.data_ptr(), out.stride(0)
            p.agg, p.ld_agg = agg.data_ptr(), agg.stride(0)
            p.att = att.data_ptr()
            p.n_channels = k
            if ops.implicit:
                p.row_scale = ops.row_scale.data_ptr()
            extra = ()
            if four:                                  # pre_S = deg * (A_low S) - S: one F-wide gather of S
                ps = torch.empty(n, f, dtype=_F32, device=dev)
                if cfg.gather_bf16 and f % 2 == 0 and f > 8:
                    sgt = cast_bf16(s_gath)
                    p.sg_bf16 = 1
                else:
                    sgt = s_gath
                p.sg, p.ld_sg = sgt.data_ptr(), sgt.stride(0)
                p.ss, p.ld_ss = s_local.data_ptr(), s_local.stride(0)
                p.deg = ops.deg.data_ptr()
                p.ps, p.ld_ps = ps.data_ptr(), ps.stride(0)
                extra = (ps, s_local)
            set_post(p)
            # the row's head statistics (mean | rstd | sigmoid | alpha per channel): 16 k bytes per row that save
            # the backward three 16-lane reductions per channel and row
            stats = torch.empty(n, 4 * k, dtype=_F32, device=dev) if any(ctx.needs_input_grad) else None
            if stats is not None:
                p.head_stats, p.ld_head_stats = stats.data_ptr(), stats.stride(0)
            ctx.head_stats = stats
            # (the row-local stage is a kernel of its own when P is given, and always with the structure channel)
            nxt = _next_proj_request(call, f, dev, row_local_only=f == 64 and (agg_given is not None or four))
            if nxt is not None:
                n_w3, n_relu, f2, n_pack = nxt
                n_zlh, _ = _narrow_tables(n, f2, f2, dev, n_pack)
                n_zi = torch.empty(n, f2, dtype=_F32, device=dev)
                p.next_w_low, p.next_w_high, p.next_w_mlp = (w.data_ptr() for w in n_w3)
                p.next_ld_w, p.next_f, p.next_relu = n_w3[0].stride(0), f2, int(n_relu)
                p.next_zlh, p.ld_next_zlh = n_zlh.data_ptr(), n_zlh.stride(0)
                p.next_zi, p.ld_next_zi = n_zi.data_ptr(), n_zi.stride(0)
            ws = ops.low.workspace(max(fp, f) if four else fp)
            with _device_ctx(dev), _Timed(f"conv_agg_{'epi' if agg_given is not None else 'fwd'}/F{f}k{k}i{f_in}"):
                st = lib.acm_conv_agg_fwd(ops.low.handle, C.byref(p), _vp(ws), ws.numel() * 4, _stream())
            _lib.check(st, "acm_conv_agg_fwd")
            if refill:
                ctx.pipe.next_table_ready = True
            if agg_holder is not None and agg_given is None:
                agg_holder["agg"] = agg
            if nxt is not None:
                call.pre_proj = (out, n_zlh, n_zi, tuple(w.data_ptr() for w in n_w3), n_relu)
            ctx.ops, ctx.cfg, ctx.f_in = ops, cfg, f_in
            # with a fused ReLU the output itself records which elements the post-op let through: the backward reads it
            # instead of regenerating the dropout mask (no extra memory: the next layer keeps the same tensor alive)
            ctx.out_mask = bool(ctx.post_relu) and ctx.post_scale is None
            if ctx.pipe is not None:
                xpad, agg = ctx.pipe.saved[0], ctx.pipe.saved[1]
            ctx.save_for_backward(xpad, agg, wl, wh, wm, mix, *vecs, *lnw, *lnb, *extra, *((out,) if ctx.out_mask else ()))
            ctx.mark_non_differentiable(att)
            return out, att
        pre = torch.empty(n, (k - 1) * f, dtype=_F32, device=dev)
        p = _lib.ConvFwd()
        p.f_out, p.n_channels = f, k
        p.relu_after, p.relu_mlp, p.layernorm = int(cfg.relu_after), int(cfg.relu_mlp), int(cfg.layernorm)
        p.scale, p.row_offset = cfg.scale, ops.row_offset
        if ops.implicit:
            p.row_scale = ops.row_scale.data_ptr()
        graph = ops.low
        if general:
            # every channel through its own operator, then the fused kernel over the identity operator as a
            # row-local epilogue: pre_L = 1*PL, pre_H = PH - 1*0, pre_S = 1*(1*PS) - 0
            pl = spmm(ops.low, zlh[:, :f])
            ph = spmm(ops.high, zlh[:, fb:fb + f])
            zero = ops.zeros(n, f)
            graph = ops.eye
            p.g_low, p.ld_g_low = pl.data_ptr(), pl.stride(0)
            p.g_high, p.ld_g_high = zero.data_ptr(), zero.stride(0)
            p.s_high, p.ld_s_high = ph.data_ptr(), ph.stride(0)
            if four:
                ps = spmm(ops.un, s_local)
                ones = ops.zeros(n, 1).new_ones(n)
                p.g_struc, p.ld_g_struc = ps.data_ptr(), ps.stride(0)
                p.s_struc, p.ld_s_struc = zero.data_ptr(), zero.stride(0)
                p.deg = ones.data_ptr()
            keep_alive = (pl, ph, zero) + ((ps, ones) if four else ())
        else:
            if cfg.gather_bf16 and 8 < f <= 64 and f % 2 == 0:
                # bf16 copy of the gathered operand(s): half the gather bytes, fp32 accumulation; the self rows
                # (s_high / s_mlp / s_struc) stay fp32
                zb = cast_bf16(zg[:, : 2 * f])
                p.gather_bf16 = 1
                p.g_low, p.ld_g_low = zb.data_ptr(), zb.stride(0)
                p.g_high, p.ld_g_high = zb.data_ptr() + 2 * f, zb.stride(0)
                if four:
                    sb = cast_bf16(s_gath)
                    p.g_struc, p.ld_g_struc = sb.data_ptr(), sb.stride(0)
            else:
                p.g_low, p.ld_g_low = zg.data_ptr(), zg.stride(0)
                p.g_high, p.ld_g_high = zg.data_ptr() + 4 * fb, zg.stride(0)
                if four:
                    p.g_struc, p.ld_g_struc = s_gath.data_ptr(), s_gath.stride(0)
                    if (f == 2 and fb == 2 and zg is zlh and zlh.stride(0) == 8 and zlh.data_ptr() % 32 == 0 and s_gath is s_local
                            and zlh.untyped_storage().nbytes() - zlh.storage_offset() * 4 >= n * 32):
                        # packed rows (this call's _narrow_tables, or the producing layer's epilogue): the parameter's rows
                        # are copied beside [Z_L | Z_H] -- one small launch for half the lines of the gather
                        torch.as_strided(zlh, (n, 2), (8, 1), zlh.storage_offset() + 4).copy_(s_local)
                        p.g_struc, p.ld_g_struc = zlh.data_ptr() + 16, 8
            p.s_high, p.ld_s_high = zlh.data_ptr() + 4 * fb, zlh.stride(0)
            if four:
                p.s_struc, p.ld_s_struc = s_local.data_ptr(), s_local.stride(0)
                p.deg = ops.deg.data_ptr()
        p.s_mlp, p.ld_s_mlp = zi.data_ptr(), zi.stride(0)
        p.att_vec = _ptr_array(vecs)
        p.ln_weight, p.ln_bias = _ptr_array(lnw), _ptr_array(lnb)
        p.att_mix = mix.data_ptr()
        p.out, p.ld_out = out.data_ptr(), out.stride(0)
        p.pre, p.ld_pre = pre.data_ptr(), pre.stride(0)
        p.att = att.data_ptr()
        set_post(p)
        ws = graph.workspace((k - 1) * f)
        # output layer + loss + K3 in one row pass (acm_conv_fwd_tail) when a training loop asked for it
        tail_req, layer_ok = call.tail, bool(tail_layer)
        ctx.tail = None
        st = None
        if (tail_req is not None and tail_req.out is None and layer_ok and not general and k == 3 and f <= 8
                and not cfg.gather_bf16 and not post_relu and post_scale is None and post_drop is None
                and any(ctx.needs_input_grad) and tail_req.labels.numel() == n
                and (graph.n_long_rows == 0 or 12.0 < graph.nnz / max(graph.n_rows, 1) <= 160.0)):
            dlog = torch.empty(n, f, dtype=_F32, device=dev)
            st3 = _k3_setup(cfg, ops, k, f, n, dev, w3[0].shape[0], pre, zi, vecs, lnw, lnb, mix, dlog, False, None, None,
                            fb=fb)
            loss = torch.empty((), dtype=_F32, device=dev)
            lo = _lib.Loss()
            lo.n_classes = f
            y = tail_req.labels.to(torch.int64).contiguous().reshape(-1)
            w_row = _as_f32c(tail_req.row_weight, "row_weight")
            lo.labels, lo.row_weight = y.data_ptr(), w_row.data_ptr()
            lo.loss, lo.dlogits, lo.ld_dlogits = loss.data_ptr(), dlog.data_ptr(), dlog.stride(0)
            q = st3["q"]
            q.defer = call.defer_ptr()
            nbytes = C.c_size_t()
            if lib.acm_conv_fwd_tail_workspace_bytes(n, f, k, C.byref(nbytes)) == 0:
                wt = torch.empty(max(nbytes.value // 4, 1), dtype=_F32, device=dev)
                with _device_ctx(dev), _Timed(f"conv_fwd_tail/F{f}k{k}"):
                    st = lib.acm_conv_fwd_tail(graph.handle, C.byref(p), C.byref(lo), C.byref(q), _vp(ws), ws.numel() * 4,
                                               _vp(wt), nbytes.value, _stream())
                if st == 0:
                    st3["keep"] = (y, w_row, wt)
                    ctx.tail = st3
                    tail_req.loss, tail_req.dz, tail_req.out = loss, dlog, out
                    if call.defer is not None:
                        call.defer.hold(wt, [loss, st3["d_mix"], *st3["d_vec"], *st3["d_lnw"], *st3["d_lnb"]],
                                        keep=[loss, st3["flat"]])
                elif st != 4:                       # ACM_EUNSUPPORTED: the layer does not qualify, three calls then
                    _lib.check(st, "acm_conv_fwd_tail")
        if st != 0:
            with _device_ctx(dev), _Timed(f"conv_fwd/F{f}k{k}"):
                st = lib.acm_conv_fwd(graph.handle, C.byref(p), _vp(ws), ws.numel() * 4, _stream())
            _lib.check(st, "acm_conv_fwd")
        ctx.ops, ctx.cfg = ops, cfg
        ctx.sparse_x = x if sparse_x else None
        # Lazy input gradient: when the input IS the output tensor of an aggregate-first layer of the same model call and
        # the model vouches that nothing else consumes it (call.hidden_private), this layer's backward may hand
        # dX = dZ Wcat^T and dW = X^T dZ to that layer's backward kernel (acm_conv_agg_bwd_t.proj_*) instead of running
        # acm_proj_bwd: the [n, F] gradient then never exists in memory.
        # ONLY under a deferral list (call.defer): this layer's dW' is then written by a LATER kernel than the one this
        # backward returns from -- exactly the contract of DeferredReductions ("every .grad is undefined until the flush";
        # the loop that owns the step checks all_adopted() and flushes before anything reads a gradient).  Without one,
        # autograd's AccumulateGrad may ADD the still unwritten dW' to an existing .grad right behind this node
        # (zero_grad(set_to_none=False), gradient accumulation, hooks), or the producer's backward may never run
        # (torch.autograd.grad on a subset): the plain GCN API therefore materialises dX and dW' here.
        ctx.lazy_producer = None
        tape = getattr(_TLS, "tape", None)
        prod = (getattr(x, "grad_fn", None) or (tape.producer(x) if tape is not None else None)) if not sparse_x else None
        if (call.hidden_private is x and prod is not None and getattr(prod, "agg_first", False) and getattr(prod, "call", None) is call
                and not zero_padded and hops == 1 and call.defer is not None):
            ctx.lazy_producer = prod
        if ctx.in_drop is not None and not (getattr(ctx, "in_drop_used", False) or getattr(ctx, "in_drop_materialised", False)):
            raise RuntimeError("acm_conv: the caller's input dropout (in_drop) was not applied by any projection path")
        ctx.save_for_backward(w3[0] if sparse_x else x, *w3, zlh, zi, pre, mix, *vecs, *lnw, *lnb)
        ctx.mark_non_differentiable(att)
        return out, att

    @staticmethod
    def backward(ctx, grad_out, _grad_att):
        if grad_out is None:
            return (None,) * 27
        lib = _lib.load()
        ops, cfg = ctx.ops, ctx.cfg
        defer = ctx.call.defer                    # the deferral list of the model call this backward belongs to
        k = cfg.n_channels
        saved = ctx.saved_tensors
        if ctx.agg_first:
            return AcmConvFunction._backward_agg(ctx, grad_out)
        x, wl_, wh_, wm_, zlh, zi, pre, mix = saved[:8]
        w3 = (wl_, wh_, wm_)
        vecs = list(saved[8:8 + k])
        lnw = list(saved[8 + k:8 + 2 * k]) if cfg.layernorm else []
        lnb = list(saved[8 + 2 * k:8 + 3 * k]) if cfg.layernorm else []
        dev = zlh.device
        n, f = zlh.shape[0], wl_.shape[1]
        grad_out = _as_f32c(grad_out, "grad_out")
        four = k == 4

        f_in_w = wl_.shape[0]
        tail = getattr(ctx, "tail", None)
        done = tail is not None and grad_out.data_ptr() == tail["grad_out"].data_ptr()
        fb = getattr(ctx, "fb", f)
        st3 = tail if done else _k3_setup(cfg, ops, k, f, n, dev, f_in_w, pre, zi, vecs, lnw, lnb, mix, grad_out,
                                          ctx.post_relu, ctx.post_scale, ctx.post_drop, fb=fb)
        q, g, dz, gs, flat, nw = st3["q"], st3["g"], st3["dz"], st3["gs"], st3["flat"], st3["nw"]
        general, ones = st3["general"], st3["ones"]
        if done:
            ctx.tail = None
        elif getattr(ctx, "mask_table", None) is not None:
            q.g_scale = None                     # the mask form's backward scales by 1 / d_i itself: G_L, G_H as they are
        d_vec, d_lnw, d_lnb, d_mix = _flat_views(flat, nw, k, f, cfg.layernorm)    # this call's own view objects
        del st3, tail
        if not done:                 # else: acm_conv_fwd_tail already ran K3 with exactly this gradient
            nbytes = C.c_size_t()
            _lib.check(lib.acm_conv_bwd_local_workspace_bytes(n, f, k, C.byref(nbytes)))
            ws = torch.empty(max(nbytes.value // 4, 1), dtype=_F32, device=dev)
            q.defer = defer.pointer() if defer is not None else None
            with _device_ctx(dev), _Timed(f"conv_bwd_local/F{f}k{k}"):
                st = lib.acm_conv_bwd_local(n, C.byref(q), _vp(ws), ws.numel() * 4, _stream())
            _lib.check(st, "acm_conv_bwd_local")
            if defer is not None:
                defer.hold(ws, [d_mix, *d_vec, *d_lnw, *d_lnb], keep=[flat])

        table = getattr(ctx, "mask_table", None)
        if table is not None:
            # the mask form's backward: dW_L, dW_H straight from dH_L, dH_H (K3's g) over the FORWARD operator and the
            # forward's table, and the row-local dW_I = X^T dZ_I in the same launch
            d_wcat = flat[:nw].view(3, f_in_w, f)
            xt = ctx.mask_x
            b = _lib.ConvAcmiiBwd()
            b.f_in, b.table = f_in_w, table.data_ptr()
            b.g_low, b.ld_g_low = g.data_ptr(), g.stride(0)
            b.g_high, b.ld_g_high = g.data_ptr() + 4 * fb, g.stride(0)
            b.g_mlp, b.ld_g_mlp = dz.data_ptr() + 8 * f, dz.stride(0)
            b.x, b.ld_x = xt.data_ptr(), xt.stride(0)
            b.self_offset = ctx.mask_self_offset
            b.row_scale = ops.row_scale.data_ptr()
            b.d_w_low, b.d_w_high, b.d_w_mlp, b.ld_dw = d_wcat[0].data_ptr(), d_wcat[1].data_ptr(), d_wcat[2].data_ptr(), f
            b.defer = defer.pointer() if defer is not None else None
            nbytes = C.c_size_t()
            _lib.check(lib.acm_conv_acmii_v_bwd_workspace_bytes(ops.low.handle, C.byref(nbytes)), "acm_conv_acmii_v_bwd_workspace_bytes")
            wsb = torch.empty(max(nbytes.value // 4, 1), dtype=_F32, device=dev)
            with _device_ctx(dev), _Timed(f"conv_acmii_v_bwd/F{f}i{f_in_w}"):
                st = lib.acm_conv_acmii_v_bwd(ops.low.handle, C.byref(b), _vp(wsb), wsb.numel() * 4, _stream())
            _lib.check(st, "acm_conv_acmii_v_bwd")
            if defer is not None:
                defer.hold(wsb, [d_wcat[0], d_wcat[1], d_wcat[2]], keep=[flat, table, g, dz, xt])
            d_struc = None
            if four:                                  # dS = A_low^T (D G_S) - G_S = P G_S - G_S (pattern-only; K3 left G_S unscaled)
                d_struc = torch.empty(n, f, dtype=_F32, device=dev)
                low_t = ops.low_t
                ws2 = low_t.workspace(f)
                o = _lib.SpmmOpts()
                o.sub, o.ld_sub = gs.data_ptr(), gs.stride(0)
                gsg = _gather_rows(ops, gs)
                if cfg.gather_bf16 and 8 < f <= 64 and f % 2 == 0:       # bf16 gathered operand (the self term stays fp32)
                    gsg = cast_bf16(gsg)
                    o.g_bf16 = 1
                with _device_ctx(dev), _Timed(f"spmm_sub/{f}"):
                    st = lib.acm_spmm_ex(low_t.handle, _vp(gsg), gsg.stride(0), f, _vp(d_struc), d_struc.stride(0),
                                         C.byref(o), _vp(ws2), ws2.numel() * 4, _stream())
                _lib.check(st, "acm_spmm_ex")
            if ops.sharded:                             # replicated parameters: sum the row-shard partials
                import torch.distributed as dist
                if defer is not None:
                    defer.allreduce(flat, ops.group)    # after the step's single flush (the all-reduce reads its sums)
                else:
                    dist.all_reduce(flat, group=ops.group)
            none4 = [None] * 4
            grads_vec = d_vec + [None] * (4 - k)
            grads_lnw = (d_lnw + [None] * (4 - k)) if cfg.layernorm else none4
            grads_lnb = (d_lnb + [None] * (4 - k)) if cfg.layernorm else none4
            return (None, d_wcat[0], d_wcat[1], d_wcat[2], grads_vec[0], grads_vec[1], grads_vec[2], grads_vec[3],
                    d_struc, d_mix, *grads_lnw, *grads_lnb, None, None, None, None, None, None, None, None, None)

        d_struc = torch.empty(n, f, dtype=_F32, device=dev) if four else None
        r = _lib.ConvBwdSpmm()
        r.f_out, r.row_offset = f, ops.row_offset
        if general:
            # transposed products channel by channel, then the fused kernel over the identity operator applies the
            # ACMII masks: dZ_L = m*(1*T_L), dZ_H = m*(T_H - 1*0), dS = 1*T_S - 0
            t_l = spmm(ops.low.transpose(), g[:, :f])
            t_h = spmm(ops.high.transpose(), g[:, fb:fb + f])
            zero = ops.zeros(n, f)
            low_t = ops.eye
            r.g_low, r.ld_g_low = t_l.data_ptr(), t_l.stride(0)
            r.g_high, r.ld_g_high = zero.data_ptr(), zero.stride(0)
            r.s_high, r.ld_s_high = t_h.data_ptr(), t_h.stride(0)
            if four:
                t_s = spmm(ops.un.transpose(), gs)
                r.g_struc, r.ld_g_struc = t_s.data_ptr(), t_s.stride(0)
                r.s_struc, r.ld_s_struc = zero.data_ptr(), zero.stride(0)
                r.inv_deg = ones.data_ptr()
                r.d_struc, r.ld_d_struc = d_struc.data_ptr(), d_struc.stride(0)
        else:
            gg = _gather_rows(ops, g)
            gsg = _gather_rows(ops, gs) if four else None
            low_t = ops.low_t
            if cfg.gather_bf16 and 8 < f <= 64 and f % 2 == 0 and fb == f:
                # bf16 copies of the gathered gradient tables: half the bytes of the fabric-bound transposed products (the
                # self terms and every sum stay fp32); opt-in, BASELINE config 3's tolerance
                gb = cast_bf16(gg[:, : 2 * f])
                r.gather_bf16 = 1
                r.g_low, r.ld_g_low = gb.data_ptr(), gb.stride(0)
                r.g_high, r.ld_g_high = gb.data_ptr() + 2 * f, gb.stride(0)
                if four:
                    gsb = cast_bf16(gsg)
                    gsg = gsb
            else:
                r.g_low, r.ld_g_low = gg.data_ptr(), gg.stride(0)
                r.g_high, r.ld_g_high = gg.data_ptr() + 4 * fb, gg.stride(0)
            r.s_high, r.ld_s_high = g.data_ptr() + 4 * fb, g.stride(0)
            if four:
                r.g_struc, r.ld_g_struc = gsg.data_ptr(), gsg.stride(0)
                r.s_struc, r.ld_s_struc = gs.data_ptr(), gs.stride(0)
                r.inv_deg = None if ops.implicit else ops.inv_deg.data_ptr()
                r.d_struc, r.ld_d_struc = d_struc.data_ptr(), d_struc.stride(0)
            if ops.implicit:
                r.self_scale = ops.self_scale.data_ptr()
        if cfg.relu_before:                       # ACMII: ReLU mask of the projected features
            r.mask_low, r.ld_mask_low = zlh.data_ptr(), zlh.stride(0)
            r.mask_high, r.ld_mask_high = zlh.data_ptr() + 4 * fb, zlh.stride(0)
        r.dz_low, r.ld_dz_low = dz.data_ptr(), dz.stride(0)
        r.dz_high, r.ld_dz_high = dz.data_ptr() + 4 * f, dz.stride(0)
        ws2 = low_t.workspace((k - 1) * f)
        with _device_ctx(dev), _Timed(f"conv_bwd_spmm/F{f}k{k}"):
            st = lib.acm_conv_bwd_spmm(low_t.handle, C.byref(r), _vp(ws2), ws2.numel() * 4, _stream())
        _lib.check(st, "acm_conv_bwd_spmm")
        if ctx.hops > 2 and ops.implicit:
            # the remaining k - 1 >= 2 transposed hops of the low channel with a pattern-only operator:
            # (P D^-1)^(k-1) t = P [D^-1 P]^(k-2) (D^-1 t) -- ONE input scaling, then k - 2 row-scaled products (the forward's
            # form) and a final plain one written over dZ_L (it reads a hop buffer), instead of a scaling pass per hop
            sc = _hop_buffer(dz, f)
            torch.mul(dz[:, :f], ops.row_scale[:, None], out=sc)
            for hop in range(ctx.hops - 2):
                sc = spmm(ops.low_t, _gather_rows(ops, sc), out=_hop_buffer(dz, f), row_scale=ops.row_scale)
            spmm(ops.low_t, _gather_rows(ops, sc), out=dz[:, :f])
        elif ctx.hops > 1:                                # the remaining k-1 transposed hops of the low channel
            t = dz[:, :f]
            last = ctx.hops - 2
            for hop in range(ctx.hops - 1):               # the last hop writes dZ_L in place unless it reads it
                t = _low_product(ops, t, transpose=True, out=dz[:, :f] if (hop == last and hop > 0) else None)
            if last == 0:
                dz[:, :f] = t

        if ctx.sparse_x is not None:                                          # dWcat = X_csr^T dZ
            xs = ctx.sparse_x
            xt = xs.csr_t
            d_wcat = spmm_v(xt, xs.values.index_select(0, xt.src_pos), dz, out=flat[:nw].view(f_in_w, 3 * f))
            d_x = None
        elif (ctx.needs_input_grad[0] and proj_bwd_supported(3 * f)
              and wl_.stride(0) == wh_.stride(0) == wm_.stride(0)):
            d_wcat = flat[:nw].view(3, f_in_w, f)                             # narrow output layer: dX and dW in one
            prod = getattr(ctx, "lazy_producer", None)
            if (prod is not None and f <= 2 and f_in_w == 64 and x.shape[1] == 64 and getattr(prod, "lazy", None) is None
                    and all(w.stride(0) == f and w.is_contiguous() for w in w3)):
                # ... left to the producing layer's backward kernel: the placeholder is what autograd carries there
                d_x = torch.empty(n, x.shape[1], dtype=_F32, device=dev)
                prod.lazy = dict(dz=dz, w3=w3, d_w=d_wcat, x=x, placeholder=d_x)
            else:
                d_x = proj_bwd(x, dz, w3, d_wcat, defer=defer)                # pass over x (acm_proj_bwd)
        else:
            d_wcat = gemm(x, dz, trans_a=True, col_blocks=3,
                          out=flat[:nw].view(3, f_in_w, f),                   # contiguous per weight
                          a_drop=_drop_spec(ctx.in_drop, ops.row_offset) if getattr(ctx, "in_drop_used", False) else None)
            d_x = gemm(dz, torch.cat(w3, dim=1), trans_b=True) if ctx.needs_input_grad[0] else None
        if d_x is not None and d_x.shape[1] != ctx.x_width:
            d_x = torch.nn.functional.pad(d_x, (0, ctx.x_width - d_x.shape[1]))
        if ops.sharded:                             # replicated parameters: sum the row-shard partials
            import torch.distributed as dist
            if defer is not None:
                defer.allreduce(flat, ops.group)    # after the step's single flush (the all-reduce reads its sums)
            else:
                dist.all_reduce(flat, group=ops.group)
        if d_wcat.dim() == 3:
            d_wl, d_wh, d_wm = d_wcat[0], d_wcat[1], d_wcat[2]
        else:
            d_wl, d_wh, d_wm = (d_wcat[:, i * f:(i + 1) * f] for i in range(3))
        none4 = [None] * 4
        grads_vec = d_vec + [None] * (4 - k)
        grads_lnw = (d_lnw + [None] * (4 - k)) if cfg.layernorm else none4
        grads_lnb = (d_lnb + [None] * (4 - k)) if cfg.layernorm else none4
        return (d_x, d_wl, d_wh, d_wm, grads_vec[0], grads_vec[1], grads_vec[2], grads_vec[3],
                d_struc, d_mix, *grads_lnw, *grads_lnb, None, None, None, None, None, None, None, None, None)


def _backward_agg(ctx, grad_out):
    """Backward of the aggregate-first forward: one row-local kernel and no SpMM for the three
    filterbank channels; the structure channel (k = 4) adds one F-wide transposed product for
    d struc_low.  Collectives: the all-reduce of the replicated-parameter gradients, plus the
    all-gather of D*G_S when sharded with k = 4."""
    lib = _lib.load()
    ops, cfg, f_in = ctx.ops, ctx.cfg, ctx.f_in
    k = cfg.n_channels
    four = k == 4
    saved = ctx.saved_tensors
    out_fwd = None
    if getattr(ctx, "out_mask", False):
        out_fwd, saved = saved[-1], saved[:-1]
    xpad, agg, wl, wh, wm, mix = saved[:6]
    vecs = list(saved[6:6 + k])
    nln = k if cfg.layernorm else 0
    lnw = list(saved[6 + k:6 + k + nln])
    lnb = list(saved[6 + k + nln:6 + k + 2 * nln])
    dev = xpad.device
    n, f, fp = xpad.shape[0], wl.shape[1], xpad.shape[1]
    lazy, ctx.lazy = getattr(ctx, "lazy", None), None
    if lazy is not None and (grad_out is not lazy["placeholder"] and grad_out.data_ptr() != lazy["placeholder"].data_ptr()):
        raise RuntimeError("acm_conv: the hidden activation marked private (CallContext.hidden_private) received a gradient "
                           "from somewhere else as well")
    fuse_proj = (lazy is not None and fp == 8 and f == 64 and out_fwd is not None and ctx.post_scale is None
                 and getattr(ctx, "head_stats", None) is not None)
    if lazy is not None and not fuse_proj:            # the kernel cannot take it: materialise dX and dW' now
        proj_bwd(lazy["x"], lazy["dz"], lazy["w3"], lazy["d_w"], defer=ctx.call.defer, dx_out=lazy["placeholder"])
        lazy = None
    grad_out = _as_f32c(grad_out, "grad_out")
    npg = 3 * f_in * f + 3 * k * f + k * k
    d_params = torch.empty(npg, dtype=_F32, device=dev)
    q = _lib.ConvAggBwd()
    q.f_in, q.f_pad, q.f_out = f_in, fp, f
    q.relu_after, q.relu_mlp, q.layernorm, q.scale = int(cfg.relu_after), int(cfg.relu_mlp), int(cfg.layernorm), cfg.scale
    q.grad_out, q.ld_grad_out = grad_out.data_ptr(), grad_out.stride(0)
    if lazy is not None:                              # the following layer's projection backward rides this launch
        dz2, w32 = lazy["dz"], lazy["w3"]
        q.grad_out = None
        q.proj_dz, q.ld_proj_dz = dz2.data_ptr(), dz2.stride(0)
        q.proj_w_low, q.proj_w_high, q.proj_w_mlp = (w.data_ptr() for w in w32)
        q.proj_ld_w, q.proj_f = w32[0].stride(0), w32[0].shape[1]
        q.proj_d_w = lazy["d_w"].data_ptr()
    q.agg, q.ld_agg = agg.data_ptr(), agg.stride(0)
    if getattr(ctx, "head_stats", None) is not None:
        q.head_stats, q.ld_head_stats = ctx.head_stats.data_ptr(), ctx.head_stats.stride(0)
    q.xs, q.ld_xs = xpad.data_ptr(), xpad.stride(0)
    q.w_low, q.w_high, q.w_mlp, q.ld_w = wl.data_ptr(), wh.data_ptr(), wm.data_ptr(), f
    q.att_vec, q.ln_weight, q.ln_bias = _ptr_array(vecs), _ptr_array(lnw), _ptr_array(lnb)
    q.att_mix = mix.data_ptr()
    q.d_params = d_params.data_ptr()
    q.post_relu = int(ctx.post_relu)
    q.n_channels = k
    if ctx.post_scale is not None:
        q.post_scale, q.ld_post_scale = ctx.post_scale.data_ptr(), ctx.post_scale.stride(0)
    spec = _drop_spec(ctx.post_drop, ops.row_offset)
    if spec is not None:
        q.post_drop = spec
    if out_fwd is not None:
        q.out, q.ld_out = out_fwd.data_ptr(), out_fwd.stride(0)
    if four:
        ps, s_local = saved[-2], saved[-1]
        gs = torch.empty(n, f, dtype=_F32, device=dev)            # D * dL/dpre_S
        q.ps, q.ld_ps = ps.data_ptr(), ps.stride(0)
        q.ss, q.ld_ss = s_local.data_ptr(), s_local.stride(0)
        q.deg = ops.deg.data_ptr()
        q.g_struc, q.ld_g_struc = gs.data_ptr(), gs.stride(0)
        q.g_struc_scale = None if ops.implicit else ops.deg.data_ptr()
    nbytes = C.c_size_t()
    _lib.check(lib.acm_conv_agg_bwd_workspace_bytes(n, f_in, f, C.byref(nbytes)))
    ws = torch.empty(max(nbytes.value // 4, 1), dtype=_F32, device=dev)
    defer = ctx.call.defer
    q.defer = defer.pointer() if defer is not None else None
    pipe = getattr(ctx, "pipe", None)
    carry = pipe is not None and pipe.next_table_ready and not pipe.next_agg_ready
    if carry:                                     # the next step's P = A_low dropout(x) rides this launch
        q.next_a = ops.low.handle
        q.next_xg, q.ld_next_xg = pipe.table().data_ptr(), pipe.table().stride(0)
        q.next_row_scale = ops.row_scale.data_ptr()
        q.next_agg, q.ld_next_agg = pipe.agg().data_ptr(), pipe.agg().stride(0)
    with _device_ctx(dev), _Timed(f"conv_agg_bwd{'+gather' if carry else ''}{'+proj' if lazy is not None else ''}/F{f}k{k}i{f_in}"):
        st = lib.acm_conv_agg_bwd(n, C.byref(q), _vp(ws), ws.numel() * 4, _stream())
    if st == 4 and (lazy is not None or carry):       # ACM_EUNSUPPORTED for this shape after all: the plain launch(es)
        if lazy is not None:
            proj_bwd(lazy["x"], lazy["dz"], lazy["w3"], lazy["d_w"], defer=defer, dx_out=lazy["placeholder"])
            q.grad_out, q.proj_dz, lazy = grad_out.data_ptr(), None, None
        if carry:                                     # the gather as its own launch, right here: the pipeline stays valid
            q.next_a, q.next_xg, q.next_row_scale, q.next_agg = None, None, None, None          # (also inside a capture, where
            spmm(ops.low, pipe.table(), out=pipe.agg(), row_scale=ops.row_scale)                 # nobody could prime() it again)
        with _device_ctx(dev), _Timed(f"conv_agg_bwd/F{f}k{k}i{f_in}"):
            st = lib.acm_conv_agg_bwd(n, C.byref(q), _vp(ws), ws.numel() * 4, _stream())
    _lib.check(st, "acm_conv_agg_bwd")
    if carry:
        pipe.next_agg_ready = True
    if defer is not None:
        defer.hold(ws, [d_params] + ([lazy["d_w"]] if lazy is not None else []),
                   keep=[d_params] + ([lazy["d_w"]._base if lazy["d_w"]._base is not None else lazy["d_w"]] if lazy is not None else []))
    d_struc = None
    if four:                                  # dS = A_low^T (D G_S) - G_S   (pattern-only: P G_S - G_S)
        gsg = _gather_rows(ops, gs)
        low_t = ops.low_t
        d_struc = torch.empty(n, f, dtype=_F32, device=dev)
        ws2 = low_t.workspace(f)
        o = _lib.SpmmOpts()
        o.sub, o.ld_sub = gs.data_ptr(), gs.stride(0)
        o.sub_scale = None if ops.implicit else ops.inv_deg.data_ptr()
        if cfg.gather_bf16 and 8 < f <= 64 and f % 2 == 0:       # bf16 gathered operand (the self term stays fp32)
            gsg = cast_bf16(gsg)
            o.g_bf16 = 1
        with _device_ctx(dev), _Timed(f"spmm_sub/{f}"):
            st = lib.acm_spmm_ex(low_t.handle, _vp(gsg), gsg.stride(0), f, _vp(d_struc), d_struc.stride(0),
                                 C.byref(o), _vp(ws2), ws2.numel() * 4, _stream())
        _lib.check(st, "acm_spmm_ex")
    if ops.sharded:
        import torch.distributed as dist
        if defer is not None:
            defer.allreduce(d_params, ops.group)
        else:
            dist.all_reduce(d_params, group=ops.group)
    wsz = f_in * f
    d_wl, d_wh, d_wm = (d_params[i * wsz:(i + 1) * wsz].view(f_in, f) for i in range(3))
    base = 3 * wsz
    pad = [None] * (4 - k)
    d_vec = [d_params[base + c * f: base + (c + 1) * f].view(f, 1) for c in range(k)] + pad
    if cfg.layernorm:
        d_lnw = [d_params[base + (k + c) * f: base + (k + 1 + c) * f] for c in range(k)] + pad
        d_lnb = [d_params[base + (2 * k + c) * f: base + (2 * k + 1 + c) * f] for c in range(k)] + pad
    else:
        d_lnw = d_lnb = [None] * 4
    d_mix = d_params[base + 3 * k * f:].view(k, k)
    return (None, d_wl, d_wh, d_wm, *d_vec, d_struc, d_mix, *d_lnw, *d_lnb, None, None, None, None, None, None, None, None, None)


AcmConvFunction._backward_agg = staticmethod(_backward_agg)


# --------------------------------------------------------------------------
# aggregate-first for WIDE dense inputs (16 < F_in <= 128), round 5
# --------------------------------------------------------------------------
AGG_WIDE_MIN_DEGREE = 12          # stored entries of A_low per row from which the wide aggregate-first form is taken


def agg_wide_supported(x, ops, cfg, f_in, f_out, post_scale=None, call=None, tail_layer=False):
    """The first layer of the arXiv-year / pokec class (ACM-Geometric/layers.py:101-104 with 128 / 65 input features, 64
    hidden): without a ReLU between projection and filter, A (X W) = (A X) W.  P = A_low drop(X) is ONE gather of F_in floats
    per edge (the literal form gathers 2 F = 128), and because the input takes no gradient the backward needs NO transposed
    gather at all:  dW_L = P^T G_L,  dW_H = X^T G_H - P^T G_H,  dW_I = X^T G_I  -- three tall-skinny products on the split-bf16
    matrix pipe.  Three-channel ACM layers; row-sharded: the ONE halo exchange of the layer is the all-gather of the dropped
    F_in-wide input rows (the literal form all-gathers 2 F-wide rows forward AND backward); tuning rewrites bit 1 switches it off."""
    from .graph import FilterOperators
    if not (tuning.HOST.rewrites & tuning.REWRITE_AGG_FIRST) or not isinstance(ops, FilterOperators):
        return False
    if not isinstance(x, torch.Tensor) or x.layout != torch.strided or x.dim() != 2 or x.dtype != _F32 or x.requires_grad:
        return False
    # (gather_dtype="bf16" takes this form too: its ONE gather reads the fp32 input, F_in x 4 bytes per edge -- no more than the
    #  2 F x 2 bytes of the literal form's bf16 tables, and exact)
    if cfg.relu_before or cfg.n_channels != 3 or f_out != 64 or not 16 < f_in <= 128 or x.shape[1] != f_in:
        return False
    if getattr(ops, "general", False) or int(getattr(ops, "hops", 1)) != 1 or x.shape[0] != ops.n_local:
        return False
    if tail_layer:
        return False
    if ops.sharded:
        # every rank must take the same form (its collectives differ from the literal form's): decided by what all ranks know,
        # the longest row block of the plan -- and row-sharded the rewrite pays at any degree (one F_in-wide halo exchange
        # instead of two 2 F-wide ones)
        import torch.distributed as dist
        return ops.n_gathered // dist.get_world_size(ops.group) >= 8192
    if x.shape[0] < 8192:
        return False
    # Where it pays (measured, profiles/r05_agg_wide.txt): the rewrite trades 4 F - F_in gathered floats per EDGE for one more
    # pass over ~3 KB per ROW (the dropped copy of X, the head as its own launch, a third more projection flops).  pokec-shaped
    # (mean degree 38, F_in 65): 12.1 -> 7.0 ms per step; arXiv-year-shaped (mean degree 15, F_in 128): 0.822 -> 0.792.
    return ops.low.nnz >= AGG_WIDE_MIN_DEGREE * x.shape[0]


class _AcmAggWide(torch.autograd.Function):
    """out, att = three-channel ACM layer in the aggregate-first form for a wide dense input (see agg_wide_supported).

    forward : [acm_dropout] -> acm_spmm_ex (P = A_low Xd) -> acm_conv_aggw_fwd (projections on the split-bf16 matrix pipe + head,
              one row-local kernel: pre_L = P W_L, pre_H = (Xd - P) W_H); with tuning rewrites bit 8 off: 2 x acm_gemm
              ([P W_L | P W_H], [Xd W_H | Xd W_I]) -> acm_conv_head_fwd
    backward: acm_conv_aggw_bwd (K3 + the three weight gradients, one kernel); with tuning rewrites bit 8 off (or a post_scale mask
              tensor): acm_conv_bwd_local (K3) -> 2 x acm_gemm TN ([P^T G_L | P^T G_H], [Xd^T G_H | Xd^T G_I])"""

    @staticmethod
    def forward(ctx, x, w_low, w_high, w_mlp, v_low, v_high, v_mlp, att_mix, lnw_low, lnw_high, lnw_mlp, lnb_low, lnb_high,
                lnb_mlp, ops, cfg, post_relu, post_scale, post_drop, call, in_drop, agg_holder=None):
        lib = _lib.load()
        ctx.set_materialize_grads(False)
        call = ctx.call = _call_or_ambient(call)
        call.next_proj = None                      # (the narrow projection hand-off rides the f_pad <= 16 kernels only)
        x = _as_f32c(x, "input")
        dev = x.device
        n, f_in = x.shape
        f, k = w_low.shape[1], 3
        fp = -(-f_in // 4) * 4                     # rows of 16-byte blocks for the gather and the split-bf16 products
        spec = _drop_spec(in_drop, ops.row_offset) if (in_drop is not None and in_drop[0] > 0) else None
        # ``agg_holder`` (layers.GraphConvolution._eval_agg_holder): P = A_low X and the padded copy of a STATIC input from the
        # previous pass over it -- every evaluation pass after the first and every training step of a model without input
        # dropout then skips the layer's gather (a third of the arXiv-year evaluation forward, two thirds of pokec's)
        if spec is not None or ops.sharded:
            agg_holder = None
        agg = None
        if spec is not None:                       # the caller's input dropout, written straight into the padded rows
            xd = torch.empty(n, fp, dtype=_F32, device=dev)
            with _device_ctx(dev), _Timed(f"dropout/{n}x{f_in}"):
                st = lib.acm_dropout(n, f_in, _vp(x), x.stride(0), _vp(xd), xd.stride(0), fp, C.byref(spec), _stream())
            _lib.check(st, "acm_dropout")
        elif fp == f_in:
            xd = x
        else:
            xd = agg_holder.get("xpad") if agg_holder is not None else None
            if xd is None or tuple(xd.shape) != (n, fp):
                xd = torch.nn.functional.pad(x, (0, fp - f_in))
                if agg_holder is not None:
                    agg_holder["xpad"], agg_holder["agg"] = xd, None
        if agg_holder is not None:
            agg = agg_holder.get("agg")
            if agg is not None and tuple(agg.shape) != (n, fp):
                agg = None
        if agg is None:
            # P = A_low Xd  [n, fp]; row-sharded: the operator's columns are the all-gathered rows (the layer's only halo exchange)
            agg = spmm(ops.low, _gather_rows(ops, xd), row_scale=ops.row_scale if ops.implicit else None)
            if agg_holder is not None:
                agg_holder["agg"] = agg
        w3 = [_as_f32c(w, "weight") for w in (w_low, w_high, w_mlp)]
        vecs = [_as_f32c(t, "att_vec") for t in (v_low, v_high, v_mlp)]
        lnw = [_as_f32c(t, "ln") for t in (lnw_low, lnw_high, lnw_mlp)] if cfg.layernorm else []
        lnb = [_as_f32c(t, "ln") for t in (lnb_low, lnb_high, lnb_mlp)] if cfg.layernorm else []
        mix = _as_f32c(att_mix, "att_vec")
        if post_scale is not None:
            post_scale = _as_f32c(post_scale, "post_scale")
        ctx.post_relu, ctx.post_scale = bool(post_relu), post_scale
        ctx.post_drop = post_drop if (post_drop is not None and post_drop[0] > 0) else None
        out = torch.empty(n, f, dtype=_F32, device=dev)
        att = torch.empty(n, 4, dtype=_F32, device=dev)
        pre = torch.empty(n, 2 * f, dtype=_F32, device=dev)
        p = _lib.ConvFwd()
        p.f_out, p.n_channels = f, k
        p.relu_after, p.relu_mlp, p.layernorm = int(cfg.relu_after), int(cfg.relu_mlp), int(cfg.layernorm)
        p.scale, p.row_offset = cfg.scale, ops.row_offset
        p.att_vec = _ptr_array(vecs)
        p.ln_weight, p.ln_bias = _ptr_array(lnw), _ptr_array(lnb)
        p.att_mix = mix.data_ptr()
        p.out, p.ld_out = out.data_ptr(), out.stride(0)
        p.pre, p.ld_pre = pre.data_ptr(), pre.stride(0)
        p.att = att.data_ptr()
        p.post_relu = int(ctx.post_relu)
        if post_scale is not None:
            p.post_scale, p.ld_post_scale = post_scale.data_ptr(), post_scale.stride(0)
        dspec = _drop_spec(ctx.post_drop, ops.row_offset)
        if dspec is not None:
            p.post_drop = dspec
        same_pitch = w3[0].stride(0) == w3[1].stride(0) == w3[2].stride(0)
        if (tuning.HOST.rewrites & tuning.REWRITE_AGGW_FUSED) and same_pitch:
            # projections + head behind the gather as ONE row-local kernel: pre_L = P W_L, pre_H = (Xd - P) W_H, Z_I = Xd W_I
            zi = torch.empty(n, f, dtype=_F32, device=dev)
            with _device_ctx(dev), _Timed(f"conv_aggw/F{f}k{k}i{f_in}"):
                st = lib.acm_conv_aggw_fwd(n, f_in, fp, _vp(agg), agg.stride(0), _vp(xd), xd.stride(0), _vp(w3[0]), _vp(w3[1]),
                                           _vp(w3[2]), w3[0].stride(0), _vp(zi), zi.stride(0), C.byref(p), _stream())
            _lib.check(st, "acm_conv_aggw_fwd")
        else:
            pad = (0, 0, 0, fp - f_in)
            wa = torch.nn.functional.pad(torch.cat((w3[0], w3[1]), 1), pad) if fp != f_in else torch.cat((w3[0], w3[1]), 1)
            wb = torch.nn.functional.pad(torch.cat((w3[1], w3[2]), 1), pad) if fp != f_in else torch.cat((w3[1], w3[2]), 1)
            za = gemm(agg, wa)                         # [P W_L | P W_H]
            zb = gemm(xd, wb)                          # [Xd W_H | Xd W_I]
            zi = zb[:, f:]
            p.g_low, p.ld_g_low = za.data_ptr(), za.stride(0)                 # "gathered" over I: pre_L = 1 * (P W_L)
            p.g_high, p.ld_g_high = za.data_ptr() + 4 * f, za.stride(0)       #                    pre_H = Xd W_H - 1 * (P W_H)
            p.s_high, p.ld_s_high = zb.data_ptr(), zb.stride(0)
            p.s_mlp, p.ld_s_mlp = zb.data_ptr() + 4 * f, zb.stride(0)
            with _device_ctx(dev), _Timed(f"conv_head/F{f}k{k}"):           # the fused epilogue as a row-local kernel of its own
                st = lib.acm_conv_head_fwd(n, C.byref(p), _stream())
            _lib.check(st, "acm_conv_head_fwd")
        ctx.ops, ctx.cfg, ctx.f_in, ctx.fp = ops, cfg, f_in, fp
        ctx.save_for_backward(xd, agg, zi, pre, mix, *vecs, *lnw, *lnb)
        ctx.mark_non_differentiable(att)
        return out, att

    @staticmethod
    def backward(ctx, grad_out, _grad_att):
        if grad_out is None:
            return (None,) * 22
        lib = _lib.load()
        ops, cfg, f_in, fp = ctx.ops, ctx.cfg, ctx.f_in, ctx.fp
        xd, agg, zi, pre, mix, *rest = ctx.saved_tensors
        k = 3
        vecs = rest[:k]
        lnw = rest[k:2 * k] if cfg.layernorm else []
        lnb = rest[2 * k:3 * k] if cfg.layernorm else []
        n, dev = xd.shape[0], xd.device
        f = pre.shape[1] // 2
        defer = ctx.call.defer
        grad_out = _as_f32c(grad_out, "grad_out")
        st3 = _k3_setup(cfg, ops, k, f, n, dev, f_in, pre, zi, vecs, lnw, lnb, mix, grad_out, ctx.post_relu, ctx.post_scale,
                        ctx.post_drop)
        q, flat, nw = st3["q"], st3["flat"], st3["nw"]
        q.g_scale = None                           # (the filter was applied before the projection: no transposed gather here)
        d_vec, d_lnw, d_lnb, d_mix = _flat_views(flat, nw, k, f, cfg.layernorm)
        q.defer = defer.pointer() if defer is not None else None
        dw = flat[:nw].view(3, f_in, f)            # the weight gradients lead the layer's flat gradient buffer
        if n == 0:                                 # a rank without rows (row-sharded, degenerate plan): zero partial sums, same collectives
            flat.zero_()
            del st3
        elif (tuning.HOST.rewrites & tuning.REWRITE_AGGW_FUSED) and ctx.post_scale is None:
            # K3 and the three weight gradients in ONE kernel: [G_L | G_H | G_I] never reach memory
            q.g_low = q.g_high = q.g_mlp = None
            nbytes = C.c_size_t()
            _lib.check(lib.acm_conv_aggw_bwd_workspace_bytes(n, fp, C.byref(nbytes)), "acm_conv_aggw_bwd_workspace_bytes")
            ws = torch.empty(max(nbytes.value // 4, 1), dtype=_F32, device=dev)
            with _device_ctx(dev), _Timed(f"conv_aggw_bwd/F{f}k{k}i{f_in}"):
                st = lib.acm_conv_aggw_bwd(n, f_in, fp, _vp(agg), agg.stride(0), _vp(xd), xd.stride(0), C.byref(q), _vp(dw[0]),
                                           _vp(dw[1]), _vp(dw[2]), f, _vp(ws), ws.numel() * 4, _stream())
            _lib.check(st, "acm_conv_aggw_bwd")
            if defer is not None:
                defer.hold(ws, [d_mix, *d_vec, *d_lnw, *d_lnb, dw], keep=[flat])
            del st3
        else:
            # K3 writes [G_L | G_H | G_I] side by side, UNSCALED; two transposed split-bf16 products read them back
            gcat = torch.empty(n, 3 * f, dtype=_F32, device=dev)
            q.g_low, q.ld_g_low = gcat.data_ptr(), gcat.stride(0)
            q.g_high, q.ld_g_high = gcat.data_ptr() + 4 * f, gcat.stride(0)
            q.g_mlp, q.ld_g_mlp = gcat.data_ptr() + 8 * f, gcat.stride(0)
            nbytes = C.c_size_t()
            _lib.check(lib.acm_conv_bwd_local_workspace_bytes(n, f, k, C.byref(nbytes)))
            ws = torch.empty(max(nbytes.value // 4, 1), dtype=_F32, device=dev)
            with _device_ctx(dev), _Timed(f"conv_bwd_local/F{f}k{k}"):
                st = lib.acm_conv_bwd_local(n, C.byref(q), _vp(ws), ws.numel() * 4, _stream())
            _lib.check(st, "acm_conv_bwd_local")
            if defer is not None:
                defer.hold(ws, [d_mix, *d_vec, *d_lnw, *d_lnb], keep=[flat])
            del st3
            a1 = gemm(agg, gcat[:, : 2 * f], trans_a=True, col_blocks=2)          # [P^T G_L | P^T G_H]   as [2, fp, f]
            a2 = gemm(xd, gcat[:, f:], trans_a=True, col_blocks=2)                # [Xd^T G_H | Xd^T G_I]
            dw[0].copy_(a1[0][:f_in])
            torch.sub(a2[0][:f_in], a1[1][:f_in], out=dw[1])
            dw[2].copy_(a2[1][:f_in])
        d_wl, d_wh, d_wm = dw[0], dw[1], dw[2]
        if ops.sharded:
            # replicated parameters: the weight gradients sit with the head's in the layer's flat buffer, ONE all-reduce sums the
            # row-shard partials (after the step's single flush when the second phases are deferred)
            import torch.distributed as dist
            if defer is not None:
                defer.allreduce(flat, ops.group)
            else:
                dist.all_reduce(flat, group=ops.group)
        none3 = [None] * 3
        return (None, d_wl, d_wh, d_wm, d_vec[0], d_vec[1], d_vec[2], d_mix,
                *(d_lnw if cfg.layernorm else none3), *(d_lnb if cfg.layernorm else none3), None, None, None, None, None, None, None,
                None)


def in_drop_supported(x, ops, cfg, f_in, f_out):
    """Whether a layer can take its caller's input dropout into its dense projection (AcmConvFunction ``in_drop``): the
    literal form on the MFMA GEMM (not aggregate-first, not the narrow streaming projection, not CSR features), an input
    that needs no gradient, shapes the row-panel GEMMs cover."""
    if isinstance(x, SparseFeatures) or not isinstance(x, torch.Tensor) or x.requires_grad or x.dim() != 2:
        return False
    if x.shape[1] != f_in or x.dtype != _F32 or not x.is_contiguous():
        return False
    agg_first = not cfg.relu_before and f_in <= 16 and f_in < f_out and f_out <= 64
    recompute = cfg.relu_before and f_out == 64 and f_in <= 8
    narrow = f_out <= 5 and f_in <= 64
    if agg_first or recompute or narrow or f_out in (2, 4, 8):
        return False
    fb = _chan_block(f_out)
    return gemm_drop_supported(x.shape[0], f_in, 2 * fb + f_out)


def acm_conv(x, params, ops, cfg, post_relu=False, post_scale=None, post_drop=None, call=None, tail_layer=False,
             agg_holder=None, in_drop=None):
    """params: dict with the reference's parameter names (see layers.GraphConvolution).
    post_relu / post_scale: optional fused ``relu(out) * post_scale`` (the caller's inter-layer
    ReLU + dropout; post_scale = keep_mask / (1 - p)).  post_drop = (p, tag, DropoutState): the same
    dropout with the mask generated in registers (acm_dropout_t) instead of read from a tensor.
    call / tail_layer / agg_holder: see AcmConvFunction.forward."""
    p = params
    if agg_wide_supported(x, ops, cfg, p["weight_low"].shape[0], p["weight_low"].shape[1], post_scale, call, tail_layer):
        return _run(_AcmAggWide, x, p["weight_low"], p["weight_high"], p["weight_mlp"], p["att_vec_low"], p["att_vec_high"],
                                 p["att_vec_mlp"], p["att_vec"], p["layer_norm_low.weight"], p["layer_norm_high.weight"],
                                 p["layer_norm_mlp.weight"], p["layer_norm_low.bias"], p["layer_norm_high.bias"],
                                 p["layer_norm_mlp.bias"], ops, cfg, post_relu, post_scale, post_drop, call, in_drop, agg_holder)
    return _run(
        AcmConvFunction, x, p["weight_low"], p["weight_high"], p["weight_mlp"], p["att_vec_low"], p["att_vec_high"],
        p["att_vec_mlp"], p["att_struc_low"], p["struc_low"], p["att_vec"],
        p["layer_norm_low.weight"], p["layer_norm_high.weight"], p["layer_norm_mlp.weight"],
        p["layer_norm_struc_low.weight"], p["layer_norm_low.bias"], p["layer_norm_high.bias"],
        p["layer_norm_mlp.bias"], p["layer_norm_struc_low.bias"], ops, cfg, post_relu, post_scale, post_drop,
        call, tail_layer, agg_holder, in_drop)
