"""Training / evaluation harness around the ACM model: the callers of the hot path
(ACM-Geometric/train.py:107-162, data_utils.py:115-168; ACM-Pytorch/train.py:49-139,
utils.py:547-574), restated so the op can be trained and measured where the reference's Python
cannot run.

* ``TrainStep``      one full-batch step: forward, fused masked NLL (acm_nll_loss), backward,
                     optimizer update; optionally captured once in a HIP graph and replayed
                     (every launch of the step is stream-ordered and allocation-free, so the
                     whole step is one graph launch instead of ~60 kernel launches).
* ``evaluate``       eval-mode forward + accuracy on index sets, on device (the reference pulls the
                     predictions to the host three times per epoch, data_utils.py:117-118).
* ``fit``            the two model-selection rules of the reference.
"""
import torch
import torch.nn.functional as F

from . import functional as AF
from .graph import FilterOperators, SparseFeatures, operators_for

_TORCH_DROPOUT = F.dropout          # to notice a patched F.dropout (mask replay in tests): see TrainStep


def row_weights(train_idx, n_rows, n_train_total=None, device=None):
    """w_i = 1/|train| on training rows else 0 (mean NLL over the training set)."""
    device = device if device is not None else train_idx.device
    w = torch.zeros(n_rows, dtype=torch.float32, device=device)
    total = n_train_total if n_train_total is not None else train_idx.numel()
    w[train_idx.to(device)] = 1.0 / float(total)
    return w


def _tape_safe(model):
    """models.GCN configurations whose training forward is a chain of this package's Functions only (nothing for a Tape to
    trip over): the two-layer ACM-GCN / ACM-GCN+ and the single-layer ACM-SGC.  ACM-GCN++ adds its residual branch with a
    torch operation unless the fused dropout's one-launch form applies; acmsnowball concatenates blocks."""
    from .models import GCN
    return type(model) is GCN and model.model_type in ("acmgcn", "acmgcnp", "acmsgc")


def _rng_snapshot(device):
    """States of the generators F.dropout draws from: torch's CPU generator and the device's."""
    dev = torch.device(device)
    cuda = torch.cuda.get_rng_state(dev) if (dev.type == "cuda" and not torch.cuda.is_current_stream_capturing()) else None
    return torch.get_rng_state(), cuda, dev


def _rng_restore(snap):
    if snap is None:
        return
    cpu, cuda, dev = snap
    torch.set_rng_state(cpu)
    if cuda is not None:
        torch.cuda.set_rng_state(cuda, dev)


def _held_entries(model):
    """What a captured pass over ``model`` reads beyond its graph's own pool: the layers' P = A_low X cache entries
    (layers.GraphConvolution._eval_agg_holder: key, input, {"agg": P, "xpad": ...}, operators) as they are right now."""
    held = []
    for m in model.modules():
        entries = getattr(m, "held_entries", None)
        if entries is not None:
            held.extend(entries())
    return held


def _capture_mode():
    """Keyword arguments of torch.cuda.graph for a capture.  With a process group alive, ProcessGroupNCCL's watchdog
    thread polls the events of earlier collectives (hipEventQuery) whenever it wakes up; under the default GLOBAL capture
    mode such a call from another thread while this thread captures is an error that aborts the process
    ("operation not permitted when stream is capturing", seen on a single-rank RCCL run).  Thread-local mode confines the
    restriction to the capturing thread."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return {"capture_error_mode": "thread_local"}
    return {}


class TrainStep:
    def __init__(self, model, optimizer, x, adj, labels, weights, adj_high=None, adj_un=None, use_graph=False,
                 fused_dropout=None, pipeline_input=None, steps_per_graph=1, flush_in_optimizer=True, small_step=None, tape=True):
        """``tape``: eager steps record this package's Functions on a functional.Tape and replay them backwards themselves
        instead of building an autograd graph (no Function.apply, no AccumulateGrad nodes, no engine hand-off: the host side of
        an eager step); a model with torch operations between its layers falls back to autograd on its first step, for good.

        ``small_step``: the fused six-launch step for small graphs (small.SmallPlan / acm_small_step): None = where it
        applies (``self.small`` is the plan, ``self.small_refused`` the reason it does not), False = never.  On that path
        ``p.grad`` stays None and neither ``model.forward`` nor ``optimizer.step`` is called (see small.SmallPlan): a model,
        parameter or optimizer with hooks registered keeps the general path.

        ``flush_in_optimizer``: with this package's FusedAdam / FusedAdamW the step's deferred gradient sums are flushed by
        the optimizer's own launch (acm_adam_config_t.pending) instead of a launch of their own.

        ``steps_per_graph`` (with ``use_graph``): capture that many consecutive optimizer steps in ONE hipGraph, so that a
        call runs them all (``steps_per_call``; the loss returned is the last one, ``losses`` holds every one) -- the ~8 us
        between two graph launches is then paid once per call instead of once per step.  Every captured step is a complete
        step (fresh counter-based masks: the step counters live on the device); a loop that looks at the model between two
        steps (train.fit's evaluation pass) keeps the default of one."""
        self.model, self.opt = model, optimizer
        self.steps_per_call = max(int(steps_per_graph), 1) if use_graph else 1
        self.losses = []
        self.x, self.adj, self.adj_high, self.adj_un = x, adj, adj_high, adj_un
        self.labels, self.weights = labels, weights
        # Relabelled operators (graph.relabel_by_degree; operators_for applies it to large graphs): the static inputs
        # of the step -- features, labels, row weights -- are moved into the operator's numbering ONCE, and the step
        # never leaves it (the loss does not care about the order of the rows).
        self._permuted = False
        ops = adj
        if not isinstance(adj, FilterOperators) and isinstance(adj, torch.Tensor) and hasattr(model, "structure_info"):
            four = model.structure_info and getattr(model, "model_type", "") in ("acmgcnp", "acmgcnpp")
            ops = operators_for(adj, adj_high, adj_un if four else None)
            # the step keeps the operator set it built (the model's forward would look the same set up again, per call, by
            # the tensors' identity): a caller that hands over the reference's adjacency TENSORS -- the drop-in loop,
            # ACM-Pytorch/train.py:95-139 -- then reaches every plan that asks for FilterOperators, the fused small-graph
            # step first of all (round 5: such a caller silently stayed on the general path)
            self.adj = ops
        if hasattr(model, "auto_csr"):
            # wide one-hot / bag-of-words features handed over dense: the CSR twin, made here once (tuning csr_features)
            self.x = x = model.auto_csr(x, ops if isinstance(ops, FilterOperators) else None)
        if isinstance(ops, FilterOperators) and ops.perm is not None and hasattr(model, "_forward"):
            self.adj = ops
            self.x = x.permute_rows(ops.perm) if isinstance(x, SparseFeatures) else x.index_select(0, ops.perm)
            self.labels, self.weights = labels.index_select(0, ops.perm), weights.index_select(0, ops.perm)
            self._permuted = True
        self.graph, self.loss = None, None
        self._held = None                   # what a captured step reads beyond its graph's pool (see _capture)
        # counter-based dropout: this loop owns the step structure (one advance per optimizer step), so the
        # model may draw its masks inside the kernels; the advance rides FusedAdam's step-counter kernel
        # (default: on, unless someone replaced F.dropout -- a mask-replay harness must keep seeing its masks)
        self._manual_advance = False
        self._defer = True
        self._tape = bool(tape)
        self._tape_proven = False           # a first taped step went through: the model's Functions are all this package's
        if self._tape and F.dropout is not _TORCH_DROPOUT and getattr(model, "dropout", 0) > 0 and not _tape_safe(model):
            # someone replaced F.dropout (a mask-replay harness: it hands out recorded masks by call position) and this model
            # may break the tape (torch operations between its layers): the redo of the step on autograd would draw the NEXT
            # masks of the harness -- its state cannot be put back like a generator's.  Such a step starts on autograd.
            self._tape = False
        from .optim import _FusedAdamBase
        self._opt_flushes = bool(flush_in_optimizer) and isinstance(optimizer, _FusedAdamBase)
        self._unflushed = None
        if fused_dropout is None:
            fused_dropout = F.dropout is _TORCH_DROPOUT
        if fused_dropout and getattr(model, "dropout", 0) > 0 and hasattr(model, "fused_dropout"):
            model.fused_dropout = True
            if model.dropout_state is None:
                model.dropout_state = AF.DropoutState(labels.device)
            if hasattr(optimizer, "also_advance"):
                optimizer.also_advance = model.dropout_state.step
            else:
                self._manual_advance = True
        self._params = list(model.parameters())          # walked every step: module.parameters() costs 0.1 ms of host time
        import inspect
        try:                                             # models.GCN takes the step's CallContext as a keyword
            self._takes_call = "call" in inspect.signature(model.forward).parameters
        except (TypeError, ValueError):
            self._takes_call = False
        # the first layer's input aggregation one step ahead, inside its own backward (functional.InputPipeline);
        # pipeline_input: None = when the configuration qualifies (InputPipeline.eligible), False = never
        self.pipe = None
        if pipeline_input is not False and not self._manual_advance and getattr(model, "fused_dropout", False) \
                and AF.InputPipeline.eligible(model, self.adj, self.x):
            self.pipe = AF.InputPipeline(self.adj, self.x, model.dropout, model.dropout_state, tag=0)
        # small graphs: the whole step behind one C-ABI call (six launches, updates applied where gradients finish)
        self.small, self.small_refused = None, "not requested"
        if small_step is not False:
            from .small import SmallPlan
            why = "the loop advances the dropout counter by hand" if self._manual_advance else \
                SmallPlan.why_not(model, self.x, self.adj, optimizer)
            if why is None and getattr(model, "dropout", 0) > 0 and not getattr(model, "fused_dropout", False):
                why = "F.dropout masks (counter-based dropout only)"
            self.small_refused = why
            if why is None:
                self.small = SmallPlan(model, self.x, self.adj, self.labels, self.weights, optimizer)
                self.pipe = None
        if use_graph:
            self._capture()

    def _one_step(self):
        """Forward + loss + backward + update (+ the dropout counter's advance): one step, eager or under capture."""
        if self.small is not None:
            loss = self.small.run()
        else:
            loss = self._forward_backward(for_optimizer=True)
            self._opt_step()
        self._count_advance()
        if self.pipe is not None:
            self.pipe.end_step()
        return loss

    def _forward_backward(self, for_optimizer=False):
        """Forward, fused loss and backward; the loss sum and the parameter-gradient sums of the backward kernels run
        as ONE deferred launch at the end (AF.deferred_reductions: four launches less per step).  That is only sound
        while nothing reads a gradient before the flush, which this method checks on every pass: every ``.grad``
        must be the tensor the backward kernels wrote (autograd adopts it when ``.grad`` is None), not a copy taken
        before the flush -- otherwise deferral is switched off for good and the step is redone.
        ``for_optimizer``: the caller runs ``_opt_step()`` next, which may take the flush into the optimizer's launch; without
        it the gradients and the loss are complete on return."""
        pipe = self.pipe
        if pipe is not None and pipe.stale():
            if self.x.is_cuda and torch.cuda.is_current_stream_capturing():
                self.pipe = pipe = None           # its buffers cannot be refilled inside a capture: the plain step
            else:
                pipe.prime()
        # this step's own context: the deferral list, the loss-tail request and the pipeline travel with the model call
        # (and, captured by the autograd Functions, with its backward) -- nothing is parked in module state
        pending = AF.DeferredReductions() if self._defer else None
        call = AF.CallContext(defer=pending, pipe=pipe)
        # eager steps: the Functions on the step's own tape (a capture keeps autograd: nothing to save in a replayed graph)
        tape = AF.Tape() if (self._tape and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())) else None
        # until a taped step has gone through once, the attempt may have to be redone on autograd: the generators' states are
        # put back first, so that the redo draws the masks (F.dropout) the aborted forward drew -- the step then IS the step an
        # autograd run of the same seeds takes (acmgcnpp trained on other masks with tape=True than with False)
        rng = _rng_snapshot(self.labels.device) if (tape is not None and not self._tape_proven) else None
        try:
            with AF.on_tape(tape):
                loss, dz, out = self._forward_loss(call)
            if pipe is not None:
                pipe.make_next()              # dropout_{t+1}(x): the operand of the gather the backward carries (only if
                                              # the forward adopted the pipeline's buffers)
            if tape is not None:
                tape.backward(out, dz)
                self._tape_proven = True
            else:
                out.backward(dz)
        except (AF.TapeBroken, RuntimeError) as err:
            if tape is not None:
                tape.release()
            if pending is not None:
                pending.discard()
            # an in-place torch operation on a taped output fails inside torch ("a leaf Variable that requires grad is being
            # used in an in-place operation") before the tape can notice: the same verdict
            broken = tape is not None and not self._tape_proven and (isinstance(err, AF.TapeBroken) or "leaf Variable" in str(err))
            if not broken:
                raise
            # torch operations between the layers: this model's steps run on autograd from now on; the step is redone
            self._tape = False
            _rng_restore(rng)
            self.opt.zero_grad(set_to_none=True)
            if pipe is not None:
                pipe.primed = False
            return self._forward_backward(for_optimizer)
        except BaseException:
            if tape is not None:
                tape.release()
            if pending is not None:
                pending.discard()
            raise
        if pending is None:
            return loss
        adopted = pending.all_adopted([loss] + [p.grad for p in self._params])
        # The optimizer's launch can flush the list itself (acm_adam_config_t.pending: one launch and one grid drain less)
        # when nothing has to come between the flush and the update: no gradient all-reduce, our own optimizer.
        self._unflushed = None
        if for_optimizer and adopted and self._opt_flushes and not pending.collectives_pending:
            self._unflushed = pending
            return loss
        pending.flush()
        if not adopted:
            self._defer = False
            self.opt.zero_grad(set_to_none=True)
            if pipe is not None:
                pipe.primed = False               # the first pass has already refilled its buffers for the next step
            return self._forward_backward(for_optimizer)
        return loss

    def _forward_loss(self, call):
        """Forward and fused loss: (loss, dloss/dlogits, logits).  The model's output layer is asked to run its row
        phase, the loss and its own row-local backward as one kernel (AF.fused_loss_tail); when it does not qualify
        (wide output, structure channel, a wrapper around the output) the loss is its own launch."""
        tail = call.tail = AF.fused_loss_tail(self.labels, self.weights)
        kw = {"call": call} if self._takes_call else {}
        if self._permuted:
            kw["rows_permuted"] = True
        if self._takes_call:
            out = self.model(self.x, self.adj, self.adj_high, self.adj_un, **kw)
        else:                                     # a wrapper without the ``call`` keyword: the thread's ambient context
            with AF.deferred_reductions_as(call.defer), AF.input_pipeline(call.pipe), tail:
                out = self.model(self.x, self.adj, self.adj_high, self.adj_un, **kw)
        call.tail = None
        if tail.matches(out):
            return tail.loss, tail.dz, out
        loss, dz = AF.nll_loss_and_grad(out, self.labels, self.weights, defer=call.defer)   # = masked_nll(...).backward(), two launches less
        return loss, dz, out

    def _eager(self):
        if not self.model.training:
            self.model.train()
        if type(self.opt).zero_grad is torch.optim.Optimizer.zero_grad:
            for group in self.opt.param_groups:      # = the stock zero_grad(set_to_none=True) without its dispatch wrapper (30 us)
                for p in group["params"]:
                    p.grad = None
        else:
            self.opt.zero_grad(set_to_none=True)
        loss = self._one_step()
        return loss                 # never hand out the autograd graph: a live AccumulateGrad node pins its
                                    # stream and breaks a later graph capture

    def _snapshot(self):
        """Values of everything a training step changes: parameters / buffers, optimizer state, dropout counter."""
        model_state = [(t, t.detach().clone()) for t in self.model.state_dict().values() if torch.is_tensor(t)]
        opt_state = [(p, {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()})
                     for p, st in self.opt.state.items()]
        ds = getattr(self.model, "dropout_state", None)
        return model_state, opt_state, ((ds.step.clone(), ds.host_steps) if ds is not None else None)

    def _restore(self, snap):
        """Put the snapshot back IN PLACE (the captured graph and the optimizer's pointer tables keep their addresses);
        optimizer state created since the snapshot is zeroed -- what a fresh optimizer starts from."""
        model_state, opt_state, drop_step = snap
        with torch.no_grad():
            for t, v in model_state:
                t.copy_(v)
            had = {id(p): st for p, st in opt_state}
            for p, st in self.opt.state.items():
                old = had.get(id(p))
                for k, v in st.items():
                    if not torch.is_tensor(v):
                        if old is not None and k in old:
                            st[k] = old[k]
                        continue
                    if old is not None and k in old:
                        v.copy_(old[k])
                    else:
                        v.zero_()
            if drop_step is not None:
                self.model.dropout_state.step.copy_(drop_step[0])
                self.model.dropout_state.host_steps = drop_step[1]
        if getattr(self, "pipe", None) is not None:
            self.pipe.primed = False            # its buffers belong to another counter value now

    def _capture(self):
        # NOTE for callers: no autograd graph of an earlier, un-captured backward through this model may still be alive
        # (e.g. a retained `out` of `out = model(x); loss(out).backward()`): its AccumulateGrad nodes are bound to the
        # stream they were created on, and the captured backward would have to synchronise with that stream --
        # torch aborts the capture (a segmentation fault in capture_end on ROCm).  Drop such references first.
        # The warm-up runs real steps (lazy handles, allocator pools, optimizer state) -- on a snapshot: the model,
        # the optimizer moments / step counts and the dropout counter are put back afterwards, so a captured
        # TrainStep starts from exactly the state an eager one starts from (fit(epochs=N) trains N steps, not N + 3).
        snap = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):           # warm-up off the capture: lazy handles, allocator pools
            for _ in range(3):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._restore(snap)
        if self.pipe is not None:
            self.pipe.prime()                   # for the restored counter, outside the capture
            torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.model.train()
        self.opt.zero_grad(set_to_none=True)
        losses = []
        with torch.cuda.graph(self.graph, **_capture_mode()):
            for k in range(self.steps_per_call):
                if k:
                    self.opt.zero_grad(set_to_none=True)
                loss = self._one_step()
                if self.small is not None and self.steps_per_call > 1:
                    loss = loss.clone()          # the plan's loss scalar is rewritten by the next step of the same graph
                losses.append(loss)
            self.loss, self.losses = loss, losses
        del loss, losses
        # A dropout-0 model reuses the first layer's P = A_low X from the layer's cache (layers._eval_agg_holder): the
        # graph bakes in the address of the P its capture read.  Only the layer's cache entry keeps that tensor alive, and
        # another forward of the same model on another input object replaces the entry -- held here (as EvalStep does), the
        # replay keeps reading valid memory whatever the layers cache afterwards (use-after-free).
        self._held = _held_entries(self.model)

    def refresh(self):
        """Re-capture after an in-place edit of the features (or of anything else the captured step read once: a captured
        step assumes ``x`` is static, like EvalStep)."""
        if self.graph is not None:
            self.graph = None
            if self.pipe is not None:
                self.pipe.primed = False
            self._capture()

    def _opt_step(self):
        pending, self._unflushed = self._unflushed, None
        if pending is not None:
            self.opt.step(pending=pending)
        else:
            self.opt.step()

    def _count_advance(self):
        """The dropout counter moves once per optimizer step: by FusedAdam's kernel (also_advance) or by hand."""
        ds = getattr(self.model, "dropout_state", None)
        if self._manual_advance:
            ds.advance()
        elif ds is not None and getattr(self.opt, "also_advance", None) is ds.step:
            ds.host_steps += 1

    def __call__(self):
        if self.graph is not None:
            self.graph.replay()
            ds = getattr(self.model, "dropout_state", None)
            if ds is not None:
                ds.host_steps += self.steps_per_call      # the replayed step(s) advanced the device counter
                if self.pipe is not None:     # ... and refilled the pipeline's buffers for the new value
                    self.pipe._host_steps = ds.host_steps
            return self.loss
        return self._eager()


@torch.no_grad()
def evaluate(model, x, adj, labels, index_sets, adj_high=None, adj_un=None):
    """Eval-mode logits and the accuracy on each index set (data_utils.py:153-168)."""
    model.eval()
    out = model(x, adj, adj_high, adj_un)
    pred = out.argmax(dim=1)
    accs = [float((pred[idx] == labels[idx]).float().mean()) for idx in index_sets]
    return out, accs


class EvalStep:
    """The per-epoch evaluation pass of the reference's loops (ACM-Geometric/train.py:138-140 + data_utils.py:153-168,
    ACM-Pytorch/train.py:129-139): eval-mode logits, the accuracy on every index set and the NLL on one of them
    (``loss_set``: the validation set).  Everything stays on the device -- the index sets become [k, n] averaging
    weights, accuracies and loss land in one small tensor -- so a call costs ONE device-to-host copy, and with
    ``use_graph`` the whole pass (the library's forward kernels + a handful of torch reductions) is a hipGraph replay.
    Returns (logits, [accuracy per index set], loss on ``loss_set``).

    Index sets may be index tensors or boolean masks.  Rows outside every set may carry the reference's "unlabeled"
    marker -1 (data_utils.rand_train_test_idx, ignore_negative): they have zero weight and their label is clamped before
    the gather.  A CAPTURED pass bakes in the addresses of what its warm-up left behind -- in particular the first layer's
    P = A_low X of an aggregate-first layer (layers.GraphConvolution._eval_agg) -- so it assumes ``x`` is not modified
    in place afterwards (call :meth:`refresh` if it was) and it keeps those tensors alive itself: another evaluation of
    the same model on other inputs may replace the layers' cache entries, the replay still reads valid memory.

    ``small_step`` (None = where it applies): small graphs run the forward as three launches behind one C-ABI call
    (small.SmallPlan) WITHOUT calling ``model.forward`` -- a model with forward hooks keeps the general path."""

    def __init__(self, model, x, adj, labels, index_sets, adj_high=None, adj_un=None, loss_set=1, use_graph=False,
                 small_step=None, fused_metrics=True):
        self.model, self.x, self.adj, self.adj_high, self.adj_un = model, x, adj, adj_high, adj_un
        self.fused_metrics, self._metrics = bool(fused_metrics), None
        self.labels = labels
        self._labels_safe = labels.clamp_min(0)             # -1 = unlabeled (never in an index set)
        n, dev = labels.shape[0], labels.device
        w = torch.zeros(len(index_sets), n, dtype=torch.float32, device=dev)
        for k, idx in enumerate(index_sets):
            idx = torch.as_tensor(idx, device=dev)
            idx = idx.nonzero().view(-1) if idx.dtype == torch.bool else idx.long()
            w[k].index_fill_(0, idx, 1.0 / max(int(idx.numel()), 1))
        self.w, self.loss_set = w, int(loss_set)
        # small graphs: the evaluation forward as one call (small.SmallPlan without an optimizer)
        self.small, self.small_refused = None, "not requested"
        if small_step is not False:
            from .graph import FilterOperators, operators_for
            from .small import SmallPlan
            ops = adj
            if not isinstance(adj, FilterOperators) and isinstance(adj, torch.Tensor) and hasattr(model, "structure_info"):
                four = model.structure_info and getattr(model, "model_type", "") in ("acmgcnp", "acmgcnpp")
                ops = operators_for(adj, adj_high, adj_un if four else None)
            xs = model.auto_csr(x, ops if isinstance(ops, FilterOperators) else None) if hasattr(model, "auto_csr") else x
            self.small_refused = SmallPlan.why_not(model, xs, ops, None, need_dropout_state=False)
            if self.small_refused is None:
                self.small = SmallPlan(model, xs, ops)
        self.graph, self.out, self.res = None, None, None
        self._held = None
        self._use_graph = bool(use_graph)
        if use_graph:
            self._capture()

    def refresh(self):
        """Re-capture after an in-place edit of the features (or of anything else the captured pass read once)."""
        if self._use_graph:
            self.graph = None
            self._capture()

    @torch.no_grad()
    def _run(self):
        self.model.eval()
        if self.small is not None:
            out = self.small.run()                       # three launches, one C-ABI call (acm_small_step, train = 0)
        else:
            out = self.model(self.x, self.adj, self.adj_high, self.adj_un)
        if self._metrics_ok(out):
            # accuracy on every index set + the NLL on one of them as ONE launch over the logits (acm_eval_metrics, ABI 28)
            # instead of argmax / compare / log_softmax / gather / matmul / sum / cat (eight torch launches)
            return out, AF.eval_metrics(out, self.labels, self.w, self.loss_set, self._metrics_buffers(out))
        correct = (out.argmax(dim=1) == self.labels).to(torch.float32)
        nll = -F.log_softmax(out, 1).gather(1, self._labels_safe.view(-1, 1)).view(-1)
        res = torch.cat([self.w @ correct, (self.w[self.loss_set] * nll).sum().view(1)])
        return out, res

    def _metrics_ok(self, out):
        return (self.fused_metrics and out.dim() == 2 and out.shape[1] <= 64 and 1 <= self.w.shape[0] <= 8
                and out.dtype == torch.float32 and out.stride(1) == 1 and self.labels.dtype == torch.int64
                and self.labels.dim() == 1 and self.labels.is_contiguous())

    def _metrics_buffers(self, out):
        """(result [k + 1], zero-initialised workspace) of acm_eval_metrics, made once per pass object: a captured pass bakes
        their addresses in."""
        if self._metrics is None:
            self._metrics = AF.eval_metrics_buffers(out.shape[0], self.w.shape[0], out.device)
        return self._metrics

    def _capture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):           # warm-up off the capture (lazy handles, allocator pools)
            for _ in range(2):
                self._run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, **_capture_mode()):
            self.out, self.res = self._run()
        # what the captured kernels read beyond the pool of the graph: the layers' evaluation-pass cache entries
        # (key, input, {"agg": P}, operators) as they are now -- held here so that a later eval-mode forward of the same
        # model on another input cannot free them under the replay
        self._held = _held_entries(self.model)

    def launch(self):
        """Enqueue the pass on the current stream WITHOUT reading its result (``read()`` does): a loop that trains several
        models side by side (``fit_concurrent``) launches every model's pass before it waits for any."""
        if self.graph is not None:
            self.graph.replay()
            self._pending = (self.out, self.res)
        else:
            self._pending = self._run()

    def read(self):
        out, res = self._pending
        vals = res.tolist()                     # the one synchronising copy of the pass
        return out, vals[:-1], vals[-1]

    def __call__(self):
        self.launch()
        return self.read()


def fit(model, optimizer, x, adj, labels, train_idx, val_idx, test_idx, epochs, rule="max_val_acc",
        early_stopping=0, adj_high=None, adj_un=None, use_graph=False, fused_dropout=None, pipeline_input=None):
    """Train and return (selected test accuracy, per-epoch history).

    ``pipeline_input``: TrainStep's input pipeline (None = where the configuration qualifies, False = never).  On by
    default since round 3: with the sixteen-rows-per-wave backward kernel carrying the gather, an epoch (training step +
    evaluation pass, both captured) of the twitch-shaped graph takes 0.463 ms with it against 0.483 ms without
    (round 2, with the older kernel: 0.551 against 0.544, so it was off here).

    rule = "max_val_acc":  test accuracy at the best validation accuracy, fixed number of epochs
                           (ACM-Geometric/train.py:139-140, logger.py:17-48)
    rule = "min_val_loss": test accuracy at the lowest validation loss, stop when the validation loss
                           exceeds the mean of the last `early_stopping` epochs
                           (ACM-Pytorch/train.py:129-139)
    """
    w = row_weights(train_idx, x.shape[0], device=x.device)
    step = TrainStep(model, optimizer, x, adj, labels, w, adj_high, adj_un, use_graph=use_graph,
                     fused_dropout=fused_dropout, pipeline_input=pipeline_input)
    best_key, selected, history = None, 0.0, []
    val_hist = []
    # the evaluation pass of every epoch: its own captured graph when the training step is one
    ev = EvalStep(model, x, adj, labels, (train_idx, val_idx, test_idx), adj_high, adj_un, loss_set=1,
                  use_graph=use_graph) if use_graph else None
    for epoch in range(epochs):
        loss = step()
        if ev is not None:
            out, (acc_tr, acc_va, acc_te), val_loss = ev()
        else:
            out, (acc_tr, acc_va, acc_te) = evaluate(model, x, adj, labels, (train_idx, val_idx, test_idx),
                                                     adj_high, adj_un)
            val_loss = float(F.nll_loss(F.log_softmax(out, 1)[val_idx], labels[val_idx]))
        history.append((float(loss), acc_tr, acc_va, acc_te, val_loss))
        key = acc_va if rule == "max_val_acc" else -val_loss
        if best_key is None or key > best_key:
            best_key, selected = key, acc_te
        if rule == "min_val_loss":
            val_hist.append(val_loss)
            if early_stopping > 0 and epoch > early_stopping:
                if val_loss > sum(val_hist[epoch - early_stopping:epoch]) / early_stopping:
                    break
    return selected, history


def fit_concurrent(runs, x, adj, labels, epochs, rule="min_val_loss", early_stopping=0, adj_high=None, adj_un=None, streams=2,
                   fused_dropout=None):
    """Several independent training runs on the same graph -- the reference's ten fixed splits, which ACM-Pytorch/train.py:49-139
    trains one after the other, each with a fresh model -- ``streams`` at a time, every run on its own stream with its own
    captured step and evaluation pass: the launches of one run fill the gaps of the other (a small-graph step is a chain of
    six short, latency-bound launches that leaves most of the chip idle).  Measured on the MI355X (scripts/probe_concurrent_splits.py):
    two runs side by side take 0.050 ms per step each on Cora (0.083 alone), 0.117 on Squirrel (0.159); more than two gain nothing.

    ``runs``: a list of ``(model, optimizer, train_idx, val_idx, test_idx)``; every run is exactly ``fit(..., use_graph=True)``
    of its model (same selection rules, same history rows, bit-identical results: the runs share no state but the read-only
    graph and features).  Returns ``[(selected test accuracy, history)]`` in the order of ``runs``.  Without a GPU stream to
    overlap on (CPU test double) the runs are trained one after the other."""
    results = [None] * len(runs)
    on_gpu = labels.is_cuda and torch.cuda.is_available()
    if not on_gpu or streams <= 1:
        for k, (model, opt, tr, va, te) in enumerate(runs):
            results[k] = fit(model, opt, x, adj, labels, tr, va, te, epochs, rule=rule, early_stopping=early_stopping,
                             adj_high=adj_high, adj_un=adj_un, use_graph=on_gpu, fused_dropout=fused_dropout)
        return results
    pending = list(range(len(runs)))
    slots = [None] * min(int(streams), len(runs))
    pool = [torch.cuda.Stream() for _ in slots]

    def start(k, stream):
        model, opt, tr, va, te = runs[k]
        with torch.cuda.stream(stream):
            w = row_weights(tr, x.shape[0], device=labels.device)
            step = TrainStep(model, opt, x, adj, labels, w, adj_high, adj_un, use_graph=True, fused_dropout=fused_dropout)
            ev = EvalStep(model, x, adj, labels, (tr, va, te), adj_high, adj_un, loss_set=1, use_graph=True)
        stream.synchronize()
        return dict(k=k, step=step, ev=ev, epoch=0, best=None, selected=0.0, history=[], vals=[])

    while pending or any(sl is not None for sl in slots):
        for i in range(len(slots)):
            if slots[i] is None and pending:
                slots[i] = start(pending.pop(0), pool[i])
        live = [(i, sl) for i, sl in enumerate(slots) if sl is not None]
        for i, sl in live:                               # every live run's epoch is enqueued before any result is awaited
            with torch.cuda.stream(pool[i]):
                sl["loss"] = sl["step"]()
                sl["ev"].launch()
        for i, sl in live:
            with torch.cuda.stream(pool[i]):
                _, (acc_tr, acc_va, acc_te), val_loss = sl["ev"].read()
                loss = float(sl["loss"])
            sl["history"].append((loss, acc_tr, acc_va, acc_te, val_loss))
            key = acc_va if rule == "max_val_acc" else -val_loss
            if sl["best"] is None or key > sl["best"]:
                sl["best"], sl["selected"] = key, acc_te
            epoch = sl["epoch"]
            done = epoch + 1 >= epochs
            if rule == "min_val_loss":
                sl["vals"].append(val_loss)
                if early_stopping > 0 and epoch > early_stopping and \
                        val_loss > sum(sl["vals"][epoch - early_stopping:epoch]) / early_stopping:
                    done = True
            sl["epoch"] = epoch + 1
            if done:
                results[sl["k"]] = (sl["selected"], sl["history"])
                slots[i] = None
    return results
