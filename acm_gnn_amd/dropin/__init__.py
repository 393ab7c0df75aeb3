"""Drop-in installation for the reference's training scripts.

Both reference scripts import the hot-path classes by module name:

    ACM-Pytorch/models/models.py:7    from models.layers import GraphConvolution, MLP
    ACM-Geometric/models.py:3         from layers import GraphConvolution, MLP

``install(dialect)`` registers this package's classes under those module names *before* the
script imports them, so ``train.py`` runs unmodified with the MI355X kernels:

    cd /path/to/ACM-GNN/ACM-Geometric
    python -m acm_gnn_amd.dropin geometric train.py --dataset twitch-gamer --method acmgcnp ...
    cd /path/to/ACM-GNN/ACM-Pytorch
    python -m acm_gnn_amd.dropin pytorch train.py --model acmgcnp --dataset_name squirrel ...

The launcher also binds ``torch.optim.Adam`` / ``torch.optim.AdamW`` to this package's FusedAdam / FusedAdamW for the run
(``train.py:112-117`` constructs them by those names): same arguments, update formulas and ``state_dict`` layout, ONE launch
per step instead of torch's ~80 -- the other half of an eager step's launches.  Where they do not apply (a CPU run of the
script, amsgrad, a tensor lr) the construction falls back to torch's own class with a warning; ``--torch-optimizer`` (before
the script's name) switches the binding off altogether (``--fused-optimizer`` is accepted for older command lines).

For ACM-Pytorch the launcher also binds ``utils.train_model`` (the script's training step, utils.py:547-574) to the fused
small-graph step where it applies (``install_fused_train_step``): Cora / Chameleon / Squirrel-class runs train in six launches
per step instead of the ~250 of the eager loop.  ``--reference-step`` keeps the reference's own function.

For ACM-Geometric it binds ``data_utils.evaluate_acmgcn`` (the per-epoch evaluation, data_utils.py:153-168) to the same forward
followed by one launch for the three accuracies instead of six device-to-host copies + numpy (``install_fast_evaluate``; exact
counts, identical numbers).  ``--reference-eval`` keeps the reference's function.

The dialect also selects the attention-LayerNorm behaviour: on for
ACM-Geometric, off for ACM-Pytorch (whose layer only normalises for the never-used spellings
"acmgcn+"/"acmgcn++").
"""
import importlib
import os
import runpy
import sys
import types

DIALECTS = {"geometric": ("layers", True), "pytorch": ("models.layers", False)}


def install(dialect):
    if dialect not in DIALECTS:
        raise ValueError(f"dialect must be one of {sorted(DIALECTS)}")
    modname, attn_ln = DIALECTS[dialect]
    from .. import layers as impl
    impl.DEFAULT_ATTN_LAYERNORM = attn_ln
    shim = types.ModuleType(modname)
    shim.__doc__ = f"acm_gnn_amd drop-in for the reference module {modname!r}"
    shim.GraphConvolution, shim.MLP = impl.GraphConvolution, impl.MLP
    shim.device = impl._default_device()
    if dialect == "pytorch":
        # `models` stays the reference's own package (models/models.py must still be found);
        # only its `layers` submodule is replaced.
        sys.path.insert(0, os.getcwd())
        pkg = importlib.import_module("models")
        pkg.layers = shim
    sys.modules[modname] = shim
    return shim


_ON_DEVICE = lambda t: t.is_cuda          # noqa: E731  (the CPU test double of the library lifts this guard)
_ADAM_POSITIONAL = ("lr", "betas", "eps", "weight_decay", "amsgrad")
_WARNED = set()


def _why_not_fused(params, args, kw):
    """None when FusedAdam / FusedAdamW can stand in for this torch.optim.Adam(W) construction, else the reason."""
    import torch
    opts = dict(zip(_ADAM_POSITIONAL, args), **kw)
    if opts.get("amsgrad") or opts.get("maximize"):
        return "amsgrad / maximize"
    if isinstance(opts.get("lr"), torch.Tensor):
        return "a tensor lr"
    tensors = []
    for item in params:
        tensors.extend(item["params"] if isinstance(item, dict) else [item])
    if not tensors:
        return "no parameters"
    if any((not _ON_DEVICE(t)) or t.dtype != torch.float32 for t in tensors):
        return "parameters that are not fp32 tensors on a GPU (a CPU run of the script)"
    return None


def _choosing(fused_cls, torch_cls):
    """A stand-in for ``torch_cls`` that constructs ``fused_cls`` where it applies and the stock optimizer otherwise (a CPU run
    of the reference script -- ACM-Pytorch without --cuda --, amsgrad, a tensor lr): the drop-in degrades, it does not crash in
    the middle of a run.  Says so once per reason."""
    class _Choose:
        def __new__(cls, params, *args, **kw):
            params = list(params)
            why = _why_not_fused(params, args, kw)
            if why is None:
                return fused_cls(params, *args, **kw)
            if (torch_cls.__name__, why) not in _WARNED:
                _WARNED.add((torch_cls.__name__, why))
                import warnings
                warnings.warn(f"acm_gnn_amd.dropin: torch.optim.{torch_cls.__name__} stays torch's own ({why})", stacklevel=2)
            return torch_cls(params, *args, **kw)
    _Choose.__name__ = _Choose.__qualname__ = torch_cls.__name__
    _Choose.fused, _Choose.stock = fused_cls, torch_cls
    return _Choose


def install_fused_optimizers():
    """Bind torch.optim.Adam / AdamW to FusedAdam / FusedAdamW (the reference constructs its optimizer by those names,
    ACM-Geometric/train.py:112-117, ACM-Pytorch/train.py:70-84) -- where they apply: fp32 parameters on a GPU, no amsgrad /
    maximize / tensor lr; anything else gets torch's own class (and a warning).  Returns the (Adam, AdamW) that were bound
    before."""
    import torch
    from ..optim import FusedAdam, FusedAdamW
    before = (torch.optim.Adam, torch.optim.AdamW)
    stock = tuple(getattr(c, "stock", c) for c in before)          # (installing twice does not wrap the wrapper)
    torch.optim.Adam, torch.optim.AdamW = _choosing(FusedAdam, stock[0]), _choosing(FusedAdamW, stock[1])
    return before


def _fused_step_refusal(model, optimizer, features, labels, criterion, dataset_name):
    """None when utils.train_model's step can run as train.TrainStep's fused small-graph step, else the reason (a string)."""
    import torch
    from ..optim import _FusedAdamBase
    if dataset_name == "deezer-europe":
        return "deezer-europe (2-D labels, AdamW)"
    if type(criterion) is not torch.nn.NLLLoss or criterion.weight is not None or criterion.reduction != "mean" \
            or criterion.ignore_index != -100:
        return "criterion is not a plain nn.NLLLoss()"
    if not isinstance(optimizer, _FusedAdamBase):
        return "optimizer is not this package's FusedAdam (the launcher's optimizer binding is off, or did not apply)"
    if not isinstance(features, torch.Tensor) or features.layout != torch.strided or not _ON_DEVICE(features) or labels.dim() != 1:
        return "features / labels: a dense feature matrix on the GPU and 1-D labels"
    if not hasattr(model, "gcns") or not hasattr(model, "dropout") or not hasattr(model, "model_type"):
        return "not the reference's GCN wrapper"
    return None


def install_fused_train_step():
    """Bind ``utils.train_model`` -- the training step of ACM-Pytorch/train.py (utils.py:547-574: model.train(), zero_grad,
    forward, log_softmax + NLLLoss on the training rows, accuracy, backward, optimizer.step()) -- to this package's fused
    small-graph step (train.TrainStep -> small.SmallPlan: the whole step as six launches behind one C-ABI call) WHERE IT
    APPLIES: a two-layer acmgcn / acmgcnp model of hidden width 64 on a graph of <= 16 384 nodes with bag-of-words features,
    ``nn.NLLLoss()``, the launcher's FusedAdam.  Same signature, same return value (100 * training accuracy, training
    loss -- both of the training-mode forward, as in the reference), parameters and optimizer state updated as
    ``optimizer.step()`` leaves them.  What differs, on purpose: the dropout masks are the library's counter-based ones
    (seeded from torch's seed) instead of torch's generator stream -- the accuracy parity of exactly this path against the
    reference is pinned by tests/test_gpu_accuracy.py (reference runs recorded with these masks injected).  Everything else
    -- other models, hidden widths, CPU runs, hooks, a scheduler-free-form optimizer -- runs the reference's own function,
    untouched.  Returns the function that was bound before (None when there is no ``utils.train_model`` to bind)."""
    try:
        utils = importlib.import_module("utils")
    except ImportError:
        return None
    ref = getattr(utils, "train_model", None)
    if ref is None or getattr(ref, "_acm_fused", False):
        return None
    import weakref

    import torch
    steps = {}                                   # id(model) -> (weakref(model), id(optimizer), TrainStep | None, key of the inputs)

    def train_model(model, optimizer, adj_low, adj_high, adj_low_unnormalized, features, labels, idx_train, criterion,
                    dataset_name):
        from .. import train as T
        from ..graph import SparseFeatures
        inputs = (id(adj_low), id(features), id(labels), id(idx_train), int(idx_train.numel()))
        entry = steps.get(id(model))
        if entry is None or entry[0]() is not model or entry[1] != id(optimizer) or entry[3] != inputs:
            step = None
            why = _fused_step_refusal(model, optimizer, features, labels, criterion, dataset_name)
            if why is None:
                had = {k: getattr(model, k) for k in ("fused_dropout", "dropout_state") if hasattr(model, k)}
                try:
                    # (the reference's GCN wrapper has no dropout switches of its own: the step only needs the attributes)
                    model.fused_dropout, model.dropout_state = getattr(model, "fused_dropout", False), getattr(model, "dropout_state", None)
                    n = labels.shape[0]
                    rows = idx_train.nonzero().view(-1) if idx_train.dtype == torch.bool else idx_train.long()    # (the fixed
                    cand = T.TrainStep(model, optimizer, SparseFeatures.auto(features), adj_low, labels,         # splits are masks)
                                       T.row_weights(rows, n), adj_high, adj_low_unnormalized, fused_dropout=True)
                    if cand.small is not None:
                        step = cand
                    else:
                        why = cand.small_refused
                except Exception as exc:          # noqa: BLE001 -- a binding must never break the script it serves
                    why = f"{type(exc).__name__}: {exc}"
                if step is None:                  # leave the model and the optimizer as the reference's own step expects them
                    if hasattr(optimizer, "also_advance"):
                        optimizer.also_advance = None
                    for k in ("fused_dropout", "dropout_state"):
                        if k in had:
                            setattr(model, k, had[k])
                        elif hasattr(model, k):
                            try:
                                delattr(model, k)
                            except AttributeError:
                                pass
            if step is None and why not in _WARNED:
                _WARNED.add(why)
                import warnings
                warnings.warn(f"acm_gnn_amd.dropin: utils.train_model stays the reference's own ({why})", stacklevel=2)
            # (a step keeps its model alive, and with it the plan's workspace: the script trains one split at a time, so the
            # entries of earlier splits are dropped -- at most two stay)
            for k in list(steps)[:-1]:
                del steps[k]
            entry = steps[id(model)] = (weakref.ref(model), id(optimizer), step, inputs)
        step = entry[2]
        if step is None:
            return ref(model, optimizer, adj_low, adj_high, adj_low_unnormalized, features, labels, idx_train, criterion,
                       dataset_name)
        model.train()
        loss = step()
        logits = step.small.logits                # the training-mode logits of the step (log_softmax keeps the arg-max)
        acc = (logits[idx_train].argmax(1) == labels[idx_train]).double().mean()
        acc_v, loss_v = torch.stack([100 * acc, loss.double()]).tolist()        # ONE synchronising copy (the reference: two .item())
        return acc_v, loss_v

    train_model._acm_fused, train_model.reference = True, ref
    utils.train_model = train_model
    return ref


def install_fast_evaluate():
    """Bind ``data_utils.evaluate_acmgcn`` -- the per-epoch evaluation of ACM-Geometric/train.py:137-138 (data_utils.py:153-168:
    eval-mode forward, then ``eval_func`` on the train / valid / test rows; ``eval_acc`` (data_utils.py:114-124) pulls labels and
    predictions to the host three times and counts in numpy) -- to the same forward followed by ONE launch over the logits
    (``acm_eval_metrics``, weights 1 on a split's rows: exact counts) and ONE four-byte-per-number copy, WHERE IT APPLIES:
    ``eval_func`` is the module's own ``eval_acc``, integer labels of shape [n] / [n, 1] on the GPU, fp32 logits of <= 64 classes.
    Same return value ``(train_acc, valid_acc, test_acc, out)`` -- the accuracies are count / len in float64 exactly as the
    reference forms them.  Anything else (``eval_rocauc``, multi-label targets, a precomputed ``result``) runs the reference's
    own function.  Returns the function that was bound before (None when there is no ``data_utils.evaluate_acmgcn``)."""
    try:
        du = importlib.import_module("data_utils")
    except ImportError:
        return None
    ref, ref_acc = getattr(du, "evaluate_acmgcn", None), getattr(du, "eval_acc", None)
    if ref is None or ref_acc is None or getattr(ref, "_acm_fused", False):
        return None
    import torch
    cache = []                                      # [(split_idx, label, labels_flat, weights, sizes, buffers)], newest last

    def evaluate_acmgcn(model, x, adj_low, adj_high, adj_low_unnormalized, dataset, split_idx, eval_func, result=None):
        label = getattr(dataset, "label", None)
        ok = (result is None and eval_func is ref_acc and isinstance(label, torch.Tensor) and _ON_DEVICE(label)
              and label.dtype == torch.int64 and (label.dim() == 1 or (label.dim() == 2 and label.shape[1] == 1))
              and isinstance(split_idx, dict) and all(isinstance(split_idx.get(k), torch.Tensor) for k in ("train", "valid", "test")))
        if not ok:
            return ref(model, x, adj_low, adj_high, adj_low_unnormalized, dataset, split_idx, eval_func, result)
        try:
            return fast(model, x, adj_low, adj_high, adj_low_unnormalized, dataset, split_idx, eval_func, label)
        except Exception as exc:                 # noqa: BLE001 -- a binding must never break the script it serves
            why = f"{type(exc).__name__}: {exc}"
            if why not in _WARNED:
                _WARNED.add(why)
                import warnings
                warnings.warn(f"acm_gnn_amd.dropin: data_utils.evaluate_acmgcn stays the reference's own ({why})", stacklevel=2)
            return ref(model, x, adj_low, adj_high, adj_low_unnormalized, dataset, split_idx, eval_func, result)

    def fast(model, x, adj_low, adj_high, adj_low_unnormalized, dataset, split_idx, eval_func, label):
        with torch.no_grad():
            model.eval()
            out = model(x, adj_low, adj_high, adj_low_unnormalized)
            if not (out.dim() == 2 and out.shape[1] <= 64 and out.dtype == torch.float32 and _ON_DEVICE(out) and out.stride(1) == 1):
                return ref(model, x, adj_low, adj_high, adj_low_unnormalized, dataset, split_idx, eval_func, out)
            from .. import functional as AF
            entry = next((e for e in cache if e[0] is split_idx and e[1] is label), None)
            if entry is None:
                n = label.shape[0]
                w = torch.zeros(3, n, dtype=torch.float32, device=label.device)
                sizes = []
                for q, k in enumerate(("train", "valid", "test")):
                    idx = split_idx[k].to(label.device)
                    idx = idx.nonzero().view(-1) if idx.dtype == torch.bool else idx.long()
                    w[q].index_fill_(0, idx, 1.0)                    # (weight 1: the sums are exact counts below 2^24)
                    sizes.append(int(idx.numel()))
                entry = (split_idx, label, label.reshape(-1).contiguous(), w, sizes, AF.eval_metrics_buffers(n, 3, label.device))
                cache.append(entry)
                del cache[:-4]
            _, _, flat, w, sizes, bufs = entry
            counts = AF.eval_metrics(out, flat, w, 1, bufs).tolist()          # the one synchronising copy of the pass
        accs = [float(counts[q]) / sizes[q] if sizes[q] else float("nan") for q in range(3)]
        return accs[0], accs[1], accs[2], out

    evaluate_acmgcn._acm_fused, evaluate_acmgcn.reference = True, ref
    du.evaluate_acmgcn = evaluate_acmgcn
    return ref


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    fused = True                                  # on by default since round 6 (it degrades to torch's own where it does not apply)
    fused_step = True                             # ... and so is the fused small-graph step behind utils.train_model (ACM-Pytorch)
    fast_eval = True                              # ... and the one-launch accuracies behind data_utils.evaluate_acmgcn (ACM-Geometric)
    for flag in ("--fused-optimizer", "--torch-optimizer", "--reference-step", "--reference-eval"):
        if flag in argv[:5] and argv.index(flag) < 5:
            argv.remove(flag)
            if flag == "--reference-step":
                fused_step = False
            elif flag == "--reference-eval":
                fast_eval = False
            else:
                fused = flag == "--fused-optimizer"
    if len(argv) < 2:
        sys.exit("usage: python -m acm_gnn_amd.dropin {geometric|pytorch} [--torch-optimizer] [--reference-step] [--reference-eval] "
                 "train.py [script args...]")
    dialect, script = argv[0], argv[1]
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)) or os.getcwd())
    install(dialect)
    if fused:
        install_fused_optimizers()
    if fused and fused_step and dialect == "pytorch":
        install_fused_train_step()
    if fast_eval and dialect == "geometric":
        install_fast_evaluate()
    sys.argv = [script] + argv[2:]
    runpy.run_path(script, run_name="__main__")
