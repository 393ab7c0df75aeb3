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
    if any((not t.is_cuda) or t.dtype != torch.float32 for t in tensors):
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


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    fused = True                                  # on by default since round 6 (it degrades to torch's own where it does not apply)
    for flag in ("--fused-optimizer", "--torch-optimizer"):
        if flag in argv[:3] and argv.index(flag) < 3:
            argv.remove(flag)
            fused = flag == "--fused-optimizer"
    if len(argv) < 2:
        sys.exit("usage: python -m acm_gnn_amd.dropin {geometric|pytorch} [--torch-optimizer] train.py [script args...]")
    dialect, script = argv[0], argv[1]
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)) or os.getcwd())
    install(dialect)
    if fused:
        install_fused_optimizers()
    sys.argv = [script] + argv[2:]
    runpy.run_path(script, run_name="__main__")
