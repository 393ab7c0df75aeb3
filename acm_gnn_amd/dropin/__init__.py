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

``--fused-optimizer`` (before the script's name) additionally binds ``torch.optim.Adam`` / ``torch.optim.AdamW`` to this
package's FusedAdam / FusedAdamW for the run (``train.py:112-117`` constructs them by those names): same arguments, update
formulas and ``state_dict`` layout, ONE launch per step instead of torch's ~80 -- the other half of an eager step's launches.

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


def install_fused_optimizers():
    """Bind torch.optim.Adam / AdamW to FusedAdam / FusedAdamW (the reference constructs its optimizer by those names,
    ACM-Geometric/train.py:112-117, ACM-Pytorch/train.py:70-84).  Returns the (Adam, AdamW) classes that were bound before."""
    import torch
    from ..optim import FusedAdam, FusedAdamW
    before = (torch.optim.Adam, torch.optim.AdamW)
    torch.optim.Adam, torch.optim.AdamW = FusedAdam, FusedAdamW
    return before


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    fused = "--fused-optimizer" in argv[:2]
    if fused:
        argv.remove("--fused-optimizer")
    if len(argv) < 2:
        sys.exit("usage: python -m acm_gnn_amd.dropin {geometric|pytorch} [--fused-optimizer] train.py [script args...]")
    dialect, script = argv[0], argv[1]
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)) or os.getcwd())
    install(dialect)
    if fused:
        install_fused_optimizers()
    sys.argv = [script] + argv[2:]
    runpy.run_path(script, run_name="__main__")
