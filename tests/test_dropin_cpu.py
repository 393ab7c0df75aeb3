"""Drop-in module installation (import mechanics only; compute is GPU-only)."""
import os
import sys
import types

import pytest


@pytest.fixture
def clean_modules():
    saved = {k: sys.modules.get(k) for k in ("layers", "models", "models.layers", "models.models")}
    yield
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    import acm_gnn_amd.layers as impl
    impl.DEFAULT_ATTN_LAYERNORM = True


def test_geometric_dialect_shadows_top_level_layers(clean_modules):
    from acm_gnn_amd import dropin, layers as impl
    shim = dropin.install("geometric")
    import layers                                   # what ACM-Geometric/models.py:3 does
    assert layers is shim and layers.GraphConvolution is impl.GraphConvolution and layers.MLP is impl.MLP
    assert impl.GraphConvolution(4, 8, 10, "acmgcnp").attn_layernorm is True


def test_pytorch_dialect_replaces_models_layers_only(clean_modules, tmp_path, monkeypatch):
    from acm_gnn_amd import dropin, layers as impl
    pkg = tmp_path / "models"                       # stand-in for the reference's models/ package
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "models.py").write_text("from models.layers import GraphConvolution, MLP\nMARK = 1\n")
    (pkg / "layers.py").write_text("raise ImportError('the reference layer must not be imported')\n")
    monkeypatch.chdir(tmp_path)
    monkeypatch.syspath_prepend(str(tmp_path))
    sys.modules.pop("models", None)
    dropin.install("pytorch")
    from models.models import GraphConvolution, MARK      # what ACM-Pytorch/train.py:13 does
    assert MARK == 1 and GraphConvolution is impl.GraphConvolution
    layer = impl.GraphConvolution(4, 8, 10, "acmgcnp")
    assert layer.attn_layernorm is False                  # quirk Q1: LN never fires in ACM-Pytorch
    assert impl.GraphConvolution(4, 8, 10, "acmgcn+").attn_layernorm is True


def test_unknown_dialect():
    from acm_gnn_amd import dropin
    with pytest.raises(ValueError):
        dropin.install("jax")


def test_the_launcher_binds_adam_and_adamw_where_the_fused_optimizers_apply(clean_modules, tmp_path, monkeypatch):
    """``python -m acm_gnn_amd.dropin geometric train.py``: the script's ``torch.optim.AdamW(...)`` / ``torch.optim.Adam(...)``
    (ACM-Geometric/train.py:112-117) construct this package's one-launch optimizers -- for fp32 parameters on a GPU, without
    amsgrad / maximize / a tensor lr; anything else (here: CPU parameters, as in a CPU run of ACM-Pytorch/train.py) gets torch's
    own class and ONE warning instead of a crash at the first step.  ``--torch-optimizer`` leaves torch.optim
    alone."""
    import warnings
    import torch
    from acm_gnn_amd import dropin, optim
    script = tmp_path / "train.py"
    script.write_text("import sys, torch\nfrom layers import GraphConvolution\n"
                      "p = [torch.nn.Parameter(torch.zeros(3))]\n"
                      "RESULT = (type(torch.optim.AdamW(p, lr=0.01, weight_decay=1e-3)).__name__,\n"
                      "          type(torch.optim.Adam(p, 0.01)).__name__, type(torch.optim.Adam(iter(p), lr=0.01)).__name__, sys.argv[1:])\n")
    before = (torch.optim.Adam, torch.optim.AdamW)
    seen = {}
    import runpy
    real = runpy.run_path
    monkeypatch.setattr(runpy, "run_path", lambda path, run_name=None: seen.update(real(path, run_name="not_main")))
    try:
        dropin._WARNED.clear()
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            dropin.main(["geometric", str(script), "--dataset", "x"])
        # CPU parameters: torch's own classes, one warning per class
        assert seen["RESULT"] == ("AdamW", "Adam", "Adam", ["--dataset", "x"])
        mine = [w for w in caught if "acm_gnn_amd.dropin" in str(w.message)]
        assert len(mine) == 2 and "CPU run" in str(mine[0].message)
        assert torch.optim.AdamW.fused is optim.FusedAdamW and torch.optim.AdamW.stock is before[1]
        # where they apply (the device check is what a GPU box answers): the fused classes, same arguments
        monkeypatch.setattr(dropin, "_why_not_fused", lambda params, args, kw: "amsgrad / maximize" if kw.get("amsgrad") else None)
        p = [torch.nn.Parameter(torch.zeros(3))]
        o = torch.optim.AdamW(p, 0.02, weight_decay=1e-3)
        assert type(o) is optim.FusedAdamW and o.param_groups[0]["lr"] == 0.02 and o.param_groups[0]["weight_decay"] == 1e-3
        assert type(torch.optim.Adam(iter(p), lr=0.01)) is optim.FusedAdam
        assert type(torch.optim.Adam(p, lr=0.01, amsgrad=True)) is before[0]
        # installing twice keeps the stock classes underneath
        dropin.install_fused_optimizers()
        assert torch.optim.Adam.stock is before[0]
    finally:
        torch.optim.Adam, torch.optim.AdamW = before
    for flags in (["--torch-optimizer"],):
        dropin.main(["geometric"] + flags + [str(script)])
        assert seen["RESULT"][:2] == ("AdamW", "Adam") and torch.optim.Adam is before[0]
    try:
        dropin.main(["geometric", "--fused-optimizer", str(script)])           # older command lines
        assert torch.optim.Adam is not before[0]
    finally:
        torch.optim.Adam, torch.optim.AdamW = before


def test_why_not_fused_reasons():
    import torch
    from acm_gnn_amd import dropin
    p = [torch.nn.Parameter(torch.zeros(3))]
    assert "CPU run" in dropin._why_not_fused(p, (), {})
    assert dropin._why_not_fused(p, (0.01, (0.9, 0.99), 1e-8, 0.0, True), {}) == "amsgrad / maximize"
    assert dropin._why_not_fused(p, (), {"lr": torch.tensor(0.1)}) == "a tensor lr"
    assert dropin._why_not_fused([], (), {}) == "no parameters"
    assert "CPU run" in dropin._why_not_fused([{"params": p, "lr": 0.1}], (), {})
