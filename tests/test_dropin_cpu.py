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


def test_fused_optimizer_flag_binds_adam_and_adamw_for_the_script(clean_modules, tmp_path, monkeypatch):
    """``python -m acm_gnn_amd.dropin geometric --fused-optimizer train.py``: the script's ``torch.optim.AdamW(...)`` /
    ``torch.optim.Adam(...)`` (ACM-Geometric/train.py:112-117) are this package's one-launch optimizers; without the flag
    they stay torch's."""
    import torch
    from acm_gnn_amd import dropin, optim
    script = tmp_path / "train.py"
    script.write_text("import sys, torch\nfrom layers import GraphConvolution\n"
                      "p = [torch.nn.Parameter(torch.zeros(3))]\n"
                      "RESULT = (type(torch.optim.AdamW(p, lr=0.01, weight_decay=1e-3)).__name__,\n"
                      "          type(torch.optim.Adam(p, lr=0.01)).__name__, sys.argv[1:])\n")
    before = (torch.optim.Adam, torch.optim.AdamW)
    seen = {}
    import runpy
    real = runpy.run_path
    monkeypatch.setattr(runpy, "run_path", lambda path, run_name=None: seen.update(real(path, run_name="not_main")))
    try:
        dropin.main(["geometric", "--fused-optimizer", str(script), "--dataset", "x"])
        assert seen["RESULT"] == ("FusedAdamW", "FusedAdam", ["--dataset", "x"])
        assert torch.optim.AdamW is optim.FusedAdamW
    finally:
        torch.optim.Adam, torch.optim.AdamW = before
    dropin.main(["geometric", str(script)])
    assert seen["RESULT"][:2] == ("AdamW", "Adam")
