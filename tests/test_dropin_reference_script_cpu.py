"""The reference's own ACM-Pytorch/train.py, UNMODIFIED, running on this package's
GraphConvolution through the drop-in shim (a11-a16 wiring check).  Only possible where the
reference checkout exists (the build container); the kernels are replaced by the numpy test
double because there is no GPU here -- what is exercised is the drop-in boundary itself:
import paths, constructor / forward signatures, parameter registration with the reference's
optimizer, train()/eval() switching, the attributes the script reads."""
import os
import runpy
import sys
import types

import pytest

import fake_lib

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "ACM-Pytorch")),
                                reason="reference checkout not present (GPU box)")


def test_reference_acm_pytorch_train_script_runs_on_the_dropin(tmp_path, monkeypatch, capsys):
    # scratch tree of symlinks: the script uses ../data, splits/ and writes ./logs
    work = tmp_path / "ACM-Pytorch"
    work.mkdir()
    for name in ("train.py", "arg_parser.py", "logger.py", "utils.py", "models", "splits"):
        os.symlink(os.path.join(REF, "ACM-Pytorch", name), work / name)
    os.symlink(os.path.join(REF, "data"), tmp_path / "data")
    os.symlink(os.path.join(REF, "BaseLogger.py"), tmp_path / "BaseLogger.py")
    monkeypatch.chdir(work)
    ref_modules = ("models", "models.layers", "models.models", "utils", "logger", "arg_parser", "BaseLogger",
                   "google_drive_downloader")
    saved = {k: sys.modules.get(k) for k in ref_modules}
    saved_path = list(sys.path)
    try:
        sys.modules["google_drive_downloader"] = types.SimpleNamespace(GoogleDriveDownloader=object)
        for m in ref_modules[:-1]:
            sys.modules.pop(m, None)
        fake_lib.install(monkeypatch)
        from acm_gnn_amd import dropin, layers as impl
        calls = {"fwd": 0}
        orig_forward = impl.GraphConvolution.forward

        def counting_forward(self, *a, **k):
            calls["fwd"] += 1
            return orig_forward(self, *a, **k)

        monkeypatch.setattr(impl.GraphConvolution, "forward", counting_forward)
        sys.path.insert(0, str(work))
        dropin.install("pytorch")
        monkeypatch.setattr(sys, "argv", ["train.py", "--model", "acmgcn", "--dataset_name", "cora",
                                          "--fixed_splits", "1", "--num_splits", "1", "--epochs", "4",
                                          "--lr", "0.01", "--weight_decay", "5e-5", "--dropout", "0.6",
                                          "--hidden", "16", "--no-cuda"])
        ns = runpy.run_path(str(work / "train.py"), run_name="__main__")
    finally:
        for k, v in saved.items():          # drop only the reference's top-level modules again
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        sys.path[:] = saved_path
        import acm_gnn_amd.layers as impl2
        impl2.DEFAULT_ATTN_LAYERNORM = True
    # 4 epochs x (train forward + eval forward) x 2 layers, all through our layer
    assert calls["fwd"] == 4 * 2 * 2
    assert 0.0 <= float(ns["result"][0]) <= 1.0
    assert ns["GCN"].__module__ == "models.models"            # the reference's own model wrapper


def test_reference_script_trains_on_the_fused_small_graph_step(tmp_path, monkeypatch, capsys):
    """Round 6: ``python -m acm_gnn_amd.dropin pytorch train.py ...`` also binds ``utils.train_model`` (the script's training
    step, utils.py:547-574) to the fused small-graph step where it applies -- the reference's OWN models.GCN (hidden 64) on
    Cora: the launcher's FusedAdam, CSR twin of the dense features, six launches per step behind one C-ABI call
    (acm_small_step; here the test double), the reference's evaluation forward untouched.  The script itself is unmodified:
    same return values of train_model, same model-selection loop, same result structure."""
    import warnings
    work = tmp_path / "ACM-Pytorch"
    work.mkdir()
    for name in ("train.py", "arg_parser.py", "logger.py", "utils.py", "models", "splits"):
        os.symlink(os.path.join(REF, "ACM-Pytorch", name), work / name)
    os.symlink(os.path.join(REF, "data"), tmp_path / "data")
    os.symlink(os.path.join(REF, "BaseLogger.py"), tmp_path / "BaseLogger.py")
    monkeypatch.chdir(work)
    ref_modules = ("models", "models.layers", "models.models", "utils", "logger", "arg_parser", "BaseLogger",
                   "google_drive_downloader")
    saved = {k: sys.modules.get(k) for k in ref_modules}
    saved_path = list(sys.path)
    import torch
    before = (torch.optim.Adam, torch.optim.AdamW)
    try:
        sys.modules["google_drive_downloader"] = types.SimpleNamespace(GoogleDriveDownloader=object)
        for m in ref_modules[:-1]:
            sys.modules.pop(m, None)
        fake = fake_lib.install(monkeypatch)
        from acm_gnn_amd import dropin, layers as impl, optim
        monkeypatch.setattr(dropin, "_ON_DEVICE", lambda t: True)           # (no GPU here: the test double takes CPU tensors)
        calls = {"fwd": 0}
        orig_forward = impl.GraphConvolution.forward

        def counting_forward(self, *a, **k):
            calls["fwd"] += 1
            return orig_forward(self, *a, **k)

        monkeypatch.setattr(impl.GraphConvolution, "forward", counting_forward)
        sys.path.insert(0, str(work))
        dropin._WARNED.clear()
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            dropin.main(["pytorch", str(work / "train.py"), "--model", "acmgcn", "--dataset_name", "cora", "--fixed_splits", "1",
                         "--num_splits", "1", "--epochs", "4", "--lr", "0.01", "--weight_decay", "5e-5", "--dropout", "0.6",
                         "--hidden", "64", "--no-cuda"])
        utils = sys.modules["utils"]
        assert getattr(utils.train_model, "_acm_fused", False) and utils.train_model.reference.__module__ == "utils"
        assert not [w for w in caught if "train_model stays" in str(w.message)], [str(w.message) for w in caught]
        # 4 training steps = 4 calls of the fused step; the layers' forward ran for the 4 evaluation passes only
        assert getattr(fake, "small_calls", 0) == 4
        assert calls["fwd"] == 4 * 2
        # hidden 16: outside the envelope -- the reference's own train_model, and a warning that says why
        calls["fwd"], fake.small_calls = 0, 0
        for m in ("utils",):
            sys.modules.pop(m, None)
        dropin._WARNED.clear()
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            dropin.main(["pytorch", str(work / "train.py"), "--model", "acmgcn", "--dataset_name", "cora", "--fixed_splits", "1",
                         "--num_splits", "1", "--epochs", "2", "--lr", "0.01", "--weight_decay", "5e-5", "--dropout", "0.6",
                         "--hidden", "16", "--no-cuda"])
        assert fake.small_calls == 0 and calls["fwd"] == 2 * 2 * 2
        assert any("train_model stays" in str(w.message) and "hidden width" in str(w.message) for w in caught)
    finally:
        torch.optim.Adam, torch.optim.AdamW = before
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        sys.path[:] = saved_path
        import acm_gnn_amd.layers as impl2
        impl2.DEFAULT_ATTN_LAYERNORM = True
