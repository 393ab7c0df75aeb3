"""pytest configuration: the ``gpu`` marker and shared fixture helpers."""
import contextlib
import glob
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(trylast=True)
def pytest_collection_modifyitems(config, items):
    """The accuracy replays run LAST: their worker processes are started when the session begins (the fixture below) and
    train underneath the other GPU tests (gate time; tests/test_gpu_accuracy.py: prefetch_replays)."""
    last = [it for it in items if os.path.basename(str(it.fspath)) == "test_gpu_accuracy.py"]
    if last and len(last) < len(items):
        keep = [it for it in items if os.path.basename(str(it.fspath)) != "test_gpu_accuracy.py"]
        items[:] = keep + last


@pytest.fixture(scope="session", autouse=True)
def _accuracy_replay_prefetch(request):
    names = [it.name for it in request.session.items if os.path.basename(str(it.fspath)) == "test_gpu_accuracy.py"]
    wanted = [nm for nm in names if nm.startswith(("test_fixed_split_accuracy", "test_bf16_gathered"))]
    if len(wanted) >= 4 and len(names) < len(request.session.items) and torch.cuda.is_available():
        import test_gpu_accuracy
        test_gpu_accuracy.prefetch_replays()
        # the oracle's CPU steps of the other tests now share the host with five worker processes: an OpenMP team as wide
        # as the machine waits for its descheduled members (a 17 s test took 70 s); a team of 48 leaves them room
        torch.set_num_threads(min(os.cpu_count() or 8, 48))
    yield


def golden_files(pattern):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def load_npz(path):
    with np.load(path, allow_pickle=False) as f:
        rec = {k: f[k] for k in f.files}
    if "cfg" in rec:
        rec["cfg"] = json.loads(str(rec["cfg"]))
    return rec


def csr_to_coo_tensor(g, prefix, n=None):
    """CSR arrays from a fixture -> coalesced float32 torch sparse COO."""
    indptr, indices = g[prefix + "_indptr"], g[prefix + "_indices"]
    vals = g.get(prefix + "_vals")
    if vals is None:
        vals = np.ones(len(indices), np.float32)
    n = n or len(indptr) - 1
    rows = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr))
    idx = torch.from_numpy(np.vstack([rows, indices]).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(vals.astype(np.float32)), (n, n)).coalesce()


def graph_tensors(dialect):
    """(adj_low, adj_high, adj_un) in the layouts the reference dialect feeds the layer."""
    g = load_npz(os.path.join(GOLDEN, f"graph_{dialect}.npz"))
    if dialect == "pytorch":
        adj_low = torch.from_numpy(g["adj_low_dense"])          # dense strided (utils.py:619-629)
    else:
        adj_low = csr_to_coo_tensor(g, "adj_low")
    return adj_low, csr_to_coo_tensor(g, "adj_high"), csr_to_coo_tensor(g, "adj_un"), g


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def tune_now(**kw):
    """``tune_now(agg_first=0, rows16=5, ...)``: set tuning switches (acm_gnn_amd.tuning: the host record and the library's
    acm_tuning_t) for the rest of the running test -- the autouse guard below restores both records afterwards.  Besides
    the real keys it takes four conveniences: ``agg_first`` / ``acmii_recompute`` / ``acmii_mask`` / ``aggw_fused`` = bits 1 / 2 / 4 / 8 of
    ``rewrites``."""
    from acm_gnn_amd import tuning
    bits = {"agg_first": tuning.REWRITE_AGG_FIRST, "acmii_recompute": tuning.REWRITE_ACMII_RECOMPUTE,
            "acmii_mask": tuning.REWRITE_ACMII_MASK, "aggw_fused": tuning.REWRITE_AGGW_FUSED}
    rew = tuning.HOST.rewrites
    for name, bit in bits.items():
        if name in kw:
            rew = (rew | bit) if int(kw.pop(name)) else (rew & ~bit)
            kw["rewrites"] = rew
    tuning.apply(**kw)


@pytest.fixture
def tune():
    """The same as a fixture (so that a test's signature says that it flips execution forms)."""
    return tune_now


@pytest.fixture(autouse=True)
def _tuning_guard():
    """Every test starts from, and leaves behind, the load-time tuning records."""
    from acm_gnn_amd import tuning
    tuning.reset(kernel_too=False)
    yield
    tuning.reset()
