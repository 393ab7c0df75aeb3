"""The oracle (oracle/acm_oracle.py) against every golden vector produced by the
imported reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import GOLDEN, golden_files, graph_tensors, load_npz, csr_to_coo_tensor
from oracle import acm_oracle as O

RT, AT = 1e-5, 1e-6


def _params(rec, prefix="param:"):
    return {k[len(prefix):]: torch.from_numpy(v.copy()).requires_grad_(True)
            for k, v in rec.items() if k.startswith(prefix)}


@pytest.mark.parametrize("path", golden_files("layer_*.npz"), ids=os.path.basename)
def test_layer_matches_reference(path):
    rec = load_npz(path)
    cfg = rec["cfg"]
    adj_low, adj_high, adj_un, _ = graph_tensors(cfg["dialect"])
    p = _params(rec)
    x = torch.from_numpy(rec["x"].copy()).requires_grad_(True)
    out, att = O.layer_forward(p, x, adj_low, adj_high, adj_un if cfg["structure_info"] else None,
                               model_type=cfg["model_type"], variant=cfg["variant"],
                               structure_info=cfg["structure_info"],
                               attn_layernorm=cfg["attn_layernorm"], return_att=True)
    out.backward(torch.from_numpy(rec["grad_out"]))
    np.testing.assert_allclose(out.detach().numpy(), rec["out"], rtol=RT, atol=AT)
    np.testing.assert_allclose(att.detach().numpy(), rec["att"], rtol=RT, atol=AT)
    np.testing.assert_allclose(x.grad.numpy(), rec["grad_x"], rtol=RT, atol=AT)
    n_checked = 0
    for k, v in rec.items():
        if k.startswith("grad:"):
            g = p[k[5:]].grad
            assert g is not None, k
            np.testing.assert_allclose(g.numpy(), v, rtol=RT, atol=AT, err_msg=k)
            n_checked += 1
    assert n_checked >= 7
    # parameters the reference left without a gradient must be unused here too
    for name, t in p.items():
        if "grad:" + name not in rec:
            assert t.grad is None or float(t.grad.abs().max()) == 0.0, name


@pytest.mark.parametrize("path", golden_files("model_*.npz"), ids=os.path.basename)
def test_model_matches_reference(path):
    rec = load_npz(path)
    cfg = rec["cfg"]
    adj_low, adj_high, adj_un, _ = graph_tensors(cfg["dialect"])
    p = _params(rec)
    masks = {k[5:]: torch.from_numpy(v.astype(np.float32)) for k, v in rec.items()
             if k.startswith("mask:")}
    logits = O.gcn_forward(p, torch.from_numpy(rec["x"]), adj_low, adj_high,
                           adj_un if cfg["structure_info"] else None,
                           model_type=cfg["model_type"], variant=cfg["variant"],
                           structure_info=cfg["structure_info"],
                           attn_layernorm=cfg["attn_layernorm"], dropout=cfg["dropout"],
                           training=True, masks=masks)
    loss = O.nll_loss_on(logits, torch.from_numpy(rec["labels"]), torch.from_numpy(rec["train_idx"]))
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), rec["logits"], rtol=RT, atol=AT)
    np.testing.assert_allclose(loss.item(), rec["loss"], rtol=RT)
    for k, v in rec.items():
        if k.startswith("grad:"):
            np.testing.assert_allclose(p[k[5:]].grad.numpy(), v, rtol=1e-4, atol=AT, err_msg=k)


def test_filters_linkx_dialect():
    """A_low = D^-1(I+A) (float64 -> float32), A_high = I - A_low, incl. raw
    self-loops (diagonal 2 before normalising) and an isolated node."""
    g = load_npz(os.path.join(GOLDEN, "graph_geometric.npz"))
    n = len(g["adj_un_indptr"]) - 1
    a = sp.csr_matrix((g["adj_un_vals"], g["adj_un_indices"], g["adj_un_indptr"]), shape=(n, n))
    low, high, un = O.filters_linkx(a)
    for t, pre in ((low, "adj_low"), (high, "adj_high"), (un, "adj_un")):
        ip, ix, v = O.coo_to_csr_arrays(t)
        np.testing.assert_array_equal(ip, g[pre + "_indptr"])
        np.testing.assert_array_equal(ix, g[pre + "_indices"])
        np.testing.assert_array_equal(v, g[pre + "_vals"])
    np.testing.assert_allclose(np.asarray(O.row_normalize_sp(sp.csr_matrix(g["feat_raw"])).todense()),
                               g["feat_rownorm"], rtol=1e-12)


def test_chained_khop_equals_the_materialised_power():
    """oracle.sgc_khop_forward (what the full-size config-5 tests compare against) == the acmsgc layer fed the dense
    A_low^3 of the reference's k-hop construction (ACM-Pytorch/utils.py:631-637, golden adj_low_pow3_dense)."""
    g = load_npz(os.path.join(GOLDEN, "graph_pytorch.npz"))
    low = torch.from_numpy(g["adj_low_dense"]).double()
    low3 = O.khop_low(low, 3)                      # the reference's construction, here in float64
    np.testing.assert_allclose(low3.numpy(), g["adj_low_pow3_dense"], rtol=1e-5, atol=1e-8)   # its fp32 golden
    n = low.shape[0]
    high = (torch.eye(n, dtype=torch.float64) - low).to_sparse()
    gen = torch.Generator().manual_seed(5)
    p = {k: v.double() for k, v in O.init_params(9, 4, n, 0, gen).items()}
    x = torch.randn(n, 9, generator=gen, dtype=torch.float64)
    ref = O.layer_forward(p, x, low3, high, model_type="acmsgc")
    got = O.sgc_khop_forward(p, x, low.to_sparse_csr(), high, 3)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-10, atol=1e-12)
    got_sx = O.sgc_khop_forward(p, x.to_sparse_csr(), low.to_sparse_csr(), high, 3)      # CSR features (Penn94)
    np.testing.assert_allclose(got_sx.numpy(), ref.numpy(), rtol=1e-10, atol=1e-12)


def test_filters_small_dialect_and_khop():
    g = load_npz(os.path.join(GOLDEN, "graph_pytorch.npz"))
    a_un = csr_to_coo_tensor(g, "adj_un").to_dense()
    low, high = O.filters_small(a_un)
    np.testing.assert_allclose(low.numpy(), g["adj_low_dense"], rtol=0, atol=0)
    ip, ix, v = O.coo_to_csr_arrays(high)
    np.testing.assert_array_equal(ix, g["adj_high_indices"])
    np.testing.assert_array_equal(v, g["adj_high_vals"])
    np.testing.assert_allclose(O.khop_low(low, 3).numpy(), g["adj_low_pow3_dense"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(O.row_normalize_dense(torch.from_numpy(g["feat_raw"])).numpy(),
                               g["feat_rownorm"], rtol=1e-6)


def _run_traj(rec, adj_low, adj_high, adj_un, x, labels, train_sel):
    cfg = rec["cfg"]
    p = _params(rec)
    # the reference optimises two unused 1x1 parameters too; they do not affect the output
    plist = list(p.values())
    opt_cls = torch.optim.Adam if cfg["optimizer"] == "adam" else torch.optim.AdamW
    opt = opt_cls(plist, lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    kw = dict(model_type=cfg["model_type"], variant=cfg["variant"],
              structure_info=cfg["structure_info"], attn_layernorm=cfg["attn_layernorm"])
    losses = []
    for _ in range(cfg["steps"]):
        opt.zero_grad()
        logits = O.gcn_forward(p, x, adj_low, adj_high, adj_un, dropout=0.0, training=True, **kw)
        loss = O.nll_loss_on(logits, labels, train_sel)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    with torch.no_grad():
        final = O.gcn_forward(p, x, adj_low, adj_high, adj_un, **kw)
    return np.asarray(losses), final.numpy()


@pytest.mark.parametrize("tag", ["acmgcn_adam", "acmgcnp_s1_adam"])
def test_cora_adam_trajectory(tag):
    """10 Adam steps through the reference's own train_model on Cora (a15)."""
    rec = load_npz(os.path.join(GOLDEN, f"traj_cora_{tag}.npz"))
    g = load_npz(os.path.join(GOLDEN, "graph_cora.npz"))
    n = int(g["n"])
    x = torch.from_numpy(sp.csr_matrix((g["feat_vals"], g["feat_indices"], g["feat_indptr"]),
                                       shape=(n, int(g["feat_dim"]))).toarray().astype(np.float32))
    adj_low = csr_to_coo_tensor(g, "adj_low").to_dense()        # small-graph dialect: dense A_low
    adj_high = csr_to_coo_tensor(g, "adj_high")
    adj_un = csr_to_coo_tensor(g, "adj_un") if rec["cfg"]["structure_info"] else None
    losses, final = _run_traj(rec, adj_low, adj_high, adj_un, x,
                              torch.from_numpy(g["labels"]), torch.from_numpy(g["train_mask"]))
    np.testing.assert_allclose(losses, rec["losses"], rtol=2e-5)
    np.testing.assert_allclose(final, rec["final_logits"], rtol=1e-3, atol=1e-4)


def test_geometric_adamw_trajectory():
    rec = load_npz(os.path.join(GOLDEN, "traj_geometric_acmgcnp_adamw.npz"))
    adj_low, adj_high, _, _ = graph_tensors("geometric")
    losses, final = _run_traj(rec, adj_low, adj_high, None, torch.from_numpy(rec["x"]),
                              torch.from_numpy(rec["labels"]), torch.from_numpy(rec["train_idx"]))
    np.testing.assert_allclose(losses, rec["losses"], rtol=2e-5)
    np.testing.assert_allclose(final, rec["final_logits"], rtol=1e-3, atol=1e-4)


def test_fused_algebra_on_real_structure():
    """The identities the HIP path relies on: A_high Z = Z - A_low Z and
    A S = D (A_low S) - S, on the Chameleon structure (raw self-loops included)."""
    g = load_npz(os.path.join(GOLDEN, "graph_chameleon.npz"))
    n = int(g["n"])
    a = sp.csr_matrix((np.ones(len(g["adj_un_indices"])), g["adj_un_indices"], g["adj_un_indptr"]),
                      shape=(n, n))
    assert a.diagonal().sum() > 0                           # quirk Q5: raw self-loops exist
    deg = np.asarray((sp.identity(n) + a).sum(1)).flatten()
    low = O.row_normalize_sp(sp.identity(n) + a).tocsr()
    high = (sp.identity(n) - low).tocsr()
    z = np.random.default_rng(0).standard_normal((n, 8))
    np.testing.assert_allclose(high @ z, z - low @ z, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(a @ z, deg[:, None] * (low @ z) - z, rtol=1e-9, atol=1e-10)
