"""Full-size parity for BASELINE configs 4 and 5: the HIP path on exactly the workloads bench.py and
scripts/bench_configs.py time, against the CPU oracle -- not on miniatures of them.

Config 4  ACM-GCN+ on the twitch-gamer-shaped graph (168 114 nodes, nnz(A_low) = 13 763 228, a 21 k-degree hub
          split into window-packed pieces, persistent 768 / 3 072-block grids, 32-bit row offsets):
          one forward + backward of the 2-layer model for (variant, structure_info) in {0,1}^2 and both node
          orders, logits / loss / every parameter gradient vs oracle.gcn_forward fed CSR operands
          (ACM-Geometric/layers.py:78-116, models.py:50-76, train.py:133-135); the same through the fused
          training-step path (loss tail + deferred reductions) and with the counter-based dropout masks
          (numpy Philox) replayed into the oracle.
Config 5  ACM-SGC 3-hop on arXiv-year- and Penn94-shaped graphs (dense and CSR features), against the oracle's
          float64 chain of products (ACM-Pytorch/utils.py:631-637; the reference cannot materialise A_low^3 here).

Tolerances (fp32 against fp32 / fp64 references, stated where asserted): forward 1e-4 of the output range,
gradients 3e-4 of each gradient's range.  The measured errors are written to
gpurun_out/fullsize_parity.json (copied to profiles/ by the round's collection script).
"""
import json
import os
import time

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import fake_lib
from oracle import acm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = {}
FWD_TOL, GRAD_TOL, LOSS_TOL = 1e-4, 3e-4, 3e-5


def _record(key, **vals):
    REPORT[key] = {k: (float(v) if isinstance(v, (int, float, np.floating)) else v) for k, v in vals.items()}
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "fullsize_parity.json"), "w") as fh:
        json.dump(REPORT, fh, indent=1, sort_keys=True)


_WL = {}


def _workload(dataset, order, normalize):
    from acm_gnn_amd import data as D
    key = (dataset, order, normalize)
    if key not in _WL:
        while len(_WL) >= 2:                          # at most two full-size workloads resident
            _WL.pop(next(iter(_WL)))
        _WL[key] = D.bench_workload(dataset, seed=0, node_order=order, normalize_features=normalize)
    return _WL[key]


def _csr_t(m, dtype=torch.float32):
    m = m.tocsr()
    m.sort_indices()
    return torch.sparse_csr_tensor(torch.from_numpy(m.indptr.astype(np.int64)), torch.from_numpy(m.indices.astype(np.int64)),
                                   torch.from_numpy(m.data).to(dtype), size=m.shape)


def _oracle_operands(wl):
    """(A_low, I - A_low, A) as torch CSR -- the reference's filters (train.py:76-81) in the layout its CPU
    products are fastest in; same values as the COO tensors the reference builds."""
    low = wl["low"]
    n = low.shape[0]
    high = (sp.identity(n, dtype=np.float32, format="csr") - low).tocsr()
    return _csr_t(low), _csr_t(high), _csr_t(wl["adj"].astype(np.float32))


def _errs(got, ref):
    d = float((got.double() - ref.double()).abs().max())
    return d, d / max(float(ref.abs().max()), 1e-30)


def _elementwise(got, ref):
    """Quantiles of the element-wise error relative to |ref| + 1e-3 rms(ref): the twitch-shaped features are
    row-normalised N(0,1) draws (train.py:69-73 divides by row sums that may be tiny), so a few rows carry logits
    1e5 times the typical one and the max-norm alone would say little about ordinary rows."""
    g, r = got.double().flatten(), ref.double().flatten()
    e = (g - r).abs() / (r.abs() + 1e-3 * r.pow(2).mean().sqrt())
    q = torch.quantile(e[torch.randperm(e.numel(), generator=torch.Generator().manual_seed(0))[:1_000_000]],
                       torch.tensor([0.5, 0.999], dtype=torch.float64))
    return float(q[0]), float(q[1]), float(e.max())


def _compare_grads(model, params, tag, rec, params64=None):
    """Every parameter gradient against the oracle.  Yardstick: GRAD_TOL of the gradient's own range plus 1e-6 (sums of
    ~1e5 terms of size ~1e-5: the fp32 noise floor of a gradient whose range is itself 1e-4).  When the float64 oracle
    gradients are given (`params64`), a gradient may instead be as far from them as 5x the fp32 oracle's OWN error:
    the row-normalised features of this workload reach 1e4, so fp32 sums of the first layer's weight gradients carry
    absolute errors of 1e-2 in any summation order -- the reference's included."""
    worst = 0.0
    for k, p in model.named_parameters():
        if k in ("fea_param", "xX_param"):
            continue
        rg = params[k].grad
        if rg is None:
            assert p.grad is None, k
            continue
        assert p.grad is not None, k
        got = p.grad.cpu()
        d, rel = _errs(got, rg)
        tol = GRAD_TOL * float(rg.abs().max()) + 1e-6
        if d >= tol and callable(params64):
            params64 = params64()                     # the float64 oracle step, only when a gradient needs the yardstick
        if d >= tol and params64 is not None:
            r64 = params64[k].grad
            d64, _ = _errs(got, r64)
            e_ref, _ = _errs(rg, r64)
            rec[f"grad64:{k}"] = [d64, e_ref]
            ok = d < tol or d64 <= max(5.0 * e_ref, tol)
        else:
            ok = d < tol
        rec[f"grad:{k}"] = rel
        worst = max(worst, rel)
        assert ok, (tag, k, d, rel, rec.get(f"grad64:{k}"))
    return worst


def _oracle_step(params, x, y, tr, ops_t, structure, variant, dtype=torch.float32, model_type="acmgcnp", **kw):
    """oracle.gcn_forward + loss + backward in `dtype`; returns (logits, loss, params with .grad)."""
    low_t, high_t, un_t = ops_t
    if dtype != torch.float32:
        conv = lambda t: torch.sparse_csr_tensor(t.crow_indices(), t.col_indices(), t.values().to(dtype), size=t.shape)  # noqa: E731
        low_t, high_t, un_t = conv(low_t), conv(high_t), conv(un_t)
    ps = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in params.items()}
    if "masks" in kw and kw["masks"]:
        kw = dict(kw, masks={k: v.to(dtype) for k, v in kw["masks"].items()})
    ref = O.gcn_forward(ps, x.to(dtype), low_t, high_t, un_t if structure else None, model_type=model_type,
                        variant=bool(variant), structure_info=structure, attn_layernorm=True, training=True, **kw)
    loss = O.nll_loss_on(ref, y, torch.from_numpy(tr))
    loss.backward()
    return ref.detach(), loss.detach(), ps


def _factors(state, p, tag, n, c, step=None):
    from oracle.philox import dropout_factors
    return dropout_factors(state.seed, int(state.step.item()) if step is None else step, tag, float(np.float32(p)), n, c)


# (variant, structure_info, node order, model): every cell of the reference's grid on the degree-ordered graph
# (ACM-Geometric/sh/run_all_settings.sh:10-14: variant x structure_info x {acmgcnp, acmgcnpp}); the random order -- the
# in-operator relabelling, other row lengths per block -- for the two cells that take different kernels
TWITCH = [(v, s, "degree", "acmgcnp") for s in (0, 1) for v in (0, 1)] + \
         [(0, 0, "degree", "acmgcnpp"), (1, 1, "degree", "acmgcnpp"), (0, 0, "random", "acmgcnp")]


@pytest.mark.parametrize("variant,structure,order,model_type", TWITCH)
def test_twitch_shaped_step_matches_oracle(variant, structure, order, model_type):
    """One forward + loss + backward of the bench model at bench size; the plain autograd route and the fused
    training-step route (output-layer tail kernel, deferred reductions) against the oracle."""
    import acm_gnn_amd
    from acm_gnn_amd import distributed as DD, functional as AF, train as T
    wl = _workload("twitch-gamer", order, not structure)
    n = wl["adj"].shape[0]
    tr = wl["splits"][0]
    x, y = torch.from_numpy(wl["x"]), torch.from_numpy(wl["y"])
    torch.manual_seed(7)
    model = acm_gnn_amd.GCN(x.shape[1], 64, 2, 2, n, 0.0, model_type, structure, variant=bool(variant),
                            attn_layernorm=True)
    with torch.no_grad():                              # non-trivial LayerNorm parameters
        for m in model.gcns:
            for nm in ("low", "high", "mlp", "struc_low"):
                getattr(m, f"layer_norm_{nm}").weight.uniform_(0.5, 1.5)
                getattr(m, f"layer_norm_{nm}").bias.uniform_(-0.5, 0.5)
    p0 = {k: v.detach().cpu().clone() for k, v in model.named_parameters() if k not in ("fea_param", "xX_param")}
    ops_t = _oracle_operands(wl)
    t0 = time.time()
    ref, ref_loss, params = _oracle_step(p0, x, y, tr, ops_t, structure, variant, dropout=0.0, model_type=model_type)
    t_oracle = time.time() - t0
    cache64 = {}

    def params64():
        if "p" not in cache64:
            cache64["p"] = _oracle_step(p0, x, y, tr, ops_t, structure, variant, dtype=torch.float64, dropout=0.0,
                                        model_type=model_type)[2]
        return cache64["p"]

    model = model.to(DEV)
    ops = DD.make_sharded_operators(wl["low"], wl["deg"], DEV, with_structure=bool(structure))
    assert ops.implicit and ops.low.n_long_rows > 0 and ops.low.nnz == 13_763_228
    xd, yd = x.to(DEV), y.to(DEV)
    w = T.row_weights(torch.from_numpy(tr).to(DEV), n)
    rec = {"oracle_s": t_oracle, "max_degree": int(ops.low.max_degree), "long_rows": int(ops.low.n_long_rows)}
    # (1) plain autograd route
    timer = AF.KernelTimer()
    AF.set_kernel_timer(timer)
    try:
        model.train()
        logits = model(xd, ops)
        loss = AF.masked_nll(logits, yd, w)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        AF.set_kernel_timer(None)
    used = sorted(set(k.split("/")[0] for k in timer.events))
    rec["kernels"] = used
    if not variant:
        assert "conv_agg_fwd" in used and any(k.startswith("conv_agg_bwd") for k in used), used       # the headline kernels, at headline size
    else:
        assert "conv_fwd" in used and "conv_bwd_spmm" in used, used          # the wide gathers
    d, rel = _errs(logits.detach().cpu(), ref.detach())
    rec["logits_abs"], rec["logits_rel"] = d, rel
    assert rel < FWD_TOL, (d, rel)                                            # FWD_TOL of the logit range
    rec["logits_elem_p50"], rec["logits_elem_p999"], rec["logits_elem_max"] = _elementwise(logits.detach().cpu(), ref.detach())
    assert rec["logits_elem_p999"] < 1e-3, rec                                # 99.9 % of the elements to 1e-3 of their own size
    rec["loss"], rec["loss_ref"] = float(loss), float(ref_loss)
    # the mean of 84 k terms spanning five orders of magnitude, summed in fp32 in different orders on both sides
    assert abs(float(loss) - float(ref_loss)) < LOSS_TOL * max(1.0, abs(float(ref_loss)))
    rec["grad_worst_rel"] = _compare_grads(model, params, "autograd", rec, params64)
    # (2) the fused training-step route: same numbers through acm_conv_fwd_tail + the deferred flush
    opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.0, weight_decay=0.0)
    step = T.TrainStep(model, opt, xd, ops, yd, w, use_graph=False, fused_dropout=False)
    opt.zero_grad(set_to_none=True)
    loss2 = step._forward_backward()
    torch.cuda.synchronize()
    assert abs(float(loss2) - float(ref_loss)) < LOSS_TOL * max(1.0, abs(float(ref_loss)))
    rec["grad_worst_rel_fused_step"] = _compare_grads(model, params, "train-step", {}, params64)
    _record(f"twitch/{model_type}/v{variant}s{structure}/{order}", **rec)


@pytest.mark.parametrize("variant,structure", [(0, 0), (1, 1)])
def test_twitch_shaped_step_with_counter_based_dropout_matches_oracle(variant, structure):
    """The benchmark's actual step (dropout 0.1 generated inside the kernels): the masks are a pure function of
    (seed, step, tag, row, col), so numpy regenerates them and the oracle replays them (masks 'x' and 'hidden',
    models.py:54,70)."""
    import acm_gnn_amd
    from acm_gnn_amd import distributed as DD, functional as AF, train as T
    p_drop = 0.1
    wl = _workload("twitch-gamer", "degree", not structure)
    n = wl["adj"].shape[0]
    tr = wl["splits"][0]
    x, y = torch.from_numpy(wl["x"]), torch.from_numpy(wl["y"])
    torch.manual_seed(8)
    model = acm_gnn_amd.GCN(x.shape[1], 64, 2, 2, n, p_drop, "acmgcnp", structure, variant=bool(variant),
                            attn_layernorm=True)
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()
              if k not in ("fea_param", "xX_param")}
    model = model.to(DEV)
    ops = DD.make_sharded_operators(wl["low"], wl["deg"], DEV, with_structure=bool(structure))
    xd, yd = x.to(DEV), y.to(DEV)
    w = T.row_weights(torch.from_numpy(tr).to(DEV), n)
    opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.0, weight_decay=0.0)
    step = T.TrainStep(model, opt, xd, ops, yd, w, use_graph=False, fused_dropout=True)
    assert model.fused_dropout
    assert (step.pipe is not None) == (not variant and not structure)     # the headline cell runs through the input pipeline
    st = model.dropout_state
    st.step.fill_(41)
    opt.zero_grad(set_to_none=True)
    loss = step._forward_backward()
    torch.cuda.synchronize()
    masks = {"x": torch.from_numpy(_factors(st, p_drop, 0, n, x.shape[1]) > 0).float(),
             "hidden": torch.from_numpy(_factors(st, p_drop, 1, n, 64) > 0).float()}
    ops_t = _oracle_operands(wl)
    p0 = {k: v.detach() for k, v in params.items()}
    _, ref_loss, params = _oracle_step(p0, x, y, tr, ops_t, structure, variant, dropout=p_drop, masks=masks)
    params64 = lambda: _oracle_step(p0, x, y, tr, ops_t, structure, variant, dtype=torch.float64, dropout=p_drop,  # noqa: E731
                                    masks=masks)[2]
    assert abs(float(loss) - float(ref_loss)) < LOSS_TOL * max(1.0, abs(float(ref_loss))), (float(loss), float(ref_loss))
    rec = {"loss": float(loss), "loss_ref": float(ref_loss)}
    rec["grad_worst_rel"] = _compare_grads(model, params, "dropout-step", rec, params64)
    _record(f"twitch-dropout/v{variant}s{structure}", **rec)


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_twitch_shaped_pipelined_steps_match_oracle(mode):
    """The headline configuration of bench.py as it is timed: train.TrainStep WITH the input pipeline (asserted), at bench
    size.  Two consecutive steps with lr = 0 (the parameters stay put, the dropout counter moves 41 -> 42 -> 43):
      * after step one, the P the first layer's backward kernel carried (acm_conv_agg_bwd_t.next_agg: 256 workgroups of
        sixteen waves, id streams with the 21 k-degree hub in pieces) must be A_low (x (.) mask_42 / (1 - p)) -- checked
        against a float64 scipy product with the mask regenerated in numpy;
      * step two -- whose forward consumed exactly that P and whose first-layer forward ran as the row-local stage alone --
        must give the oracle's loss and every parameter gradient under the replayed masks of counter 42
        (ACM-Geometric/models.py:54,70; train.py:133-135).
    'graph': the same through the captured step (hipGraph replays), which is what bench.py's headline number times."""
    import acm_gnn_amd
    from acm_gnn_amd import distributed as DD, functional as AF, train as T
    p_drop = 0.1
    wl = _workload("twitch-gamer", "degree", True)
    n = wl["adj"].shape[0]
    tr = wl["splits"][0]
    x, y = torch.from_numpy(wl["x"]), torch.from_numpy(wl["y"])
    torch.manual_seed(9)
    model = acm_gnn_amd.GCN(x.shape[1], 64, 2, 2, n, p_drop, "acmgcnp", 0, variant=False, attn_layernorm=True)
    p0 = {k: v.detach().cpu().clone() for k, v in model.named_parameters() if k not in ("fea_param", "xX_param")}
    model = model.to(DEV)
    ops = DD.make_sharded_operators(wl["low"], wl["deg"], DEV)
    xd, yd = x.to(DEV), y.to(DEV)
    w = T.row_weights(torch.from_numpy(tr).to(DEV), n)
    opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.0, weight_decay=0.0)
    model.dropout_state = st = AF.DropoutState(torch.device(DEV), seed=1234)
    st.step.fill_(41)
    timer = AF.KernelTimer()
    AF.set_kernel_timer(timer if mode == "eager" else None)      # (no timing events inside a capture)
    try:
        step = T.TrainStep(model, opt, xd, ops, yd, w, use_graph=(mode == "graph"), fused_dropout=True)
        assert step.pipe is not None, "the benchmark configuration must qualify for the input pipeline"
        assert int(st.step.item()) == 41                 # (a capture's warm-up is undone)
        step()                                           # counter 41 -> 42; its backward carried P_42
        torch.cuda.synchronize()
    finally:
        AF.set_kernel_timer(None)
    assert int(st.step.item()) == 42 and step.pipe.primed and not step.pipe.stale()
    if mode == "eager":
        used = sorted(set(k.split("/")[0] for k in timer.events))
        assert any(k.startswith("conv_agg_bwd+gather") for k in used) and "conv_agg_epi" in used, used
    # ---- the carried P against float64
    f42 = _factors(st, p_drop, 0, n, x.shape[1], step=42)
    xdrop = wl["x"].astype(np.float64) * f42
    p_ref = wl["low"].astype(np.float64) @ xdrop
    got_tab = step.pipe.filled[0].cpu().double().numpy()
    got_p = step.pipe.filled[1].cpu().double().numpy()
    assert np.array_equal(got_tab[:, :x.shape[1]], (wl["x"] * f42.astype(np.float32)).astype(np.float64)) and not got_tab[:, x.shape[1]:].any()
    scale = float(np.abs(p_ref).max())
    err_p = float(np.abs(got_p[:, :x.shape[1]] - p_ref).max())
    row_err = np.abs(got_p[:, :x.shape[1]] - p_ref).max(1) / (np.abs(wl["low"]).astype(np.float64) @ np.abs(xdrop)).max(1).clip(1e-30)
    rec = {"carried_P_max_abs": err_p, "carried_P_range": scale, "carried_P_rowwise_rel": float(row_err.max())}
    assert err_p < 1e-5 * scale and not got_p[:, x.shape[1]:].any(), rec
    assert float(row_err.max()) < 2e-5, rec              # every row to fp32 summation error of its own terms
    # ---- step two against the oracle under the masks of counter 42
    if mode == "graph":
        loss2 = step()                                   # a replay: gradients stay in the captured .grad tensors
    else:
        opt.zero_grad(set_to_none=True)
        loss2 = step._forward_backward()
    torch.cuda.synchronize()
    masks = {"x": torch.from_numpy(f42 > 0).float(), "hidden": torch.from_numpy(_factors(st, p_drop, 1, n, 64, step=42) > 0).float()}
    ops_t = _oracle_operands(wl)
    _, ref_loss, params = _oracle_step(p0, x, y, tr, ops_t, 0, 0, dropout=p_drop, masks=masks)
    params64 = lambda: _oracle_step(p0, x, y, tr, ops_t, 0, 0, dtype=torch.float64, dropout=p_drop, masks=masks)[2]  # noqa: E731
    rec["loss"], rec["loss_ref"] = float(loss2), float(ref_loss)
    assert abs(float(loss2) - float(ref_loss)) < LOSS_TOL * max(1.0, abs(float(ref_loss))), rec
    rec["grad_worst_rel"] = _compare_grads(model, params, "pipelined-" + mode, rec, params64)
    _record(f"twitch-pipelined/{mode}", **rec)


@pytest.mark.parametrize("dataset,sparse_x", [("arxiv-year", False), ("penn94", True), ("penn94", False), ("penn94", "auto")])
def test_three_hop_acm_sgc_at_linkx_size(dataset, sparse_x):
    """BASELINE config 5 on one GPU: GCN('acmsgc') with FilterOperators.hops = 3 (chain of 1-hop products, forward
    and transposed backward) against the oracle's float64 chain."""
    import acm_gnn_amd
    from acm_gnn_amd import distributed as DD, functional as AF, train as T
    from acm_gnn_amd.graph import SparseFeatures
    wl = _workload(dataset, "random", True)
    n = wl["adj"].shape[0]
    tr = wl["splits"][0]
    y = torch.from_numpy(wl["y"])
    n_cls = int(wl["y"].max()) + 1
    f_in = wl["x"].shape[1]
    torch.manual_seed(9)
    model = acm_gnn_amd.GCN(f_in, 64, n_cls, 2, n, 0.0, "acmsgc", 0)
    layer = model.gcns[0]
    p64 = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in layer.named_parameters()}
    low64 = _csr_t(wl["low"].astype(np.float64), torch.float64)
    high64 = _csr_t((sp.identity(n, dtype=np.float64, format="csr") - wl["low"].astype(np.float64)).tocsr(), torch.float64)
    # sparse_x: True = the caller hands over SparseFeatures; "auto" = dense one-hot features, the model makes the CSR twin
    # itself (tuning csr_features, the default); False = the dense projection (key off for Penn94, whose features qualify)
    auto = sparse_x == "auto"
    x64 = _csr_t(sp.csr_matrix(wl["x"].astype(np.float64)), torch.float64) if sparse_x else torch.from_numpy(wl["x"]).double()
    ref = O.sgc_khop_forward(p64, x64, low64, high64, 3)
    ref_loss = O.nll_loss_on(ref, y, torch.from_numpy(tr))
    ref_loss.backward()

    model = model.to(DEV)
    ops = DD.make_sharded_operators(wl["low"], wl["deg"], DEV)
    ops.hops = 3
    xd = (SparseFeatures.from_scipy(sp.csr_matrix(wl["x"]), DEV) if sparse_x is True else torch.from_numpy(wl["x"]).to(DEV))
    w = T.row_weights(torch.from_numpy(tr).to(DEV), n)
    model.train()
    from acm_gnn_amd import tuning
    with tuning.override(csr_features=256 if auto else 0):
        logits = model(xd, ops)
        loss = AF.masked_nll(logits, y.to(DEV), w)
        loss.backward()
    torch.cuda.synchronize()
    assert (getattr(xd, "_acm_csr_twin", (0, None))[1] is not None) == auto
    d, rel = _errs(logits.detach().cpu(), ref.detach())
    rec = {"logits_abs": d, "logits_rel": rel, "loss": float(loss), "loss_ref": float(ref_loss)}
    assert rel < FWD_TOL, (d, rel)
    assert abs(float(loss) - float(ref_loss)) < LOSS_TOL * max(1.0, abs(float(ref_loss)))
    worst = 0.0
    for k, p in layer.named_parameters():
        rg = p64[k].grad
        if rg is None:
            assert p.grad is None, k
            continue
        dd, r = _errs(p.grad.cpu(), rg)
        rec[f"grad:{k}"] = r
        worst = max(worst, r)
        assert r < GRAD_TOL, (k, dd, r)
    rec["grad_worst_rel"] = worst
    _record(f"sgc3hop/{dataset}/{'auto-csr' if auto else 'csr' if sparse_x else 'dense'}-features", **rec)
