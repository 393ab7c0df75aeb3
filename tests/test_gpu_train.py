"""Train-step level parity on the MI355X: fused loss kernel, the reference's recorded 10-step
Adam/AdamW trajectories (a15), eager vs hipGraph-replayed steps."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.nn.functional as F

from conftest import GOLDEN, csr_to_coo_tensor, graph_tensors, load_npz

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,c", [(1, 1), (5, 2), (1000, 2), (777, 5), (4096, 7), (300, 64)])
def test_masked_nll_kernel(n, c):
    from acm_gnn_amd import functional as AF, train as T
    g = torch.Generator().manual_seed(n + c)
    z = (torch.randn(n, c, generator=g) * 3).to(DEV).requires_grad_(True)
    y = torch.randint(0, c, (n,), generator=g).to(DEV)
    idx = torch.randperm(n, generator=g)[: max(1, n // 2)].to(DEV)
    loss = AF.masked_nll(z, y, T.row_weights(idx, n))
    (loss * 1.5).backward()
    zr = z.detach().clone().requires_grad_(True)
    ref = F.nll_loss(F.log_softmax(zr.double(), 1)[idx], y[idx])
    (ref * 1.5).backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    torch.testing.assert_close(z.grad, zr.grad, rtol=1e-5, atol=1e-7)
    assert float(z.grad[torch.ones(n, dtype=torch.bool, device=DEV).index_fill(0, idx, False)].abs().max() if n > idx.numel() else 0) == 0
    loss2 = AF.masked_nll(z.detach(), y, T.row_weights(idx, n))
    assert loss2.item() == loss.item()                       # deterministic reduction


def _set_params(module, rec):
    sd = module.state_dict()
    for k, v in rec.items():
        if k.startswith("param:"):
            sd[k[6:]].copy_(torch.from_numpy(v))


def _trajectory(rec, adj_low, adj_high, adj_un, x, labels, train_idx, use_graph, fused=False):
    from acm_gnn_amd import GCN, train as T
    from acm_gnn_amd.graph import clear_cache
    clear_cache()
    cfg = rec["cfg"]
    n = x.shape[0]
    model = GCN(x.shape[1], cfg["hidden"], int(labels.max()) + 1, 2, n, 0.0, cfg["model_type"], cfg["structure_info"],
                variant=bool(cfg["variant"]), attn_layernorm=bool(cfg["attn_layernorm"])).to(DEV)
    _set_params(model, rec)
    if fused:                                   # acm_adam_step against the reference's recorded trajectory
        from acm_gnn_amd import FusedAdam, FusedAdamW
        opt = (FusedAdam if cfg["optimizer"] == "adam" else FusedAdamW)(model.parameters(), lr=cfg["lr"],
                                                                         weight_decay=cfg["weight_decay"])
    else:
        opt_cls = torch.optim.Adam if cfg["optimizer"] == "adam" else torch.optim.AdamW
        opt = opt_cls(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"], capturable=use_graph)
    w = T.row_weights(train_idx.to(DEV), n)
    step = T.TrainStep(model, opt, x.to(DEV), adj_low.to(DEV), labels.to(DEV), w, adj_high.to(DEV),
                       adj_un.to(DEV) if adj_un is not None else None, use_graph=False)
    losses = [float(step()) for _ in range(cfg["steps"])]
    out, _ = T.evaluate(model, x.to(DEV), adj_low.to(DEV), labels.to(DEV), (train_idx.to(DEV),), adj_high.to(DEV),
                        adj_un.to(DEV) if adj_un is not None else None)
    return np.asarray(losses), out.cpu().numpy()


@pytest.mark.parametrize("fused", [False, True], ids=["torch_adam", "fused_adam"])
@pytest.mark.parametrize("tag", ["acmgcn_adam", "acmgcnp_s1_adam"])
def test_cora_adam_trajectory_matches_reference(tag, fused):
    """10 Adam steps recorded from the reference's own train_model on Cora (dense A_low dialect)."""
    rec = load_npz(os.path.join(GOLDEN, f"traj_cora_{tag}.npz"))
    g = load_npz(os.path.join(GOLDEN, "graph_cora.npz"))
    n = int(g["n"])
    x = torch.from_numpy(sp.csr_matrix((g["feat_vals"], g["feat_indices"], g["feat_indptr"]),
                                       shape=(n, int(g["feat_dim"]))).toarray().astype(np.float32))
    adj_low = csr_to_coo_tensor(g, "adj_low").to_dense()
    adj_high = csr_to_coo_tensor(g, "adj_high")
    adj_un = csr_to_coo_tensor(g, "adj_un") if rec["cfg"]["structure_info"] else None
    train_idx = torch.from_numpy(np.nonzero(g["train_mask"])[0])
    losses, final = _trajectory(rec, adj_low, adj_high, adj_un, x, torch.from_numpy(g["labels"]), train_idx, False, fused)
    np.testing.assert_allclose(losses, rec["losses"], rtol=5e-5)
    np.testing.assert_allclose(final, rec["final_logits"], rtol=5e-3, atol=2e-3)   # Adam amplifies 1e-7 gradient noise


@pytest.mark.parametrize("fused", [False, True], ids=["torch_adamw", "fused_adamw"])
def test_geometric_adamw_trajectory_matches_reference(fused):
    rec = load_npz(os.path.join(GOLDEN, "traj_geometric_acmgcnp_adamw.npz"))
    low, high, _, _ = graph_tensors("geometric")
    losses, final = _trajectory(rec, low, high, None, torch.from_numpy(rec["x"]), torch.from_numpy(rec["labels"]),
                                torch.from_numpy(rec["train_idx"]), False, fused)
    np.testing.assert_allclose(losses, rec["losses"], rtol=5e-5)
    np.testing.assert_allclose(final, rec["final_logits"], rtol=5e-3, atol=2e-3)   # Adam amplifies 1e-7 gradient noise


@pytest.mark.parametrize("fused", [False, True], ids=["torch_adamw", "fused_adamw"])
def test_graph_replayed_step_equals_eager(fused):
    """The whole step captured once in a HIP graph and replayed gives the same losses, with torch's
    optimizer and with the one-launch acm_adam_step; the two optimizers give the same losses too."""
    from acm_gnn_amd import GCN, data as D, train as T
    from acm_gnn_amd.graph import CsrGraph, FilterOperators
    adj, x_np, y_np, (tr, _, _), _ = D.synthetic_dataset("tiny", seed=1)
    low, deg = D.build_filters(adj)
    ops = FilterOperators(CsrGraph.from_scipy(low, DEV))
    x, y = torch.from_numpy(D.row_normalize_features(x_np)).to(DEV), torch.from_numpy(y_np).to(DEV)
    w = T.row_weights(torch.from_numpy(tr).to(DEV), x.shape[0])
    res = []
    for use_graph in (False, True):
        torch.manual_seed(0)
        model = GCN(7, 64, 2, 2, x.shape[0], 0.0, "acmgcnp", 0, variant=False).to(DEV)
        if fused:
            from acm_gnn_amd import FusedAdamW
            opt = FusedAdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
        else:
            opt = torch.optim.AdamW(model.parameters(), lr=0.01, weight_decay=1e-3, capturable=True)
        if use_graph:
            state = {k: v.clone() for k, v in model.state_dict().items()}
        step = T.TrainStep(model, opt, x, ops, y, w, use_graph=use_graph)
        if use_graph:                       # capture warm-up advanced the weights: rewind
            model.load_state_dict(state)
            for st in opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        res.append([float(step()) for _ in range(6)])
    np.testing.assert_allclose(res[1], res[0], rtol=1e-5)
    _LOSSES[fused] = res[0]
    if len(_LOSSES) == 2:
        np.testing.assert_allclose(_LOSSES[True], _LOSSES[False], rtol=2e-5)


_LOSSES = {}


def test_captured_eval_step_and_fit_with_graphs():
    """train.EvalStep as a hipGraph replay equals the eager evaluation, and fit(use_graph=True) -- captured training step
    + captured evaluation pass per epoch -- reproduces the history of the eager loop (same fused dropout, same optimizer)."""
    import acm_gnn_amd
    from acm_gnn_amd import data as D, distributed as DD, train as T
    from acm_gnn_amd.graph import clear_cache
    clear_cache()
    adj, x_np, y_np, (tr, va, te), n = D.synthetic_dataset("tiny", seed=3)
    low, deg = D.build_filters(adj)
    ops = DD.make_sharded_operators(low, deg, DEV)
    x, y = torch.from_numpy(x_np).to(DEV), torch.from_numpy(y_np).to(DEV)
    sets = tuple(torch.from_numpy(s).to(DEV) for s in (tr, va, te))
    n_cls = int(y_np.max()) + 1
    hist = {}
    for use_graph in (False, True):
        torch.manual_seed(1)
        model = acm_gnn_amd.GCN(x.shape[1], 64, n_cls, 2, n, 0.3, "acmgcnp", 0, variant=False).to(DEV)
        opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.02, weight_decay=1e-3)
        if use_graph:
            ev = T.EvalStep(model, x, ops, y, sets, loss_set=1, use_graph=True)
            ev_eager = T.EvalStep(model, x, ops, y, sets, loss_set=1)
            (o1, a1, l1), (o2, a2, l2) = ev(), ev_eager()
            assert torch.equal(o1, o2) and a1 == a2 and l1 == l2
        acc, h = T.fit(model, opt, x, ops, y, *sets, epochs=8, rule="min_val_loss", early_stopping=0,
                       use_graph=use_graph, fused_dropout=True)
        hist[use_graph] = (acc, h)
    (acc_e, h_e), (acc_g, h_g) = hist[False], hist[True]
    assert len(h_e) == len(h_g) == 8
    for e, g in zip(h_e, h_g):
        np.testing.assert_allclose(g, e, rtol=2e-4, atol=2e-5)
    assert abs(acc_e - acc_g) < 1e-6


def test_eval_passes_reuse_the_aggregated_input_on_the_gpu(monkeypatch):
    """acm_conv_agg_fwd_t.agg_given: the second evaluation pass over the same input skips the gather and returns the very
    same logits (the row-local stage reads the P the first pass stored); an in-place edit of the input takes the gather
    again; the captured EvalStep holds the shortened pass."""
    import acm_gnn_amd
    from acm_gnn_amd import data as D, distributed as DD, functional as AF, train as T
    from acm_gnn_amd.graph import clear_cache
    clear_cache()
    adj, x_np, y_np, (tr, va, te), n = D.synthetic_dataset("tiny", seed=4)
    low, deg = D.build_filters(adj)
    ops = DD.make_sharded_operators(low, deg, DEV)
    x, y = torch.from_numpy(x_np).to(DEV), torch.from_numpy(y_np).to(DEV)
    torch.manual_seed(2)
    model = acm_gnn_amd.GCN(x.shape[1], 64, int(y_np.max()) + 1, 2, n, 0.3, "acmgcnp", 0, variant=False).to(DEV)
    model.eval()
    calls = []
    orig = AF._gather_rows
    monkeypatch.setattr(AF, "_gather_rows", lambda o, t: (calls.append(tuple(t.shape)), orig(o, t))[1])
    with torch.no_grad():
        o1 = model(x, ops)
        n1 = len(calls)
        o2 = model(x, ops)
        o2b = model(x, ops)
        # the shortened pass runs the row-local stage as its own kernel (another instantiation of the same code): equal
        # to the fused kernel's result to rounding, and bit-identical from one shortened pass to the next
        tol = 1e-5 * max(1.0, float(o1.abs().max()))
        assert float((o1 - o2).abs().max()) < tol and torch.equal(o2, o2b)
        assert len(calls) - n1 < n1                     # the layer-1 gather of the input is gone
        for _l in model.gcns: _l.eval_agg_cache = False
        o0 = model(x, ops)
        assert torch.equal(o0, o1)                      # the pass that gathers is unchanged
        for _l in model.gcns: _l.eval_agg_cache = True
        x.add_(0.25)
        o3 = model(x, ops)
        assert float((o3 - o1).abs().max()) > 1e-3
        for _l in model.gcns: _l.eval_agg_cache = False
        assert float((model(x, ops) - o3).abs().max()) < tol
    sets = tuple(torch.from_numpy(s).to(DEV) for s in (tr, va, te))
    for _l in model.gcns: _l.eval_agg_cache = True
    ev_g = T.EvalStep(model, x, ops, y, sets, use_graph=True)
    ev_e = T.EvalStep(model, x, ops, y, sets)
    (og, ag, lg), (oe, ae, le) = ev_g(), ev_e()
    assert torch.equal(og, oe) and ag == ae and lg == le          # both are shortened passes by now


def _pipeline_case(n, avg, seed, relabel=False, with_structure=False):
    """A random graph in the input pipeline's regime (dense input of 7 features, 12 < nnz / n <= 160) + a fresh model."""
    from acm_gnn_amd import data as D
    from acm_gnn_amd.distributed import make_sharded_operators
    rng = np.random.default_rng(seed)
    m = n * avg // 2
    w = 1.0 / (np.arange(n) + 10.0) ** 0.6                       # skewed degrees: a few long rows (pieces of id streams)
    r, c = rng.choice(n, m, p=w / w.sum()), rng.integers(0, n, m)
    adj = sp.csr_matrix((np.ones(m, np.float32), (r, c)), shape=(n, n))
    adj = ((adj + adj.T) > 0).astype(np.float32).tocsr()
    adj.setdiag(0)
    adj.eliminate_zeros()
    low, deg = D.build_filters(adj)
    ops = make_sharded_operators(low, deg, torch.device(DEV), relabel=relabel, with_structure=with_structure)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 7, generator=g).to(DEV)
    y = torch.randint(0, 2, (n,), generator=g).to(DEV)
    return ops, x, y


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
@pytest.mark.parametrize("n,avg,model_type", [(3000, 40, "acmgcnp"), (20000, 24, "acmgcnp"), (3000, 40, "acmgcn"),
                                              (3001, 40, "acmgcnp"),           # 3001: degree relabelling inside the operator
                                              (3000, 40, "acmgcnpp")])         # the residual Linear reads the table too
def test_input_pipeline_matches_plain_step(monkeypatch, n, avg, model_type, use_graph, tune):
    """TrainStep with the input pipeline (the next step's P = A_low dropout(x) gathered by two extra waves per SIMD of the
    first layer's backward kernel; acm_conv_agg_bwd_t.next_agg, acm_conv_agg_fwd_t.agg_given / agg_copy,
    acm_dropout_t.step_offset) against the plain step: same masks, same losses and parameters up to the summation
    order of the gather; P itself against acm_spmm.  The small case has fewer stream waves than CUs and rows of
    several pieces; the large one is past the default size threshold."""
    from acm_gnn_amd import GCN, FusedAdamW, functional as AF, train as T
    tune(pipeline=1024)
    ops, x, y = _pipeline_case(n, avg, seed=n, relabel=n == 3001)
    assert (ops.perm is not None) == (n == 3001)
    w = T.row_weights(torch.arange(0, n, 3, device=DEV), n)

    def run(pipeline):
        torch.manual_seed(0)
        model = GCN(7, 64, 2, 2, n, 0.2, model_type, 0, variant=False, attn_layernorm=model_type == "acmgcnp").to(DEV)
        model.dropout_state = AF.DropoutState(torch.device(DEV), seed=99)
        opt = FusedAdamW(model.parameters(), lr=0.01)
        step = T.TrainStep(model, opt, x, ops, y, w, use_graph=use_graph, pipeline_input=pipeline)
        losses = [float(step()) for _ in range(8)]
        return step, model, losses

    step_a, model_a, loss_a = run(False)
    step_b, model_b, loss_b = run(None)
    assert step_a.pipe is None and step_b.pipe is not None and step_b.pipe.primed
    np.testing.assert_allclose(loss_b, loss_a, rtol=2e-4, atol=1e-5)
    for (k, pa), (_, pb) in zip(model_a.state_dict().items(), model_b.state_dict().items()):
        torch.testing.assert_close(pb, pa, rtol=5e-3, atol=5e-4, msg=k)
    # the buffers after step 8: filled = operands of step 9 (dropout with the current counter), saved = those of step 8
    pipe = step_b.pipe
    want_table = AF.dropout(step_b.x, 0.2, model_b.dropout_state, tag=0, pad_to=8)      # (step_b.x: in the operator's numbering)
    torch.testing.assert_close(pipe.filled[0], want_table, rtol=0, atol=0)
    want_p = AF.spmm(ops.low, want_table, row_scale=ops.row_scale)
    torch.testing.assert_close(pipe.filled[1], want_p, rtol=1e-5, atol=1e-5)
    assert float((pipe.saved[0] - pipe.filled[0]).abs().max()) > 0         # a different mask


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
def test_input_pipeline_whose_backward_cannot_carry_the_gather(use_graph, tune):
    """rows16 without the backward bit: acm_conv_agg_bwd declines next_agg (ACM_EUNSUPPORTED) and the host launches the
    gather itself right there -- also inside a capture, where a pipeline left unprimed would replay a stale P for ever
    (found with scripts/probe_step_env.py in round 4: the captured step was 25 us "faster" and wrong)."""
    from acm_gnn_amd import GCN, FusedAdamW, functional as AF, train as T
    n = 3000
    ops, x, y = _pipeline_case(n, 40, seed=n)
    w = T.row_weights(torch.arange(0, n, 3, device=DEV), n)

    def run(**tuning_items):
        tune(pipeline=1024, **tuning_items)
        torch.manual_seed(0)
        model = GCN(7, 64, 2, 2, n, 0.2, "acmgcnp", 0, variant=False, attn_layernorm=True).to(DEV)
        model.dropout_state = AF.DropoutState(torch.device(DEV), seed=99)
        step = T.TrainStep(model, FusedAdamW(model.parameters(), lr=0.01), x, ops, y, w, use_graph=use_graph)
        return step, model, [float(step()) for _ in range(8)]

    _, model_a, loss_a = run()
    step_b, model_b, loss_b = run(rows16=5)
    assert step_b.pipe is not None and step_b.pipe.primed
    np.testing.assert_allclose(loss_b, loss_a, rtol=2e-4, atol=1e-5)
    for (k, pa), (_, pb) in zip(model_a.state_dict().items(), model_b.state_dict().items()):
        torch.testing.assert_close(pb, pa, rtol=5e-3, atol=5e-4, msg=k)
    want_table = AF.dropout(step_b.x, 0.2, model_b.dropout_state, tag=0, pad_to=8)
    torch.testing.assert_close(step_b.pipe.filled[0], want_table, rtol=0, atol=0)
    torch.testing.assert_close(step_b.pipe.filled[1], AF.spmm(ops.low, want_table, row_scale=ops.row_scale), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("use_graph", [True, False], ids=["graph", "eager"])
def test_fit_with_the_input_pipeline_equals_fit_without(monkeypatch, use_graph, tune):
    """train.fit (captured training step + captured evaluation pass per epoch) with the input pipeline in its training
    half -- the default -- against pipeline_input=False: the evaluation pass in between neither disturbs the look-ahead
    buffers nor advances the dropout counter; same histories, same selected accuracy."""
    from acm_gnn_amd import GCN, FusedAdamW, functional as AF, train as T
    tune(pipeline=1024)
    n = 6000
    ops, x, y = _pipeline_case(n, 30, seed=11)
    idx = torch.randperm(n, generator=torch.Generator().manual_seed(3)).to(DEV)
    sets = (idx[: n // 2], idx[n // 2: 3 * n // 4], idx[3 * n // 4:])

    def run(pipeline):
        torch.manual_seed(0)
        model = GCN(7, 64, 2, 2, n, 0.3, "acmgcnp", 0, variant=False, attn_layernorm=True).to(DEV)
        model.dropout_state = AF.DropoutState(torch.device(DEV), seed=5)
        opt = FusedAdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
        return T.fit(model, opt, x, ops, y, *sets, epochs=12, use_graph=use_graph, pipeline_input=pipeline)

    acc_a, hist_a = run(False)
    acc_b, hist_b = run(None)
    assert abs(acc_a - acc_b) < 5e-3
    a = np.array([[float(v) for v in h.values()] if isinstance(h, dict) else [float(v) for v in h] for h in hist_a])
    b = np.array([[float(v) for v in h.values()] if isinstance(h, dict) else [float(v) for v in h] for h in hist_b])
    np.testing.assert_allclose(b, a, rtol=5e-3, atol=5e-3)


@pytest.mark.parametrize("model_type,s", [("acmgcnp", 0), ("acmgcnpp", 1)])
def test_fit_of_the_acmii_variant_captured_equals_eager_and_the_fp32_kernels(model_type, s, tune):
    """train.fit with the reference's default variant (ACMII, ACM-Geometric/parse.py:57) on a narrow input: the mask form of the
    first layer (acm_conv_acmii_v.hip: the table is rebuilt inside every captured step and evaluation pass, the item streams
    come from the warm-up) captured against eager, and against the fp32-MFMA forward + transposed-gather backward it
    replaces (rewrites without bit 4): same histories."""
    from acm_gnn_amd import GCN, FusedAdamW, functional as AF, train as T
    n = 5000
    ops, x, y = _pipeline_case(n, 30, seed=17, with_structure=bool(s))
    idx = torch.randperm(n, generator=torch.Generator().manual_seed(3)).to(DEV)
    sets = (idx[: n // 2], idx[n // 2: 3 * n // 4], idx[3 * n // 4:])

    def run(use_graph, mask):
        tune(acmii_mask=int(mask))
        torch.manual_seed(0)
        model = GCN(7, 64, 2, 2, n, 0.3, model_type, s, variant=True, attn_layernorm=True).to(DEV)
        model.dropout_state = AF.DropoutState(torch.device(DEV), seed=5)
        opt = FusedAdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
        timer = AF.KernelTimer()
        AF.set_kernel_timer(timer if not use_graph else None)
        try:
            acc, hist = T.fit(model, opt, x, ops, y, *sets, epochs=10, use_graph=use_graph)
        finally:
            AF.set_kernel_timer(None)
        used = set(k.split("/")[0] for k in timer.events)
        rows = np.array([[float(v) for v in h.values()] if isinstance(h, dict) else [float(v) for v in h] for h in hist])
        return acc, rows, used

    acc_e, hist_e, used = run(False, True)
    assert {"acmii_table", "conv_acmii_v_fwd", "conv_acmii_v_bwd"} <= used, sorted(used)
    acc_g, hist_g, _ = run(True, True)
    acc_o, hist_o, used_o = run(False, False)
    assert "conv_acmii_fwd" in used_o and "conv_acmii_v_fwd" not in used_o, sorted(used_o)
    np.testing.assert_allclose(hist_g, hist_e, rtol=5e-3, atol=5e-3)
    # (another summation order in both passes, ten optimizer steps on random labels: accuracies move by a few nodes)
    np.testing.assert_allclose(hist_o, hist_e, rtol=2e-2, atol=2e-2)
    assert abs(acc_g - acc_e) < 5e-3 and abs(acc_o - acc_e) < 2e-2


def test_carried_gather_leaves_the_backward_unchanged():
    """acm_conv_agg_bwd with and without next_agg: the parameter gradients agree (eight instead of four slabs per
    workgroup: another summation order), and the carried P equals the stand-alone product bit for bit across launches."""
    from acm_gnn_amd import GraphConvolution, functional as AF
    n = 6000
    ops, x, _ = _pipeline_case(n, 30, seed=3)
    assert ops.low.build_streams(n_waves=1024) and ops.low.stream_waves % 4 == 0
    torch.manual_seed(1)
    layer = GraphConvolution(7, 64, n, "acmgcnp", variant=0, structure_info=0, attn_layernorm=True).to(DEV)
    st = AF.DropoutState(torch.device(DEV), seed=5)
    pipe = AF.InputPipeline(ops, x, 0.3, st)
    pipe.prime()
    gout = torch.randn(n, 64, device=DEV)
    params = [p for p in layer.parameters() if p.requires_grad]

    def grads(carry):
        for p in params:
            p.grad = None
        xin = pipe.table() if carry else pipe.table().clone()
        out = layer(xin, ops, None, post_relu=True, call=AF.CallContext(pipe=pipe if carry else None))
        if carry:
            assert pipe.make_next()
        out.backward(gout)
        return [p.grad.clone() for p in params if p.grad is not None], out.detach().clone()

    g_plain, out_plain = grads(False)
    g_carry, out_carry = grads(True)
    assert pipe.next_agg_ready
    torch.testing.assert_close(out_carry, out_plain, rtol=1e-5, atol=1e-5)
    for a, b in zip(g_carry, g_plain):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5)
    st.step += 1                                                  # what the optimizer's step would do
    want = AF.spmm(ops.low, AF.dropout(x, 0.3, st, tag=0, pad_to=8), row_scale=ops.row_scale)
    torch.testing.assert_close(pipe.filled[1], want, rtol=1e-5, atol=1e-5)
    first = pipe.filled[1].clone()
    st.step -= 1
    pipe.prime()
    pipe.end_step()
    pipe.primed = True
    g_again, _ = grads(True)
    assert torch.equal(pipe.filled[1], first)                     # deterministic: slot order, not arrival order
    for a, b in zip(g_again, g_carry):
        assert torch.equal(a, b)


@pytest.mark.parametrize("pipeline", [None, False], ids=["pipelined", "plain"])
def test_several_steps_per_captured_graph_equal_single_step_replays(pipeline, tune):
    """TrainStep(steps_per_graph=K): K consecutive optimizer steps in ONE hipGraph (fresh counter-based masks per step: the
    step counters live on the device; the input pipeline's hand-over from step to step inside the graph) -- two calls
    equal 2 K replays of the single-step graph: every loss, every parameter, the dropout counter."""
    from acm_gnn_amd import GCN, FusedAdamW, functional as AF, train as T
    tune(pipeline=1024)
    n, K = 5000, 4
    ops, x, y = _pipeline_case(n, 30, seed=21)
    w = T.row_weights(torch.arange(0, n, 3, device=DEV), n)

    def run(spg):
        torch.manual_seed(0)
        model = GCN(7, 64, 2, 2, n, 0.25, "acmgcnp", 0, variant=False, attn_layernorm=True).to(DEV)
        model.dropout_state = AF.DropoutState(torch.device(DEV), seed=31)
        opt = FusedAdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
        step = T.TrainStep(model, opt, x, ops, y, w, use_graph=True, pipeline_input=pipeline, steps_per_graph=spg)
        assert step.steps_per_call == spg and (step.pipe is not None) == (pipeline is None)
        losses = []
        for _ in range(2 * K // spg):
            step()
            losses += [float(v) for v in step.losses] if spg > 1 else [float(step.loss)]
        return losses, [p.detach().clone() for p in model.parameters()], int(model.dropout_state.step.item())

    la, pa, ca = run(1)
    lb, pb, cb = run(K)
    assert ca == cb == 2 * K and len(la) == len(lb) == 2 * K
    assert la == lb, (la, lb)
    assert all(torch.equal(u, v) for u, v in zip(pa, pb))


def test_drop_in_route_with_dense_one_hot_features_takes_the_csr_projection():
    """layers.GraphConvolution._csr_input on the device, in the reference's loop shape (ACM-Geometric/train.py:119-140 around
    models.py:52-76: F.dropout of the dense features, first layer; an evaluation pass per epoch): [16 384, 1 024] one-hot
    features (2^24 elements: the per-step support check runs).  After the first evaluation pass every training pass
    projects the dropped copy from the reference structure; output and weight gradients equal the dense projection's on
    the same dropped tensor, and a tensor with an entry outside the structure keeps the dense projection."""
    import numpy as np
    import scipy.sparse as sp
    import torch.nn.functional as F
    from acm_gnn_amd import GraphConvolution, SparseFeatures, layers as L, tuning
    from acm_gnn_amd.graph import clear_cache
    from oracle import acm_oracle as O
    clear_cache()
    n, f_in = 16384, 1024
    rng = np.random.default_rng(3)
    a = sp.random(n, n, density=8.0 / n, random_state=rng, format="csr", dtype=np.float32)
    a.data[:] = 1.0
    a = ((a + a.T) > 0).astype(np.float32).tocsr()
    a.setdiag(0)
    a.eliminate_zeros()
    low, high, _ = O.filters_linkx(a)
    low, high = low.to(DEV), high.to(DEV)
    x = torch.zeros(n, f_in)
    x[torch.arange(n).repeat_interleave(4), torch.from_numpy(rng.integers(0, f_in, 4 * n))] = 1.0
    x = (x / x.sum(1, keepdim=True)).to(DEV)
    torch.manual_seed(4)
    layer = GraphConvolution(f_in, 64, n, "acmgcnp").to(DEV)
    seen = []
    conv = L.AF.acm_conv
    L.AF.acm_conv = lambda inp, *a_, **k: (seen.append(type(inp).__name__), conv(inp, *a_, **k))[1]
    try:
        def train_pass(inp):
            layer.train()
            layer.zero_grad()
            out = layer(inp, low, high, None)
            out.square().sum().backward()
            return out.detach(), {k: p.grad.clone() for k, p in layer.named_parameters() if p.grad is not None}

        xd = F.dropout(x, 0.5, training=True)
        train_pass(xd)                                       # first epoch: no reference structure yet
        layer.eval()
        with torch.no_grad():
            e_csr = layer(x, low, high, None)
            with tuning.override(csr_features=0):
                e_dense = layer(x.clone(), low, high, None)
        assert seen == ["Tensor", "SparseFeatures", "Tensor"], seen
        assert isinstance(SparseFeatures.known_twin(x), SparseFeatures)
        scale = float(e_dense.abs().max())
        assert float((e_csr - e_dense).abs().max()) < 2e-5 * max(1.0, scale)
        del seen[:]
        out_c, g_c = train_pass(xd)
        with tuning.override(csr_features=0):
            out_d, g_d = train_pass(xd)
        assert seen == ["SparseFeatures", "Tensor"], seen
        assert float((out_c - out_d).abs().max()) < 2e-5 * max(1.0, float(out_d.abs().max()))
        assert set(g_c) == set(g_d)
        for k, v in g_d.items():
            assert float((g_c[k] - v).abs().max()) < 1e-4 * max(1.0, float(v.abs().max())), k
        bad = xd.clone()
        bad[5, int((x[5] == 0).nonzero()[0])] = 0.25
        del seen[:]
        train_pass(bad)
        assert seen == ["Tensor"], seen
    finally:
        L.AF.acm_conv = conv
        clear_cache()


@pytest.mark.gpu
def test_captured_dropout_free_fit_on_relabelled_operators_survives_an_emptied_cache():
    """fit(use_graph=True) of a dropout-0 model on relabelled operators.  The captured training step
    reads the first layer's P = A_low X at the address its capture saw; the EvalStep built next hands the model a fresh
    permuted copy of x, which used to replace the layer's only cache entry and free that P.  TrainStep now holds the
    entries it captured with (and training / evaluation inputs have their own): emptying the allocator's cache and
    churning memory between the replays must not change a single loss against the eager loop."""
    from acm_gnn_amd import GCN, FusedAdamW, train as T
    n = 3001
    ops, x, y = _pipeline_case(n, 40, seed=7, relabel=True)
    assert ops.perm is not None
    idx = torch.randperm(n, generator=torch.Generator().manual_seed(1)).to(DEV)
    tr, va, te = idx[:1500], idx[1500:2250], idx[2250:]

    def run(use_graph):
        torch.manual_seed(0)
        model = GCN(7, 64, 2, 2, n, 0.0, "acmgcnp", 0, variant=False, attn_layernorm=True).to(DEV)
        opt = FusedAdamW(model.parameters(), lr=0.01)
        w = T.row_weights(tr, n, device=DEV)
        step = T.TrainStep(model, opt, x, ops, y, w, use_graph=use_graph)
        ev = T.EvalStep(model, x, ops, y, (tr, va, te), use_graph=use_graph)
        if use_graph:
            assert step._held, "the captured step holds no cache entry: nothing keeps its P alive"
        out = []
        for _ in range(6):
            loss = float(step())
            torch.cuda.empty_cache()
            junk = [torch.full((n, 8), float("nan"), device=DEV) for _ in range(8)]      # whatever is free gets overwritten
            del junk
            _, accs, vloss = ev()
            out.append((loss, vloss) + tuple(accs))
        return np.asarray(out)

    np.testing.assert_allclose(run(True), run(False), rtol=2e-5, atol=1e-6)


@pytest.mark.gpu
def test_kernel_timer_brackets_captured_launches_with_event_record_nodes():
    """functional.KernelTimer(external=True): a launch made under capture sits between two event-record nodes of the
    hipGraph (added through hipGraphAddEventRecordNode: ROCm rejects external event records, torch refuses them), every
    replay re-records them, and the replayed losses are those of a capture without the timer (bench.py's roofline.avg_ms)."""
    from acm_gnn_amd import GCN, FusedAdamW, data as D, functional as AF, train as T
    from acm_gnn_amd.graph import CsrGraph, FilterOperators
    adj, x_np, y_np, (tr, _, _), _ = D.synthetic_dataset("tiny", seed=1)
    low, deg = D.build_filters(adj)
    ops = FilterOperators(CsrGraph.from_scipy(low, DEV))
    x, y = torch.from_numpy(D.row_normalize_features(x_np)).to(DEV), torch.from_numpy(y_np).to(DEV)
    w = T.row_weights(torch.from_numpy(tr).to(DEV), x.shape[0])
    losses = []
    for timed in (False, True):
        torch.manual_seed(0)
        model = GCN(7, 64, 2, 2, x.shape[0], 0.0, "acmgcnp", 0, variant=False).to(DEV)
        opt = FusedAdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
        probe = AF.KernelTimer(only="conv_", external=True) if timed else None
        AF.set_kernel_timer(probe)
        try:
            step = T.TrainStep(model, opt, x, ops, y, w, use_graph=True, small_step=False)
        finally:
            AF.set_kernel_timer(None)
        losses.append([float(step()) for _ in range(4)])
        if timed:
            assert probe.captured
            for label in probe.captured:
                for _ in range(2):
                    step()
                    ms = probe.captured_ms(label)
                    assert len(ms) == len(probe.captured[label]) and all(0.0 < v < 5.0 for v in ms), (label, ms)
    np.testing.assert_allclose(losses[1], losses[0], rtol=1e-6)
