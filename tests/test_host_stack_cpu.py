"""Host logic on CPU: the ctypes structs, autograd wiring, GraphConvolution / GCN and the
operator cache, driven through the numpy test double of the C ABI (tests/fake_lib.py) and
checked against the reference goldens.  The kernels themselves are checked on the GPU box."""
import os

import numpy as np
import pytest
import torch

import torch.nn.functional as F

import fake_lib
from conftest import GOLDEN, golden_files, graph_tensors, load_npz, tune_now as tune

FWD = dict(rtol=1e-5, atol=1e-5)


def _close(actual, desired, what, rtol=1e-4, atol=5e-5):
    atol = atol * max(1.0, float(np.abs(desired).max()))
    np.testing.assert_allclose(actual.detach().numpy(), desired, err_msg=what, rtol=rtol, atol=atol)


def _set_params(module, rec):
    sd = module.state_dict()
    for k, v in rec.items():
        if k.startswith("param:"):
            assert tuple(sd[k[6:]].shape) == tuple(v.shape), k
            sd[k[6:]].copy_(torch.from_numpy(v))


@pytest.mark.parametrize("path", golden_files("layer_*.npz"), ids=os.path.basename)
def test_layer_host_path_against_golden(path, monkeypatch):
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GraphConvolution
    rec = load_npz(path)
    cfg = rec["cfg"]
    low, high, un, _ = graph_tensors(cfg["dialect"])
    layer = GraphConvolution(cfg["f_in"], cfg["f_out"], rec["x"].shape[0], cfg["model_type"], variant=cfg["variant"],
                             structure_info=cfg["structure_info"], attn_layernorm=bool(cfg["attn_layernorm"]))
    _set_params(layer, rec)
    x = torch.from_numpy(rec["x"].copy()).requires_grad_(True)
    out = layer(x, low, high, un if cfg["structure_info"] else None)
    out.backward(torch.from_numpy(rec["grad_out"]))
    _close(out, rec["out"], "out", **FWD)
    _close(x.grad, rec["grad_x"], "grad_x")
    named = dict(layer.named_parameters())
    for k, v in rec.items():
        if k.startswith("grad:"):
            assert named[k[5:]].grad is not None, k
            _close(named[k[5:]].grad, v, k)
    for name, p in named.items():
        if "grad:" + name not in rec:
            assert p.grad is None, name


@pytest.mark.parametrize("path", golden_files("model_*.npz"), ids=os.path.basename)
def test_model_host_path_against_golden(path, monkeypatch):
    import torch.nn.functional as F
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN
    rec = load_npz(path)
    cfg = rec["cfg"]
    low, high, un, _ = graph_tensors(cfg["dialect"])
    model = GCN(cfg["f_in"], cfg["hidden"], cfg["classes"], 2, rec["x"].shape[0], cfg["dropout"], cfg["model_type"],
                cfg["structure_info"], variant=cfg["variant"], attn_layernorm=bool(cfg["attn_layernorm"]))
    _set_params(model, rec)
    order = ["x"] + (["xX"] if cfg["model_type"] == "acmgcnpp" else []) + ["hidden"]
    masks = [torch.from_numpy(rec["mask:" + nm].astype(np.float32)) for nm in order if "mask:" + nm in rec]

    def replay(inp, p=0.5, training=True, inplace=False):
        return inp if (not training or p == 0.0) else inp * masks.pop(0) / (1.0 - p)

    monkeypatch.setattr(F, "dropout", replay)
    model.train()
    logits = model(torch.from_numpy(rec["x"]), low, high, un if cfg["structure_info"] else None)
    idx, labels = torch.from_numpy(rec["train_idx"]), torch.from_numpy(rec["labels"])
    loss = F.nll_loss(F.log_softmax(logits, dim=1)[idx], labels[idx])
    loss.backward()
    _close(logits, rec["logits"], "logits", **FWD)
    named = dict(model.named_parameters())
    for k, v in rec.items():
        if k.startswith("grad:"):
            _close(named[k[5:]].grad, v, k)


def test_aggregate_first_dispatch_rules(monkeypatch, tune):
    """First-layer shape (F_in = 7 < F = 64, no input gradient) takes the aggregate-first entry
    points, with or without the structure channel; ACMII / differentiable input take the literal ones."""
    fake = fake_lib.install(monkeypatch)
    from acm_gnn_amd import GraphConvolution
    calls = []
    for name in ("acm_conv_fwd", "acm_conv_agg_fwd", "acm_conv_agg_bwd", "acm_conv_bwd_spmm", "acm_gemm",
                 "acm_spmm_ex"):
        orig = getattr(fake, name)
        monkeypatch.setattr(fake, name, (lambda o, n: lambda *a: (calls.append(n), o(*a))[1])(orig, name))
    low, high, un, _ = graph_tensors("geometric")
    n = low.shape[0]

    def run(model_type, variant, s, x_grad, f_in=7, f_out=64):
        calls.clear()
        layer = GraphConvolution(f_in, f_out, n, model_type, variant=variant, structure_info=s, attn_layernorm=True)
        x = torch.randn(n, f_in, requires_grad=x_grad)
        layer(x, low, high, un if s else None).sum().backward()
        return set(calls)

    assert run("acmgcnp", 0, 0, False) == {"acm_conv_agg_fwd", "acm_conv_agg_bwd"}
    assert "acm_conv_agg_fwd" not in run("acmgcnp", 1, 0, False)
    assert run("acmgcnp", 0, 1, False) == {"acm_conv_agg_fwd", "acm_conv_agg_bwd", "acm_spmm_ex"}
    assert "acm_conv_agg_fwd" not in run("acmgcnp", 0, 0, True)
    assert "acm_conv_agg_fwd" not in run("acmgcnp", 0, 0, False, f_in=64, f_out=2)
    tune(agg_first=0)
    assert run("acmgcnp", 0, 0, False) == {"acm_gemm", "acm_conv_fwd", "acm_conv_bwd_spmm"}


@pytest.mark.parametrize("ln", [False, True])
@pytest.mark.parametrize("f_in,f_out", [(7, 64), (3, 24), (16, 40)])
def test_aggregate_first_with_structure_equals_literal_and_oracle(f_in, f_out, ln, monkeypatch, tune):
    """ABI v5: the 4-channel aggregate-first path (pre_S = deg * (A_low S) - S, dS = A_low^T (D G_S) - G_S)
    against the literal 3F-wide path and the oracle, forward and every parameter gradient."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    import acm_oracle as oracle
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GraphConvolution
    low, high, un, _ = graph_tensors("geometric")
    n = low.shape[0]
    torch.manual_seed(5)
    layer = GraphConvolution(f_in, f_out, n, "acmgcnp", variant=0, structure_info=1, attn_layernorm=ln)
    x = torch.randn(n, f_in)
    go = torch.randn(n, f_out)
    mask = (torch.rand(n, f_out) > 0.3).float() / 0.7

    def run(agg):
        tune(agg_first=int(bool(agg)))
        layer.zero_grad()
        out = layer(x, low, high, un, post_relu=True, post_scale=mask)
        out.backward(go)
        return out.detach().clone(), {k: v.grad.clone() for k, v in layer.named_parameters() if v.grad is not None}, \
            layer.att_struc_vec_low.detach().clone() if hasattr(layer, "att_struc_vec_low") else None

    out_a, g_a, att_a = run(True)
    out_l, g_l, att_l = run(False)
    _close(out_a, out_l.numpy(), "out", **FWD)
    assert set(g_a) == set(g_l) and "struc_low" in g_a
    for k in g_l:
        _close(g_a[k], g_l[k].numpy(), k)
    params = {k: v.detach().clone().double().requires_grad_(True) for k, v in layer.named_parameters()}
    ref = oracle.layer_forward(params, x.double(), low.double(), high.double(), un.double(), model_type="acmgcnp",
                               variant=0, structure_info=1, attn_layernorm=ln)
    ref = torch.relu(ref) * mask.double()
    ref.backward(go.double())
    _close(out_a, ref.detach().float().numpy(), "out vs oracle", **FWD)
    for k in g_a:
        _close(g_a[k], params[k].grad.float().numpy(), k + " vs oracle")


def test_operator_cache_and_filter_verification(monkeypatch):
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import graph
    low, high, un, _ = graph_tensors("geometric")
    a = graph.operators_for(low, high, un)
    assert graph.operators_for(low, high, un) is a                     # cached by storage identity
    assert a.deg is not None and a.low.n_rows == low.shape[0]
    # filters that are not (A_low, I - A_low[, D A_low - I]) are not an error: they select the general
    # two-operator path (exercised in test_general_operator_pair_host_path)
    bad_high = (high * 2.0).coalesce()
    g1 = graph.operators_for(low, bad_high, None)
    assert g1.general and g1.high is not None and not a.general
    bad_un = (un * 3.0).coalesce()
    g2 = graph.operators_for(low, high, bad_un)
    assert g2.general and g2.un is not None


def test_operator_cache_rejects_stale_and_modified_sources(monkeypatch):
    """A cache hit needs the tensors the entry was built from to be alive and unmodified: a freed adjacency whose
    address the allocator hands to a new one of the same shape / nnz, or an in-place re-normalisation, must rebuild."""
    import gc
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import graph
    low, high, un, _ = graph_tensors("geometric")
    a = graph.operators_for(low, high, None)
    assert graph.operators_for(low, high, None) is a
    assert graph.operators_for(low.detach(), high, None) is a              # another wrapper of the same live storage
    low._values().mul_(1.0)                                                # in-place write: version counter moves
    b = graph.operators_for(low, high, None)
    assert b is not a and graph.operators_for(low, high, None) is b
    # a dead source: simulate address reuse by keeping the key and killing the tensor the entry refers to
    key = next(iter(graph._CACHE))
    ops, states = graph._CACHE[key]
    tmp = low.clone()
    graph._CACHE[key] = (ops, (graph._source_state(tmp),) + tuple(states[1:]))
    del tmp
    gc.collect()
    c = graph.operators_for(low, high, None)
    assert c is not b
    assert all(st is None or st[0]() is not None for _, sts in graph._CACHE.values() for st in sts)


def test_captured_train_step_starts_from_the_eager_state(monkeypatch):
    """TrainStep's capture warm-up runs real optimizer steps; they must leave no trace (parameters, Adam moments and
    step counts, dropout counter).  Exercised on the host with the capture itself stubbed out."""
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, FusedAdamW, train as T
    from acm_gnn_amd import functional as AF
    low, high, un, g = graph_tensors("geometric")
    n = low.shape[0]
    torch.manual_seed(0)
    model = GCN(12, 16, 3, 2, n, 0.5, "acmgcnpp", 0, init_layers_X=2)      # with BatchNorm buffers in mlpX
    opt = FusedAdamW(model.parameters(), lr=0.05)
    x, y = torch.randn(n, 12), torch.randint(0, 3, (n,))
    w = T.row_weights(torch.arange(0, n, 2), n)
    step = T.TrainStep(model, opt, x, low, y, w, high, None, use_graph=False, fused_dropout=True)
    step()                                                                  # one real step: non-trivial Adam state
    before = {k: v.clone() for k, v in model.state_dict().items()}
    opt_before = {id(p): {k: v.clone() for k, v in st.items()} for p, st in opt.state.items()}
    drop_before = int(model.dropout_state.step.item())
    snap = step._snapshot()
    for _ in range(3):
        step._eager()
    assert any(not torch.equal(before[k], v) for k, v in model.state_dict().items())
    step._restore(snap)
    for k, v in model.state_dict().items():
        assert torch.equal(before[k], v), k
    for p, st in opt.state.items():
        for k, v in st.items():
            assert torch.equal(opt_before[id(p)][k], v), k
    assert int(model.dropout_state.step.item()) == drop_before
    # fresh optimizer: state created by the warm-up is zeroed in place
    opt2 = FusedAdamW(model.parameters(), lr=0.05)
    step2 = T.TrainStep(model, opt2, x, low, y, w, high, None, use_graph=False, fused_dropout=True)
    snap2 = step2._snapshot()
    step2._eager()
    step2._restore(snap2)
    assert all(float(v.abs().sum()) == 0.0 for st in opt2.state.values() for v in st.values())


def _model_grads(model):
    return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}


def test_acmgcnpp_residual_kernel_dense_and_csr_features(monkeypatch):
    """The ACM-GCN++ residual branch on the library's kernels (acm_linear_fwd / acm_spmm_v + acm_bias_act /
    acm_bias_act_bwd): dense features against the oracle (models.py:26-27,55-56,73), CSR features against dense."""
    fake_lib.install(monkeypatch)
    from oracle import acm_oracle as oracle
    from acm_gnn_amd import GCN, SparseFeatures
    low, high, un, g = graph_tensors("geometric")
    n = low.shape[0]
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(n, 12, generator=gen) * (torch.rand(n, 12, generator=gen) < 0.3)
    y = torch.randint(0, 3, (n,), generator=gen)
    idx = torch.arange(0, n, 2)
    outs = {}
    for kind in ("dense", "csr"):
        torch.manual_seed(1)
        model = GCN(12, 16, 3, 2, n, 0.0, "acmgcnpp", 1, variant=True, attn_layernorm=True)
        xin = SparseFeatures.from_torch(x) if kind == "csr" else x
        out = model(xin, low, high, un)
        F.nll_loss(F.log_softmax(out, 1)[idx], y[idx]).backward()
        outs[kind] = (out.detach(), _model_grads(model), model)
    _close(outs["csr"][0], outs["dense"][0].numpy(), "csr vs dense logits", **FWD)
    for k, v in outs["dense"][1].items():
        _close(outs["csr"][1][k], v.numpy(), "csr vs dense " + k)
    model = outs["dense"][2]
    params = {k: v.detach().clone().double().requires_grad_(True) for k, v in model.named_parameters()
              if k not in ("fea_param", "xX_param")}
    ref = oracle.gcn_forward(params, x.double(), low.double(), high.double(), un.double(), model_type="acmgcnpp",
                             variant=True, structure_info=1, attn_layernorm=True)
    oracle.nll_loss_on(ref, y, idx).backward()
    _close(outs["dense"][0], ref.detach().float().numpy(), "logits vs oracle", **FWD)
    for k in ("mlpX.lins.0.weight", "mlpX.lins.0.bias", "gcns.0.weight_low", "gcns.1.weight_mlp"):
        _close(outs["dense"][1][k], params[k].grad.float().numpy(), k + " vs oracle")


@pytest.mark.parametrize("variant,nlayers", [(False, 2), (True, 3)])
def test_snowball_model_matches_the_reference_wiring(variant, nlayers, monkeypatch):
    """acmsnowball (ACM-Geometric/models.py:38-39,57-64 with the missing nnodes supplied, quirk Q2) against the oracle's
    literal restatement of that forward."""
    fake_lib.install(monkeypatch)
    from oracle import acm_oracle as oracle
    from acm_gnn_amd import GCN
    low, high, un, g = graph_tensors("geometric")
    n = low.shape[0]
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(n, 9, generator=gen)
    y = torch.randint(0, 4, (n,), generator=gen)
    idx = torch.arange(1, n, 3)
    torch.manual_seed(2)
    model = GCN(9, 8, 4, nlayers, n, 0.0, "acmsnowball", 0, variant=variant)
    assert [m.in_features for m in model.gcns] == [9 + 8 * k for k in range(nlayers + 1)]
    out = model(x, low, high, un)
    F.nll_loss(F.log_softmax(out, 1)[idx], y[idx]).backward()
    params = {k: v.detach().clone().double().requires_grad_(True) for k, v in model.named_parameters()
              if k not in ("fea_param", "xX_param")}
    ref = oracle.snowball_forward(params, x.double(), low.double(), high.double(), nlayers=nlayers, variant=variant)
    oracle.nll_loss_on(ref, y, idx).backward()
    _close(out, ref.detach().float().numpy(), "snowball logits", **FWD)
    for k, p in model.named_parameters():
        if k in params and params[k].grad is not None:
            assert p.grad is not None, k
            _close(p.grad, params[k].grad.float().numpy(), k)


@pytest.mark.parametrize("f_in,s_info", [(7, 0), (3, 0), (7, 1)])
def test_acmii_recompute_host_path_equals_literal(f_in, s_info, monkeypatch, tune):
    """Host plumbing of the ACMII recompute-on-gather route (functional.AcmConvFunction -> acm_conv_acmii_fwd, then the
    literal backward on the tensors that call saved) against the literal route and the oracle."""
    fake_lib.install(monkeypatch)
    from oracle import acm_oracle as oracle
    from acm_gnn_amd import GraphConvolution
    from acm_gnn_amd.graph import clear_cache
    low, high, un, g = graph_tensors("geometric")
    n = low.shape[0]
    gen = torch.Generator().manual_seed(11)
    x, go = torch.randn(n, f_in, generator=gen), torch.randn(n, 64, generator=gen)
    res = {}
    for mode in ("1", "0"):
        tune(acmii_recompute=int(mode))
        clear_cache()
        torch.manual_seed(3)
        layer = GraphConvolution(f_in, 64, n, "acmgcnp", variant=True, structure_info=s_info, attn_layernorm=True)
        out = layer(x, low, high, un if s_info else None)
        out.backward(go)
        res[mode] = (out.detach(), _model_grads(layer), layer)
    _close(res["1"][0], res["0"][0].numpy(), "recompute vs literal", **FWD)
    for k, v in res["0"][1].items():
        _close(res["1"][1][k], v.numpy(), "recompute vs literal " + k)
    params = {k: v.detach().clone().double().requires_grad_(True) for k, v in res["1"][2].named_parameters()}
    ref = oracle.layer_forward(params, x.double(), low.double(), high.double(), un.double() if s_info else None,
                               model_type="acmgcnp", variant=True, structure_info=s_info, attn_layernorm=True)
    ref.backward(go.double())
    _close(res["1"][0], ref.detach().float().numpy(), "recompute vs oracle", **FWD)
    for k, v in res["1"][1].items():
        _close(v, params[k].grad.float().numpy(), k + " vs oracle")


@pytest.mark.parametrize("model_type,variant,s,sparse_x", [("acmgcnp", 0, 1, False), ("acmgcnp", 1, 0, False),
                                                           ("acmgcnpp", 0, 0, True), ("acmsnowball", 1, 0, False)])
def test_in_operator_relabelling_is_transparent(model_type, variant, s, sparse_x, monkeypatch, tune):
    """graph.relabel_by_degree: the operators live in a degree-sorted numbering, layers / GCN / TrainStep translate rows
    at the boundary -- logits, attention weights and every gradient (struc_low included) equal the un-relabelled run."""
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, SparseFeatures, graph, train as T
    low, high, un, g = graph_tensors("geometric")
    n = low.shape[0]
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(n, 9, generator=gen) * (torch.rand(n, 9, generator=gen) < 0.5)
    y = torch.randint(0, 3, (n,), generator=gen)
    w = T.row_weights(torch.arange(0, n, 2), n)
    res = {}
    for mode in ("0", "1"):
        tune(relabel={"auto": -1}.get(mode, None) if mode == "auto" else int(mode))
        graph.clear_cache()
        ops = graph.operators_for(low, high, un if s else None)
        assert (ops.perm is not None) == (mode == "1")
        if mode == "1":
            d = np.diff(graph.explicit_arrays(ops)[0].numpy())        # (a raw self-loop is listed twice in the pattern-only form)
            assert np.all(np.diff(d) <= 0) and sorted(ops.perm.tolist()) == list(range(n))
        torch.manual_seed(5)
        model = GCN(9, 16, 3, 2, n, 0.0, model_type, s, variant=bool(variant), attn_layernorm=True)
        xin = SparseFeatures.from_torch(x) if sparse_x else x
        out = model(xin, low, high, un)                                 # model-level translation (tensors, like the reference)
        T.F.nll_loss(T.F.log_softmax(out, 1)[::2], y[::2]).backward()
        att = model.gcns[0].att_low.detach().clone()
        grads = _model_grads(model)
        model.zero_grad(set_to_none=True)
        out_l = model.gcns[0](xin, low, high, un if s else None)        # layer-level translation (the drop-in route)
        step = T.TrainStep(model, torch.optim.SGD(model.parameters(), lr=0.0), xin, low, y, w, high, un, fused_dropout=False)
        assert step._permuted == (mode == "1")
        loss = step()
        res[mode] = (out.detach(), grads, out_l.detach(), float(loss), _model_grads(model), att)
    _close(res["1"][0], res["0"][0].numpy(), "logits", **FWD)
    _close(res["1"][2], res["0"][2].numpy(), "layer output", **FWD)
    _close(res["1"][5], res["0"][5].numpy(), "attention weights", **FWD)
    assert abs(res["1"][3] - res["0"][3]) < 1e-6
    for k, v in res["0"][1].items():
        _close(res["1"][1][k], v.numpy(), k)
    for k, v in res["0"][4].items():
        _close(res["1"][4][k], v.numpy(), "train step " + k)


def test_structure_info_with_acmgcn_is_an_error(monkeypatch):
    """Reference quirk Q3: att_vec is 4x4 but acmgcn mixes 3 channels -> RuntimeError there too."""
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GraphConvolution
    low, high, un, _ = graph_tensors("geometric")
    layer = GraphConvolution(12, 16, low.shape[0], "acmgcn", structure_info=1)
    with pytest.raises(RuntimeError, match="att_vec"):
        layer(torch.randn(low.shape[0], 12), low, high, un)


def test_no_cpu_fallback_in_product_path():
    """Without the test double a CPU tensor is refused loudly."""
    from acm_gnn_amd import GraphConvolution, functional
    from acm_gnn_amd.graph import clear_cache
    clear_cache()
    low, high, _, _ = graph_tensors("geometric")
    layer = GraphConvolution(12, 16, low.shape[0], "acmgcn").cpu()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layer(torch.randn(low.shape[0], 12), low, high, None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        functional.gemm(torch.randn(4, 4), torch.randn(4, 4))


def test_trivial_branches(monkeypatch):
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GraphConvolution
    low, high, _, g = graph_tensors("pytorch")          # dense adj_low, as torch.mm requires
    x = torch.randn(low.shape[0], 12)
    mlp = GraphConvolution(12, 5, low.shape[0], "mlp")
    torch.testing.assert_close(mlp(x, low, high, None), x @ mlp.weight_mlp, rtol=1e-5, atol=1e-5)
    gcn = GraphConvolution(12, 5, low.shape[0], "gcn")
    torch.testing.assert_close(gcn(x, low, high, None), low @ (x @ gcn.weight_low), rtol=1e-5, atol=1e-5)


def test_masked_nll_equals_log_softmax_nll(monkeypatch):
    import torch.nn.functional as F
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import functional as AF, train as T
    g = torch.Generator().manual_seed(0)
    z = torch.randn(50, 5, generator=g, requires_grad=True)
    y = torch.randint(0, 5, (50,), generator=g)
    idx = torch.randperm(50, generator=g)[:20]
    loss = AF.masked_nll(z, y, T.row_weights(idx, 50))
    (loss * 2.0).backward()
    zr = z.detach().clone().requires_grad_(True)
    ref = F.nll_loss(F.log_softmax(zr, 1)[idx], y[idx])
    (ref * 2.0).backward()
    torch.testing.assert_close(loss.detach(), ref.detach(), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(z.grad, zr.grad, rtol=1e-5, atol=1e-7)


def test_fit_selection_rules(monkeypatch):
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, train as T
    low, high, un, _ = graph_tensors("geometric")
    n = low.shape[0]
    torch.manual_seed(0)
    x = torch.randn(n, 7)
    y = (x[:, 0] > 0).long()
    perm = torch.randperm(n)
    tr, va, te = perm[:48], perm[48:72], perm[72:]
    for rule, es in (("max_val_acc", 0), ("min_val_loss", 3)):
        model = GCN(7, 16, 2, 2, n, 0.0, "acmgcnp", 0, variant=False)
        opt = torch.optim.Adam(model.parameters(), lr=0.05)
        acc, hist = T.fit(model, opt, x, low, y, tr, va, te, epochs=12, rule=rule, early_stopping=es, adj_high=high)
        assert 0.0 <= acc <= 1.0 and 1 <= len(hist) <= 12
        assert hist[-1][0] < hist[0][0]                       # the loss goes down
        if rule == "max_val_acc":
            best = max(range(len(hist)), key=lambda i: (hist[i][2], -i))
            assert acc == hist[best][3]


@pytest.mark.parametrize("model_type,variant,s", [("acmgcn", 0, 0), ("acmgcnp", 1, 1)])
def test_sparse_feature_projection_equals_dense(model_type, variant, s, monkeypatch):
    """CSR X through acm_spmm_v (forward) and its transpose with permuted values (dW) == dense X."""
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, SparseFeatures
    low, high, un, _ = graph_tensors("geometric")
    n = low.shape[0]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, 40, generator=g) * (torch.rand(n, 40, generator=g) < 0.1)
    y = torch.randint(0, 3, (n,), generator=g)
    res = []
    for sparse in (False, True):
        torch.manual_seed(0)
        model = GCN(40, 16, 3, 2, n, 0.0, model_type, s, variant=bool(variant), attn_layernorm=True)
        inp = SparseFeatures.from_torch(x) if sparse else x
        out = model(inp, low, high, un if s else None)
        torch.nn.functional.cross_entropy(out, y).backward()
        res.append((out.detach(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    torch.testing.assert_close(res[1][0], res[0][0], rtol=1e-5, atol=1e-6)
    assert res[0][1].keys() == res[1][1].keys()
    for k in res[0][1]:
        torch.testing.assert_close(res[1][1][k], res[0][1][k], rtol=1e-4, atol=1e-6, msg=lambda m, k=k: f"{k}: {m}")
    # torch sparse tensors are accepted directly
    torch.manual_seed(0)
    model = GCN(40, 16, 3, 2, n, 0.0, model_type, s, variant=bool(variant), attn_layernorm=True)
    torch.testing.assert_close(model(x.to_sparse(), low, high, un if s else None).detach(), res[0][0], rtol=1e-5, atol=1e-6)


def _general_case(model_type, variant, s, ln, dev, khop):
    """Filters that are NOT (A_low, I - A_low): k-hop low-pass with an un-powered high-pass (the reference's
    ACM-SGC hops > 1, ACM-Pytorch/utils.py:631-637) and a re-weighted raw adjacency."""
    import scipy.sparse as sp
    from oracle import acm_oracle as O
    from acm_gnn_amd import GraphConvolution
    from acm_gnn_amd.graph import clear_cache
    clear_cache()
    low, high, un, g = graph_tensors("geometric")
    n = low.shape[0]
    low_k = low.to_dense()
    for _ in range(khop - 1):
        low_k = low_k @ low.to_dense()
    low_k = low_k.to_sparse()
    un_w = (un * 0.5).coalesce()
    torch.manual_seed(2)
    layer = GraphConvolution(10, 16, n, model_type, variant=variant, structure_info=s, attn_layernorm=ln)
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in layer.named_parameters()}
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(n, 10, generator=gen)
    gout = torch.randn(n, 16, generator=gen)
    xr = x.clone().requires_grad_(True)
    ref = O.layer_forward(params, xr, low_k, high, un_w if s else None, model_type=model_type, variant=variant,
                          structure_info=s, attn_layernorm=ln)
    ref.backward(gout)
    layer = layer.to(dev)
    xd = x.to(dev).requires_grad_(True)
    out = layer(xd, low_k.to(dev), high.to(dev), un_w.to(dev) if s else None)
    out.backward(gout.to(dev))
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-3, atol=5e-5)
    for k, p in layer.named_parameters():
        if params[k].grad is None:
            assert p.grad is None, k
        else:
            tol = 1e-4 * max(1.0, float(params[k].grad.abs().max()))
            assert float((p.grad.cpu() - params[k].grad).abs().max()) < tol, k


def _khop_chain_case(dev, hops, f_out, implicit, monkeypatch):
    """ACM-SGC with ``ops.hops = k`` (chain of 1-hop products, nothing materialised) == the oracle fed the dense
    A_low^k the reference builds (ACM-Pytorch/utils.py:631-637), forward and every gradient."""
    from oracle import acm_oracle as O
    from acm_gnn_amd import GraphConvolution
    from acm_gnn_amd.graph import clear_cache, operators_for
    tune(implicit=int(bool(implicit)))
    clear_cache()
    low, high, un, g = graph_tensors("geometric")
    n = low.shape[0]
    low_k = O.khop_low(low.to_dense(), hops).to_sparse()
    torch.manual_seed(2)
    layer = GraphConvolution(10, f_out, n, "acmsgc")
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in layer.named_parameters()}
    gen = torch.Generator().manual_seed(3)
    x, gout = torch.randn(n, 10, generator=gen), torch.randn(n, f_out, generator=gen)
    xr = x.clone().requires_grad_(True)
    ref = O.layer_forward(params, xr, low_k, high, None, model_type="acmsgc", variant=0, structure_info=0,
                          attn_layernorm=False)
    ref.backward(gout)
    layer = layer.to(dev)
    ops = operators_for(low.to(dev), high.to(dev), None)
    assert ops.implicit == implicit and not ops.general
    ops.hops = hops
    xd = x.to(dev).requires_grad_(True)
    out = layer(xd, ops)
    out.backward(gout.to(dev))
    ops.hops = 1
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-3, atol=5e-5)
    for k, p in layer.named_parameters():
        if params[k].grad is None:
            assert p.grad is None, k
        else:
            tol = 1e-4 * max(1.0, float(params[k].grad.abs().max()))
            assert float((p.grad.cpu() - params[k].grad).abs().max()) < tol, k


@pytest.mark.parametrize("hops,f_out,implicit", [(3, 5, True), (2, 16, True), (3, 5, False)])
def test_acmsgc_khop_chain_host_path(hops, f_out, implicit, monkeypatch):
    fake_lib.install(monkeypatch)
    _khop_chain_case("cpu", hops, f_out, implicit, monkeypatch)


def test_khop_is_refused_where_it_is_not_defined(monkeypatch):
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GraphConvolution
    from acm_gnn_amd.graph import operators_for
    low, high, un, _ = graph_tensors("geometric")
    ops = operators_for(low, high, None)
    ops.hops = 2
    try:
        with pytest.raises(NotImplementedError, match="ACM-SGC chain"):
            GraphConvolution(10, 16, low.shape[0], "acmgcn")(torch.randn(low.shape[0], 10), ops)
    finally:
        ops.hops = 1


@pytest.mark.parametrize("model_type,variant,s,ln,khop", [("acmsgc", 0, 0, False, 3), ("acmgcnp", 1, 1, True, 2),
                                                          ("acmgcn", 0, 0, False, 2)])
def test_general_operator_pair_host_path(model_type, variant, s, ln, khop, monkeypatch):
    fake = fake_lib.install(monkeypatch)
    calls = []
    orig = fake.acm_spmm_v
    monkeypatch.setattr(fake, "acm_spmm_v", lambda *a: (calls.append(1), orig(*a))[1])
    _general_case(model_type, variant, s, ln, "cpu", khop)
    assert len(calls) >= 4            # separate products per channel, forward and transposed


def test_bf16_gather_option_host_path(monkeypatch):
    """gather_dtype='bf16' routes through acm_cast_bf16 and differs from fp32 only at the bf16 rounding level."""
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GraphConvolution
    low, high, un, _ = graph_tensors("geometric")
    n = low.shape[0]
    x = torch.randn(n, 20, generator=torch.Generator().manual_seed(0))
    outs = {}
    for dt in ("fp32", "bf16"):
        torch.manual_seed(1)
        layer = GraphConvolution(20, 64, n, "acmgcnp", variant=1, structure_info=1, attn_layernorm=True, gather_dtype=dt)
        outs[dt] = layer(x, low, high, un).detach()
    err = (outs["bf16"] - outs["fp32"]).abs().max().item()
    assert 0 < err < 2e-2 * max(1.0, outs["fp32"].abs().max().item())
    with pytest.raises(ValueError):
        GraphConvolution(20, 64, n, "acmgcn", gather_dtype="fp8")._config()


@pytest.mark.parametrize("n_cls", [1, 2, 3])
def test_next_layer_projection_rides_the_row_local_stage(n_cls, monkeypatch, tune):
    """models.GCN names the output layer while it calls the hidden one; where the hidden layer's row-local stage runs as
    its own kernel (P = A_low X given: here the second evaluation pass over an unmodified input; in training the input
    pipeline) it carries the output layer's narrow projection (acm_conv_agg_fwd_t.next_*) and the output layer does not
    launch acm_proj_fwd -- for F' <= 2 only, with the same logits; the fused gather kernel never carries it."""
    fake = fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN
    low, high, un, _ = graph_tensors("geometric")
    n = low.shape[0]
    calls = []
    for name in ("acm_proj_fwd_at", "acm_conv_agg_fwd"):
        orig = getattr(fake, name)
        if name == "acm_conv_agg_fwd":
            monkeypatch.setattr(fake, name, (lambda o: lambda h, pp, *a: (calls.append(("agg", int(pp._obj.agg_given), int(pp._obj.next_f))), o(h, pp, *a))[1])(orig))
        else:
            monkeypatch.setattr(fake, name, (lambda o, nm: lambda *a: (calls.append(nm), o(*a))[1])(orig, name))
    x = torch.randn(n, 7, generator=torch.Generator().manual_seed(1))
    torch.manual_seed(5)
    model = GCN(7, 64, n_cls, 2, n, 0.3, "acmgcnp", 0, variant=0, attn_layernorm=True)
    model.eval()
    with torch.no_grad():
        o1 = model(x, low, high)
        first = list(calls)
        calls.clear()
        o2 = model(x, low, high)
        second = list(calls)
        tune(rows16=6)                                 # without the sixteen-rows-per-wave stage nothing carries it
        calls.clear()
        o3 = model(x, low, high)
        third = list(calls)
    assert first[0] == ("agg", 0, 0) and "acm_proj_fwd_at" in first
    assert second[0][:2] == ("agg", 1) and (second[0][2] == n_cls) == (n_cls <= 2)
    assert ("acm_proj_fwd_at" not in second) == (n_cls <= 2)
    assert third[0] == ("agg", 1, 0) and "acm_proj_fwd_at" in third
    for o in (o2, o3):
        np.testing.assert_allclose(o.numpy(), o1.numpy(), rtol=1e-5, atol=1e-5 * max(1.0, float(o1.abs().max())))


def test_eval_step_equals_evaluate(monkeypatch):
    """train.EvalStep (accuracies and validation NLL reduced on the device, one host copy) against train.evaluate +
    F.nll_loss, the code path fit() uses without a captured graph."""
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, train as T
    low, high, un, _ = graph_tensors("geometric")
    n = low.shape[0]
    torch.manual_seed(3)
    x = torch.randn(n, 7)
    y = torch.randint(0, 3, (n,))
    perm = torch.randperm(n)
    sets = (perm[:40], perm[40:70], perm[70:])
    model = GCN(7, 16, 3, 2, n, 0.5, "acmgcnp", 0, variant=False)
    ev = T.EvalStep(model, x, low, y, sets, adj_high=high, loss_set=1)
    out, accs, val_loss = ev()
    ref_out, ref_accs = T.evaluate(model, x, low, y, sets, adj_high=high)
    # (the second pass reuses P = A_low X of the first: the numpy double computes it in float64 the first time)
    np.testing.assert_allclose(out.numpy(), ref_out.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(accs, ref_accs, rtol=1e-5, atol=1e-6)
    ref_loss = float(F.nll_loss(F.log_softmax(ref_out, 1)[sets[1]], y[sets[1]]))
    assert abs(val_loss - ref_loss) < 1e-5 * max(1.0, abs(ref_loss))
    assert not model.training


def test_eval_passes_reuse_the_aggregated_input(monkeypatch):
    """An eval-mode, no-grad pass over the same unmodified input reuses P = A_low X of the previous pass
    (acm_conv_agg_fwd_t.agg_given): no gather the second time, same logits; an in-place edit of the input or a
    training-mode call takes the gather again."""
    fake = fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, functional as AF
    low, high, un, _ = graph_tensors("geometric")
    n = low.shape[0]
    given = []
    orig = fake.acm_conv_agg_fwd
    monkeypatch.setattr(fake, "acm_conv_agg_fwd", lambda h, pp, *a: (given.append(int(pp._obj.agg_given)), orig(h, pp, *a))[1])
    torch.manual_seed(2)
    model = GCN(7, 64, 2, 2, n, 0.4, "acmgcnp", 0, variant=False)
    x = torch.randn(n, 7)
    model.eval()
    with torch.no_grad():
        o1 = model(x, low, high)
        o2 = model(x, low, high)
        assert given == [0, 1] and torch.allclose(o1, o2, rtol=1e-5, atol=1e-6)
        x.mul_(1.5)                                    # in-place edit: the version counter moves
        o3 = model(x, low, high)
        assert given == [0, 1, 0] and not torch.equal(o1, o3)
        o4 = model(x.clone(), low, high)               # another tensor
        assert given[-1] == 0 and torch.allclose(o3, o4, rtol=1e-5, atol=1e-6)
    model(x, low, high)                                # autograd on: never from the cache
    model.train()
    with torch.no_grad():
        model(x, low, high)
    assert given[-2:] == [0, 0]
    for _l in model.gcns: _l.eval_agg_cache = False
    model.eval()
    with torch.no_grad():
        model(x, low, high), model(x, low, high)
    assert given[-2:] == [0, 0]


def _dense_graph_ops(n=160, avg=20, seed=5):
    """FilterOperators of a random graph dense enough for the input pipeline's regime (12 < nnz / n <= 160)."""
    import scipy.sparse as sp
    from acm_gnn_amd import data as D
    from acm_gnn_amd.distributed import make_sharded_operators
    rng = np.random.default_rng(seed)
    m = n * avg // 2
    r, c = rng.integers(0, n, m), rng.integers(0, n, m)
    adj = sp.csr_matrix((np.ones(m, np.float32), (r, c)), shape=(n, n))
    adj = ((adj + adj.T) > 0).astype(np.float32).tocsr()
    adj.setdiag(0)
    adj.eliminate_zeros()
    low, deg = D.build_filters(adj)
    return make_sharded_operators(low, deg, torch.device("cpu")), n


def test_input_pipeline_with_the_residual_branch_of_acmgcnpp(monkeypatch, tune):
    """ACM-GCN++ under the input pipeline: the residual Linear (models.py:55-56) reads the pipeline's table in its forward
    and -- because the first layer's forward refills that table with the NEXT step's dropped input -- this step's rows from
    the pipeline's saved copy in its backward.  Losses and parameters equal the plain step's."""
    fake_lib.install(monkeypatch)
    tune(pipeline=32)
    from acm_gnn_amd import GCN, FusedAdamW, functional as AF, train as T
    ops, n = _dense_graph_ops()
    x, y = torch.randn(n, 7, generator=torch.Generator().manual_seed(1)), torch.randint(0, 2, (n,), generator=torch.Generator().manual_seed(2))
    w = T.row_weights(torch.arange(0, n, 2), n)

    def run(pipeline):
        torch.manual_seed(0)
        model = GCN(7, 64, 2, 2, n, 0.3, "acmgcnpp", 0, variant=False, attn_layernorm=True)
        model.dropout_state = AF.DropoutState(torch.device("cpu"), seed=77)
        opt = FusedAdamW(model.parameters(), lr=0.02)
        step = T.TrainStep(model, opt, x, ops, y, w, use_graph=False, fused_dropout=True, pipeline_input=pipeline)
        losses = [float(step()) for _ in range(5)]
        return step, losses, {k: v.clone() for k, v in model.state_dict().items()}

    step_a, loss_a, sd_a = run(False)
    step_b, loss_b, sd_b = run(None)
    assert step_a.pipe is None and step_b.pipe is not None and step_b.pipe.primed
    np.testing.assert_allclose(loss_b, loss_a, rtol=1e-5, atol=1e-6)
    for k in sd_a:
        np.testing.assert_allclose(sd_b[k].numpy(), sd_a[k].numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


def test_input_pipeline_equals_plain_train_step(monkeypatch, tune):
    """train.TrainStep with the input pipeline (functional.InputPipeline: the first layer's P = A_low dropout(x) of step
    t + 1 gathered inside the layer's backward of step t, acm_conv_agg_bwd_t.next_agg / acm_dropout_t.step_offset)
    against the plain step: same losses and parameters; the forward runs with agg_given, every backward carries the
    gather, prime() runs once."""
    fake = fake_lib.install(monkeypatch)
    tune(pipeline=32)
    from acm_gnn_amd import GCN, FusedAdamW, functional as AF, train as T
    ops, n = _dense_graph_ops()
    x, y = torch.randn(n, 7, generator=torch.Generator().manual_seed(1)), torch.randint(0, 2, (n,), generator=torch.Generator().manual_seed(2))
    w = T.row_weights(torch.arange(0, n, 2), n)
    calls = {"given": [], "carried": [], "spmm": 0, "refill": [], "dropout": 0}
    fwd, bwd, spmm_ex, drop = fake.acm_conv_agg_fwd, fake.acm_conv_agg_bwd, fake.acm_spmm_ex, fake.acm_dropout
    monkeypatch.setattr(fake, "acm_conv_agg_fwd", lambda h, pp, *a: (calls["given"].append(int(pp._obj.agg_given)),
                                                                    calls["refill"].append(bool(pp._obj.next_x)), fwd(h, pp, *a))[2])
    monkeypatch.setattr(fake, "acm_dropout", lambda *a: (calls.__setitem__("dropout", calls["dropout"] + 1), drop(*a))[1])
    monkeypatch.setattr(fake, "acm_conv_agg_bwd", lambda nn, qq, *a: (calls["carried"].append(bool(qq._obj.next_agg)), bwd(nn, qq, *a))[1])

    def run(pipeline):
        torch.manual_seed(0)
        model = GCN(7, 64, 2, 2, n, 0.3, "acmgcnp", 0, variant=False, attn_layernorm=True)
        model.dropout_state = AF.DropoutState(torch.device("cpu"), seed=77)
        opt = FusedAdamW(model.parameters(), lr=0.02)
        step = T.TrainStep(model, opt, x, ops, y, w, use_graph=False, fused_dropout=True, pipeline_input=pipeline)
        losses = [float(step()) for _ in range(5)]
        return step, losses, {k: v.clone() for k, v in model.state_dict().items()}

    step_a, loss_a, sd_a = run(False)
    assert step_a.pipe is None and calls["given"] == [0] * 5 and calls["carried"] == [False] * 5
    calls["given"].clear(), calls["carried"].clear()
    step_b, loss_b, sd_b = run(None)
    assert step_b.pipe is not None and ops.low.stream_waves == 40          # four gather waves per 16 rows
    assert calls["given"] == [1] * 5 and calls["carried"] == [True] * 5
    # ... and (ABI 22) refills the table with the next step's dropped input itself: one acm_dropout launch in prime(), none
    # per step (the plain run above: one per step)
    assert calls["refill"][-5:] == [True] * 5 and calls["dropout"] == 5 + 1, calls
    np.testing.assert_allclose(loss_b, loss_a, rtol=1e-5, atol=1e-6)
    for k in sd_a:
        np.testing.assert_allclose(sd_b[k].numpy(), sd_a[k].numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
    # the counter set from outside (TrainStep._restore after a capture's warm-up): the buffers are re-primed, in step again
    snap = step_b._snapshot()
    step_b._eager()
    step_b._restore(snap)
    assert step_b.pipe.stale()
    again = float(step_b())
    step_a._restore(step_a._snapshot())
    np.testing.assert_allclose(again, float(step_a()), rtol=1e-5, atol=1e-6)
    # an in-place edit of the features between eager steps is noticed (version counter) and P recomputed
    x.mul_(0.5)
    assert step_b.pipe.stale()
    np.testing.assert_allclose(float(step_b()), float(step_a()), rtol=1e-4, atol=1e-6)     # (seven fp32 steps apart by now)


def test_input_pipeline_survives_a_redone_step(monkeypatch, tune):
    """TrainStep redoes a step without deferred reductions when a gradient was not adopted; by then the first pass has
    refilled the pipeline's buffers for the NEXT step, so the redo must refill them for this one: same losses as the
    plain step."""
    fake_lib.install(monkeypatch)
    tune(pipeline=32)
    from acm_gnn_amd import GCN, FusedAdamW, functional as AF, train as T
    ops, n = _dense_graph_ops(seed=9)
    x, y = torch.randn(n, 7, generator=torch.Generator().manual_seed(4)), torch.randint(0, 2, (n,), generator=torch.Generator().manual_seed(5))
    w = T.row_weights(torch.arange(0, n, 2), n)

    def run(pipeline, refuse_first):
        torch.manual_seed(0)
        model = GCN(7, 64, 2, 2, n, 0.3, "acmgcnp", 0, variant=False, attn_layernorm=True)
        model.dropout_state = AF.DropoutState(torch.device("cpu"), seed=11)
        step = T.TrainStep(model, FusedAdamW(model.parameters(), lr=0.02), x, ops, y, w, use_graph=False, fused_dropout=True,
                           pipeline_input=pipeline)
        if refuse_first:
            real, calls = AF.DeferredReductions.all_adopted, []
            monkeypatch.setattr(AF.DeferredReductions, "all_adopted",
                                lambda self, t: (calls.append(1), real(self, t) and len(calls) > 1)[1])
        losses = [float(step()) for _ in range(4)]
        if refuse_first:
            monkeypatch.undo()
            fake_lib.install(monkeypatch)
            tune(pipeline=32)
            assert not step._defer
        return step, losses

    _, plain = run(False, False)
    step, redone = run(None, True)
    assert step.pipe is not None
    np.testing.assert_allclose(redone, plain, rtol=1e-5, atol=1e-6)


def test_input_pipeline_notices_steps_made_by_someone_else(monkeypatch, tune):
    """Two TrainSteps on one model (bench.py keeps an eager and a captured one): when the other one has advanced the
    dropout counter, the pipelined step's look-ahead buffers belong to a past step -- DropoutState.host_steps tells."""
    fake_lib.install(monkeypatch)
    tune(pipeline=32)
    from acm_gnn_amd import GCN, FusedAdamW, functional as AF, train as T
    ops, n = _dense_graph_ops(seed=13)
    x, y = torch.randn(n, 7, generator=torch.Generator().manual_seed(6)), torch.randint(0, 2, (n,), generator=torch.Generator().manual_seed(7))
    w = T.row_weights(torch.arange(0, n, 2), n)

    def fresh():
        torch.manual_seed(0)
        model = GCN(7, 64, 2, 2, n, 0.3, "acmgcnp", 0, variant=False, attn_layernorm=True)
        model.dropout_state = AF.DropoutState(torch.device("cpu"), seed=21)
        return model, FusedAdamW(model.parameters(), lr=0.02)

    model, opt = fresh()
    plain = T.TrainStep(model, opt, x, ops, y, w, fused_dropout=True, pipeline_input=False)
    want = [float(plain()) for _ in range(4)]
    model, opt = fresh()
    piped = T.TrainStep(model, opt, x, ops, y, w, fused_dropout=True)
    other = T.TrainStep(model, opt, x, ops, y, w, fused_dropout=True, pipeline_input=False)
    assert piped.pipe is not None and other.pipe is None
    got = [float(piped()), float(piped()), float(other())]
    assert piped.pipe.stale()
    got.append(float(piped()))
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


def _two_step_setups(n, ops):
    """Two different models (ACM-GCN+ with the input pipeline, ACM-GCN++ without) on the same operators."""
    from acm_gnn_amd import GCN, FusedAdamW, functional as AF, train as T
    x = torch.randn(n, 7, generator=torch.Generator().manual_seed(1))
    y = torch.randint(0, 2, (n,), generator=torch.Generator().manual_seed(2))
    w = T.row_weights(torch.arange(0, n, 2), n)

    def make(kind):
        torch.manual_seed(3)
        mt = "acmgcnp" if kind == 0 else "acmgcnpp"
        model = GCN(7, 64, 2, 2, n, 0.3, mt, 0, variant=False, attn_layernorm=True)
        model.dropout_state = AF.DropoutState(torch.device("cpu"), seed=50 + kind)
        return T.TrainStep(model, FusedAdamW(model.parameters(), lr=0.02), x, ops, y, w, use_graph=False, fused_dropout=True,
                           pipeline_input=None if kind == 0 else False)
    return make


def test_two_train_steps_interleaved_equal_running_them_apart(monkeypatch, tune):
    """The per-call context (functional.CallContext: deferral list, loss-tail request, input pipeline, projection
    hand-off) travels with each model call: two TrainSteps of different models stepped alternately -- and with the forward
    of one between the forward and the backward of the other -- give exactly the losses and parameters of the two run
    one after the other."""
    fake_lib.install(monkeypatch)
    tune(pipeline=32)
    from acm_gnn_amd import functional as AF
    ops, n = _dense_graph_ops(seed=5)
    make = _two_step_setups(n, ops)
    apart = []
    for kind in (0, 1):
        step = make(kind)
        apart.append(([float(step()) for _ in range(4)], {k: v.clone() for k, v in step.model.state_dict().items()}))
    a, b = make(0), make(1)
    assert a.pipe is not None and b.pipe is None
    la, lb = [], []
    for _ in range(4):
        la.append(float(a()))
        lb.append(float(b()))
    assert la == apart[0][0] and lb == apart[1][0]
    for step, (_, sd) in ((a, apart[0]), (b, apart[1])):
        for k, v in step.model.state_dict().items():
            assert torch.equal(v, sd[k]), k
    # a forward of model B between the forward and the backward of model A (what a hand-off parked in module state cannot
    # survive): drive the two halves of A's step by hand
    a2, b2 = make(0), make(1)
    ref = make(0)
    for _ in range(2):
        a2.model.train()
        a2.opt.zero_grad(set_to_none=True)
        if a2.pipe.stale():
            a2.pipe.prime()
        call = AF.CallContext(defer=AF.DeferredReductions(), pipe=a2.pipe)
        loss, dz, out = a2._forward_loss(call)
        lb2 = float(b2())                                     # a whole step of the other model in between
        assert a2.pipe.make_next()
        out.backward(dz)
        call.defer.flush()
        a2.opt.step()
        a2._count_advance()
        a2.pipe.end_step()
        assert float(loss) == float(ref()) and np.isfinite(lb2)
    assert AF._ambient().defer is None and AF._ambient().tail is None and AF._ambient().pipe is None


def test_two_train_steps_in_threads_equal_running_them_apart(monkeypatch, tune):
    fake_lib.install(monkeypatch)
    tune(pipeline=32)
    import threading
    ops, n = _dense_graph_ops(seed=6)
    make = _two_step_setups(n, ops)
    apart = []
    for kind in (0, 1):
        step = make(kind)
        apart.append([float(step()) for _ in range(6)])
    steps = [make(0), make(1)]
    got, errs = [None, None], []
    barrier = threading.Barrier(2)

    def work(i):
        try:
            out = []
            for _ in range(6):
                barrier.wait(timeout=60)
                out.append(float(steps[i]()))
            got[i] = out
        except BaseException as e:                      # noqa: BLE001
            errs.append(e)
            barrier.abort()

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert got[0] == apart[0] and got[1] == apart[1]


def test_pipeline_is_not_refilled_when_the_forward_did_not_adopt_it(monkeypatch, tune):
    """make_next() must not overwrite the pipeline's table when the layer went another way (here
    ACM_AGG_FIRST=0: the literal path saves the table itself for dW): the step then equals the plain step."""
    fake_lib.install(monkeypatch)
    tune(pipeline=32)
    from acm_gnn_amd import GCN, FusedAdamW, functional as AF, train as T
    ops, n = _dense_graph_ops(seed=7)
    x, y = torch.randn(n, 7, generator=torch.Generator().manual_seed(1)), torch.randint(0, 2, (n,), generator=torch.Generator().manual_seed(2))
    w = T.row_weights(torch.arange(0, n, 2), n)

    def run(pipeline):
        torch.manual_seed(0)
        model = GCN(7, 64, 2, 2, n, 0.3, "acmgcnp", 0, variant=False, attn_layernorm=True)
        model.dropout_state = AF.DropoutState(torch.device("cpu"), seed=21)
        step = T.TrainStep(model, FusedAdamW(model.parameters(), lr=0.02), x, ops, y, w, use_graph=False, fused_dropout=True,
                           pipeline_input=pipeline)
        if pipeline is None:
            assert step.pipe is not None
        tune(agg_first=0)               # after eligibility was decided: the layer now goes the literal way
        out = [float(step()) for _ in range(3)]
        tune(agg_first=1)
        return out

    np.testing.assert_allclose(run(None), run(False), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("fused_dropout", [False, True], ids=["mask-tensors", "counter-dropout"])
def test_output_layer_projection_backward_rides_the_hidden_layers_backward(monkeypatch, fused_dropout):
    """acm_conv_agg_bwd_t.proj_* (ABI 20): in the two-layer models the output layer's dX = dZ Wcat^T and dW = X^T dZ are
    left to the hidden layer's row-local backward kernel -- models.GCN vouches that the hidden activations feed nothing
    else (CallContext.hidden_private) -- so acm_proj_bwd is not launched and the [n, 64] gradient never exists.  ONLY under
    a deferral list (the contract "every .grad is undefined until the flush": ADVICE r03 -- without one, autograd may
    accumulate the still unwritten dW' into an existing .grad); same gradients as the plain call; a model whose hidden
    tensor is not private (ACM-GCN++: the residual is added in between) keeps the two launches."""
    fake = fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, functional as AF
    ops, n = _dense_graph_ops(seed=3)
    x = torch.randn(n, 7, generator=torch.Generator().manual_seed(1))
    calls = []
    for name in ("acm_proj_bwd", "acm_conv_agg_bwd"):
        orig = getattr(fake, name)
        if name == "acm_conv_agg_bwd":
            monkeypatch.setattr(fake, name, (lambda o: lambda nn, qq, *a: (calls.append("agg_bwd+proj" if qq._obj.proj_dz else "agg_bwd"), o(nn, qq, *a))[1])(orig))
        else:
            monkeypatch.setattr(fake, name, (lambda o: lambda *a: (calls.append("proj_bwd"), o(*a))[1])(orig))

    def run(model_type, lazy):
        calls.clear()
        torch.manual_seed(5)
        model = GCN(7, 64, 2, 2, n, 0.3, model_type, 0, variant=0, attn_layernorm=True)
        model.train()
        if fused_dropout:
            model.fused_dropout, model.dropout_state = True, AF.DropoutState(torch.device("cpu"), seed=9)
        else:
            torch.manual_seed(11)
        if lazy:                                       # the loop owns the step: defer, backward, flush, then read
            with AF.deferred_reductions() as pending:
                out = model(x, ops)
                out.square().sum().backward()
                assert pending.all_adopted([p.grad for p in model.parameters()])
        else:
            out = model(x, ops)
            out.square().sum().backward()
        return out.detach(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, list(calls)

    out_a, g_a, calls_a = run("acmgcnp", True)
    out_b, g_b, calls_b = run("acmgcnp", False)
    # (with F.dropout mask tensors the hidden layer's post-op is a post_scale tensor: its backward regenerates the mixed row
    #  instead of reading `out`, the kernel that carries the projection does not cover that form, and the hand-off is
    #  materialised by acm_proj_bwd inside the hidden layer's backward -- same launches as without it)
    assert calls_a == (["agg_bwd+proj"] if fused_dropout else ["proj_bwd", "agg_bwd"]) and calls_b == ["proj_bwd", "agg_bwd"], (calls_a, calls_b)
    assert torch.equal(out_a, out_b) and g_a.keys() == g_b.keys()
    for k in g_a:
        torch.testing.assert_close(g_a[k], g_b[k], rtol=1e-5, atol=1e-6, msg=lambda m, k=k: f"{k}: {m}")
    _, _, calls_c = run("acmgcnpp", True)
    assert "agg_bwd+proj" not in calls_c and "proj_bwd" in calls_c, calls_c


def test_plain_backward_into_existing_grads_accumulates_finished_values(monkeypatch):
    """with ``zero_grad(set_to_none=False)`` / gradient accumulation autograd ADDS what the output
    layer's backward returns to the existing ``.grad`` right behind that node -- so outside a deferral list the layer must
    return finished dW' (acm_proj_bwd), never views the hidden layer's kernel fills later.  Two backward passes into
    pre-zeroed grads equal twice the single-pass gradient; torch.autograd.grad on the output layer's weights alone (the
    hidden layer's backward never runs) is finite and equal too."""
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, functional as AF
    ops, n = _dense_graph_ops(seed=3)
    x = torch.randn(n, 7, generator=torch.Generator().manual_seed(1))
    torch.manual_seed(5)
    model = GCN(7, 64, 2, 2, n, 0.3, "acmgcnp", 0, variant=0, attn_layernorm=True)
    model.train()
    model.fused_dropout, model.dropout_state = True, AF.DropoutState(torch.device("cpu"), seed=9)
    model(x, ops).square().sum().backward()
    once = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    for p in model.parameters():
        if p.grad is not None:
            p.grad.zero_()                              # zero_grad(set_to_none=False)
    for _ in range(2):
        model(x, ops).square().sum().backward()
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), k
            torch.testing.assert_close(p.grad, 2 * once[k], rtol=1e-5, atol=1e-6, msg=lambda m, k=k: f"{k}: {m}")
    w2 = [model.gcns[1].weight_low, model.gcns[1].weight_high, model.gcns[1].weight_mlp]
    g2 = torch.autograd.grad(model(x, ops).square().sum(), w2)
    for g, w, nm in zip(g2, w2, ("weight_low", "weight_high", "weight_mlp")):
        torch.testing.assert_close(g, once[f"gcns.1.{nm}"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("f_in,forms", [(20, 7), (128, 5), (128, 0)])
def test_input_dropout_of_the_khop_layer_is_never_lost(monkeypatch, tune, f_in, forms):
    """models.GCN leaves the input dropout to the first layer (in_drop) whenever the dense projection
    CAN draw the mask in its operand load -- but the k-hop layer (acmsgc, hops = 3) projects into two tables, and when
    acm_proj3 declines (fewer than 32 features; the split-bf16 kernels switched off) that path has no dropout in the load:
    the layer then applies acm_dropout itself.  Output and gradients equal the run with the dropped input handed in."""
    fake = fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, functional as AF
    tune(gemm_forms=forms)
    ops, n = _dense_graph_ops(n=8192, avg=14, seed=2)
    ops.hops = 3
    x = torch.randn(n, f_in, generator=torch.Generator().manual_seed(1))
    calls = []
    for name in ("acm_gemm_drop", "acm_proj3", "acm_dropout"):
        orig = getattr(fake, name)
        monkeypatch.setattr(fake, name, (lambda o, nm: lambda *a: (calls.append(nm), o(*a))[1])(orig, name))
    def run(layer_takes_it):
        if not layer_takes_it:                          # the model then applies acm_dropout itself and hands the dropped x down
            monkeypatch.setattr(AF, "in_drop_supported", lambda *a, **k: False)
        calls.clear()
        torch.manual_seed(5)
        model = GCN(f_in, 64, 5, 2, n, 0.4, "acmsgc", 0, variant=0, attn_layernorm=True)
        model.train()
        model.fused_dropout, model.dropout_state = True, AF.DropoutState(torch.device("cpu"), seed=9)
        out = model(x, ops)
        out.square().sum().backward()
        return out.detach(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, list(calls)

    out_a, g_a, calls_a = run(True)
    out_b, g_b, calls_b = run(False)
    assert calls_a.count("acm_dropout") == 1 and calls_b.count("acm_dropout") == 1, (calls_a, calls_b)    # never lost, never twice
    assert "acm_gemm_drop" not in calls_a
    torch.testing.assert_close(out_a, out_b, rtol=1e-6, atol=1e-6 * float(out_b.abs().max()))
    assert g_a.keys() == g_b.keys()
    for k in g_a:
        torch.testing.assert_close(g_a[k], g_b[k], rtol=1e-5, atol=1e-5 * float(g_b[k].abs().max()), msg=lambda m, k=k: f"{k}: {m}")


@pytest.mark.parametrize("model_type,hops", [("acmgcnp", 1), ("acmsgc", 3)])
def test_input_dropout_rides_the_dense_projection(monkeypatch, model_type, hops, tune):
    """acm_proj3 (ABI 21) / acm_gemm_drop (ABI 20): for a wide dense input the first layer's projection Z = drop(X) W -- the
    three weight matrices read in place, no torch.cat -- and its backward dW = drop(X)^T dZ draw the input-dropout mask while
    they load X (ACM-Geometric/models.py:54 + layers.py:86-88), so the separate acm_dropout pass and the dropped copy of X
    disappear; same loss and gradients as without the row-panel kernels (acm_tuning_t.gemm_forms bit 1 cleared)."""
    fake = fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, functional as AF
    ops, n = _dense_graph_ops(n=8192, avg=14, seed=2)
    ops.hops = hops
    x = torch.randn(n, 128, generator=torch.Generator().manual_seed(1))
    calls = []
    for name in ("acm_gemm_drop", "acm_proj3", "acm_dropout"):
        orig = getattr(fake, name)
        monkeypatch.setattr(fake, name, (lambda o, nm: lambda *a: (calls.append(nm), o(*a))[1])(orig, name))

    def run(fused):
        tune(agg_first=0)                              # the LITERAL form (round 5: aggregate-first takes 16 < F_in <= 128 by default)
        if not fused:
            tune(gemm_forms=6)                         # no row-panel kernels: no dropout in the operand loads either
        calls.clear()
        torch.manual_seed(5)
        model = GCN(128, 64, 5, 2, n, 0.4, model_type, 0, variant=0, attn_layernorm=True)
        model.train()
        model.fused_dropout, model.dropout_state = True, AF.DropoutState(torch.device("cpu"), seed=9)
        out = model(x, ops)
        out.square().sum().backward()
        return out.detach(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, list(calls)

    out_a, g_a, calls_a = run(True)
    out_b, g_b, calls_b = run(False)
    # forward (the 64 -> 5 output layer of the 2-layer model projects through acm_proj3 as well), backward
    assert calls_a.count("acm_proj3") == (2 if hops == 1 else 1) and calls_a.count("acm_gemm_drop") == 1 and "acm_dropout" not in calls_a, calls_a
    assert "acm_gemm_drop" not in calls_b and calls_b.count("acm_dropout") >= 1, calls_b
    torch.testing.assert_close(out_a, out_b, rtol=1e-6, atol=1e-6 * float(out_b.abs().max()))
    assert g_a.keys() == g_b.keys()
    for k in g_a:
        torch.testing.assert_close(g_a[k], g_b[k], rtol=1e-5, atol=1e-5 * float(g_b[k].abs().max()), msg=lambda m, k=k: f"{k}: {m}")


@pytest.mark.parametrize("s", [0, 1])
def test_training_without_input_dropout_gathers_the_static_input_once(s, monkeypatch):
    """Round 4: a model without input dropout (the reference's twitch-gamer ACM-GCN+ runs: --dropout 0) hands the SAME
    unmodified feature tensor to its first layer every step, so P = A_low X (and the zero-padded copy of X) of the first
    step serve every later one -- forward with agg_given, backward from the kept P -- until the features are edited in
    place (version counter).  Losses and gradients equal a run with the cache switched off."""
    fake = fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN
    ops, n = _dense_graph_ops(seed=4)
    if s:
        import scipy.sparse as sp
        from acm_gnn_amd.graph import CsrGraph, FilterOperators
        ip, ix, _ = ops.low.arrays()
        pat = sp.csr_matrix((np.ones(len(ix), np.float32), ix.numpy(), ip.numpy()), shape=(n, n))
        deg = np.asarray(pat.sum(1)).ravel().astype(np.float32)
        low = sp.diags(1.0 / deg) @ pat
        ops = FilterOperators(CsrGraph.from_scipy(sp.csr_matrix(low).astype(np.float32), "cpu"), deg=torch.from_numpy(deg))
    x = torch.randn(n, 7, generator=torch.Generator().manual_seed(1))
    given = []
    orig = fake.acm_conv_agg_fwd
    monkeypatch.setattr(fake, "acm_conv_agg_fwd", lambda h, pp, *a: (given.append(int(pp._obj.agg_given)), orig(h, pp, *a))[1])

    def run(cache):
        given.clear()
        torch.manual_seed(5)
        model = GCN(7, 64, 2, 2, n, 0.0, "acmgcnp", s, variant=0, attn_layernorm=True)
        for layer in model.gcns:
            layer.eval_agg_cache = cache
        model.train()
        opt = torch.optim.SGD(model.parameters(), lr=0.05)
        losses = []
        for it in range(4):
            if it == 3:
                x.mul_(1.5)                               # an in-place edit: the next step gathers again
            opt.zero_grad()
            loss = model(x, ops).square().mean()
            loss.backward()
            opt.step()
            losses.append(float(loss))
        x.div_(1.5)
        return losses, [p.detach().clone() for p in model.parameters()], list(given)

    la, pa, ga = run(True)
    lb, pb, gb = run(False)
    assert ga == [0, 1, 1, 0] and gb == [0, 0, 0, 0], (ga, gb)
    np.testing.assert_allclose(la, lb, rtol=1e-5)
    for u, v in zip(pa, pb):
        torch.testing.assert_close(u, v, rtol=1e-4, atol=1e-6)


def test_wide_sparse_features_handed_over_dense_take_the_csr_route(monkeypatch):
    """graph.SparseFeatures.auto / GCN.auto_csr (tuning key csr_features): the reference's loaders hand one-hot / bag-of-words
    features over dense (ACM-Geometric/train.py:66-67); the model projects them from a CSR twin made once per tensor.  Same
    logits and gradients as with the key off (dense route), the twin is made once, and the refusals hold: dense-valued
    inputs, narrow inputs, inputs that need a gradient, acmsnowball, a patched F.dropout."""
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, SparseFeatures, graph, tuning, train as T
    low, high, un, g = graph_tensors("geometric")
    n = low.shape[0]
    gen = torch.Generator().manual_seed(5)
    f_in = 300
    x = (torch.rand(n, f_in, generator=gen) < 0.02).float() * torch.rand(n, f_in, generator=gen)
    y = torch.randint(0, 3, (n,), generator=gen)
    idx = torch.arange(0, n, 2)
    made = []
    real = SparseFeatures.from_torch.__func__
    monkeypatch.setattr(SparseFeatures, "from_torch", classmethod(lambda cls, t: (made.append(1), real(cls, t))[1]))
    outs = {}
    for key in (256, 0):
        with tuning.override(csr_features=key):
            for method in ("acmgcnp", "acmgcnpp", "acmsgc"):
                torch.manual_seed(1)
                model = GCN(f_in, 16, 3, 2, n, 0.0, method, 0, variant=False, attn_layernorm=True)
                seen = []
                layer = model.gcns[0]
                fwd = layer.forward
                layer.forward = lambda inp, *a, _f=fwd, _s=seen, **k: (_s.append(type(inp).__name__), _f(inp, *a, **k))[1]
                for _ in range(2):
                    model.zero_grad()
                    out = model(x, low, high, un)
                    F.nll_loss(F.log_softmax(out, 1)[idx], y[idx]).backward()
                assert seen == (["SparseFeatures"] * 2 if key else ["Tensor"] * 2), (key, method, seen)
                outs[key, method] = (out.detach(), _model_grads(model))
    assert len(made) == 1, made                       # one twin per tensor object, across models and calls
    for method in ("acmgcnp", "acmgcnpp", "acmsgc"):
        _close(outs[256, method][0], outs[0, method][0].numpy(), method + ": csr twin vs dense logits", **FWD)
        for k, v in outs[0, method][1].items():
            _close(outs[256, method][1][k], v.numpy(), method + ": csr twin vs dense " + k)
    # train.TrainStep asks the model before it permutes the rows (relabelled operators): same loss as the dense route
    w = T.row_weights(idx, n)
    losses = {}
    for key in (256, 0):
        with tuning.override(csr_features=key, relabel=1):
            graph.clear_cache()
            torch.manual_seed(1)
            model = GCN(f_in, 16, 3, 2, n, 0.0, "acmgcnp", 0, attn_layernorm=True)
            step = T.TrainStep(model, torch.optim.SGD(model.parameters(), lr=0.0), x, low, y, w, high, un)
            assert step._permuted and isinstance(step.x, SparseFeatures) == bool(key)
            losses[key] = float(step())
    graph.clear_cache()
    assert abs(losses[256] - losses[0]) < 1e-6, losses
    assert len(made) == 1, made
    # an in-place edit of the features is noticed (tensor version): a new twin
    x.mul_(2.0)
    assert isinstance(SparseFeatures.auto(x), SparseFeatures) and len(made) == 2
    # refusals
    assert SparseFeatures.auto(torch.randn(n, f_in)) .__class__ is torch.Tensor                 # dense values
    assert SparseFeatures.auto(x[:, :100].contiguous()).__class__ is torch.Tensor               # narrow
    assert SparseFeatures.auto(x.clone().requires_grad_(True)).__class__ is torch.Tensor        # needs a gradient
    snow = GCN(f_in, 8, 3, 2, n, 0.0, "acmsnowball", 0)
    assert snow.auto_csr(x, None) is x
    model = GCN(f_in, 16, 3, 2, n, 0.5, "acmgcnp", 0)
    monkeypatch.setattr(F, "dropout", lambda t, p=0.5, training=True, inplace=False: t)         # a mask-replay harness
    assert model.auto_csr(x, None) is x
    monkeypatch.undo()


def test_drop_in_route_projects_dropped_dense_features_from_the_reference_structure(monkeypatch):
    """layers.GraphConvolution._csr_input: the reference's own GCN applies F.dropout to the dense features and hands a NEW
    tensor to the first layer every step (ACM-Geometric/models.py:54).  An evaluation pass (the loader's tensor itself)
    makes the CSR twin and leaves it as the layer's reference structure; a training pass takes that structure with the
    dropped copy's values after the support check -- same output and gradients as the dense projection; a tensor with an
    entry outside the structure, small inputs and the first training pass (no structure yet) stay dense."""
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GraphConvolution, SparseFeatures, tuning
    low, high, un, g = graph_tensors("geometric")
    n = low.shape[0]
    gen = torch.Generator().manual_seed(6)
    f_in = 320
    x = (torch.rand(n, f_in, generator=gen) < 0.02).float()
    torch.manual_seed(2)
    layer = GraphConvolution(f_in, 16, n, "acmgcnp", structure_info=0)
    seen = []
    import acm_gnn_amd.layers as L
    real = L.AF.acm_conv
    monkeypatch.setattr(L.AF, "acm_conv", lambda inp, *a, **k: (seen.append(type(inp).__name__), real(inp, *a, **k))[1])
    monkeypatch.setattr(GraphConvolution, "CSR_CHECK_MIN_ELEMENTS", 1)

    def train_pass(inp):
        layer.train()
        layer.zero_grad()
        out = layer(inp, low, high, un)
        out.square().sum().backward()
        return out.detach(), {k: p.grad.clone() for k, p in layer.named_parameters() if p.grad is not None}

    xd = F.dropout(x, 0.5, training=True)
    train_pass(xd)                                            # no reference structure yet
    assert seen == ["Tensor"]
    layer.eval()
    with torch.no_grad():
        e1 = layer(x, low, high, un)
        e2 = layer(x, low, high, un)
    assert seen[1:] == ["SparseFeatures"] * 2 and torch.equal(e1, e2)
    with tuning.override(csr_features=0):
        with torch.no_grad():
            e0 = layer(x.clone(), low, high, un)
    _close(e1, e0.numpy(), "evaluation: csr twin vs dense", **FWD)
    del seen[:]
    out_c, g_c = train_pass(xd)
    assert seen == ["SparseFeatures"]
    with tuning.override(csr_features=0):
        out_d, g_d = train_pass(xd)
    assert seen == ["SparseFeatures", "Tensor"]
    _close(out_c, out_d.numpy(), "training: reference structure vs dense", **FWD)
    for k, v in g_d.items():
        _close(g_c[k], v.numpy(), "training: reference structure vs dense, " + k)
    # an entry outside the structure: not a masked copy of the features -> dense
    bad = xd.clone()
    bad[0, int((x[0] == 0).nonzero()[0])] = 1.0
    del seen[:]
    train_pass(bad)
    assert seen == ["Tensor"]
    # small inputs are not checked per step
    monkeypatch.setattr(GraphConvolution, "CSR_CHECK_MIN_ELEMENTS", 1 << 24)
    train_pass(xd)
    assert seen == ["Tensor", "Tensor"]


def test_training_and_evaluation_inputs_keep_separate_p_cache_entries(monkeypatch):
    """a dropout-0 training step reuses the first layer's P = A_low X from the layer's cache, and a
    captured step bakes in that tensor's address.  With relabelled operators the evaluation pass hands the layer a FRESH
    permuted copy of x: with one cache entry per layer that replaced the training entry and freed the P a captured replay
    still reads.  Now (a) training and evaluation have their own entries -- neither evicts the other, so the gather runs
    once per role, not once per pass -- and (b) a step that captures holds the entries (`_held_entries`)."""
    import weakref
    fake = fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, train as T
    ops, n = _dense_graph_ops(seed=9)
    x = torch.randn(n, 7, generator=torch.Generator().manual_seed(2))
    gathers = []
    orig = fake.acm_conv_agg_fwd
    monkeypatch.setattr(fake, "acm_conv_agg_fwd", lambda h, pp, *a: (gathers.append(int(pp._obj.agg_given)), orig(h, pp, *a))[1])
    torch.manual_seed(3)
    model = GCN(7, 64, 2, 2, n, 0.0, "acmgcnp", 0, variant=0, attn_layernorm=True)
    opt = torch.optim.SGD(model.parameters(), lr=0.05)

    def train_pass():
        model.train()
        opt.zero_grad()
        model(x, ops).square().mean().backward()
        opt.step()

    def eval_pass():
        model.eval()
        with torch.no_grad():
            return model(x.clone(), ops)             # a fresh tensor object every pass, as x.index_select(perm) is

    train_pass()
    p_train = weakref.ref(model.gcns[0].held_entries()[0][2]["agg"])
    assert p_train() is not None
    held = T._held_entries(model)                    # what TrainStep._capture keeps
    gathers.clear()
    eval_pass()
    train_pass()
    eval_pass()
    train_pass()
    # the training entry survived both evaluation passes: its P was given (1) both times; the evaluation input is a new
    # object each pass and gathers (0) -- by design
    assert gathers[0::2] == [0, 0] and gathers[1::2] == [1, 1], gathers
    assert p_train() is not None and held[0][2]["agg"] is p_train()
    roles = model.gcns[0].__dict__["_eval_agg"]
    assert set(roles) == {"train", "eval"}


@pytest.mark.parametrize("model_type,s,variant,p_drop", [("acmgcnp", 1, 0, 0.5), ("acmgcn", 0, 1, 0.0)])
def test_small_graph_step_host_path_equals_the_general_path(model_type, s, variant, p_drop, monkeypatch):
    """small.SmallPlan (acm_small_step, ABI 25) behind train.TrainStep / EvalStep: a small graph with CSR features and this
    package's FusedAdam takes the fused step -- one C-ABI call per step, parameters and the optimizer's own state tensors
    updated in place, the dropout counter advanced -- and lands where the general path (autograd Functions + acm_adam_step)
    lands; anything outside the envelope says why and stays on the general path."""
    import scipy.sparse as sp
    fake = fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, FusedAdamW, SparseFeatures, functional as AF, train as T
    from acm_gnn_amd.graph import CsrGraph, FilterOperators
    from acm_gnn_amd.small import SmallPlan
    ops0, n = _dense_graph_ops(n=90, avg=8, seed=12)
    ip, ix, _ = ops0.low.arrays()
    pat = sp.csr_matrix((np.ones(len(ix), np.float32), ix.numpy(), ip.numpy()), shape=(n, n))
    deg = np.asarray(pat.sum(1)).ravel().astype(np.float32)
    from acm_gnn_amd import graph as G
    ops = G.as_implicit(FilterOperators(CsrGraph.from_scipy(sp.csr_matrix(sp.diags(1.0 / deg) @ pat).astype(np.float32), "cpu"),
                                        deg=torch.from_numpy(deg)))
    assert ops.implicit
    rng = np.random.default_rng(0)
    x_np = ((rng.random((n, 50)) < 0.08) * rng.uniform(0.5, 1.5, (n, 50))).astype(np.float32)
    xs = SparseFeatures.from_scipy(sp.csr_matrix(x_np), "cpu")
    y = torch.from_numpy(rng.integers(0, 3, n))
    w = T.row_weights(torch.arange(0, n, 2), n)

    def run(small):
        torch.manual_seed(1)
        model = GCN(50, 64, 3, 2, n, p_drop, model_type, s, variant=bool(variant), attn_layernorm=model_type == "acmgcnp")
        model.dropout_state = AF.DropoutState("cpu", seed=21) if p_drop > 0 else None
        opt = FusedAdamW(model.parameters(), lr=0.02, weight_decay=1e-2)
        step = T.TrainStep(model, opt, xs, ops, y, w, small_step=None if small else False)
        assert (step.small is not None) == small, step.small_refused
        calls = getattr(fake, "small_calls", 0)
        losses = [float(step()) for _ in range(5)]
        assert getattr(fake, "small_calls", 0) - calls == (5 if small else 0)
        ev = T.EvalStep(model, xs, ops, y, (torch.arange(0, n, 2), torch.arange(1, n, 2)), small_step=None if small else False)
        assert (ev.small is not None) == small
        out, accs, vloss = ev()
        return model, opt, losses, out, accs, vloss

    ma, oa, la, out_a, acc_a, vl_a = run(True)
    mb, ob, lb, out_b, acc_b, vl_b = run(False)
    np.testing.assert_allclose(la, lb, rtol=2e-4, atol=1e-6)
    for (k, pa), (_, pb) in zip(ma.state_dict().items(), mb.state_dict().items()):
        torch.testing.assert_close(pa, pb, rtol=2e-3, atol=2e-5, msg=k)
    torch.testing.assert_close(out_a, out_b, rtol=1e-3, atol=1e-4)
    assert acc_a == pytest.approx(acc_b, abs=0.03) and vl_a == pytest.approx(vl_b, rel=1e-3)
    # the optimizer's state is the plan's state: same tensors, same step counts as optimizer.step() leaves
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        sa, sb = oa.state.get(pa, {}), ob.state.get(pb, {})
        assert set(sa) == set(sb), "the fused step touched a parameter the reference's autograd leaves without gradient"
        if sa:
            assert float(sa["step"]) == float(sb["step"]) == 5.0
    if p_drop > 0:
        assert int(ma.dropout_state.step.item()) == 5 and ma.dropout_state.host_steps == 5
    # after a forward the layers carry the mixing weights, like the reference's (layers.py:91,107)
    assert ma.gcns[0].att_low.shape == (n, 1) and float(ma.gcns[1].att_mlp.abs().sum()) > 0
    # outside the envelope: says why
    assert "dense features" in SmallPlan.why_not(ma, torch.from_numpy(x_np), ops)
    with __import__("acm_gnn_amd").tuning.override(small_step=0):
        assert "switched off" in SmallPlan.why_not(ma, xs, ops)
    m3 = GCN(50, 32, 3, 2, n, 0.0, "acmgcn", 0)
    assert "hidden width" in SmallPlan.why_not(m3, xs, ops)
    m4 = GCN(50, 64, 3, 2, n, 0.0, "acmgcnpp", 0)
    assert "model_type" in SmallPlan.why_not(m4, xs, ops)
    assert "optimizer" in SmallPlan.why_not(ma, xs, ops, torch.optim.Adam(ma.parameters()))


def test_train_step_keeps_the_operators_it_builds_from_adjacency_tensors(monkeypatch):
    """a caller that hands TrainStep the reference's TENSORS (ACM-Pytorch dialect: dense adj_low, sparse
    adj_high / adj_low_unnormalized, dense bag-of-words features; ACM-Pytorch/train.py:95-139) gets the operator set built
    once and kept (``step.adj``), the CSR twin of the features, and therefore the fused small-graph step -- the same plan,
    and the same numbers, as a caller that passes FilterOperators + SparseFeatures himself."""
    import scipy.sparse as sp
    fake = fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, FusedAdam, SparseFeatures, functional as AF, train as T
    from acm_gnn_amd.graph import FilterOperators, operators_for
    n = 120
    rng = np.random.default_rng(4)
    a = sp.random(n, n, density=0.06, random_state=3, format="csr")
    a = ((a + a.T) > 0).astype(np.float32).tocsr()
    a.setdiag(0)
    a.eliminate_zeros()
    a_un = torch.from_numpy(a.toarray())
    rowsum = (torch.eye(n) + a_un).sum(1)
    adj_low = torch.mm(torch.diag(torch.pow(rowsum, -1)), torch.eye(n) + a_un)        # utils.normalize_tensor
    adj_high = (torch.eye(n) - adj_low).to_sparse()
    x = torch.from_numpy(((rng.random((n, 400)) < 0.03) * 1.0).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, 3, n))
    w = T.row_weights(torch.arange(0, n, 2), n)

    def run(tensors):
        torch.manual_seed(1)
        model = GCN(400, 64, 3, 1, n, 0.5, "acmgcnp", 1, variant=False, attn_layernorm=False)
        model.dropout_state = AF.DropoutState("cpu", seed=9)
        opt = FusedAdam(model.parameters(), lr=0.01, weight_decay=1e-4)
        if tensors:
            step = T.TrainStep(model, opt, x, adj_low, y, w, adj_high, a_un.to_sparse())
        else:
            ops = operators_for(adj_low, adj_high, a_un.to_sparse())
            step = T.TrainStep(model, opt, SparseFeatures.from_torch(x), ops, y, w)
        assert isinstance(step.adj, FilterOperators) and isinstance(step.x, SparseFeatures)
        assert step.small is not None, step.small_refused
        calls = getattr(fake, "small_calls", 0)
        losses = [float(step()) for _ in range(3)]
        assert getattr(fake, "small_calls", 0) - calls == 3
        return losses, [p.detach().clone() for p in model.parameters()]

    la, pa = run(True)
    lb, pb = run(False)
    assert la == lb
    for u, v in zip(pa, pb):
        assert torch.equal(u, v)
    # and EvalStep from the same tensors
    torch.manual_seed(1)
    model = GCN(400, 64, 3, 1, n, 0.5, "acmgcnp", 1, variant=False, attn_layernorm=False)
    ev = T.EvalStep(model, x, adj_low, y, (torch.arange(0, n, 2),), adj_high, a_un.to_sparse(), loss_set=0)
    assert ev.small is not None, ev.small_refused


@pytest.mark.parametrize("fused", [True, False], ids=["one_kernel", "products_then_head"])
@pytest.mark.parametrize("model_type,f_in,p_drop,ln", [("acmgcnp", 128, 0.4, True), ("acmgcn", 65, 0.0, False), ("acmgcnp", 40, 0.3, True)])
def test_aggregate_first_for_wide_dense_inputs_equals_the_literal_form(model_type, f_in, p_drop, ln, fused, monkeypatch, tune):
    """Round 5 (functional._AcmAggWide): a first layer with 16 < F_in <= 128 dense features, no ReLU before the filter and an
    input that takes no gradient gathers P = A_low drop(X) once (F_in floats per edge instead of 2 F) and needs NO transposed
    gather in its backward: dW_L = P^T G_L, dW_H = X^T G_H - P^T G_H, dW_I = X^T G_I.  Same logits and gradients as the
    literal project-then-gather form (tuning rewrites bit 1 off); F_in that is no multiple of 4 is padded (pokec: 65).
    ``fused``: projections + head behind the gather as one call (acm_conv_aggw_fwd, rewrites bit 8) or two products + the head."""
    fake = fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, functional as AF
    ops, n = _dense_graph_ops(n=8192, avg=30, seed=3)          # (the form is taken from a mean degree of 12 on)
    x = torch.randn(n, f_in, generator=torch.Generator().manual_seed(2))
    calls = []
    for name in ("acm_conv_bwd_spmm", "acm_spmm_ex", "acm_spmm", "acm_conv_aggw_fwd", "acm_conv_head_fwd", "acm_conv_aggw_bwd"):
        orig = getattr(fake, name)
        monkeypatch.setattr(fake, name, (lambda o, nm: lambda *a: (calls.append(nm), o(*a))[1])(orig, name))

    def run(agg):
        tune(agg_first=int(agg), aggw_fused=int(fused))
        calls.clear()
        torch.manual_seed(4)
        model = GCN(f_in, 64, 3, 2, n, p_drop, model_type, 0, variant=0, attn_layernorm=ln)
        model.train()
        if p_drop > 0:
            model.fused_dropout, model.dropout_state = True, AF.DropoutState(torch.device("cpu"), seed=11)
        out = model(x, ops)
        out.square().sum().backward()
        return out.detach(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, list(calls)

    out_a, g_a, calls_a = run(True)
    out_b, g_b, calls_b = run(False)
    # one gather forward for the first layer, none backward (the output layer keeps its two)
    assert calls_a.count("acm_conv_bwd_spmm") == 1 and calls_b.count("acm_conv_bwd_spmm") == 2, (calls_a, calls_b)
    assert len([c for c in calls_a if c.startswith("acm_spmm")]) == 1
    assert (calls_a.count("acm_conv_aggw_fwd"), calls_a.count("acm_conv_head_fwd"), calls_a.count("acm_conv_aggw_bwd")) == \
        ((1, 0, 1) if fused else (0, 1, 0))
    torch.testing.assert_close(out_a, out_b, rtol=1e-5, atol=1e-5 * float(out_b.abs().max()))
    assert g_a.keys() == g_b.keys()
    for k in g_a:
        torch.testing.assert_close(g_a[k], g_b[k], rtol=1e-4, atol=1e-5 * float(g_b[k].abs().max()), msg=lambda m, k=k: f"{k}: {m}")


@pytest.mark.parametrize("f_in", [40, 65])
def test_wide_aggregate_first_layer_gathers_a_static_input_once(f_in, monkeypatch):
    """functional._AcmAggWide with layers.GraphConvolution's P cache: evaluation passes over the same unmodified features and
    training steps of a model without input dropout gather P = A_low X once (F_in 65: the zero-padded copy is kept with it); an
    in-place edit of the features, another tensor or a model WITH input dropout in training mode takes the gather again."""
    fake = fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, functional as AF
    ops, n = _dense_graph_ops(n=8192, avg=30, seed=3)
    gathers = []
    for name in ("acm_spmm_ex", "acm_spmm"):
        orig = getattr(fake, name)
        monkeypatch.setattr(fake, name, (lambda o: lambda h, g, ldg, width, *a: (gathers.append(int(width)), o(h, g, ldg, width, *a))[1])(orig))
    fp = -(-f_in // 4) * 4
    torch.manual_seed(2)
    model = GCN(f_in, 64, 3, 2, n, 0.0, "acmgcnp", 0, variant=False)
    x = torch.randn(n, f_in, generator=torch.Generator().manual_seed(5))
    model.eval()
    with torch.no_grad():
        o1 = model(x, ops)
        o2 = model(x, ops)
        assert gathers.count(fp) == 1 and torch.equal(o1, o2)
        x.mul_(1.5)
        o3 = model(x, ops)
        assert gathers.count(fp) == 2 and not torch.equal(o1, o3)
        model(x.clone(), ops)
        assert gathers.count(fp) == 3
    # training without input dropout: one gather for the whole run, gradients as without the cache
    model.train()
    gathers.clear()
    grads = []
    for cache in (True, True, False):
        for layer in model.gcns:
            layer.eval_agg_cache = cache
        model.zero_grad(set_to_none=True)
        model(x, ops).square().sum().backward()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    assert gathers.count(fp) == 2                              # first cached step + the uncached one
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]) and torch.equal(grads[0][k], grads[2][k]), k
    # with input dropout (counter-based) every training step draws a new input: never from the cache
    for layer in model.gcns:
        layer.eval_agg_cache = True
    model.dropout, model.fused_dropout, model.dropout_state = 0.3, True, AF.DropoutState(torch.device("cpu"), seed=3)
    gathers.clear()
    for _ in range(2):
        model(x, ops).square().sum().backward()
    assert gathers.count(fp) == 2


def test_small_plan_refuses_hooked_models_and_optimizers(monkeypatch):
    """the fused small-graph step calls neither model.forward nor optimizer.step and leaves p.grad None -- a hook
    on any of them would silently never fire, so such a model / optimizer keeps the general path."""
    import scipy.sparse as sp
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, FusedAdam, SparseFeatures, train as T
    from acm_gnn_amd.small import SmallPlan
    ops, n = _dense_graph_ops(n=90, avg=8, seed=12)
    rng = np.random.default_rng(0)
    xs = SparseFeatures.from_scipy(sp.csr_matrix(((rng.random((n, 50)) < 0.08) * 1.0).astype(np.float32)), "cpu")
    y, w = torch.from_numpy(rng.integers(0, 3, n)), T.row_weights(torch.arange(0, n, 2), n)

    def fresh():
        torch.manual_seed(1)
        m = GCN(50, 64, 3, 2, n, 0.0, "acmgcn", 0)
        return m, FusedAdam(m.parameters(), lr=0.01)

    m, o = fresh()
    assert SmallPlan.why_not(m, xs, ops, o) is None
    h = m.gcns[0].register_forward_hook(lambda *a: None)
    assert "hooks" in SmallPlan.why_not(m, xs, ops, o)
    step = T.TrainStep(m, o, xs, ops, y, w)
    assert step.small is None and "hooks" in step.small_refused
    h.remove()
    assert SmallPlan.why_not(m, xs, ops, o) is None
    m, o = fresh()
    m.gcns[1].weight_low.register_hook(lambda g: g)
    assert "hooks" in SmallPlan.why_not(m, xs, ops, o)
    m, o = fresh()
    o.register_step_post_hook(lambda *a: None)
    assert "hooks" in SmallPlan.why_not(m, xs, ops, o)
    assert SmallPlan.why_not(m, xs, ops, None, need_dropout_state=False) is None         # (an evaluation plan has no optimizer)


def test_layer_applied_input_dropout_never_reuses_a_cached_aggregate(monkeypatch):
    """``GraphConvolution.forward(..., input_drop=(p, tag, state))`` on an aggregate-first layer whose projection
    cannot carry the dropout drops the input itself -- the operand of P = A_low X then changes with the step counter while the
    raw input tensor (the key of the layer's P cache) does not: the cache must stay out of it."""
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import functional as AF
    from acm_gnn_amd.layers import GraphConvolution
    ops, n = _dense_graph_ops()
    x = torch.randn(n, 7, generator=torch.Generator().manual_seed(1))
    monkeypatch.setattr(AF, "in_drop_supported", lambda *a, **k: False)
    torch.manual_seed(0)
    layer = GraphConvolution(7, 64, n, "acmgcnp", attn_layernorm=True)
    layer.eval()                                           # (no gradient: the P cache is at its most eager)
    st = AF.DropoutState("cpu", seed=5)
    with torch.no_grad():
        a0 = layer(x, ops, input_drop=(0.5, 0, st)).clone()
        again = layer(x, ops, input_drop=(0.5, 0, st)).clone()
        st.advance()
        a1 = layer(x, ops, input_drop=(0.5, 0, st)).clone()
        layer.__dict__.pop("_eval_agg", None)              # a layer without history, same counter value
        b1 = layer(x, ops, input_drop=(0.5, 0, st)).clone()
        plain0 = layer(x, ops).clone()                      # without dropout the cache serves the second pass
        plain1 = layer(x, ops).clone()
    assert torch.equal(a0, again) and not torch.equal(a0, a1)
    assert torch.equal(a1, b1), "the second step used the first step's P"
    torch.testing.assert_close(plain0, plain1)
    assert layer.held_entries()


def test_eval_step_takes_the_one_launch_metrics(monkeypatch):
    """train.EvalStep: accuracy per index set + validation NLL from ONE library call (acm_eval_metrics, ABI 28) instead of
    eight torch launches; same numbers as the torch ops, unlabeled rows (-1) outside the sets allowed, and anything the call
    does not cover (more than 64 classes, more than eight sets) keeps the torch ops."""
    fake = fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, train as T
    ops, n = _dense_graph_ops(n=200, avg=10, seed=2)
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(n, 7, generator=g), torch.randint(0, 3, (n,), generator=g)
    sets = (torch.arange(0, 80), torch.arange(80, 140), torch.arange(140, 190))
    y[190:] = -1
    torch.manual_seed(0)
    model = GCN(7, 64, 3, 2, n, 0.2, "acmgcnp", 0, attn_layernorm=True)
    a = T.EvalStep(model, x, ops, y, sets)
    b = T.EvalStep(model, x, ops, y, sets, fused_metrics=False)
    calls = getattr(fake, "eval_metrics_calls", 0)
    (oa, acc_a, la), (ob, acc_b, lb) = a(), b()
    assert getattr(fake, "eval_metrics_calls", 0) - calls == 1
    torch.testing.assert_close(oa, ob)
    np.testing.assert_allclose(acc_a, acc_b, rtol=1e-6)
    np.testing.assert_allclose(la, lb, rtol=1e-5)
    many = T.EvalStep(model, x, ops, y, tuple(torch.arange(q, q + 10) for q in range(0, 90, 10)))      # nine sets
    many()
    assert getattr(fake, "eval_metrics_calls", 0) - calls == 1


def test_fit_concurrent_without_streams_is_fit_run_by_run(monkeypatch):
    """train.fit_concurrent on the CPU double (no stream to overlap on): every run is fit() of its model, in order."""
    import scipy.sparse as sp
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, FusedAdam, SparseFeatures, train as T
    ops, n = _dense_graph_ops(n=90, avg=8, seed=12)
    rng = np.random.default_rng(0)
    xs = SparseFeatures.from_scipy(sp.csr_matrix(((rng.random((n, 50)) < 0.08) * 1.0).astype(np.float32)), "cpu")
    y = torch.from_numpy(rng.integers(0, 3, n))

    def make(k):
        torch.manual_seed(k)
        m = GCN(50, 64, 3, 2, n, 0.0, "acmgcn", 0)
        idx = torch.randperm(n, generator=torch.Generator().manual_seed(k))
        return m, FusedAdam(m.parameters(), lr=0.01), idx[:40], idx[40:65], idx[65:]

    got = T.fit_concurrent([make(k) for k in range(3)], xs, ops, y, epochs=4, rule="max_val_acc")
    for k in range(3):
        m, opt, tr, va, te = make(k)
        sel, hist = T.fit(m, opt, xs, ops, y, tr, va, te, 4, rule="max_val_acc")
        assert got[k][0] == sel and got[k][1] == hist
