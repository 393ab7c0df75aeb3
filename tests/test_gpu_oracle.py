"""HIP layer vs the CPU oracle on seeded inputs: both execution forms (literal project-then-
aggregate and aggregate-first), real graph structures, split long rows.  MI355X box."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import GOLDEN, load_npz, tune_now as tune
from oracle import acm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(n, seed, density=0.05, hub=True):
    rng = np.random.default_rng(seed)
    a = sp.random(n, n, density=density, random_state=rng, format="csr")
    a = ((a + a.T) > 0).astype(np.float64).tolil()
    if hub:
        a[0, 1:] = 1.0
        a[1:, 0] = 1.0
    a[n - 1, :] = 0
    a[:, n - 1] = 0
    a[3, 3] = 1.0                                    # raw self-loop (quirk Q5)
    return sp.csr_matrix(a)


def _run_both(model_type, variant, structure_info, ln, n, f_in, f_out, seed, x_grad, monkeypatch, agg,
              adj=None, chunk_env=None, implicit=True):
    from acm_gnn_amd import GraphConvolution
    from acm_gnn_amd.graph import clear_cache, operators_for
    tune(agg_first=int(bool(agg)))
    tune(implicit=int(bool(implicit)))
    clear_cache()
    adj = adj if adj is not None else _graph(n, seed)
    n = adj.shape[0]
    low, high, un = O.filters_linkx(adj)
    torch.manual_seed(seed)
    layer = GraphConvolution(f_in, f_out, n, model_type, variant=variant, structure_info=structure_info,
                             attn_layernorm=ln)
    with torch.no_grad():
        for nm in ("low", "high", "mlp", "struc_low"):
            getattr(layer, f"layer_norm_{nm}").weight.uniform_(0.5, 1.5)
            getattr(layer, f"layer_norm_{nm}").bias.uniform_(-0.5, 0.5)
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in layer.named_parameters()}
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(n, f_in, generator=g)
    gout = torch.randn(n, f_out, generator=g)
    # oracle
    xr = x.clone().requires_grad_(x_grad)
    ref, ref_att = O.layer_forward(params, xr, low, high, un if structure_info else None, model_type=model_type,
                                   variant=variant, structure_info=structure_info, attn_layernorm=ln,
                                   return_att=True)
    ref.backward(gout)
    # HIP
    layer = layer.to(DEV)
    xd = x.to(DEV).requires_grad_(x_grad)
    lowd, highd, und = low.to(DEV), high.to(DEV), un.to(DEV) if structure_info else None
    assert operators_for(lowd, highd, und).implicit == implicit      # filters_linkx graphs always qualify
    out = layer(xd, lowd, highd, und)
    out.backward(gout.to(DEV))
    scale = max(1.0, float(ref.abs().max()))
    assert float((out.detach().cpu() - ref.detach()).abs().max()) < 2e-5 * scale
    att = torch.cat([layer.att_low, layer.att_high, layer.att_mlp], 1).cpu()
    assert float((att - ref_att[:, :3].detach()).abs().max()) < 2e-5
    if structure_info:
        assert float((layer.att_struc_vec_low.cpu() - ref_att[:, 3:4].detach()).abs().max()) < 2e-5
    for k, p in layer.named_parameters():
        rg = params[k].grad
        if rg is None:
            assert p.grad is None, k
            continue
        assert p.grad is not None, k
        tol = 1e-4 * max(1.0, float(rg.abs().max()))
        assert float((p.grad.cpu() - rg).abs().max()) < tol, (k, float((p.grad.cpu() - rg).abs().max()), tol)
    if x_grad:
        tol = 1e-4 * max(1.0, float(xr.grad.abs().max()))
        assert float((xd.grad.cpu() - xr.grad).abs().max()) < tol
    return out.detach().cpu()


AGG_CASES = [("acmgcn", False, 3, 16), ("acmgcn", False, 7, 64), ("acmgcnp", True, 7, 64), ("acmgcnp", True, 8, 33),
             ("acmgcnp", True, 4, 5), ("acmgcnp", False, 12, 64), ("acmgcnp", True, 16, 64), ("acmsgc", False, 7, 64),
             ("acmgcnpp", True, 1, 2)]


@pytest.mark.parametrize("model_type,ln,f_in,f_out", AGG_CASES)
def test_aggregate_first_matches_oracle_and_literal(model_type, ln, f_in, f_out, monkeypatch):
    from acm_gnn_amd import functional as AF
    timer = AF.KernelTimer()
    AF.set_kernel_timer(timer)
    try:
        a = _run_both(model_type, 0, 0, ln, 300, f_in, f_out, 11, False, monkeypatch, agg=True)
        used = set(k.split("/")[0] for k in timer.events)
        assert "conv_agg_fwd" in used and "conv_agg_bwd" in used and "conv_fwd" not in used, used
        timer.events.clear()
        b = _run_both(model_type, 0, 0, ln, 300, f_in, f_out, 11, False, monkeypatch, agg=False)
        used = set(k.split("/")[0] for k in timer.events)
        assert "conv_fwd" in used and "conv_agg_fwd" not in used, used
    finally:
        AF.set_kernel_timer(None)
    assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("form", ["fused", "two-stage", "two-stage-4rows"])
@pytest.mark.parametrize("f_in,f_out,hub", [(7, 64, True), (3, 16, True), (8, 33, False)])
def test_aggregate_first_kernel_forms(form, f_in, f_out, hub, monkeypatch, tune):
    """The implementations of the three-channel aggregate-first forward -- one kernel (two lanes per neighbour for
    32-byte rows, one lane per neighbour for 16-byte rows), gather + the sixteen-rows-per-wave row-local stage, and
    gather + the four-rows-per-wave stage with the long rows' partial sums added by that stage -- against the oracle, on
    a graph with a split hub row."""
    if form != "fused":
        tune(agg_fused=0)
    if form == "two-stage-4rows":
        tune(rows16=6)
    adj = _graph(700, 21, density=0.04, hub=hub)
    _run_both("acmgcnp", 0, 0, True, 700, f_in, f_out, 21, False, monkeypatch, agg=True, adj=adj)
    _run_both("acmgcn", 0, 0, False, 700, f_in, f_out, 22, False, monkeypatch, agg=True, adj=adj, implicit=False)


@pytest.mark.parametrize("model_type,s,f_in,f_out,agg,implicit", [
    ("acmgcnp", 0, 7, 64, True, True),      # aggregate-first: the fused kernel declines rows of several windows -> two stages
    ("acmgcnp", 1, 7, 64, True, True),      # ... with the structure channel (narrow P gather + wide S gather)
    ("acmgcnp", 0, 64, 2, False, True),     # literal two-class layer: EpiRaw + row kernel forward, EpiBwd backward
    ("acmgcnp", 1, 64, 2, False, True),     # ... four channels: the pair-lane kernel over packed 32-byte rows
    ("acmgcnp", 0, 40, 5, False, False),    # blocks of 8 columns, explicit values
    ("acmgcn", 0, 33, 3, False, True)])
def test_rows_of_several_windows_match_oracle(model_type, s, f_in, f_out, agg, implicit, monkeypatch, tune):
    """Work lists with rows that sixteen pieces of 4 x chunk do not cover (acm_csr.cpp, build_items: such a row takes
    several whole windows, each window leaves its sum in a partial slot, spmm_fixup_windows_kernel adds them) -- forced
    here with chunk = 8 on a 700-node graph whose node 0 has 699 neighbours (3 windows of 16 pieces), every other row
    an ordinary long row of <= 16 pieces.  Layer forward + backward against the oracle for the narrow gathers of both
    forms."""
    from acm_gnn_amd.graph import CsrGraph
    tune(chunk=8)
    adj = _graph(700, 31, density=0.04, hub=True)
    low = O.filters_linkx(adj)[0].coalesce()
    idx = low.indices().numpy()
    probe = CsrGraph.from_scipy(sp.csr_matrix((low.values().numpy(), (idx[0], idx[1])), shape=tuple(low.shape)), DEV)
    assert probe.chunk == 8 and probe.max_degree >= 699
    assert probe.n_partial_slots % 16 == 0 and probe.n_items >= 700 + 48 - 1    # the hub alone: 48 pieces in 3 windows
    _run_both(model_type, 0, s, True, 700, f_in, f_out, 31, False, monkeypatch, agg=agg, adj=adj, implicit=implicit)


@pytest.mark.parametrize("ln", [True, False], ids=["ln", "no-ln"])
@pytest.mark.parametrize("s,f_in", [(0, 3), (0, 7), (0, 12), (1, 4), (1, 7), (1, 8), (1, 13), (1, 16)])
def test_sixteen_rows_per_wave_stages_cover_channels_and_pad_widths(s, f_in, ln, monkeypatch, tune):
    """Round 4: agg_epi16_kernel / agg_bwd16_kernel for three AND four channels (the structure channel
    arrives as finished rows) and f_pad = 4, 8, 16 -- the two-stage forward + the row-local backward against the oracle
    (forward, mixing weights, every gradient incl. d struc_low), on a graph with a split hub row and a row count that is
    not a multiple of 16; and against the four-rows-per-wave kernels they replace."""
    from acm_gnn_amd import functional as AF
    adj = _graph(333, 17, density=0.05, hub=True)
    tune(agg_fused=0)                                  # the row-local stage as its own kernel for three channels too
    timer = AF.KernelTimer()
    AF.set_kernel_timer(timer)
    try:
        a = _run_both("acmgcnp", 0, s, ln, 333, f_in, 64, 31, False, monkeypatch, agg=True, adj=adj)
        used = set(k.split("/")[0] for k in timer.events)
        assert {"conv_agg_fwd", "conv_agg_bwd"} <= used and "conv_fwd" not in used, used
    finally:
        AF.set_kernel_timer(None)
    tune(rows16=4)
    b = _run_both("acmgcnp", 0, s, ln, 333, f_in, 64, 31, False, monkeypatch, agg=True, adj=adj)
    assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))


AGG_STRUC_CASES = [(True, 7, 64), (False, 7, 64), (True, 3, 24), (True, 16, 40), (True, 4, 5), (False, 12, 33)]


@pytest.mark.parametrize("ln,f_in,f_out", AGG_STRUC_CASES)
def test_aggregate_first_with_structure_channel(ln, f_in, f_out, monkeypatch):
    """ABI v5: four channels in aggregate-first order (pre_S = deg (A_low S) - S; dS through acm_spmm_ex)."""
    from acm_gnn_amd import functional as AF
    timer = AF.KernelTimer()
    AF.set_kernel_timer(timer)
    try:
        a = _run_both("acmgcnp", 0, 1, ln, 300, f_in, f_out, 13, False, monkeypatch, agg=True)
        used = set(k.split("/")[0] for k in timer.events)
        assert {"conv_agg_fwd", "conv_agg_bwd", "spmm_sub"} <= used and "conv_fwd" not in used, used
        timer.events.clear()
        b = _run_both("acmgcnp", 0, 1, ln, 300, f_in, f_out, 13, False, monkeypatch, agg=False)
        used = set(k.split("/")[0] for k in timer.events)
        assert "conv_fwd" in used and "conv_agg_fwd" not in used, used
    finally:
        AF.set_kernel_timer(None)
    assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))


def test_spmm_ex_entry_point():
    """Y = relu?(row_scale * (A G) - sub_scale * SUB) against scipy: explicit values and pattern-only handles
    (with a repeated column), split long rows, every width class, bf16 operand."""
    import ctypes as C
    from acm_gnn_amd import _lib
    from acm_gnn_amd.functional import cast_bf16
    from acm_gnn_amd.graph import CsrGraph
    rng = np.random.default_rng(0)
    n = 700
    a = sp.csr_matrix(_graph(n, 4, density=0.02), dtype=np.float32)
    a.sort_indices()
    lib = _lib.load()
    for unit in (False, True):
        if unit:
            # repeat the first stored column of every non-empty row: the multigraph encoding of a raw self-loop
            ip, ix = a.indptr, a.indices
            rep = np.ones(a.nnz, np.int64)
            rep[ip[:-1][np.diff(ip) > 0]] = 2
            ix2 = np.repeat(ix, rep)
            ip2 = np.concatenate([[0], np.cumsum(np.bincount(np.repeat(np.arange(n), np.diff(ip)), weights=rep, minlength=n))]).astype(np.int32)
            g = CsrGraph.from_csr(torch.from_numpy(ip2).to(DEV), torch.from_numpy(ix2.astype(np.int32)).to(DEV), None, n, chunk=64)
            assert g.pattern_only and g.arrays()[2] is None
            ref_a = sp.csr_matrix((np.ones(len(ix2)), ix2, ip2), shape=(n, n))       # duplicates add up in the product
        else:
            a.data = rng.standard_normal(a.nnz).astype(np.float32)
            g = CsrGraph.from_scipy(a, DEV, chunk=64)
            ref_a = a.astype(np.float64)
        for width in (2, 5, 8, 24, 64, 100):
            G = rng.standard_normal((n, width)).astype(np.float32)
            S = rng.standard_normal((n, width)).astype(np.float32)
            rs = rng.uniform(0.1, 2.0, n).astype(np.float32)
            ss = rng.uniform(0.1, 2.0, n).astype(np.float32)
            for use_rs, use_sub, use_ss, relu, bf16 in ((0, 0, 0, 0, 0), (1, 0, 0, 1, 0), (1, 1, 1, 0, 0), (0, 1, 0, 1, 0),
                                                        (1, 1, 0, 0, 1)):
                if bf16 and not (8 < width <= 64 and width % 2 == 0):
                    continue
                Gd, Sd = torch.from_numpy(G).to(DEV), torch.from_numpy(S).to(DEV)
                rd, sd = torch.from_numpy(rs).to(DEV), torch.from_numpy(ss).to(DEV)
                Gref = G
                if bf16:
                    Gb = cast_bf16(Gd)
                    Gref = Gb.float().cpu().numpy()                       # bf16 -> fp32 is exact
                o = _lib.SpmmOpts()
                o.row_scale = rd.data_ptr() if use_rs else None
                o.sub, o.ld_sub = (Sd.data_ptr(), width) if use_sub else (None, 0)
                o.sub_scale = sd.data_ptr() if use_ss else None
                o.relu, o.g_bf16 = relu, bf16
                Y = torch.empty(n, width, device=DEV)
                ws = g.workspace(width)
                src = Gb if bf16 else Gd
                st = lib.acm_spmm_ex(g.handle, src.data_ptr(), src.stride(0), width, Y.data_ptr(), width, C.byref(o),
                                     ws.data_ptr(), ws.numel() * 4, torch.cuda.current_stream().cuda_stream)
                _lib.check(st, "acm_spmm_ex")
                ref = ref_a @ Gref.astype(np.float64)
                if use_rs:
                    ref = rs[:, None] * ref
                if use_sub:
                    ref = ref - (ss[:, None] if use_ss else 1.0) * S
                if relu:
                    ref = np.maximum(ref, 0)
                err = np.abs(Y.cpu().numpy() - ref).max()
                assert err < 3e-5 * max(1.0, np.abs(ref).max()), (unit, width, use_rs, use_sub, use_ss, relu, bf16, err)


@pytest.mark.parametrize("model_type,variant,s,ln,f_in,f_out,x_grad,agg", [
    ("acmgcnp", 0, 1, True, 7, 64, False, True), ("acmgcnp", 0, 0, True, 7, 64, False, True),
    ("acmgcnp", 0, 1, True, 33, 64, True, False), ("acmgcnp", 1, 1, True, 64, 5, True, False),
    ("acmgcn", 0, 0, False, 40, 2, True, False), ("acmgcnp", 1, 1, True, 20, 100, True, False),
    ("acmsgc", 0, 0, False, 30, 7, True, False),
    # four channels of two columns: the packed 32-byte rows of the pair-lane narrow gather (forward and transposed), with
    # values and pattern-only, hub row split into window-packed pieces
    ("acmgcnp", 0, 1, True, 64, 2, True, False), ("acmgcnp", 1, 1, True, 64, 2, True, False)])
def test_explicit_value_form_matches_oracle(model_type, variant, s, ln, f_in, f_out, x_grad, agg, monkeypatch):
    """ACM_IMPLICIT=0: the explicit (id, value) operators -- the form every other test leaves for the
    pattern-only one -- against the oracle, and equal to the pattern-only result."""
    a = _run_both(model_type, variant, s, ln, 400, f_in, f_out, 3, x_grad, monkeypatch, agg=agg, implicit=False)
    b = _run_both(model_type, variant, s, ln, 400, f_in, f_out, 3, x_grad, monkeypatch, agg=agg, implicit=True)
    assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("hops,f_out,implicit", [(3, 5, True), (2, 64, True), (3, 2, False)])
def test_acmsgc_khop_chain(hops, f_out, implicit, monkeypatch):
    """ACM-SGC k-hop (BASELINE config 5) as a chain of 1-hop products vs the dense A_low^k of the reference."""
    from test_host_stack_cpu import _khop_chain_case
    _khop_chain_case(DEV, hops, f_out, implicit, monkeypatch)


def test_aggregate_first_not_used_when_illegal(monkeypatch):
    """ACMII (ReLU between projection and filter) or an input that needs a gradient must take
    the literal path."""
    from acm_gnn_amd import functional as AF
    for kwargs in (dict(model_type="acmgcn", variant=1, structure_info=0, x_grad=False),
                   dict(model_type="acmgcnp", variant=1, structure_info=1, x_grad=False),
                   dict(model_type="acmgcn", variant=0, structure_info=0, x_grad=True)):
        timer = AF.KernelTimer()
        AF.set_kernel_timer(timer)
        try:
            _run_both(kwargs["model_type"], kwargs["variant"], kwargs["structure_info"], True, 200, 7, 64, 5,
                      kwargs["x_grad"], monkeypatch, agg=True)
            assert not any(k.startswith("conv_agg") for k in timer.events), kwargs
        finally:
            AF.set_kernel_timer(None)


@pytest.mark.parametrize("model_type,variant,s,ln,f_in,f_out", [
    ("acmgcn", 0, 0, False, 40, 64), ("acmgcn", 1, 0, False, 40, 64), ("acmgcnp", 0, 1, True, 33, 64),
    ("acmgcnp", 1, 1, True, 64, 5), ("acmgcnp", 0, 0, True, 64, 2), ("acmgcnp", 1, 1, True, 20, 100),
    ("acmgcnp", 0, 1, True, 24, 130), ("acmsgc", 0, 0, False, 30, 7), ("acmgcnpp", 0, 0, True, 17, 16)])
def test_literal_path_matches_oracle(model_type, variant, s, ln, f_in, f_out, monkeypatch):
    _run_both(model_type, variant, s, ln, 400, f_in, f_out, 3, True, monkeypatch, agg=False)


@pytest.mark.parametrize("name,f_out,s", [("chameleon", 64, 1), ("chameleon", 5, 0), ("squirrel", 64, 0)])
def test_real_structures(name, f_out, s, monkeypatch):
    """Real graphs (raw self-loops, degree up to 1.9k => long rows are split across work items)."""
    g = load_npz(os.path.join(GOLDEN, f"graph_{name}.npz"))
    n = int(g["n"])
    adj = sp.csr_matrix((np.ones(len(g["adj_un_indices"])), g["adj_un_indices"], g["adj_un_indptr"]), shape=(n, n))
    _run_both("acmgcnp", 0, s, True, n, 24, f_out, 2, True, monkeypatch, agg=False, adj=adj)
    _run_both("acmgcnp", 0, s, True, n, 7, 64, 2, False, monkeypatch, agg=True, adj=adj)
    _run_both("acmgcnp", 0, s, True, n, 24, f_out, 2, True, monkeypatch, agg=False, adj=adj, implicit=False)


@pytest.mark.parametrize("model_type,variant,s,f_out", [("acmgcn", 0, 0, 64), ("acmgcnp", 1, 1, 64), ("acmgcnp", 0, 0, 5)])
def test_sparse_feature_projection_matches_dense_path(model_type, variant, s, f_out, monkeypatch, tune):
    """CSR features (acm_spmm_v forward, transposed handle with src_pos-permuted values backward) vs the
    dense-X GEMM path and vs the oracle, incl. wide F_in and a hub feature column."""
    from acm_gnn_amd import GraphConvolution, SparseFeatures
    from acm_gnn_amd.graph import clear_cache
    tune(agg_first=1)
    clear_cache()
    n, f_in = 300, 700
    adj = _graph(n, 21)
    low, high, un = O.filters_linkx(adj)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(n, f_in, generator=g) * (torch.rand(n, f_in, generator=g) < 0.02)
    x[:, 5] = 1.0                                              # a feature every node has (long transposed row)
    x[7] = 0.0                                                 # a node without features (empty row)
    gout = torch.randn(n, f_out, generator=g)
    torch.manual_seed(1)
    layer = GraphConvolution(f_in, f_out, n, model_type, variant=variant, structure_info=s, attn_layernorm=True)
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in layer.named_parameters()}
    ref = O.layer_forward(params, x, low, high, un if s else None, model_type=model_type, variant=variant,
                          structure_info=s, attn_layernorm=True)
    ref.backward(gout)
    layer = layer.to(DEV)
    adj_d = (low.to(DEV), high.to(DEV), un.to(DEV) if s else None)
    outs = []
    for inp in (x.to(DEV), SparseFeatures.from_torch(x.to(DEV)), x.to_sparse().to(DEV)):
        layer.zero_grad(set_to_none=True)
        out = layer(inp, *adj_d)
        out.backward(gout.to(DEV))
        outs.append((out.detach().cpu(), {k: p.grad.cpu().clone() for k, p in layer.named_parameters() if p.grad is not None}))
    for out, grads in outs:
        assert float((out - ref.detach()).abs().max()) < 2e-5 * max(1.0, float(ref.detach().abs().max()))
        for k, gv in grads.items():
            rg = params[k].grad
            assert float((gv - rg).abs().max()) < 1e-4 * max(1.0, float(rg.abs().max())), k


@pytest.mark.parametrize("model_type,variant,s,ln,khop", [("acmsgc", 0, 0, False, 3), ("acmgcnp", 1, 1, True, 2),
                                                          ("acmgcn", 0, 0, False, 2)])
def test_general_operator_pair(model_type, variant, s, ln, khop):
    """k-hop ACM-SGC as the reference feeds it (A_low^k with an un-powered adj_high) and re-weighted raw
    adjacency: the drop-in takes the general two-operator path and still matches the oracle."""
    from test_host_stack_cpu import _general_case
    _general_case(model_type, variant, s, ln, DEV, khop)


# ---------------------------------------------------------------------------------------------------------
# the trivial layer branches, the ACM-GCN++ residual kernel, acmsnowball
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model_type", ["mlp", "sgc", "gcn"])
@pytest.mark.parametrize("n,f_in,f_out", [(300, 7, 64), (513, 130, 5), (64, 33, 1)])
def test_trivial_branches_match_oracle(model_type, n, f_in, f_out):
    """layers.py:80-85: 'mlp' = X W_I, 'sgc' / 'gcn' = torch.mm(adj_low, X W_L) with a DENSE adj_low -- on acm_gemm,
    forward and both gradients against the oracle."""
    from acm_gnn_amd import GraphConvolution
    adj = _graph(n, 5, hub=False)
    low = torch.from_numpy(np.asarray(O.row_normalize_sp(sp.identity(n) + adj).todense()).astype(np.float32))
    torch.manual_seed(3)
    layer = GraphConvolution(f_in, f_out, n, model_type)
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in layer.named_parameters()}
    g = torch.Generator().manual_seed(4)
    x, gout = torch.randn(n, f_in, generator=g), torch.randn(n, f_out, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = O.layer_forward(params, xr, low, None, None, model_type=model_type)
    ref.backward(gout)
    layer = layer.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    out = layer(xd, low.to(DEV), None, None)
    out.backward(gout.to(DEV))
    assert float((out.detach().cpu() - ref.detach()).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    wname = "weight_mlp" if model_type == "mlp" else "weight_low"
    for k, p in layer.named_parameters():
        if k == wname:
            rg = params[k].grad
            assert float((p.grad.cpu() - rg).abs().max()) < 1e-4 * max(1.0, float(rg.abs().max())), k
        else:
            assert p.grad is None and params[k].grad is None, k          # untouched parameters get no gradient
    assert float((xd.grad.cpu() - xr.grad).abs().max()) < 1e-4 * max(1.0, float(xr.grad.abs().max()))
    if model_type != "mlp":
        with pytest.raises(RuntimeError, match="dense"):
            layer(xd, low.to_sparse().to(DEV), None, None)                # the reference's torch.mm needs a dense adj_low too


def _philox_mask(state, p, tag, n, c):
    import fake_lib

    class D:
        pass
    dd = D()
    dd.p, dd.tag, dd.seed, dd.row_offset = np.float32(p), tag, state.seed, 0
    host = np.array([int(state.step.item())], np.int64)
    dd.step = host.ctypes.data
    return torch.from_numpy(fake_lib.dropout_factors(dd, n, c) > 0).float()


@pytest.mark.parametrize("variant,structure,f_in,p_drop,sparse_x", [(0, 0, 7, 0.3, False), (1, 1, 40, 0.4, False),
                                                                    (0, 1, 300, 0.0, True), (1, 0, 300, 0.0, True),
                                                                    (0, 0, 7, 0.0, False)])
def test_acmgcnpp_residual_branch_matches_oracle(variant, structure, f_in, p_drop, sparse_x, monkeypatch):
    """ACM-GCN++ (models.py:26-27,55-56,73) end to end: the residual Linear + ReLU + dropout as acm_linear_fwd (bias,
    ReLU and the counter-based mask in the GEMM epilogue; CSR features: acm_spmm_v + acm_bias_act) and its backward
    (acm_bias_act_bwd), against the oracle fed the numpy-regenerated masks (tags 0 input, 1 hidden, 2 residual)."""
    from acm_gnn_amd import GCN, SparseFeatures, functional as AF
    from acm_gnn_amd.graph import clear_cache
    clear_cache()
    n = 400
    adj = _graph(n, 9)
    low, high, un = O.filters_linkx(adj)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n, f_in, generator=g)
    if sparse_x:
        x = x * (torch.rand(n, f_in, generator=g) < 0.05)
    y = torch.randint(0, 3, (n,), generator=g)
    idx = torch.arange(0, n, 2)
    torch.manual_seed(5)
    model = GCN(f_in, 64, 3, 2, n, p_drop, "acmgcnpp", structure, variant=bool(variant), attn_layernorm=True)
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()
              if k not in ("fea_param", "xX_param")}
    model = model.to(DEV)
    masks = None
    if p_drop:
        model.fused_dropout, model.dropout_state = True, AF.DropoutState(DEV, seed=99)
        model.dropout_state.step.fill_(7)
        st = model.dropout_state
        masks = {"x": _philox_mask(st, p_drop, 0, n, f_in), "hidden": _philox_mask(st, p_drop, 1, n, 64),
                 "xX": _philox_mask(st, p_drop, 2, n, 64)}
    model.train()
    xin = SparseFeatures.from_torch(x.to(DEV)) if sparse_x else x.to(DEV)
    timer = AF.KernelTimer()
    AF.set_kernel_timer(timer)
    try:
        out = model(xin, low.to(DEV), high.to(DEV), un.to(DEV) if structure else None)
        loss = torch.nn.functional.nll_loss(torch.log_softmax(out, 1)[idx.to(DEV)], y.to(DEV)[idx.to(DEV)])
        loss.backward()
    finally:
        AF.set_kernel_timer(None)
    used = set(k.split("/")[0] for k in timer.events)
    # the residual Linear: a narrow dense input rides one launch behind the first layer (fea1 + xX: acm_linear_fwd_add, masks
    # recomputed by acm_linear_bwd_recompute); else acm_linear_fwd / acm_spmm_v + acm_bias_act, and acm_bias_act_bwd + a product
    narrow = not sparse_x and f_in <= 16
    fused = narrow and p_drop > 0           # (the training forward with counter-based dropout; without dropout: xX first, one-pass backward)
    assert ("bias_act" in used) == sparse_x and ("linear_fwd" in used) == (not sparse_x and not fused), used
    assert ({"linear_fwd_add", "linear_bwd_recompute"} <= used) == fused and ("linear_bwd" in used) == (narrow and not fused), used
    assert ("bias_act_bwd" in used) == (not narrow), used
    ref = O.gcn_forward(params, x, low, high, un if structure else None, model_type="acmgcnpp", variant=bool(variant),
                        structure_info=structure, attn_layernorm=True, dropout=p_drop, training=True, masks=masks)
    ref_loss = O.nll_loss_on(ref, y, idx)
    ref_loss.backward()
    assert float((out.detach().cpu() - ref.detach()).abs().max()) < 3e-5 * max(1.0, float(ref.abs().max()))
    for k, p in model.named_parameters():
        if k not in params:
            continue
        rg = params[k].grad
        if rg is None:
            assert p.grad is None, k
            continue
        assert float((p.grad.cpu() - rg).abs().max()) < 1e-4 * max(1.0, float(rg.abs().max())), k


@pytest.mark.parametrize("variant,nlayers,p_drop", [(0, 2, 0.0), (1, 3, 0.0), (0, 2, 0.35)])
def test_snowball_matches_oracle(variant, nlayers, p_drop):
    """acmsnowball (models.py:38-39,57-64 with the layer's missing nnodes supplied -- quirk Q2): dense stack of ACM
    layers, each hidden layer's ReLU + dropout in its epilogue, against the oracle's literal restatement."""
    from acm_gnn_amd import GCN, functional as AF
    from acm_gnn_amd.graph import clear_cache
    clear_cache()
    n = 350
    adj = _graph(n, 12)
    low, high, _ = O.filters_linkx(adj)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(n, 11, generator=g)
    y = torch.randint(0, 4, (n,), generator=g)
    idx = torch.arange(1, n, 2)
    torch.manual_seed(8)
    model = GCN(11, 24, 4, nlayers, n, p_drop, "acmsnowball", 0, variant=bool(variant))
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()
              if k not in ("fea_param", "xX_param")}
    model = model.to(DEV)
    masks = None
    if p_drop:
        model.fused_dropout, model.dropout_state = True, AF.DropoutState(DEV, seed=5)
        model.dropout_state.step.fill_(3)
        st = model.dropout_state
        masks = {"x": _philox_mask(st, p_drop, 0, n, 11)}
        masks.update({f"h{k}": _philox_mask(st, p_drop, 1 + k, n, 24) for k in range(nlayers)})
    model.train()
    out = model(x.to(DEV), low.to(DEV), high.to(DEV), None)
    loss = torch.nn.functional.nll_loss(torch.log_softmax(out, 1)[idx.to(DEV)], y.to(DEV)[idx.to(DEV)])
    loss.backward()
    ref = O.snowball_forward(params, x, low, high, nlayers=nlayers, variant=bool(variant), dropout=p_drop, training=True,
                             masks=masks)
    O.nll_loss_on(ref, y, idx).backward()
    assert float((out.detach().cpu() - ref.detach()).abs().max()) < 3e-5 * max(1.0, float(ref.abs().max()))
    for k, p in model.named_parameters():
        if k in params and params[k].grad is not None:
            rg = params[k].grad
            assert p.grad is not None, k
            assert float((p.grad.cpu() - rg).abs().max()) < 1e-4 * max(1.0, float(rg.abs().max())), k


@pytest.mark.parametrize("model_type,ln,f_in,implicit,s", [("acmgcnp", True, 7, True, 0), ("acmgcn", False, 8, True, 0),
                                                           ("acmgcnp", True, 3, False, 0), ("acmgcnpp", False, 1, True, 0),
                                                           ("acmgcnp", True, 7, True, 1), ("acmgcnpp", False, 5, False, 1)])
def test_acmii_recompute_on_gather_matches_oracle_and_literal(model_type, ln, f_in, implicit, s, monkeypatch, tune):
    """ACMII first layer with a narrow input (layers.py:94-99): acm_conv_acmii_fwd gathers the 32-byte input rows and
    recomputes relu(x_j [W_L | W_H]) per edge on the matrix pipe; against the oracle (forward, every gradient) and
    against the literal project-then-gather form, on a graph with a split hub row."""
    from acm_gnn_amd import functional as AF
    adj = _graph(700, 31, density=0.04, hub=True)
    timer = AF.KernelTimer()
    AF.set_kernel_timer(timer)
    try:
        tune(acmii_recompute=1, acmii_mask=0)
        a = _run_both(model_type, 1, s, ln, 700, f_in, 64, 31, False, monkeypatch, agg=True, adj=adj, implicit=implicit)
        used = set(k.split("/")[0] for k in timer.events)
        assert "conv_acmii_fwd" in used and "conv_fwd" not in used and "conv_bwd_spmm" in used, used
        timer.events.clear()
        # the mask form of the same layer (acm_conv_acmii_v.hip: bf16 matrix pipe, weight gradients without a transposed
        # product) where it applies -- a pattern-only operator; elsewhere the switch changes nothing
        tune(acmii_mask=1)
        c = _run_both(model_type, 1, s, ln, 700, f_in, 64, 31, False, monkeypatch, agg=True, adj=adj, implicit=implicit)
        used = set(k.split("/")[0] for k in timer.events)
        if implicit:
            assert {"acmii_table", "conv_acmii_v_fwd", "conv_acmii_v_bwd"} <= used and "conv_bwd_spmm" not in used, used
            assert ("spmm_sub" in used) == bool(s), used
        else:
            assert "conv_acmii_fwd" in used and "conv_bwd_spmm" in used and "conv_acmii_v_fwd" not in used, used
        assert float((a - c).abs().max()) < 2e-5 * max(1.0, float(a.abs().max()))
        timer.events.clear()
        tune(acmii_recompute=0)
        b = _run_both(model_type, 1, s, ln, 700, f_in, 64, 31, False, monkeypatch, agg=True, adj=adj, implicit=implicit)
        used = set(k.split("/")[0] for k in timer.events)
        assert "conv_fwd" in used and "conv_acmii_fwd" not in used, used
    finally:
        AF.set_kernel_timer(None)
    assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("model_type,variant,s,f_out", [("acmgcnp", 1, 0, 64), ("acmgcnp", 1, 1, 64), ("acmgcnp", 0, 1, 40),
                                                        ("acmgcn", 1, 0, 130)])
def test_channel_per_pass_backward_matches_oracle(model_type, variant, s, f_out, monkeypatch, tune):
    """K4 with one gathered channel per pass (the form large graphs take: acm_conv_bwd_spmm, EpiBwdLow / High / Struc)
    forced on a small graph with a split hub row, against the oracle; and equal to the single-pass form."""
    adj = _graph(600, 41, density=0.04, hub=True)
    tune(bwd_split=1)
    a = _run_both(model_type, variant, s, True, 600, 30, f_out, 41, True, monkeypatch, agg=False, adj=adj)
    tune(bwd_split=0)
    b = _run_both(model_type, variant, s, True, 600, 30, f_out, 41, True, monkeypatch, agg=False, adj=adj)
    assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("n", [5, 400, 4099])
@pytest.mark.parametrize("model_type,variant,ln,implicit,p_drop", [("acmgcnp", 0, True, True, 0.0), ("acmgcn", 1, False, True, 0.3),
                                                                   ("acmgcnp", 1, True, False, 0.4), ("acmgcnp", 0, True, True, 0.25)])
def test_sixteen_rows_per_wave_local_backward_equals_the_older_kernel(n, model_type, variant, ln, implicit, p_drop, monkeypatch, tune):
    """acm_conv_local16.hip (K3 of the literal layer at F = 64, k = 3: sixteen rows per wave, 16-byte accesses, the head
    recomputed with in-lane sums, the post-op undone by recomputing the mixed row's sign and regenerating the Philox mask)
    against conv_bwd_local_grouped_kernel (ACM_LOCAL16_OFF=1): every gradient of the layer, with and without LayerNorm, both
    variants, pattern-only and explicit operators (g_scale), fused ReLU + counter-based dropout, row counts that are not a
    multiple of 16 / smaller than one wave step -- and against the oracle for the plain case."""
    from acm_gnn_amd import GraphConvolution, functional as AF
    from acm_gnn_amd.graph import clear_cache
    tune(agg_first=0)
    tune(implicit=int(bool(implicit)))
    adj = _graph(max(n, 4), 11, density=min(0.5, 20.0 / max(n, 4)), hub=n > 10)
    n = adj.shape[0]
    low, high, _ = O.filters_linkx(adj)
    g = torch.Generator().manual_seed(n)
    x, gout = torch.randn(n, 24, generator=g).to(DEV), torch.randn(n, 64, generator=g).to(DEV)

    def run(off):
        if off:
            tune(rows16=3)
        else:
            tune(rows16=7)
        clear_cache()
        torch.manual_seed(2)
        layer = GraphConvolution(24, 64, n, model_type, variant=variant, structure_info=0, attn_layernorm=ln).to(DEV)
        st = AF.DropoutState(torch.device(DEV), seed=4)
        xd = x.clone().requires_grad_(True)
        kw = dict(post_relu=True, post_drop=(p_drop, 1, st)) if p_drop else {}
        timer = AF.KernelTimer()
        AF.set_kernel_timer(timer)
        out = layer(xd, low.to(DEV), high.to(DEV), None, **kw)
        out.backward(gout)
        AF.set_kernel_timer(None)
        assert any(k.startswith("conv_bwd_local/F64k3") for k in timer.summary()), list(timer.summary())
        return out.detach(), xd.grad, {k: p.grad for k, p in layer.named_parameters() if p.grad is not None}

    out_a, dx_a, g_a = run(True)
    out_b, dx_b, g_b = run(False)
    assert torch.equal(out_a, out_b)
    assert g_a.keys() == g_b.keys()
    for k in g_a:
        tol = 2e-5 * max(1.0, float(g_a[k].abs().max()))
        assert float((g_a[k] - g_b[k]).abs().max()) < tol, (k, float((g_a[k] - g_b[k]).abs().max()), tol)
    assert float((dx_a - dx_b).abs().max()) < 2e-5 * max(1.0, float(dx_a.abs().max()))
    if not p_drop and n == 400:
        _run_both(model_type, variant, 0, ln, 400, 24, 64, 5, True, monkeypatch, agg=False, implicit=implicit)
