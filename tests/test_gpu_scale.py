"""Scale headroom: the reference's grid includes pokec (1.63 M nodes / 30.6 M edges, 65 features) and
snap-patents (2.92 M nodes, 13.98 M DIRECTED edges, 269 features; ACM-Geometric/sh/run_all_settings.sh:2, train.py:66-67
`--directed` keeps the edge list as it is) -- ten to seventeen times the rows of the benchmark graph.

  * a training step (forward + NLL + backward) on graphs of those shapes -- pattern-only operator for pokec, explicit
    (id, value) operator + explicit transposed CSR for the directed one -- against the CPU oracle fed CSR operands:
    logits, loss and every parameter gradient;
  * every `n_rows * ld >= 2^31` guard of the row-local kernels (32-bit element offsets inside the kernels): driven through
    the C ABI with operands whose leading dimension is large enough to trip it at 70 k rows -- the sixteen-rows-per-wave
    forward stage must decline and the four-rows-per-wave kernel (64-bit row addresses) must return the same numbers as for
    compact operands; the entry points without a 64-bit route must fail loudly (ACM_EUNSUPPORTED), never index out of range.

The graphs are drawn on the GPU (a power-law pairing with torch: the numpy Chung-Lu generator of acm_gnn_amd.data needs two
minutes for 30 M edges; what matters here is the SIZE and the degree skew, not the exact edge count)."""
import ctypes as C
import os
import time

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import acm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _powerlaw_graph_on_gpu(n, n_edges, max_deg, seed, directed):
    """scipy CSR (0/1, no self loops) of ~n_edges distinct (i, j) pairs with power-law endpoints; symmetric unless directed."""
    from acm_gnn_amd import data as D
    g = torch.Generator(device=DEV).manual_seed(seed)
    w = torch.from_numpy(D._powerlaw_weights(n, 2.0 * n_edges / n, max_deg)).to(DEV)
    cdf = torch.cumsum(w / w.sum(), 0)
    perm = torch.randperm(n, generator=g, device=DEV)
    u = perm[torch.searchsorted(cdf, torch.rand(n_edges, generator=g, device=DEV, dtype=torch.float64)).clamp_(0, n - 1)]
    v = perm[torch.searchsorted(cdf, torch.rand(n_edges, generator=g, device=DEV, dtype=torch.float64)).clamp_(0, n - 1)]
    keep = u != v
    u, v = u[keep], v[keep]
    if not directed:
        u, v = torch.cat([u, v]), torch.cat([v, u])
    keys = torch.unique(u * n + v)                         # sorted: row-major order
    rows, cols = (keys // n), (keys % n)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=DEV)
    indptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
    return sp.csr_matrix((np.ones(keys.numel(), np.float32), cols.cpu().numpy().astype(np.int32),
                          indptr.cpu().numpy().astype(np.int64)), shape=(n, n))


def _csr_t(m):
    m = m.tocsr()
    m.sort_indices()
    return torch.sparse_csr_tensor(torch.from_numpy(m.indptr.astype(np.int64)), torch.from_numpy(m.indices.astype(np.int64)),
                                   torch.from_numpy(m.data.astype(np.float32)), size=m.shape)


@pytest.mark.parametrize("name,n,n_edges,max_deg,f_in,n_cls,directed", [
    ("pokec", 1_632_803, 30_622_564, 14_854, 65, 2, False),
    ("snap-patents", 2_923_922, 13_975_788, 800, 269, 5, True)])
def test_training_step_at_linkx_scale_matches_oracle(name, n, n_edges, max_deg, f_in, n_cls, directed):
    import acm_gnn_amd
    from acm_gnn_amd import data as D, functional as AF
    from acm_gnn_amd.graph import CsrGraph, FilterOperators, as_implicit, relabel_by_degree
    t0 = time.time()
    adj = _powerlaw_graph_on_gpu(n, n_edges, max_deg, seed=3, directed=directed)
    low, deg = D.build_filters(adj)                        # A_low = D^-1 (A + I) (train.py:76-81), any A
    rng = np.random.default_rng(1)
    x = torch.from_numpy(D.row_normalize_features(np.abs(rng.standard_normal((n, f_in))).astype(np.float32)))
    y = torch.from_numpy(rng.integers(0, n_cls, n).astype(np.int64))
    tr = np.sort(rng.permutation(n)[: n // 2])
    t_prep = time.time() - t0
    torch.manual_seed(5)
    model = acm_gnn_amd.GCN(f_in, 64, n_cls, 2, n, 0.0, "acmgcnp", 0, variant=False, attn_layernorm=True)
    p0 = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters() if k not in ("fea_param", "xX_param")}
    # ---- the HIP path: the operators as operators_for would build them (pattern-only where the pattern is symmetric, the
    # explicit form + an explicit transposed CSR otherwise; degree relabelling inside the operator for graphs this size)
    ops = relabel_by_degree(as_implicit(FilterOperators(CsrGraph.from_scipy(low, DEV))))
    assert ops.implicit == (not directed) and ops.perm is not None
    model = model.to(DEV)
    xd, yd = x.to(DEV), y.to(DEV)
    w = torch.zeros(n, device=DEV)
    w[torch.from_numpy(tr).to(DEV)] = 1.0 / len(tr)
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
    assert torch.isfinite(logits).all()
    # ---- the oracle on the host cores, CSR operands
    t0 = time.time()
    torch.set_num_threads(min(os.cpu_count() or 8, 64))
    high = (sp.identity(n, dtype=np.float32, format="csr") - low).tocsr()
    ref = O.gcn_forward(p0, x, _csr_t(low), _csr_t(high), None, model_type="acmgcnp", variant=False, structure_info=0,
                        attn_layernorm=True, dropout=0.0, training=True)
    ref_loss = O.nll_loss_on(ref, y, torch.from_numpy(tr))
    if directed:                     # gradients where the transposed operator is a structure of its own; the pattern-only
        ref_loss.backward()          # backward at 10x the benchmark's rows is the twitch-shaped one of test_gpu_fullsize.py
    t_oracle = time.time() - t0
    got = logits.detach().cpu()
    scale = float(ref.detach().abs().max())
    err = float((got - ref.detach()).abs().max())
    assert err < 1e-4 * scale, (name, err, scale)
    assert abs(float(loss) - float(ref_loss)) < 3e-5 * max(1.0, abs(float(ref_loss)))
    worst = 0.0
    for k, prm in model.named_parameters():
        if k not in p0:
            continue
        rg = p0[k].grad
        if rg is None:
            assert prm.grad is None or not directed, k
            assert prm.grad is None or torch.isfinite(prm.grad).all(), k
            continue
        d = float((prm.grad.cpu() - rg).abs().max())
        tol = 3e-4 * float(rg.abs().max()) + 1e-6
        worst = max(worst, d / max(float(rg.abs().max()), 1e-30))
        assert d < tol, (name, k, d, tol)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    import json
    with open(os.path.join(out, f"scale_parity_{name}.json"), "w") as fh:
        json.dump({"nodes": n, "nnz_A_low": int(low.nnz), "max_degree": int(np.diff(low.indptr).max()), "f_in": f_in,
                   "operator": "pattern-only" if ops.implicit else "explicit + transposed CSR", "prep_s": round(t_prep, 1),
                   "oracle_step_s": round(t_oracle, 1), "logits_max_err_over_range": err / scale, "grad_worst_rel": worst,
                   "kernels": sorted(set(k.split("/")[0] for k in timer.events))}, fh, indent=1)


# ---------------------------------------------------------------- the 2^31 guards, through the C ABI
def _agg_case(n, k, seed=0):
    """Operands of the aggregate-first row-local stage over a GIVEN P = A_low X (agg_given): no graph needed."""
    g = torch.Generator().manual_seed(seed)
    f_in, F = 7, 64
    t = dict(agg=torch.randn(n, 8, generator=g), xs=torch.randn(n, 8, generator=g), w=[torch.randn(f_in, F, generator=g) * 0.3 for _ in range(3)],
             vec=[torch.randn(F, generator=g) * 0.2 for _ in range(k)], lnw=[torch.rand(F, generator=g) + 0.5 for _ in range(k)],
             lnb=[torch.randn(F, generator=g) * 0.1 for _ in range(k)], mix=torch.randn(k, k, generator=g),
             ps=torch.randn(n, F, generator=g), ss=torch.randn(n, F, generator=g), deg=torch.rand(n, generator=g) * 5 + 1,
             grad_out=torch.randn(n, F, generator=g))
    t["agg"][:, 7] = 0
    t["xs"][:, 7] = 0
    return {kk: ([u.to(DEV) for u in v] if isinstance(v, list) else v.to(DEV)) for kk, v in t.items()}


def _wide(t, ld):
    """The same [n, c] values inside an [n, ld] buffer: a row pitch of `ld` floats."""
    buf = torch.zeros(t.shape[0], ld, device=DEV)
    view = buf[:, : t.shape[1]]
    view.copy_(t)
    return view


def _agg_fwd(lib, _lib, AF, handle, c, k, out, stats, wide_ps=None):
    p = _lib.ConvAggFwd()
    p.f_in, p.f_pad, p.f_out, p.relu_after, p.relu_mlp, p.layernorm, p.scale = 7, 8, 64, 1, 1, 1, 3.0 if k == 3 else 1.0
    p.xs, p.ld_xs, p.xg, p.ld_xg = c["xs"].data_ptr(), 8, c["xs"].data_ptr(), 8
    p.w_low, p.w_high, p.w_mlp = (t.data_ptr() for t in c["w"])
    p.ld_w = 64
    p.att_vec, p.ln_weight, p.ln_bias = AF._ptr_array(c["vec"]), AF._ptr_array(c["lnw"]), AF._ptr_array(c["lnb"])
    p.att_mix = c["mix"].data_ptr()
    p.out, p.ld_out = out.data_ptr(), out.stride(0)
    p.agg, p.ld_agg, p.agg_given = c["agg"].data_ptr(), 8, 1
    att = torch.empty(out.shape[0], 4, device=DEV)
    p.att, p.n_channels = att.data_ptr(), k
    p.head_stats, p.ld_head_stats = stats.data_ptr(), stats.stride(0)
    if k == 4:
        ps = wide_ps if wide_ps is not None else c["ps"]
        p.sg, p.ld_sg = c["ss"].data_ptr(), 64          # (over the identity operator: A_low S = S, written to ps)
        p.ps, p.ld_ps, p.ss, p.ld_ss, p.deg = ps.data_ptr(), ps.stride(0), c["ss"].data_ptr(), 64, c["deg"].data_ptr()
    ws = torch.empty(1 << 20, device=DEV)
    st = lib.acm_conv_agg_fwd(handle, C.byref(p), C.c_void_p(ws.data_ptr()), ws.numel() * 4, None)
    torch.cuda.synchronize()
    return st, att


def test_rows_times_pitch_beyond_2_31_takes_the_64_bit_kernels_or_fails_loudly():
    from acm_gnn_amd import _lib, functional as AF
    from acm_gnn_amd.graph import CsrGraph
    lib = _lib.load()
    n, ld = 70_000, 32_768                                # n * ld = 2.29e9 >= 2^31 - 1
    assert n * ld >= 2 ** 31 - 1
    eye = CsrGraph.from_scipy(sp.identity(n, dtype=np.float32, format="csr"), DEV)      # (agg_given: the operator is not walked)
    # (1) forward row-local stage, three channels: acm_agg_epi16 declines (acm_conv_agg16.hip: n_rows * ld_max >= INT32_MAX),
    #     agg_epilogue_kernel (64-bit rows) produces the same values as the sixteen-rows-per-wave kernel on compact operands
    c = _agg_case(n, 3)
    out_c, st_c = torch.empty(n, 64, device=DEV), torch.empty(n, 12, device=DEV)
    rc, att_c = _agg_fwd(lib, _lib, AF, eye.handle, c, 3, out_c, st_c)
    assert rc == 0, lib.acm_last_error()
    out_w, st_w = _wide(torch.zeros(n, 64, device=DEV), ld), torch.empty(n, 12, device=DEV)
    rc, att_w = _agg_fwd(lib, _lib, AF, eye.handle, c, 3, out_w, st_w)
    assert rc == 0, lib.acm_last_error()
    scale = float(out_c.abs().max())
    assert float((out_w - out_c).abs().max()) < 2e-5 * scale and float((att_w - att_c).abs().max()) < 2e-5
    assert float((st_w - st_c).abs().max()) < 1e-4 * max(1.0, float(st_c.abs().max()))
    # (2) ... with the structure channel: ps at a pitch that overflows 32-bit offsets has NO 64-bit route in acm_conv_agg_fwd
    #     (agg_fwd_row indexes ps / ss with 32-bit offsets): the call must refuse, not read out of range
    c4 = _agg_case(n, 4, seed=1)
    out4, st4 = torch.empty(n, 64, device=DEV), torch.empty(n, 16, device=DEV)
    rc, _ = _agg_fwd(lib, _lib, AF, eye.handle, c4, 4, out4, st4, wide_ps=_wide(c4["ps"], ld))
    assert rc == 4 and b"32-bit" in lib.acm_last_error(), (rc, lib.acm_last_error())
    rc, _ = _agg_fwd(lib, _lib, AF, eye.handle, c4, 4, out4, st4)
    assert rc == 0 and torch.isfinite(out4).all()
    # (3) backward: a gradient at that pitch is refused by acm_conv_agg_bwd as a whole (its kernels keep 32-bit offsets)
    q = _lib.ConvAggBwd()
    q.f_in, q.f_pad, q.f_out, q.relu_after, q.relu_mlp, q.layernorm, q.scale, q.n_channels = 7, 8, 64, 1, 1, 1, 3.0, 3
    go_w = _wide(c["grad_out"], ld)
    q.grad_out, q.ld_grad_out = go_w.data_ptr(), go_w.stride(0)
    q.agg, q.ld_agg, q.xs, q.ld_xs = c["agg"].data_ptr(), 8, c["xs"].data_ptr(), 8
    q.w_low, q.w_high, q.w_mlp = (t.data_ptr() for t in c["w"])
    q.ld_w = 64
    q.att_vec, q.ln_weight, q.ln_bias = AF._ptr_array(c["vec"]), AF._ptr_array(c["lnw"]), AF._ptr_array(c["lnb"])
    q.att_mix = c["mix"].data_ptr()
    q.head_stats, q.ld_head_stats = st_c.data_ptr(), 12
    d_params = torch.empty(3 * 7 * 64 + 9 * 64 + 9, device=DEV)
    q.d_params = d_params.data_ptr()
    nbytes = C.c_size_t()
    assert lib.acm_conv_agg_bwd_workspace_bytes(n, 7, 64, C.byref(nbytes)) == 0
    ws = torch.empty(nbytes.value // 4 + 1, device=DEV)
    rc = lib.acm_conv_agg_bwd(n, C.byref(q), C.c_void_p(ws.data_ptr()), ws.numel() * 4, None)
    assert rc == 4 and b"32-bit" in lib.acm_last_error(), (rc, lib.acm_last_error())
    q.grad_out, q.ld_grad_out = c["grad_out"].data_ptr(), 64
    rc = lib.acm_conv_agg_bwd(n, C.byref(q), C.c_void_p(ws.data_ptr()), ws.numel() * 4, None)
    torch.cuda.synchronize()
    assert rc == 0 and torch.isfinite(d_params).all()
    # (4) K3 of the literal layer: acm_conv_bwd_local refuses a pitch beyond 32-bit offsets as a whole (both of its kernels
    #     index with them), and runs on the compact operands
    g = torch.Generator().manual_seed(4)
    pre, zi = torch.randn(n, 128, generator=g).to(DEV), torch.randn(n, 64, generator=g).to(DEV)
    for tag, go in (("compact", c["grad_out"]), ("wide", go_w)):
        b = _lib.ConvBwdLocal()
        b.f_out, b.n_channels, b.relu_after, b.relu_mlp, b.layernorm, b.scale = 64, 3, 1, 1, 1, 3.0
        b.grad_out, b.ld_grad_out = go.data_ptr(), go.stride(0)
        b.pre, b.ld_pre, b.s_mlp, b.ld_s_mlp = pre.data_ptr(), 128, zi.data_ptr(), 64
        b.att_vec, b.ln_weight, b.ln_bias = AF._ptr_array(c["vec"]), AF._ptr_array(c["lnw"]), AF._ptr_array(c["lnb"])
        b.att_mix = c["mix"].data_ptr()
        gl, dz = torch.empty(n, 128, device=DEV), torch.empty(n, 192, device=DEV)
        b.g_low, b.ld_g_low, b.g_high, b.ld_g_high = gl.data_ptr(), 128, gl.data_ptr() + 256, 128
        b.g_mlp, b.ld_g_mlp = dz.data_ptr() + 512, 192
        flat = torch.empty(9 * 64 + 9, device=DEV)
        dv = [flat[i * 64:(i + 1) * 64] for i in range(3)]
        dlw = [flat[(3 + i) * 64:(4 + i) * 64] for i in range(3)]
        dlb = [flat[(6 + i) * 64:(7 + i) * 64] for i in range(3)]
        b.d_att_vec, b.d_ln_weight, b.d_ln_bias = AF._ptr_array(dv), AF._ptr_array(dlw), AF._ptr_array(dlb)
        b.d_att_mix = flat[9 * 64:].data_ptr()
        assert lib.acm_conv_bwd_local_workspace_bytes(n, 64, 3, C.byref(nbytes)) == 0
        wsb = torch.empty(nbytes.value // 4 + 1, device=DEV)
        rc = lib.acm_conv_bwd_local(n, C.byref(b), C.c_void_p(wsb.data_ptr()), wsb.numel() * 4, None)
        torch.cuda.synchronize()
        if tag == "wide":
            assert rc == 4 and b"2^31" in lib.acm_last_error(), (rc, lib.acm_last_error())
        else:
            assert rc == 0 and torch.isfinite(gl).all() and torch.isfinite(flat).all(), lib.acm_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False], ids=["one_kernel", "products_then_head"])
@pytest.mark.parametrize("model_type,f_in,p_drop,ln", [("acmgcnp", 128, 0.3, True), ("acmgcn", 65, 0.0, False), ("acmgcnp", 40, 0.2, True)])
def test_aggregate_first_for_wide_inputs_matches_the_literal_form_and_the_oracle(model_type, f_in, p_drop, ln, fused, tune):
    """functional._AcmAggWide (round 5): first layers with 16 < F_in <= 128 dense features (arXiv-year 128, pokec 65) gather
    P = A_low drop(X) once and run no transposed gather in the backward.  Logits, loss and every gradient against the literal
    form (tuning rewrites bit 1 off) at fp32 re-association level, and the logits against the oracle on sampled rows.
    ``fused``: projections + head behind the gather as ONE kernel (acm_conv_aggw_fwd: split-bf16 MFMA products feeding the
    sixteen-rows head in registers) or as two products + acm_conv_head_fwd (tuning rewrites bit 8 off)."""
    import scipy.sparse as sp
    from acm_gnn_amd import GCN, data as D, functional as AF, train as T
    from acm_gnn_amd.distributed import make_sharded_operators
    from oracle import acm_oracle as O
    n = 30000
    rng = np.random.default_rng(5)
    m = n * 18                                       # mean degree ~36: the form is taken from 12 on
    r, c = rng.integers(0, n, m), rng.integers(0, n, m)
    adj = sp.csr_matrix((np.ones(m, np.float32), (r, c)), shape=(n, n))
    adj = ((adj + adj.T) > 0).astype(np.float32).tocsr()
    adj.setdiag(0)
    adj.eliminate_zeros()
    low, deg = D.build_filters(adj)
    ops = make_sharded_operators(low, deg, torch.device(DEV))
    x = torch.randn(n, f_in, generator=torch.Generator().manual_seed(1)).to(DEV)
    y = torch.randint(0, 4, (n,), generator=torch.Generator().manual_seed(2)).to(DEV)
    w = T.row_weights(torch.arange(0, n, 2, device=DEV), n)

    def run(agg):
        tune(agg_first=int(agg), aggw_fused=int(fused))
        torch.manual_seed(3)
        model = GCN(f_in, 64, 4, 2, n, p_drop, model_type, 0, variant=False, attn_layernorm=ln).to(DEV)
        model.train()
        if p_drop > 0:
            model.fused_dropout, model.dropout_state = True, AF.DropoutState(torch.device(DEV), seed=7)
        timer = AF.KernelTimer()
        AF.set_kernel_timer(timer)
        out = model(x, ops)
        loss = AF.masked_nll(out, y, w)
        loss.backward()
        AF.set_kernel_timer(None)
        labels = set(timer.summary())
        return model, out.detach(), float(loss), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, labels

    ma, out_a, loss_a, g_a, lab_a = run(True)
    mb, out_b, loss_b, g_b, lab_b = run(False)
    assert any(k.startswith("conv_aggw/" if fused else "conv_head/") for k in lab_a), lab_a
    assert not any(k.startswith(("conv_head/", "conv_aggw/")) for k in lab_b)
    assert sum(k.startswith("conv_bwd_spmm/") for k in lab_a) == 1 and sum(k.startswith("conv_bwd_spmm/") for k in lab_b) == 2
    scale = float(out_b.abs().max())
    assert float((out_a - out_b).abs().max()) <= 2e-5 * scale + 1e-6
    np.testing.assert_allclose(loss_a, loss_b, rtol=2e-5)
    assert g_a.keys() == g_b.keys()
    for k in g_a:
        tol = 2e-4 * float(g_b[k].abs().max()) + 1e-8
        assert float((g_a[k] - g_b[k]).abs().max()) <= tol, k
    if p_drop == 0:                                  # eval-mode logits against the oracle (CPU, CSR operands) on sampled rows
        ma.eval()
        with torch.no_grad():
            got = ma(x, ops).cpu()
        params = {k: v.detach().cpu() for k, v in ma.state_dict().items() if k.startswith("gcns.")}
        lo = torch.sparse_csr_tensor(torch.from_numpy(low.indptr.astype(np.int64)), torch.from_numpy(low.indices.astype(np.int64)),
                                     torch.from_numpy(low.data.astype(np.float32)), size=low.shape)
        hi_sp = (sp.identity(n, format="csr") - low).tocsr()
        hi = torch.sparse_csr_tensor(torch.from_numpy(hi_sp.indptr.astype(np.int64)), torch.from_numpy(hi_sp.indices.astype(np.int64)),
                                     torch.from_numpy(hi_sp.data.astype(np.float32)), size=low.shape)
        ref = O.gcn_forward(params, x.cpu(), lo, hi, None, model_type=model_type, variant=False, structure_info=0, attn_layernorm=ln)
        assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n,f_in,ln,relu_after,post", [(4099, 128, True, True, "drop"), (1000, 65, False, True, "scale"), (37, 20, True, False, None),
                                                       (20000, 100, True, True, None), (257, 5, True, True, None),
                                                       (64, 16, False, True, "drop"), (1, 33, True, True, None)])
def test_wide_aggregate_first_kernel_against_float64(n, f_in, ln, relu_after, post):
    """acm_conv_aggw_fwd through the C ABI on random rows: pre_L = P W_L, pre_H = (Xd - P) W_H, Z_I = Xd W_I at fp32 accuracy
    (float64 products as the referee: the split-bf16 products keep six of nine partial products), then the head of
    acm_conv_head_fwd on the SAME pre-activations (the stored ones), so that the head is compared at fp32 re-association level
    whatever the ReLU does to entries near zero.  Ragged row counts, padded F_in, post-op forms."""
    import ctypes as C
    from acm_gnn_amd import _lib, functional as AF
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + f_in)
    fp = -(-f_in // 4) * 4
    P = torch.zeros(n, fp)
    X = torch.zeros(n, fp)
    P[:, :f_in] = torch.randn(n, f_in, generator=g)
    X[:, :f_in] = torch.randn(n, f_in, generator=g)
    W = [torch.randn(f_in, 64, generator=g) * 0.2 for _ in range(3)]
    vecs = [torch.randn(64, generator=g) for _ in range(3)]
    lnw = [torch.rand(64, generator=g) + 0.5 for _ in range(3)]
    lnb = [torch.randn(64, generator=g) * 0.1 for _ in range(3)]
    mix = torch.randn(3, 3, generator=g)
    dev = lambda t: t.to(DEV).contiguous()
    Pd, Xd, Wd, vd, lwd, lbd, mixd = dev(P), dev(X), [dev(w) for w in W], [dev(v) for v in vecs], [dev(t) for t in lnw], \
        [dev(t) for t in lnb], dev(mix)
    keep = dev((torch.rand(n, 64, generator=g) > 0.3).float() / 0.7) if post == "scale" else None
    state = AF.DropoutState(torch.device(DEV), seed=5) if post == "drop" else None

    def block(out, pre, att):
        p = _lib.ConvFwd()
        p.f_out, p.n_channels, p.relu_after, p.relu_mlp, p.layernorm, p.scale = 64, 3, int(relu_after), 1, int(ln), 3.0
        p.att_vec = AF._ptr_array(vd)
        if ln:
            p.ln_weight, p.ln_bias = AF._ptr_array(lwd), AF._ptr_array(lbd)
        p.att_mix = mixd.data_ptr()
        p.out, p.ld_out, p.pre, p.ld_pre, p.att = out.data_ptr(), 64, pre.data_ptr(), 128, att.data_ptr()
        p.post_relu = 1
        if keep is not None:
            p.post_scale, p.ld_post_scale = keep.data_ptr(), 64
        if state is not None:
            p.post_drop = AF._drop_spec((0.4, 1, state), 0)
        return p
    out, pre, att = torch.empty(n, 64, device=DEV), torch.empty(n, 128, device=DEV), torch.empty(n, 4, device=DEV)
    zi = torch.empty(n, 64, device=DEV)
    p = block(out, pre, att)
    st = lib.acm_conv_aggw_fwd(n, f_in, fp, Pd.data_ptr(), fp, Xd.data_ptr(), fp, Wd[0].data_ptr(), Wd[1].data_ptr(), Wd[2].data_ptr(),
                               64, zi.data_ptr(), 64, C.byref(p), AF._stream())
    _lib.check(st, "acm_conv_aggw_fwd")
    torch.cuda.synchronize()
    P64, X64, W64 = P[:, :f_in].double(), X[:, :f_in].double(), [w.double() for w in W]
    want = [P64 @ W64[0], (X64 - P64) @ W64[1], X64 @ W64[2]]
    got = [pre[:, :64].cpu().double(), pre[:, 64:].cpu().double(), zi.cpu().double()]
    for wv, gv, name in zip(want, got, ("pre_L", "pre_H", "Z_I")):
        # an fp32 FMA chain over K terms errs by ~K eps |terms|: the split products must stay inside that
        bound = 4e-7 * float((P64.abs() + X64.abs()).max() * max(w.abs().max() for w in W64)) * f_in
        assert float((wv - gv).abs().max()) <= bound, (name, float((wv - gv).abs().max()), bound)
    # the head on the stored pre-activations: acm_conv_head_fwd (g_low = pre_L, s_high - g_high = pre_H - 0, s_mlp = Z_I)
    out2, pre2, att2 = torch.empty_like(out), torch.empty_like(pre), torch.empty_like(att)
    zero = torch.zeros(n, 64, device=DEV)
    q = block(out2, pre2, att2)
    q.g_low, q.ld_g_low, q.g_high, q.ld_g_high = pre.data_ptr(), 128, zero.data_ptr(), 64
    q.s_high, q.ld_s_high, q.s_mlp, q.ld_s_mlp = pre.data_ptr() + 256, 128, zi.data_ptr(), 64
    _lib.check(lib.acm_conv_head_fwd(n, C.byref(q), AF._stream()), "acm_conv_head_fwd")
    torch.cuda.synchronize()
    assert torch.equal(pre2, pre)
    torch.testing.assert_close(att, att2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(out, out2, rtol=2e-5, atol=2e-6 * float(out2.abs().max()))
    assert (out2 == 0).float().mean() < 0.9 and torch.isfinite(out).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n,f_in,ln,post", [(4099, 128, True, "drop"), (1000, 65, False, None), (37, 20, True, None), (20000, 100, True, "relu"),
                                            (129, 5, True, None), (70000, 128, True, "drop")])
def test_wide_aggregate_first_backward_kernel_against_k3_and_float64_products(n, f_in, ln, post):
    """acm_conv_aggw_bwd through the C ABI: its head-parameter gradients against acm_conv_bwd_local's on the same inputs
    (re-association level), its weight gradients against float64 products of P, Xd - P, Xd with the G tables acm_conv_bwd_local
    wrote (the fused kernel never stores G).  Ragged row counts (slabs of 128), padded F_in, several slabs per workgroup."""
    import ctypes as C
    from acm_gnn_amd import _lib, functional as AF
    lib = _lib.load()
    g = torch.Generator().manual_seed(3 * n + f_in)
    fp = -(-f_in // 4) * 4
    P, X = torch.zeros(n, fp), torch.zeros(n, fp)
    P[:, :f_in] = torch.randn(n, f_in, generator=g)
    X[:, :f_in] = torch.randn(n, f_in, generator=g)
    pre = torch.randn(n, 128, generator=g)
    zi = torch.randn(n, 64, generator=g)
    go = torch.randn(n, 64, generator=g)
    vecs = [torch.randn(64, generator=g) for _ in range(3)]
    lnw = [torch.rand(64, generator=g) + 0.5 for _ in range(3)]
    lnb = [torch.randn(64, generator=g) * 0.1 for _ in range(3)]
    mix = torch.randn(3, 3, generator=g)
    dev = lambda t: t.to(DEV).contiguous()
    Pd, Xd, pred, zid, god, mixd = dev(P), dev(X), dev(pre), dev(zi), dev(go), dev(mix)
    vd, lwd, lbd = [dev(v) for v in vecs], [dev(t) for t in lnw], [dev(t) for t in lnb]
    state = AF.DropoutState(torch.device(DEV), seed=5) if post == "drop" else None

    def block(flat):
        q = _lib.ConvBwdLocal()
        q.f_out, q.n_channels, q.relu_after, q.relu_mlp, q.layernorm, q.scale = 64, 3, 1, 1, int(ln), 3.0
        q.grad_out, q.ld_grad_out, q.pre, q.ld_pre, q.s_mlp, q.ld_s_mlp = god.data_ptr(), 64, pred.data_ptr(), 128, zid.data_ptr(), 64
        q.att_vec = AF._ptr_array(vd)
        if ln:
            q.ln_weight, q.ln_bias = AF._ptr_array(lwd), AF._ptr_array(lbd)
        q.att_mix = mixd.data_ptr()
        d_vec, d_lnw, d_lnb, d_mix = AF._flat_views(flat, 0, 3, 64, ln)
        q.d_att_vec = AF._ptr_array(d_vec)
        if ln:
            q.d_ln_weight, q.d_ln_bias = AF._ptr_array(d_lnw), AF._ptr_array(d_lnb)
        q.d_att_mix = d_mix.data_ptr()
        q.post_relu = int(post is not None)
        if state is not None:
            q.post_drop = AF._drop_spec((0.4, 1, state), 0)
        return q
    nflat = 3 * 64 + (6 * 64 if ln else 0) + 9
    # reference: K3 with its G tables in memory
    flat_ref = torch.zeros(nflat, device=DEV)
    gcat = torch.empty(n, 192, device=DEV)
    q = block(flat_ref)
    q.g_low, q.ld_g_low, q.g_high, q.ld_g_high, q.g_mlp, q.ld_g_mlp = gcat.data_ptr(), 192, gcat.data_ptr() + 256, 192, gcat.data_ptr() + 512, 192
    nb = C.c_size_t()
    _lib.check(lib.acm_conv_bwd_local_workspace_bytes(n, 64, 3, C.byref(nb)))
    ws = torch.empty(max(nb.value // 4, 1), device=DEV)
    _lib.check(lib.acm_conv_bwd_local(n, C.byref(q), ws.data_ptr(), ws.numel() * 4, AF._stream()), "acm_conv_bwd_local")
    # the fused kernel
    flat = torch.zeros(nflat, device=DEV)
    dw = torch.full((3, f_in, 64), float("nan"), device=DEV)
    q2 = block(flat)
    _lib.check(lib.acm_conv_aggw_bwd_workspace_bytes(n, fp, C.byref(nb)))
    ws2 = torch.empty(max(nb.value // 4, 1), device=DEV)
    st = lib.acm_conv_aggw_bwd(n, f_in, fp, Pd.data_ptr(), fp, Xd.data_ptr(), fp, C.byref(q2), dw[0].data_ptr(), dw[1].data_ptr(),
                               dw[2].data_ptr(), 64, ws2.data_ptr(), ws2.numel() * 4, AF._stream())
    _lib.check(st, "acm_conv_aggw_bwd")
    torch.cuda.synchronize()
    assert torch.isfinite(dw).all() and torch.isfinite(flat).all()
    scale = float(flat_ref.abs().max())
    assert float((flat - flat_ref).abs().max()) <= 2e-5 * scale * max(1.0, (n / 4096) ** 0.5) + 1e-6
    G = gcat.cpu().double()
    P64, X64 = P[:, :f_in].double(), X[:, :f_in].double()
    want = torch.stack([P64.T @ G[:, :64], (X64 - P64).T @ G[:, 64:128], X64.T @ G[:, 128:]])
    # a sum of n products of O(1) factors in fp32: ~ eps * sqrt(n) * |term| per element (the accumulators are fp32)
    mag = float((P64.abs().max() + X64.abs().max()) * G.abs().max())
    assert float((dw.cpu().double() - want).abs().max()) <= 3e-7 * mag * n ** 0.5 + 1e-6 * float(want.abs().max())
