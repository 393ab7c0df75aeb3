"""HIP kernels vs. CPU references through the C ABI (runs on the MI355X box)."""
import os

import numpy as np
import pytest
from acm_gnn_amd import tuning
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _rand_csr(n_rows, n_cols, density, seed, hub_rows=(), empty_rows=()):
    rng = np.random.default_rng(seed)
    m = sp.random(n_rows, n_cols, density=density, random_state=rng, format="lil", dtype=np.float32)
    for r in hub_rows:
        cols = rng.choice(n_cols, size=min(n_cols, hub_rows[r]), replace=False)
        m[r, cols] = rng.standard_normal(len(cols)).astype(np.float32)
    for r in empty_rows:
        m[r, :] = 0
    m = m.tocsr()
    m.eliminate_zeros()
    m.sort_indices()
    return m


@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (5, 3, 2), (64, 64, 32), (65, 63, 33), (300, 192, 7),
                                   (1000, 6, 64), (7, 192, 5000), (64, 6, 3000), (130, 70, 2089),
                                   (16, 256, 100), (17, 17, 17), (2048, 21, 129), (3, 700, 9)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_gemm_matches_fp64(m, n, k, ta, tb):
    from acm_gnn_amd import functional as AF
    g = torch.Generator().manual_seed(m * 131 + n * 17 + k + 2 * ta + tb)
    a = torch.randn((k, m) if ta else (m, k), generator=g)
    b = torch.randn((n, k) if tb else (k, n), generator=g)
    ref = (a.double().T if ta else a.double()) @ (b.double().T if tb else b.double())
    out = AF.gemm(a.to(DEV), b.to(DEV), trans_a=bool(ta), trans_b=bool(tb)).cpu()
    scale = (a.abs().double().T if ta else a.abs().double()) @ (b.abs().double().T if tb else b.abs().double())
    err = (out.double() - ref).abs()
    assert float((err / (scale + 1e-30)).max()) < 2e-6, float(err.max())
    out_relu = AF.gemm(a.to(DEV), b.to(DEV), trans_a=bool(ta), trans_b=bool(tb), relu=True).cpu()
    assert torch.equal(out_relu, out.clamp_min(0))


@pytest.mark.parametrize("m,n,k,blocks", [(64, 6, 3000, 3), (7, 192, 5000, 3), (64, 15, 5201, 3), (20, 12, 40, 4), (300, 192, 7, 3)])
def test_gemm_column_block_output(m, n, k, blocks):
    """acm_gemm_blocks: the same product delivered as contiguous column blocks (split-K and direct stores)."""
    from acm_gnn_amd import functional as AF
    g = torch.Generator().manual_seed(m + n + k)
    a, b = torch.randn(k, m, generator=g), torch.randn(k, n, generator=g)
    whole = AF.gemm(a.to(DEV), b.to(DEV), trans_a=True)
    parts = AF.gemm(a.to(DEV), b.to(DEV), trans_a=True, col_blocks=blocks)
    assert parts.shape == (blocks, m, n // blocks) and parts.is_contiguous()
    assert torch.equal(torch.cat(list(parts), dim=1), whole)


@pytest.mark.parametrize("n,f_in,q", [(1000, 64, 6), (777, 64, 15), (513, 40, 9), (300, 130, 3), (4097, 64, 12), (5, 7, 6),
                                      (168114, 64, 6)])
def test_proj_bwd_matches_fp64(n, f_in, q):
    """acm_proj_bwd: dX = dZ W^T and dW = X^T dZ in one pass, vs fp64; dW delivered as three column blocks."""
    from acm_gnn_amd import functional as AF
    g = torch.Generator().manual_seed(n + f_in + q)
    x, dz, w = torch.randn(n, f_in, generator=g), torch.randn(n, q, generator=g), torch.randn(f_in, q, generator=g)
    dw = torch.empty(3, f_in, q // 3, device=DEV)
    w3 = [w[:, j * (q // 3):(j + 1) * (q // 3)].contiguous().to(DEV) for j in range(3)]
    dx = AF.proj_bwd(x.to(DEV), dz.to(DEV), w3, dw)
    ref_dx = dz.double() @ w.double().T
    ref_dw = x.double().T @ dz.double()
    sc_dx = dz.abs().double() @ w.abs().double().T
    sc_dw = x.abs().double().T @ dz.abs().double()
    assert float(((dx.cpu().double() - ref_dx).abs() / (sc_dx + 1e-30)).max()) < 2e-6
    got = torch.cat(list(dw.cpu()), dim=1).double()
    assert float(((got - ref_dw).abs() / (sc_dw + 1e-30)).max()) < 3e-6
    dw2 = torch.empty_like(dw)
    dx2 = AF.proj_bwd(x.to(DEV), dz.to(DEV), w3, dw2)
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2)            # deterministic
    # unaligned / strided operands take the scalar path
    xs = torch.randn(n, f_in + 3, generator=g)[:, 1:f_in + 1]
    dxs = AF.proj_bwd(xs.to(DEV), dz.to(DEV), w3, dw2)
    assert float(((dxs.cpu().double() - ref_dx).abs() / (sc_dx + 1e-30)).max()) < 2e-6


@pytest.mark.parametrize("n,f_in,f", [(1000, 64, 2), (777, 64, 5), (513, 40, 3), (300, 130, 1), (4097, 64, 4), (5, 7, 2),
                                      (168114, 64, 2), (200, 200, 8)])
@pytest.mark.parametrize("relu", [False, True])
def test_proj_fwd_matches_fp64(n, f_in, f, relu):
    """acm_proj_fwd: [Z_lh | Z_i] = relu?(X [W_L | W_H | W_I]) from the three weight matrices, vs fp64."""
    from acm_gnn_amd import functional as AF
    g = torch.Generator().manual_seed(n + f_in + f)
    x = torch.randn(n, f_in, generator=g)
    w3 = [torch.randn(f_in, f, generator=g) for _ in range(3)]
    zlh, zi = torch.empty(n, 2 * f, device=DEV), torch.empty(n, f, device=DEV)
    AF.proj_fwd(x.to(DEV), [w.to(DEV) for w in w3], zlh, zi, relu=relu)
    ref = x.double() @ torch.cat(w3, 1).double()
    scale = x.abs().double() @ torch.cat(w3, 1).abs().double()
    if relu:
        ref = ref.clamp_min(0)
    got = torch.cat([zlh, zi], 1).cpu().double()
    assert float(((got - ref).abs() / (scale + 1e-30)).max()) < 2e-6


@pytest.mark.parametrize("m,n,k,split", [(1000, 6, 64, 4), (168114, 6, 64, 4), (300, 24, 7, 16), (64, 12, 3000, 8), (5, 3, 2, 2)])
def test_gemm_two_matrix_output(m, n, k, split, monkeypatch, tune):
    """acm_gemm_split: columns [0, split) to one matrix, the rest to another (direct and split-K stores)."""
    from acm_gnn_amd import functional as AF
    g = torch.Generator().manual_seed(m + n + k)
    a, b = torch.randn(m, k, generator=g), torch.randn(k, n, generator=g)
    tune(gemm_forms=tuning.kernel()["gemm_forms"] & ~6)          # the single-output product on the same kernel family (fp32 MFMA chain)
    whole = AF.gemm(a.to(DEV), b.to(DEV), relu=True)
    tune(gemm_forms=tuning.kernel()["gemm_forms"] | 6)
    o1 = torch.full((m, split + 3), 7.0, device=DEV)[:, :split]          # strided destinations keep their padding
    o2 = torch.full((m, n - split), 7.0, device=DEV)
    AF.gemm_split(a.to(DEV), b.to(DEV), o1, o2, relu=True)
    assert torch.equal(o1, whole[:, :split]) and torch.equal(o2, whole[:, split:])


def test_gemm_is_an_fmaf_chain_in_k_order():
    """f32 MFMA == k-ordered fmaf chain: integer-valued inputs must be exact."""
    from acm_gnn_amd import functional as AF
    g = torch.Generator().manual_seed(0)
    a = torch.randint(-8, 9, (200, 77), generator=g).float()
    b = torch.randint(-8, 9, (77, 45), generator=g).float()
    assert torch.equal(AF.gemm(a.to(DEV), b.to(DEV)).cpu(), a @ b)


def test_gemm_strided_views_and_errors():
    from acm_gnn_amd import functional as AF
    a = torch.randn(50, 30, device=DEV)
    b = torch.randn(30, 40, device=DEV)
    out = torch.zeros(50, 64, device=DEV)
    AF.gemm(a, b, out=out[:, 8:48])
    torch.testing.assert_close(out[:, 8:48].cpu(), (a.cpu().double() @ b.cpu().double()).float(), rtol=1e-5, atol=1e-5)
    assert float(out[:, :8].abs().max()) == 0 and float(out[:, 48:].abs().max()) == 0
    with pytest.raises(ValueError):
        AF.gemm(a, torch.randn(31, 4, device=DEV))
    with pytest.raises(RuntimeError):
        AF.gemm(a.cpu(), b)


@pytest.mark.parametrize("width", [1, 2, 3, 4, 5, 7, 8, 9, 16, 33, 64, 65, 128, 192, 256, 300])
@pytest.mark.parametrize("chunk", [0, 16])
def test_spmm_matches_scipy(width, chunk):
    from acm_gnn_amd import functional as AF
    from acm_gnn_amd.graph import CsrGraph
    m = _rand_csr(517, 400, 0.03, seed=width, hub_rows={3: 390, 100: 200, 516: 70}, empty_rows=(0, 7, 515))
    g = CsrGraph.from_scipy(m, DEV, chunk=chunk)
    assert g.nnz == m.nnz and g.max_degree == int(np.diff(m.indptr).max())
    if chunk:
        assert g.n_long_rows >= 3 and g.n_partial_slots >= 6
    dense = torch.randn(400, width, generator=torch.Generator().manual_seed(1))
    out = AF.spmm(g, dense.to(DEV)).cpu().numpy()
    ref = m.astype(np.float64) @ dense.double().numpy()
    scale = abs(m).astype(np.float64) @ dense.abs().double().numpy()
    assert np.max(np.abs(out - ref) / (scale + 1e-20) * (scale > 0)) < 2e-6
    assert np.all(out[[0, 7, 515]] == 0)                       # empty rows write exact zeros
    # determinism: same bits on a second launch (no float atomics anywhere)
    out2 = AF.spmm(g, dense.to(DEV)).cpu().numpy()
    assert np.array_equal(out, out2)


@pytest.mark.parametrize("chunk,hubs", [(16, {3: 390, 100: 200, 516: 70}), (16, {0: 17, 1: 17, 2: 300, 9: 33}),
                                        (64, {5: 400, 6: 65, 7: 129, 8: 1000}), (0, {1: 390}),
                                        (8, {2: 1100, 7: 600, 20: 300, 400: 90})])
def test_work_list_layout(chunk, hubs):
    """Long rows are min(16, ceil(deg / chunk)) pieces, packed into windows of 16 work items that no row straddles
    (the kernels that take one window per workgroup round finish those rows through LDS); a row that sixteen pieces of
    4 x chunk do not cover takes ceil(deg / (32 chunk)) whole windows (the last case: 5 and 3 windows for the rows of
    1100 and 600): the counts the library reports equal the restatement in tests/fake_lib.py."""
    import ctypes as C
    import fake_lib
    from acm_gnn_amd import _lib
    from acm_gnn_amd.graph import CsrGraph
    m = _rand_csr(1200, 1100, 0.01, seed=4, hub_rows=hubs)
    g = CsrGraph.from_scipy(m, DEV, chunk=chunk)
    fake = fake_lib.FakeLib()
    h = C.c_void_p()
    ip, ix, v = (np.ascontiguousarray(a) for a in (m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float32)))
    assert fake.acm_csr_create(m.shape[0], m.shape[1], m.nnz, ip.ctypes.data, ix.ctypes.data, v.ctypes.data, chunk, C.byref(h)) == 0
    info = _lib.CsrInfo()
    assert fake.acm_csr_info(h, C.byref(info)) == 0
    assert (g.chunk, g.n_long_rows, g.n_partial_slots, g.n_items, g.max_degree) == \
        (info.chunk, info.n_long_rows, info.n_partial_slots, info.n_items, info.max_degree)
    assert g.n_partial_slots % 16 == 0 and (g.n_long_rows == 0) == (g.n_partial_slots == 0)
    for width in (2, 8, 64):                                   # narrow (cooperative) and wide (slots + fix-up) gathers
        from acm_gnn_amd import functional as AF
        dense = torch.randn(1100, width, generator=torch.Generator().manual_seed(width))
        out = AF.spmm(g, dense.to(DEV)).cpu().numpy()
        ref = m.astype(np.float64) @ dense.double().numpy()
        scale = abs(m).astype(np.float64) @ dense.abs().double().numpy()
        assert np.max(np.abs(out - ref) / (scale + 1e-20) * (scale > 0)) < 2e-6


def test_spmm_ignores_nonfinite_rows_it_never_references():
    from acm_gnn_amd import functional as AF
    from acm_gnn_amd.graph import CsrGraph
    m = sp.csr_matrix(np.array([[0, 1, 1, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 0]], np.float32))
    g = CsrGraph.from_scipy(m, DEV)
    for width in (2, 4, 8, 64):
        dense = torch.ones(4, width)
        dense[0] = float("inf")
        dense[3] = float("nan")
        out = AF.spmm(g, dense.to(DEV)).cpu()
        assert torch.isfinite(out).all()


def test_transpose_and_slice_handles():
    from acm_gnn_amd import functional as AF
    from acm_gnn_amd.graph import CsrGraph
    m = _rand_csr(300, 211, 0.05, seed=5, hub_rows={9: 200})
    g = CsrGraph.from_scipy(m, DEV, chunk=32)
    gt = g.transpose()
    ip, ix, v = (t.cpu().numpy() for t in gt.arrays())
    mt = m.T.tocsr()
    mt.sort_indices()
    assert np.array_equal(ip, mt.indptr) and np.array_equal(ix, mt.indices) and np.array_equal(v, mt.data)
    dense = torch.randn(300, 16, generator=torch.Generator().manual_seed(2))
    ref = mt.astype(np.float64) @ dense.double().numpy()
    np.testing.assert_allclose(AF.spmm(gt, dense.to(DEV)).cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    sl = g.slice_rows(100, 250)
    d2 = torch.randn(211, 8, generator=torch.Generator().manual_seed(3))
    np.testing.assert_allclose(AF.spmm(sl, d2.to(DEV)).cpu().numpy(),
                               m[100:250].astype(np.float64) @ d2.double().numpy(), rtol=1e-5, atol=1e-5)


def test_from_torch_layouts_agree():
    from acm_gnn_amd.graph import CsrGraph
    m = _rand_csr(64, 64, 0.1, seed=8)
    coo = m.tocoo()
    perm = np.random.default_rng(0).permutation(coo.nnz)           # un-coalesced order
    idx = torch.from_numpy(np.vstack([coo.row[perm], coo.col[perm]]).astype(np.int64))
    t_coo = torch.sparse_coo_tensor(idx, torch.from_numpy(coo.data[perm]), (64, 64)).to(DEV)
    t_dense = torch.from_numpy(m.toarray()).to(DEV)
    t_csr = t_dense.to_sparse_csr()
    ref = None
    for t in (t_coo, t_dense, t_csr):
        arrs = [a.cpu().numpy() for a in CsrGraph.from_torch(t).arrays()]
        if ref is None:
            ref = arrs
            assert np.array_equal(arrs[0], m.indptr) and np.array_equal(arrs[1], m.indices)
        for a, b in zip(arrs, ref):
            assert np.array_equal(a, b)


def test_create_rejects_bad_input():
    from acm_gnn_amd.graph import CsrGraph
    ip = torch.tensor([0, 2, 1], dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="ACM_ESHAPE"):
        CsrGraph.from_csr(ip, torch.zeros(1, dtype=torch.int32, device=DEV), torch.ones(1, device=DEV), 4)
    ip = torch.tensor([0, 1], dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="out of range"):
        CsrGraph.from_csr(ip, torch.tensor([9], dtype=torch.int32, device=DEV), torch.ones(1, device=DEV), 4)


def test_device_filter_construction_is_bit_exact(tmp_path):
    """edge list -> A_low on the GPU == the oracle's restatement of the reference (float64 scipy -> float32),
    including duplicate edges, raw self-loops (diagonal 2/d) and isolated nodes; plus the on-disk cache."""
    from oracle import acm_oracle as O
    from acm_gnn_amd import graph as G, functional as AF
    rng = np.random.default_rng(0)
    n, e = 3000, 40000
    src, dst = rng.integers(0, n - 5, e), rng.integers(0, n - 5, e)       # last 5 nodes isolated
    src[:50] = dst[:50]                                                   # raw self-loops
    src[50:100], dst[50:100] = src[100:150], dst[100:150]                 # duplicate edges
    ei = torch.from_numpy(np.vstack([src, dst])).to(DEV)
    ops = G.filters_from_edge_index(ei, n)
    a = sp.coo_matrix((np.ones(2 * e), (np.concatenate([src, dst]), np.concatenate([dst, src]))), shape=(n, n)).tocsr()
    a.data[:] = 1.0                                                       # to_undirected coalesces duplicates
    ref_low, _, _ = O.filters_linkx(a)
    ip, ix, v = O.coo_to_csr_arrays(ref_low)
    assert ops.implicit and ops.low.pattern_only            # D^-1 (I + A) always has the pattern-only form
    assert ops.low.nnz == len(ix) + int((a.diagonal() > 0).sum())   # raw self-loops list their column twice
    got = [t.cpu().numpy() for t in G.explicit_arrays(ops)]
    assert np.array_equal(got[0], ip) and np.array_equal(got[1], ix)
    assert np.array_equal(got[2], v)                                      # bit-exact values
    assert np.array_equal(ops.deg.cpu().numpy(), np.asarray((sp.identity(n) + a).sum(1)).ravel().astype(np.float32))
    path = str(tmp_path / "ops.npz")
    G.save_operators(path, ops)
    back = G.load_operators(path, DEV)
    x = torch.randn(n, 8, device=DEV)
    assert back.implicit and torch.equal(back.row_scale, ops.row_scale)
    assert torch.equal(AF.spmm(back.low, x), AF.spmm(ops.low, x)) and torch.equal(back.deg, ops.deg)
    # the pattern-only product, row-scaled, is the explicit one
    with tuning.override(implicit=0):
        expl = G.filters_from_edge_index(ei, n)
    assert not expl.implicit
    want = AF.spmm(expl.low, x)
    got2 = ops.row_scale[:, None] * AF.spmm(ops.low, x)
    assert float((got2 - want).abs().max()) < 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("width", [16, 64, 68, 100, 128, 132, 192, 200, 256, 320])
@pytest.mark.parametrize("mode", ["vec", "scalar"])
def test_wide_gather_forms_match_scipy(width, mode, monkeypatch, tune):
    """The vector form of the wide gather (dwordx4 rows, four neighbours per instruction; spmm_vec_kernel) forced on,
    and the dword-per-lane form, for widths that fill column blocks exactly, partially (no read may leave the row:
    the last table row ends the allocation), and beyond 256 columns (two kernel passes); split hub rows, empty rows,
    pattern-only handles and a non-finite row the operator never references."""
    from acm_gnn_amd import functional as AF
    from acm_gnn_amd.graph import CsrGraph
    tune(wide_form=2 if mode == "vec" else 1)
    m = _rand_csr(400, 333, 0.06, seed=width, hub_rows={5: 300, 399: 150}, empty_rows=(0, 17))
    m[:, 332] = 0                                                   # nobody references the last table row ...
    m = sp.csr_matrix(m)
    m.eliminate_zeros()
    dense = torch.randn(333, width, generator=torch.Generator().manual_seed(width))
    dense[332] = float("nan")                                       # ... which is poisoned
    ref = m.astype(np.float64) @ np.nan_to_num(dense.double().numpy())
    for chunk in (0, 32):
        g = CsrGraph.from_scipy(m, DEV, chunk=chunk)
        out = AF.spmm(g, dense.to(DEV)).cpu().numpy()
        assert np.isfinite(out).all()
        np.testing.assert_allclose(out, ref, rtol=1e-5, atol=4e-5)      # hub rows: sums of 300 N(0,1)-sized terms
    ip, ix, _ = (t for t in CsrGraph.from_scipy(m, DEV).arrays())
    pat = CsrGraph.from_csr(ip, ix, None, 333)
    ones = sp.csr_matrix((np.ones(m.nnz), m.indices, m.indptr), shape=m.shape)
    np.testing.assert_allclose(AF.spmm(pat, dense.to(DEV)).cpu().numpy(), ones @ np.nan_to_num(dense.double().numpy()),
                               rtol=1e-5, atol=4e-5)


ROWS_NN = [(9000, 192, 128), (4100, 21, 100), (20000, 15, 64), (168114, 192, 128), (70000, 64, 300), (8192, 180, 17)]    # (no split-K in the tile kernel)


@pytest.mark.parametrize("m,n,k", ROWS_NN)
def test_row_panel_gemm_equals_the_tile_kernel_bit_for_bit(m, n, k, monkeypatch, tune):
    """acm_gemm_rows.hip (NN): a tall A (>= 4096 rows) takes the row-panel kernel -- A read once, all N columns per
    workgroup.  Both kernels are k-ordered fmaf chains on the fp32 MFMA, so their results are IDENTICAL, ragged shapes and
    the ReLU epilogue included; and integer inputs are exact."""
    from acm_gnn_amd import functional as AF
    g = torch.Generator().manual_seed(m + n + k)
    a, b = torch.randn(m, k, generator=g).to(DEV), torch.randn(k, n, generator=g).to(DEV)
    tune(gemm_forms=tuning.kernel()["gemm_forms"] | 8)                   # (by default only the shapes it wins on, or with a dropout)
    tune(gemm_forms=tuning.kernel()["gemm_forms"] & ~6)                       # (K <= 128 otherwise takes the split-bf16 kernels, below)
    new = AF.gemm(a, b, relu=True)
    tune(gemm_forms=tuning.kernel()["gemm_forms"] & ~9)
    old = AF.gemm(a, b, relu=True)
    tune(gemm_forms=tuning.kernel()["gemm_forms"] | 9)
    assert torch.equal(new, old)
    ai, bi = torch.randint(-8, 9, (m, k), generator=g).float(), torch.randint(-8, 9, (k, n), generator=g).float()
    assert torch.equal(AF.gemm(ai.to(DEV), bi.to(DEV)).cpu(), ai @ bi)
    out = torch.full((m, n + 5), 3.0, device=DEV)                     # a strided destination keeps its padding
    AF.gemm(a, b, out=out[:, 2:2 + n])
    assert torch.equal(out[:, 2:2 + n], AF.gemm(a, b)) and float((out[:, :2] - 3).abs().max()) == 0 and float((out[:, 2 + n:] - 3).abs().max()) == 0


@pytest.mark.parametrize("rows,f_in,n,blocks", [(9000, 128, 192, 3), (20001, 100, 15, 3), (168114, 128, 192, 3), (8192, 17, 21, 0),
                                                (40000, 64, 180, 0)])
def test_row_panel_transposed_gemm_matches_fp64(rows, f_in, n, blocks, tune):
    """acm_gemm_rows.hip (TN): dW = X^T dZ over >= 8192 rows as one K x N slab per workgroup + the deterministic slab sum;
    vs float64, as column blocks too, and bit-identical from launch to launch."""
    from acm_gnn_amd import functional as AF
    tune(gemm_forms=9)                                  # row-panel kernels for every shape they cover, no split-bf16
    g = torch.Generator().manual_seed(rows + f_in + n)
    x, dz = torch.randn(rows, f_in, generator=g), torch.randn(rows, n, generator=g)
    ref = x.double().T @ dz.double()
    scale = x.abs().double().T @ dz.abs().double()
    got = AF.gemm(x.to(DEV), dz.to(DEV), trans_a=True)
    assert float(((got.cpu().double() - ref).abs() / (scale + 1e-30)).max()) < 3e-6
    assert torch.equal(got, AF.gemm(x.to(DEV), dz.to(DEV), trans_a=True))
    if blocks:
        parts = AF.gemm(x.to(DEV), dz.to(DEV), trans_a=True, col_blocks=blocks)
        assert torch.equal(torch.cat(list(parts), dim=1), got)


def _rel_err(got, a64, b64):
    return float(((got.cpu().double() - a64 @ b64).abs() / (a64.abs() @ b64.abs() + 1e-30)).max())


@pytest.mark.parametrize("m,n,k", [(9000, 192, 128), (8200, 21, 100), (20000, 15, 64), (168114, 192, 128), (10000, 180, 36),
                                   (8192, 70, 32)])
def test_split_bf16_projection_keeps_fp32_accuracy(m, n, k, monkeypatch, tune):
    """acm_gemm_bx3.hip (NN): Z = X W for a tall X of <= 128 columns on the bf16 matrix pipe, every fp32 operand split
    EXACTLY into three bf16 numbers and six of the nine partial products kept: the error against float64 (relative to
    sum |x||w|) stays at the level of the fp32 fmaf chain of the other kernels -- with entries spanning seven orders of
    magnitude -- small integers are exact, the ReLU epilogue, ragged N / K and strided destinations work, and the result
    is the same from launch to launch."""
    from acm_gnn_amd import functional as AF
    g = torch.Generator().manual_seed(m + n + k)
    a, b = torch.randn(m, k, generator=g), torch.randn(k, n, generator=g)
    a[::7] *= 1e3
    a[::5] *= 1e-4
    a, b = a.to(DEV), b.to(DEV)
    new = AF.gemm(a, b)
    tune(gemm_forms=tuning.kernel()["gemm_forms"] & ~6)
    old = AF.gemm(a, b)
    tune(gemm_forms=tuning.kernel()["gemm_forms"] | 6)
    assert not torch.equal(new, old)                                   # (another kernel did run)
    a64, b64 = a.cpu().double(), b.cpu().double()
    e_new, e_old = _rel_err(new, a64, b64), _rel_err(old, a64, b64)
    assert e_new < 1e-6 and e_new < 1.5 * e_old + 1e-8, (e_new, e_old)
    assert torch.equal(new, AF.gemm(a, b))
    assert torch.equal(AF.gemm(a, b, relu=True), new.clamp_min(0))
    ai, bi = torch.randint(-8, 9, (m, k), generator=g).float(), torch.randint(-8, 9, (k, n), generator=g).float()
    assert torch.equal(AF.gemm(ai.to(DEV), bi.to(DEV)).cpu(), ai @ bi)
    out = torch.full((m, n + 5), 3.0, device=DEV)                     # a strided (and 16-byte misaligned) destination
    AF.gemm(a, b, out=out[:, 2:2 + n])
    assert torch.equal(out[:, 2:2 + n], new) and float((out[:, :2] - 3).abs().max()) == 0 and float((out[:, 2 + n:] - 3).abs().max()) == 0


def test_split_bf16_products_with_extreme_and_nonfinite_inputs(tune):
    """the split-bf16 kernels are the DEFAULT for tall fp32 products, so what they do outside the comfortable
    range is part of the fp32 path's contract.  (1) Entries from 1e-30 to 1e30 (the splits keep fp32's exponent range):
    error vs float64 relative to sum |x||w| at fp32 level.  (2) A non-finite entry poisons exactly what it poisons in the
    fp32 kernel -- the output row of that X row (NN), the dW row of that feature (TN) -- and nothing else: every other
    output element is bit-identical to the run with the entry replaced by 1.  The poisoned elements are non-finite in both
    kernels; their CLASS may differ (inf - bf16(inf) is NaN: the split turns an inf operand into NaN where the fmaf chain
    keeps +-inf), which is stated here rather than hidden.  (3) Subnormal operands contribute at most their own magnitude
    (the matrix pipe may flush them): absolute error <= 1e-37 x K."""
    from acm_gnn_amd import functional as AF
    g = torch.Generator().manual_seed(11)
    m, k, n = 9000, 64, 21
    # (1) extreme range, rows scaled by powers of ten from 1e-30 to 1e30 (products stay below fp32's maximum)
    x, w = torch.randn(m, k, generator=g), torch.randn(k, n, generator=g)
    scale = 10.0 ** torch.linspace(-30, 30, m).round()
    x = x * scale[:, None]
    z = AF.gemm(x.to(DEV), w.to(DEV))
    ref = x.double() @ w.double()
    mag = x.double().abs() @ w.double().abs()
    assert torch.isfinite(z).all()
    assert float(((z.cpu().double() - ref).abs() / mag).max()) < 1e-6
    dz = torch.randn(m, n, generator=g)
    xs = torch.randn(m, k, generator=g) * (10.0 ** torch.linspace(-15, 15, m).round())[:, None]
    dw = AF.gemm(xs.to(DEV), dz.to(DEV), trans_a=True)
    ref = xs.double().T @ dz.double()
    mag = xs.double().abs().T @ dz.double().abs()
    assert torch.isfinite(dw).all() and float(((dw.cpu().double() - ref).abs() / mag).max()) < 1e-6
    # (2) non-finite entries
    x = torch.randn(m, k, generator=g)
    clean = x.clone()
    bad = {(5, 3): float("inf"), (77, 0): float("nan"), (8999, 63): float("-inf")}
    for (r, c), v in bad.items():
        x[r, c], clean[r, c] = v, 1.0
    for forms in ("split-bf16", "fp32"):
        if forms == "fp32":
            tune(gemm_forms=tuning.kernel()["gemm_forms"] & ~6)
        z_bad, z_ok = AF.gemm(x.to(DEV), w.to(DEV)).cpu(), AF.gemm(clean.to(DEV), w.to(DEV)).cpu()
        rows = sorted({r for r, _ in bad})
        assert not torch.isfinite(z_bad[rows]).any(), forms                   # every element of a poisoned row (w has no zeros)
        keep = torch.ones(m, dtype=torch.bool)
        keep[rows] = False
        assert torch.equal(z_bad[keep], z_ok[keep]), forms
        d_bad, d_ok = AF.gemm(x.to(DEV), dz.to(DEV), trans_a=True).cpu(), AF.gemm(clean.to(DEV), dz.to(DEV), trans_a=True).cpu()
        feats = sorted({c for _, c in bad})
        assert not torch.isfinite(d_bad[feats]).any(), forms
        keep = torch.ones(k, dtype=torch.bool)
        keep[feats] = False
        assert torch.equal(d_bad[keep], d_ok[keep]), forms
    tune(gemm_forms=tuning.kernel()["gemm_forms"] | 6)
    # (3) subnormal operands
    tiny = torch.full((m, k), 1e-40)
    z = AF.gemm(tiny.to(DEV), torch.ones(k, n, device=DEV))
    assert float((z.cpu().double() - 1e-40 * k).abs().max()) <= 1e-37 * k


@pytest.mark.parametrize("rows,f_in,n,blocks", [(9000, 128, 192, 3), (20001, 100, 15, 3), (168114, 128, 192, 3), (8192, 33, 21, 0),
                                                (40000, 64, 180, 0), (16500, 2089, 192, 3), (16400, 300, 70, 0), (41554, 1030, 21, 3)])
def test_split_bf16_weight_gradient_keeps_fp32_accuracy(rows, f_in, n, blocks, monkeypatch, tune):
    """acm_gemm_bx3.hip (TN): dW = X^T dZ, the contraction over the rows: tiles split while they are staged, operands read
    from LDS as packed row pairs; vs float64 at the fp32 kernels' level, exact on small integers, deterministic, column
    blocks."""
    from acm_gnn_amd import functional as AF
    g = torch.Generator().manual_seed(rows + f_in + n)
    x, dz = torch.randn(rows, f_in, generator=g), torch.randn(rows, n, generator=g)
    x[::3] *= 1e2
    dz[::11] *= 1e-3
    x, dz = x.to(DEV), dz.to(DEV)
    new = AF.gemm(x, dz, trans_a=True)
    tune(gemm_forms=tuning.kernel()["gemm_forms"] & ~6)
    old = AF.gemm(x, dz, trans_a=True)
    tune(gemm_forms=tuning.kernel()["gemm_forms"] | 6)
    assert not torch.equal(new, old)
    x64, dz64 = x.cpu().double(), dz.cpu().double()
    e_new, e_old = _rel_err(new, x64.T, dz64), _rel_err(old, x64.T, dz64)
    assert e_new < 1e-6 and e_new < 2.0 * e_old + 1e-8, (e_new, e_old)
    assert torch.equal(new, AF.gemm(x, dz, trans_a=True))
    xi, zi = torch.randint(-3, 4, (rows, f_in), generator=g).float(), torch.randint(-3, 4, (rows, n), generator=g).float()
    assert torch.equal(AF.gemm(xi.to(DEV), zi.to(DEV), trans_a=True).cpu().double(), xi.double().T @ zi.double())
    if blocks:
        parts = AF.gemm(x, dz, trans_a=True, col_blocks=blocks)
        assert torch.equal(torch.cat(list(parts), dim=1), new)


@pytest.mark.parametrize("rows,f_in,n", [(9000, 128, 192), (20001, 100, 15), (8200, 17, 21), (168114, 128, 192)])
def test_gemm_with_the_input_dropout_in_the_tile_load(rows, f_in, n):
    """acm_gemm_drop: Z = drop(X) W and dW = drop(X)^T dZ with the counter-based mask drawn while X is staged equal the
    products of the dropped copy acm_dropout writes (same mask: numpy Philox, oracle/philox.py) -- NN bit for bit (same
    fmaf chain on the same operand values), TN to fp32 summation order."""
    from acm_gnn_amd import functional as AF
    from oracle.philox import dropout_factors
    g = torch.Generator().manual_seed(rows + n)
    x, w, dz = torch.randn(rows, f_in, generator=g).to(DEV), torch.randn(f_in, n, generator=g).to(DEV), torch.randn(rows, n, generator=g).to(DEV)
    st = AF.DropoutState(torch.device(DEV), seed=77)
    st.step.fill_(5)
    p, tag, off = 0.35, 0, 1000
    xd = AF.dropout(x, p, st, tag=tag, row_offset=off)                  # the dropped copy (acm_dropout)
    keep = torch.from_numpy(dropout_factors(st.seed, 5, tag, p, rows, f_in, row_offset=off) > 0)
    assert torch.equal(xd.cpu() != 0, keep & (x.cpu() != 0))            # ... is the numpy mask
    spec = st.spec(p, tag, off)
    z = AF.gemm(x, w, relu=True, a_drop=spec)
    assert torch.equal(z, AF.gemm(xd, w, relu=True))
    dw = AF.gemm(x, dz, trans_a=True, a_drop=spec)
    ref = xd.cpu().double().T @ dz.cpu().double()
    scale = xd.cpu().abs().double().T @ dz.cpu().abs().double()
    assert float(((dw.cpu().double() - ref).abs() / (scale + 1e-30)).max()) < 3e-6
    with tuning.override(gemm_forms=tuning.kernel()["gemm_forms"] | 8):     # the same kernel on the dropped copy: identical
        assert torch.equal(dw, AF.gemm(xd, dz, trans_a=True))


@pytest.mark.parametrize("n,k,f,fb,split", [(9000, 128, 64, 64, 0), (9000, 64, 5, 8, 16), (20000, 100, 5, 8, 0), (8192, 32, 2, 2, 4),
                                            (168114, 64, 2, 2, 4), (10000, 128, 7, 8, 16)])
def test_projection_from_three_weight_matrices_in_place(n, k, f, fb, split):
    """acm_proj3 (ABI 21): [Z_L 0 | Z_H 0 | Z_I] = relu?(drop?(X) [W_L 0 | W_H 0 | W_I]) with the weights read in place --
    bit-identical to the same kernel on the packed matrix torch.cat builds (zero columns between the channel blocks), one
    or two output tables, the ReLU, the dropout drawn in the load against the product of the dropped copy; shapes outside
    the envelope are refused without a launch."""
    from acm_gnn_amd import functional as AF
    g = torch.Generator().manual_seed(n + k + f)
    x = torch.randn(n, k, generator=g).to(DEV)
    w3 = [torch.randn(k, f, generator=g).to(DEV) for _ in range(3)]
    zpad = torch.zeros(k, fb - f, device=DEV)
    wcat = torch.cat([w3[0], zpad, w3[1], zpad, w3[2]], 1).contiguous()
    ncols = 2 * fb + f
    for relu in (False, True):
        want = AF.gemm(x, wcat, relu=relu)
        if split:
            o1, o2 = torch.full((n, split), 7.0, device=DEV), torch.full((n, ncols - split + 3), 7.0, device=DEV)[:, : ncols - split]
            assert AF.proj3(x, w3, fb, o1, o2, relu=relu)
            assert torch.equal(o1, want[:, :split]) and torch.equal(o2, want[:, split:])
        else:
            out = torch.full((n, ncols + 4), 7.0, device=DEV)
            assert AF.proj3(x, w3, fb, out[:, :ncols], relu=relu)
            assert torch.equal(out[:, :ncols], want) and float((out[:, ncols:] - 7).abs().max()) == 0
    st = AF.DropoutState(torch.device(DEV), seed=3)
    st.step.fill_(2)
    spec = st.spec(0.3, 0, 500)
    xd = AF.dropout(x, 0.3, st, tag=0, row_offset=500)
    out = torch.empty(n, ncols, device=DEV)
    assert AF.proj3(x, w3, fb, out, x_drop=spec)
    assert torch.equal(out, AF.gemm(xd, wcat))
    a64 = x.cpu().double()
    err = ((AF.gemm(x, wcat).cpu().double() - a64 @ wcat.cpu().double()).abs() / (a64.abs() @ wcat.cpu().double().abs() + 1e-30)).max()
    assert float(err) < 1e-6
    # outside the envelope: nothing launched
    small = torch.randn(100, k, device=DEV)
    assert AF.proj3(small, w3, fb, torch.empty(100, ncols, device=DEV)) is False
    assert AF.proj3(x[:, : k - 1], [w[: k - 1] for w in w3], fb, torch.empty(n, ncols, device=DEV)) is False


def test_wide_weight_gradient_with_the_dropout_in_the_operand_load():
    """acm_gemm_bx3.hip (TN) for inputs wider than 128 features (workgroups own blocks of 128 output rows): dW = drop(X)^T dZ
    with the mask drawn while X is staged equals the product of the dropped copy (Philox block = column mod 16 + 16 * (column
    / 64), any column) bit for bit."""
    from acm_gnn_amd import functional as AF
    g = torch.Generator().manual_seed(5)
    rows, f_in, n = 16400, 452, 40                      # (the wide form starts at 16 384 rows)
    x, dz = torch.randn(rows, f_in, generator=g).to(DEV), torch.randn(rows, n, generator=g).to(DEV)
    st = AF.DropoutState(torch.device(DEV), seed=21)
    st.step.fill_(3)
    spec = st.spec(0.3, 0, 77)
    xd = AF.dropout(x, 0.3, st, tag=0, row_offset=77)
    dw = AF.gemm(x, dz, trans_a=True, a_drop=spec)
    assert torch.equal(dw, AF.gemm(xd, dz, trans_a=True))
    ref = xd.cpu().double().T @ dz.cpu().double()
    scale = xd.cpu().abs().double().T @ dz.cpu().abs().double()
    assert float(((dw.cpu().double() - ref).abs() / (scale + 1e-30)).max()) < 1e-6


@pytest.mark.parametrize("n,f_in,f_out,relu,p", [(1, 1, 2, 1, 0.0), (1000, 7, 64, 1, 0.3), (70001, 16, 200, 0, 0.4), (4099, 9, 33, 1, 0.0),
                                                 (168114, 7, 64, 1, 0.1)])
def test_linear_backward_in_one_pass_matches_fp64(n, f_in, f_out, relu, p):
    """acm_linear_bwd (the ACM-GCN++ residual Linear on a narrow dense input, models.py:26-27,55-56 backward): dW = G^T X and
    db = column sums of G with G = dY * keep * [Y > 0] formed in registers -- against float64 and against the two calls it
    replaces (acm_bias_act_bwd + acm_gemm), twice over bit for bit; padded pitches."""
    import ctypes as C
    from acm_gnn_amd import _lib, functional as AF
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + f_in)
    ldx, ldy = f_in + (n % 2), f_out + 3
    x = torch.randn(n, ldx, generator=g).to(DEV)
    pre = torch.randn(n, ldy, generator=g)
    keep = (torch.rand(n, ldy, generator=g) >= p).float() / (1.0 - p) if p else torch.ones(n, ldy)
    y = ((pre.clamp_min(0) if relu else pre) * keep).to(DEV)
    dy = torch.randn(n, ldy, generator=g).to(DEV)
    ks = 1.0 / (1.0 - p)
    nb = C.c_size_t()
    _lib.check(lib.acm_linear_bwd_workspace_bytes(n, f_in, f_out, C.byref(nb)))
    ws = torch.empty(nb.value // 4, device=DEV)
    outs = []
    for _ in range(2):
        dw = torch.full((f_out, f_in + 1), float("nan"), device=DEV)
        db = torch.full((f_out,), float("nan"), device=DEV)
        _lib.check(lib.acm_linear_bwd(n, f_in, f_out, x.data_ptr(), ldx, y.data_ptr(), ldy, dy.data_ptr(), ldy, ks, relu, dw.data_ptr(),
                                      f_in + 1, db.data_ptr(), ws.data_ptr(), nb.value, None, None), "acm_linear_bwd")
        torch.cuda.synchronize()
        outs.append((dw.clone(), db.clone()))
    assert torch.equal(outs[0][0][:, :f_in], outs[1][0][:, :f_in]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isnan(outs[0][0][:, f_in]).all()                       # the pitch column is not written
    y64, dy64, x64 = (t.cpu().double() for t in (y[:, :f_out], dy[:, :f_out], x[:, :f_in]))
    g64 = torch.where(y64 > 0, dy64 * ks, torch.zeros_like(dy64)) if relu else (
        torch.where(y64 != 0, dy64 * ks, torch.zeros_like(dy64)) if p else dy64)
    want_w, want_b = g64.t() @ x64, g64.sum(0)
    scale = max(1.0, float(want_w.abs().max()))
    assert float((outs[0][0][:, :f_in].cpu().double() - want_w).abs().max()) < 2e-5 * scale
    assert float((outs[0][1].cpu().double() - want_b).abs().max()) < 2e-5 * max(1.0, float(want_b.abs().max()))
    assert lib.acm_linear_bwd_workspace_bytes(n, 17, f_out, C.byref(nb)) == 4


@pytest.mark.parametrize("f_in,f_out,relu,p", [(7, 64, 1, 0.3), (16, 200, 0, 0.0), (1, 2, 1, 0.5), (9, 65, 1, 0.0)])
def test_linear_forward_of_a_narrow_input_streams_with_the_bits_of_the_gemm_route(f_in, f_out, relu, p):
    """acm_linear_fwd with f_in <= 16 and >= 1024 rows runs the streaming kernel of acm_linear.hip; fewer rows run the MFMA
    tile GEMM with the same epilogue.  Same fmaf chain in k order, same bias -> ReLU -> dropout: the first rows of a long
    call equal a short call bit for bit (the counter-based mask depends on the row index only), and both match float64."""
    import ctypes as C
    from acm_gnn_amd import _lib, functional as AF
    lib = _lib.load()
    n, m = 70003, 1000
    g = torch.Generator().manual_seed(f_in * 100 + f_out)
    x = torch.randn(n, f_in + 1, generator=g).to(DEV)
    w = torch.randn(f_out, f_in, generator=g).to(DEV)
    b = torch.randn(f_out, generator=g).to(DEV)
    state = AF.DropoutState(DEV, seed=11)
    spec = AF._drop_spec((p, 3, state), 0) if p else None
    outs = []
    for rows in (n, m):
        y = torch.full((rows, f_out + 2), float("nan"), device=DEV)
        nb = C.c_size_t()
        _lib.check(lib.acm_gemm_workspace_bytes(0, 1, rows, f_out, f_in, C.byref(nb)))
        ws = torch.empty(max(nb.value // 4, 1), device=DEV)
        _lib.check(lib.acm_linear_fwd(rows, f_in, f_out, x.data_ptr(), f_in + 1, w.data_ptr(), f_in, b.data_ptr(), relu,
                                      C.byref(spec) if spec is not None else None, y.data_ptr(), f_out + 2, ws.data_ptr(), nb.value, None),
                   "acm_linear_fwd")
        torch.cuda.synchronize()
        outs.append(y)
    assert torch.equal(outs[0][:m, :f_out], outs[1][:, :f_out])
    assert torch.isnan(outs[0][:, f_out:]).all()
    pre = x[:, :f_in].cpu().double() @ w.cpu().double().t() + b.cpu().double()
    if relu:
        pre = pre.clamp_min(0)
    got = outs[0][:, :f_out].cpu().double()
    kept = got != 0 if p else torch.ones_like(got, dtype=torch.bool)
    scale = 1.0 / (1.0 - p)
    assert float(((got - pre * scale) * kept).abs().max()) < 2e-5 * max(1.0, float(pre.abs().max()))
    if p:
        frac = float(((got == 0) & (pre > 1e-6 if relu else pre.abs() > 1e-6)).double().mean() / max(float((pre > 1e-6 if relu else pre.abs() > 1e-6).double().mean()), 1e-9))
        assert abs(frac - p) < 0.02


@pytest.mark.parametrize("n,f_in,f_out,relu,p", [(1, 1, 2, 1, 0.0), (1000, 7, 64, 1, 0.3), (70001, 16, 200, 0, 0.4), (168114, 7, 64, 1, 0.1)])
def test_residual_add_and_its_recomputing_backward(n, f_in, f_out, relu, p):
    """acm_linear_fwd_add (Y = add + dropout(relu(X W^T + b)), in place over add as well) against acm_linear_fwd + an add, bit
    for bit; acm_linear_bwd_recompute (masks formed again from X, W, b and the counter) against acm_linear_bwd on the plain
    forward's output, bit for bit -- the recomputed pre-activation and factors ARE the forward's."""
    import ctypes as C
    from acm_gnn_amd import _lib, functional as AF
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + f_out)
    x = torch.randn(n, f_in + 1, generator=g).to(DEV)
    w = torch.randn(f_out, f_in, generator=g).to(DEV)
    b = torch.randn(f_out, generator=g).to(DEV)
    add = torch.randn(n, f_out + 1, generator=g).to(DEV)
    dy = torch.randn(n, f_out, generator=g).to(DEV)
    state = AF.DropoutState(DEV, seed=5)
    spec = AF._drop_spec((p, 2, state), 0) if p else None
    sp = C.byref(spec) if spec is not None else None
    # plain forward (the streaming route needs >= 1024 rows; below that the GEMM route -- same bits)
    y = torch.empty(n, f_out, device=DEV)
    nb = C.c_size_t()
    _lib.check(lib.acm_gemm_workspace_bytes(0, 1, n, f_out, f_in, C.byref(nb)))
    ws = torch.empty(max(nb.value // 4, 1), device=DEV)
    _lib.check(lib.acm_linear_fwd(n, f_in, f_out, x.data_ptr(), f_in + 1, w.data_ptr(), f_in, b.data_ptr(), relu, sp, y.data_ptr(), f_out,
                                  ws.data_ptr(), nb.value, None), "acm_linear_fwd")
    out = torch.full((n, f_out), float("nan"), device=DEV)
    _lib.check(lib.acm_linear_fwd_add(n, f_in, f_out, x.data_ptr(), f_in + 1, w.data_ptr(), f_in, b.data_ptr(), relu, sp, add.data_ptr(),
                                      f_out + 1, out.data_ptr(), f_out, None), "acm_linear_fwd_add")
    assert torch.equal(out, y + add[:, :f_out])
    inplace = add.clone()
    _lib.check(lib.acm_linear_fwd_add(n, f_in, f_out, x.data_ptr(), f_in + 1, w.data_ptr(), f_in, b.data_ptr(), relu, sp, inplace.data_ptr(),
                                      f_out + 1, inplace.data_ptr(), f_out + 1, None), "acm_linear_fwd_add")
    assert torch.equal(inplace[:, :f_out], out) and torch.equal(inplace[:, f_out], add[:, f_out])
    # backward: masks read off y against masks recomputed
    nb2 = C.c_size_t()
    _lib.check(lib.acm_linear_bwd_workspace_bytes(n, f_in, f_out, C.byref(nb2)))
    ws2 = torch.empty(nb2.value // 4, device=DEV)
    res = []
    for which in (0, 1):
        dw, db = torch.full((f_out, f_in), float("nan"), device=DEV), torch.full((f_out,), float("nan"), device=DEV)
        if which == 0:
            st = lib.acm_linear_bwd(n, f_in, f_out, x.data_ptr(), f_in + 1, y.data_ptr(), f_out, dy.data_ptr(), f_out, 1.0 / (1.0 - p), relu,
                                    dw.data_ptr(), f_in, db.data_ptr(), ws2.data_ptr(), nb2.value, None, None)
        else:
            st = lib.acm_linear_bwd_recompute(n, f_in, f_out, x.data_ptr(), f_in + 1, w.data_ptr(), f_in, b.data_ptr(), relu, sp,
                                              dy.data_ptr(), f_out, dw.data_ptr(), f_in, db.data_ptr(), ws2.data_ptr(), nb2.value, None, None)
        _lib.check(st, "acm_linear_bwd*")
        torch.cuda.synchronize()
        res.append((dw, db))
    if relu or not p:                    # (without a ReLU the mask read off y also drops kept exact zeros: measure zero)
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    else:
        torch.testing.assert_close(res[1][0], res[0][0], rtol=1e-5, atol=1e-5)
