"""The mask form of the ACMII first layer (acm_conv_acmii_v.hip) through the C ABI: the table, the forward against the
fp32-MFMA kernel it replaces, the weight gradients against a float64 restatement of ACM-Geometric/layers.py:94-99's autograd,
on graphs with hub rows (pieces + fix-up), rows without neighbours, item counts that are not a multiple of four, and the
error paths.  The layer-level parity (oracle, goldens, literal form) is in test_gpu_oracle.py / test_gpu_golden.py /
test_gpu_fullsize.py, which run this form by default."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _pattern(n, density, seed, hub=0, empty=()):
    rng = np.random.default_rng(seed)
    a = sp.random(n, n, density=density, random_state=rng, format="lil")
    a = ((a + a.T) > 0).astype(np.float32).tolil()
    if hub:
        a[0, 1:hub] = 1
        a[1:hub, 0] = 1
    for r in empty:
        a[r, :] = 0
    return sp.csr_matrix(a)


def _bf16_parts(row_dwords):
    """[hi (4 dwords) | mid | lo / 2 | masks] -> float32 hi, mid, lo / 2 of the eight features"""
    out = []
    for part in range(3):
        d = row_dwords[:, 4 * part:4 * part + 4].astype(np.uint32)
        lo16, hi16 = (d & 0xFFFF), (d >> 16)
        vals = np.empty((d.shape[0], 8), np.uint32)
        vals[:, 0::2], vals[:, 1::2] = lo16 << 16, hi16 << 16
        out.append(vals.view(np.float32))
    return out


def _table(lib, x8, wl, wh, f_in):
    from acm_gnn_amd import _lib
    n = x8.shape[0]
    nb = C.c_size_t()
    _lib.check(lib.acm_acmii_table_bytes(n, C.byref(nb)))
    assert nb.value == (n + 1) * 64
    table = torch.empty(nb.value // 4, dtype=torch.int32, device=DEV)
    _lib.check(lib.acm_acmii_table(n, f_in, x8.data_ptr(), x8.stride(0), wl.data_ptr(), wh.data_ptr(), 64, table.data_ptr(), nb.value,
                                   None), "acm_acmii_table")
    return table


@pytest.mark.parametrize("n,f_in", [(1, 1), (37, 7), (5000, 8), (70001, 3)])
def test_table_rows_are_exact_splits_and_the_masks_of_the_projection(n, f_in):
    from acm_gnn_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(n)
    x8 = torch.zeros(n, 8)
    x8[:, :f_in] = torch.randn(n, f_in, generator=g) * torch.logspace(-6, 6, n)[:, None]      # a wide dynamic range
    wl, wh = torch.randn(f_in, 64, generator=g), torch.randn(f_in, 64, generator=g)
    table = _table(lib, x8.to(DEV), wl.to(DEV), wh.to(DEV), f_in)
    rows = table.cpu().numpy().view(np.uint32).reshape(n + 1, 16)
    assert not rows[n].any()                                         # the zero row idle slots fetch
    hi, mid, lo = _bf16_parts(rows[:n])
    xs = x8.numpy()
    assert np.array_equal((hi.astype(np.float64) + mid + 2.0 * lo).astype(np.float32), xs)    # x = hi + mid + 2 (lo / 2), exactly
    assert np.array_equal(hi.view(np.uint32) & 0xFFFF, np.zeros_like(hi, np.uint32))
    # mask byte m of a row: bit 4 ch + t = [x W_ch[:, 16 t + m] > 0]; a sign may differ only where z is rounding noise
    mb = rows[:n, 12:16].copy().view(np.uint8).reshape(n, 16)
    x64 = xs[:, :f_in].astype(np.float64)
    for ch, w in enumerate((wl, wh)):
        z = x64 @ w.numpy().astype(np.float64)
        scale = np.abs(x64) @ np.abs(w.numpy().astype(np.float64)) + 1e-300
        for t in range(4):
            got = (mb >> (4 * ch + t)) & 1
            want = z[:, 16 * t:16 * t + 16] > 0
            bad = got != want
            assert not (bad & (np.abs(z[:, 16 * t:16 * t + 16]) > 1e-5 * scale[:, 16 * t:16 * t + 16])).any()
            assert bad.mean() < 1e-3


def _layer_inputs(n, f_in, seed):
    g = torch.Generator().manual_seed(seed)
    x8 = torch.zeros(n, 8)
    x8[:, :f_in] = torch.randn(n, f_in, generator=g)
    w = [torch.randn(f_in, 64, generator=g) * 0.5 for _ in range(3)]
    vecs = [torch.randn(64, 1, generator=g) * 0.3 for _ in range(3)]
    lnw = [torch.rand(64, generator=g) + 0.5 for _ in range(3)]
    lnb = [torch.randn(64, generator=g) * 0.2 for _ in range(3)]
    mix = torch.randn(3, 3, generator=g) * 0.4
    return x8, w, vecs, lnw, lnb, mix


def _fwd_struct(_lib, n, f_in, x8, w, vecs, lnw, lnb, mix, rs, ln, with_zlh):
    out, pre, att = (torch.full((n, c), float("nan"), device=DEV) for c in (64, 128, 4))
    zlh, zi = torch.full((n, 128), float("nan"), device=DEV), torch.full((n, 64), float("nan"), device=DEV)
    p = _lib.ConvAcmiiFwd()
    p.f_in, p.f_pad, p.f_out, p.layernorm, p.scale, p.n_channels = f_in, 8, 64, int(ln), 3.0, 3
    p.xg, p.ld_xg, p.xs, p.ld_xs = x8.data_ptr(), 8, x8.data_ptr(), 8
    p.w_low, p.w_high, p.w_mlp, p.ld_w = w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(), 64
    for c in range(3):
        p.att_vec[c] = vecs[c].data_ptr()
        if ln:
            p.ln_weight[c], p.ln_bias[c] = lnw[c].data_ptr(), lnb[c].data_ptr()
    p.att_mix = mix.data_ptr()
    p.out, p.ld_out, p.pre, p.ld_pre, p.att = out.data_ptr(), 64, pre.data_ptr(), 128, att.data_ptr()
    if with_zlh:
        p.zlh, p.ld_zlh = zlh.data_ptr(), 128
    p.zi, p.ld_zi = zi.data_ptr(), 64
    p.row_scale = rs.data_ptr()
    return p, dict(out=out, pre=pre, att=att, zlh=zlh, zi=zi)


@pytest.mark.parametrize("n,density,hub,f_in,ln,chunk", [(403, 0.03, 0, 7, True, 0), (1501, 0.01, 900, 8, True, 128),
                                                         (2048, 0.004, 1500, 1, False, 256), (9, 0.5, 0, 5, True, 0)])
def test_forward_and_weight_gradients_through_the_c_abi(n, density, hub, f_in, ln, chunk):
    from acm_gnn_amd import _lib
    from acm_gnn_amd.graph import CsrGraph
    lib = _lib.load()
    a = _pattern(n, density, n, hub=hub, empty=(n - 1, n // 2))
    ip, ix, _ = CsrGraph.from_scipy(a, DEV).arrays()
    gph = CsrGraph.from_csr(ip, ix, None, n, chunk=chunk)               # pattern-only
    assert (gph.n_long_rows > 0) == bool(hub)
    deg = np.maximum(np.asarray(a.sum(1)).ravel(), 1.0)
    rs = torch.tensor(1.0 / deg, dtype=torch.float32, device=DEV)
    x8, w, vecs, lnw, lnb, mix = _layer_inputs(n, f_in, n + 1)
    x8, mix = x8.to(DEV), mix.to(DEV).contiguous()
    w, vecs, lnw, lnb = ([t.to(DEV).contiguous() for t in lst] for lst in (w, vecs, lnw, lnb))
    nbytes = C.c_size_t()
    _lib.check(lib.acm_conv_acmii_fwd_workspace_bytes(gph.handle, C.byref(nbytes)))
    ws = torch.empty(max(nbytes.value // 4, 1), device=DEV)
    # the fp32-MFMA kernel
    p0, ref = _fwd_struct(_lib, n, f_in, x8, w, vecs, lnw, lnb, mix, rs, ln, True)
    _lib.check(lib.acm_conv_acmii_fwd(gph.handle, C.byref(p0), ws.data_ptr(), ws.numel() * 4, None), "acm_conv_acmii_fwd")
    # the mask form: without item streams it refuses ...
    table = _table(lib, x8, w[0], w[1], f_in)
    p1, got = _fwd_struct(_lib, n, f_in, x8, w, vecs, lnw, lnb, mix, rs, ln, bool(hub))
    assert lib.acm_conv_acmii_v_fwd(gph.handle, C.byref(p1), table.data_ptr(), ws.data_ptr(), ws.numel() * 4, None) == 1   # ACM_EINVAL
    assert b"item streams" in lib.acm_last_error()
    # ... with them (a wave count that is not a divisor of anything) it matches, twice over bit for bit
    assert gph.build_item_streams(n_waves=24) and gph.item_stream_waves % 4 == 0
    outs = []
    for _ in range(2):
        for t in got.values():
            t.fill_(float("nan"))
        _lib.check(lib.acm_conv_acmii_v_fwd(gph.handle, C.byref(p1), table.data_ptr(), ws.data_ptr(), ws.numel() * 4, None),
                   "acm_conv_acmii_v_fwd")
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in got.items()})
    for k in ("out", "pre", "att", "zi"):
        assert torch.equal(outs[0][k], outs[1][k]), k
        scale = max(1.0, float(ref[k].abs().max()))
        assert float((outs[0][k] - ref[k]).abs().max()) < 2e-5 * scale, (k, float((outs[0][k] - ref[k]).abs().max()))
    assert torch.isfinite(outs[0]["out"]).all()
    if hub:
        assert float((outs[0]["zlh"][:, 64:] - ref["zlh"][:, 64:]).abs().max()) < 2e-5
    # weight gradients against float64: dW_L = X^T (m_L o A^T G_L), dW_H = X^T (m_H o (G_H - A^T G_H)), dW_I = X^T dZ_I
    gen = torch.Generator().manual_seed(5)
    gl, gh, gi = (torch.randn(n, 64, generator=gen).to(DEV) for _ in range(3))
    dw = torch.full((3, f_in, 64), float("nan"), device=DEV)
    b = _lib.ConvAcmiiBwd()
    b.f_in, b.table = f_in, table.data_ptr()
    b.g_low, b.ld_g_low, b.g_high, b.ld_g_high, b.g_mlp, b.ld_g_mlp = gl.data_ptr(), 64, gh.data_ptr(), 64, gi.data_ptr(), 64
    b.x, b.ld_x, b.row_scale = x8.data_ptr(), 8, rs.data_ptr()
    b.self_offset = 1                                                    # rows beyond the table: refused
    nb0 = C.c_size_t()
    _lib.check(lib.acm_conv_acmii_v_bwd_workspace_bytes(gph.handle, C.byref(nb0)))
    ws0 = torch.empty(nb0.value // 4, device=DEV)
    b.d_w_low = b.d_w_high = b.d_w_mlp = ws0.data_ptr()
    b.ld_dw = 64
    assert lib.acm_conv_acmii_v_bwd(gph.handle, C.byref(b), ws0.data_ptr(), ws0.numel() * 4, None) == 1
    b.self_offset = 0
    b.d_w_low, b.d_w_high, b.d_w_mlp, b.ld_dw = dw[0].data_ptr(), dw[1].data_ptr(), dw[2].data_ptr(), 64
    nb2 = C.c_size_t()
    _lib.check(lib.acm_conv_acmii_v_bwd_workspace_bytes(gph.handle, C.byref(nb2)))
    ws2 = torch.empty(nb2.value // 4, device=DEV)
    _lib.check(lib.acm_conv_acmii_v_bwd(gph.handle, C.byref(b), ws2.data_ptr(), ws2.numel() * 4, None), "acm_conv_acmii_v_bwd")
    first = dw.clone()
    dw.fill_(float("nan"))
    _lib.check(lib.acm_conv_acmii_v_bwd(gph.handle, C.byref(b), ws2.data_ptr(), ws2.numel() * 4, None), "acm_conv_acmii_v_bwd")
    assert torch.equal(first, dw)                                        # deterministic
    x64 = x8[:, :f_in].cpu().numpy().astype(np.float64)
    al = sp.diags(1.0 / deg) @ a.astype(np.float64)
    # the masks as the table holds them (a sign on rounding noise may differ from any other evaluation of x W)
    mb = table.cpu().numpy().view(np.uint32).reshape(n + 1, 16)[:n, 12:16].copy().view(np.uint8).reshape(n, 16)
    masks = np.zeros((2, n, 64))
    for ch in range(2):
        for t in range(4):
            masks[ch][:, 16 * t:16 * t + 16] = (mb >> (4 * ch + t)) & 1
    g64 = [t.cpu().numpy().astype(np.float64) for t in (gl, gh, gi)]
    want = [x64.T @ (masks[0] * (al.T @ g64[0])), x64.T @ (masks[1] * (g64[1] - al.T @ g64[1])), x64.T @ g64[2]]
    for c in range(3):
        tol = 2e-5 * max(1.0, float(np.abs(want[c]).max()))
        assert float(np.abs(first[c].cpu().numpy() - want[c]).max()) < tol, c


def test_mask_form_declines_what_it_does_not_cover():
    from acm_gnn_amd import _lib
    from acm_gnn_amd.graph import CsrGraph
    lib = _lib.load()
    a = _pattern(200, 0.05, 3)
    valued = CsrGraph.from_scipy(a, DEV)                                # explicit values
    assert not valued.build_item_streams()
    x8 = torch.zeros(200, 8, device=DEV)
    w = torch.zeros(9, 64, device=DEV)
    tb = torch.empty(201 * 16, dtype=torch.int32, device=DEV)
    assert lib.acm_acmii_table(200, 9, x8.data_ptr(), 8, w.data_ptr(), w.data_ptr(), 64, tb.data_ptr(), tb.numel() * 4, None) == 4
    assert lib.acm_acmii_table(200, 7, x8.data_ptr(), 8, w.data_ptr(), w.data_ptr(), 64, tb.data_ptr(), 64, None) == 5    # ACM_ENOMEM
