"""A numpy stand-in for libacm_hip.so -- TEST DOUBLE, used only by the CPU test-suite.

It implements the C ABI of include/acm_hip.h entry point by entry point on *host* pointers
(float64 arithmetic, float32 storage) so that everything above the ABI -- the ctypes structs,
the autograd Function, GraphConvolution / GCN, the row-shard plan and its gloo collectives -- can
be exercised without a GPU.  Product code never imports this module; tests install it with
``install(monkeypatch)``.  It is also an executable statement of what each entry point computes.
"""
import contextlib
import ctypes as C

import numpy as np

LN_EPS = 1e-5


def _view(ptr, rows, cols, ld, dtype=np.float32):
    if not ptr or rows == 0 or cols == 0:
        return np.zeros((rows, cols), dtype)
    ptr = ptr.value if isinstance(ptr, C.c_void_p) else int(ptr)
    n = (rows - 1) * ld + cols
    base = np.ctypeslib.as_array((np.ctypeslib.as_ctypes_type(dtype) * n).from_address(ptr))
    return np.lib.stride_tricks.as_strided(base, shape=(rows, cols), strides=(ld * base.itemsize, base.itemsize))


def _vec(ptr, n, dtype=np.float32):
    return _view(ptr, 1, n, n, dtype)[0]


class _Csr:
    def __init__(self, indptr, indices, vals, n_cols, chunk):
        self.indptr, self.indices, self.vals = indptr.copy(), indices.copy(), vals.copy()
        self.n_rows, self.n_cols = len(indptr) - 1, n_cols
        auto = 128
        while auto < 1024 and 2 * auto <= len(indices) // 8192:
            auto *= 2
        self.chunk = chunk or auto

    def dense_mul(self, g):
        import scipy.sparse as sp
        m = sp.csr_matrix((self.vals.astype(np.float64), self.indices, self.indptr), shape=(self.n_rows, self.n_cols))
        return m @ g.astype(np.float64)


from oracle.philox import philox7_words, dropout_factors as _keep_factors  # noqa: E402  (the numpy restatement of the mask)


def dropout_factors(d, n_rows, n_cols):
    """[n_rows, n_cols] factors (0 or 1/(1-p)) of an acm_dropout_t (ctypes struct or None)."""
    if d is None or d.p <= 0:
        return np.ones((n_rows, n_cols))
    step = int(_vec(d.step, 1, np.int64)[0]) + int(getattr(d, "step_offset", 0))
    return _keep_factors(d.seed, step, d.tag, float(np.float32(d.p)), n_rows, n_cols, row_offset=int(d.row_offset))


def _post_fwd(p, out, n, F):
    if p.post_relu:
        out = np.maximum(out, 0)
    if p.post_scale:
        out = out * _view(p.post_scale, n, F, p.ld_post_scale)
    if p.post_drop.p > 0:
        out = out * dropout_factors(p.post_drop, n, F)
    return out


def _post_bwd(p, dO, raw_out, n, F):
    if p.post_relu:
        dO = np.where(raw_out > 0, dO, 0.0)
    if p.post_scale:
        dO = dO * _view(p.post_scale, n, F, p.ld_post_scale)
    if p.post_drop.p > 0:
        dO = dO * dropout_factors(p.post_drop, n, F)
    return dO


def _head(H, k, layernorm, vecs, lnw, lnb, mix):
    """H: list of k arrays [n, F] (float64). Returns dict with everything the backward needs."""
    n, F = H[0].shape
    hn, xhat, rstd, s = [], [], [], []
    for c in range(k):
        if layernorm:
            mean = H[c].mean(1, keepdims=True)
            var = ((H[c] - mean) ** 2).mean(1, keepdims=True)
            r = 1.0 / np.sqrt(var + LN_EPS)
            xh = (H[c] - mean) * r
            h = xh * lnw[c][None, :] + lnb[c][None, :]
        else:
            r, xh, h = np.ones((n, 1)), np.zeros_like(H[c]), H[c]
        hn.append(h)
        xhat.append(xh)
        rstd.append(r)
        s.append(h @ vecs[c])
    g = 1.0 / (1.0 + np.exp(-np.stack(s, 1)))                  # [n, k]
    logits = g @ mix / k
    logits = logits - logits.max(1, keepdims=True)
    e = np.exp(logits)
    alpha = e / e.sum(1, keepdims=True)
    return dict(hn=hn, xhat=xhat, rstd=rstd, g=g, alpha=alpha)


def _head_backward(H, hd, dO, k, layernorm, vecs, lnw, mix, scale):
    n, F = H[0].shape
    alpha, g = hd["alpha"], hd["g"]
    dalpha = scale * np.stack([(dO * H[c]).sum(1) for c in range(k)], 1)
    dot = (alpha * dalpha).sum(1, keepdims=True)
    dlogit = alpha * (dalpha - dot)
    dg = dlogit @ mix.T / k
    d_mix = g.T @ dlogit / k
    ds = dg * g * (1.0 - g)
    dH, d_vec, d_lnw, d_lnb = [], [], [], []
    for c in range(k):
        d_vec.append((ds[:, c:c + 1] * hd["hn"][c]).sum(0))
        if layernorm:
            dhn = ds[:, c:c + 1] * vecs[c][None, :]
            d_lnw.append((dhn * hd["xhat"][c]).sum(0))
            d_lnb.append(dhn.sum(0))
            dxh = dhn * lnw[c][None, :]
            m1 = dxh.mean(1, keepdims=True)
            m2 = (dxh * hd["xhat"][c]).mean(1, keepdims=True)
            dH.append(scale * alpha[:, c:c + 1] * dO + hd["rstd"][c] * (dxh - m1 - hd["xhat"][c] * m2))
        else:
            dH.append(scale * alpha[:, c:c + 1] * dO + ds[:, c:c + 1] * vecs[c][None, :])
    return dH, d_vec, d_lnw, d_lnb, d_mix


class FakeLib:
    def __init__(self):
        self._handles, self._next, self._err = {}, 1, b""
        self._pending = {}
        self._tuning = dict(self.TUNING_DEFAULTS)

    # ---- plumbing ---------------------------------------------------------
    def acm_version(self):
        return 23

    # ---- the tuning record (acm_tuning_t): plain storage + the library's validation; the entry points below consult it
    # where the library's dispatch does (gemm_forms, rows16)
    TUNING_DEFAULTS = dict(chunk=0, wide_form=0, bwd_split=-1, rows16=7, agg_fused=1, gemm_forms=7)

    def acm_tuning_get(self, out):
        t = out._obj
        for k, v in self._tuning.items():
            setattr(t, k, v)
        return 0

    def acm_tuning_set(self, ptr):
        if not ptr:
            self._tuning = dict(self.TUNING_DEFAULTS)
            return 0
        t = ptr._obj
        new = {k: int(getattr(t, k)) for k in self.TUNING_DEFAULTS}
        ok = ((new["chunk"] == 0 or (8 <= new["chunk"] <= 4096 and new["chunk"] & (new["chunk"] - 1) == 0))
              and 0 <= new["wide_form"] <= 3 and -1 <= new["bwd_split"] <= 1 and 0 <= new["rows16"] <= 7
              and new["agg_fused"] in (0, 1) and 0 <= new["gemm_forms"] <= 15)
        if not ok:
            self._err = b"acm_tuning_set: field out of range"
            return 1
        self._tuning = new
        return 0

    def acm_shard_plan(self, n_rows, indptr, world, row_cost, bounds):
        """numpy restatement of the host routine: nearest row boundary to p / world of the cost prefix."""
        ip = _vec(indptr, n_rows + 1, np.int64).astype(object)
        out = _vec(bounds, world + 1, np.int64)
        cost = [int(ip[r]) + int(row_cost) * r for r in range(n_rows + 1)]
        out[0] = 0
        for p in range(1, world):
            target = cost[n_rows] * p // world
            r = int(out[p - 1])
            while r < n_rows and cost[r] < target:
                r += 1
            if r > out[p - 1] and target - cost[r - 1] < cost[r] - target:
                r -= 1
            out[p] = r
        out[world] = n_rows
        return 0

    # ---- deferred second phases (acm_reduce_list_t): results of a deferred call are poisoned with NaN until the
    # flush, so a consumer that reads them too early fails its test instead of passing by accident
    def _emit(self, defer, writes):
        """writes: [(destination ndarray view, value)]"""
        if not defer:
            for dst, val in writes:
                dst[...] = val
            return 0
        from acm_gnn_amd._lib import ReduceList
        addr = defer.value if isinstance(defer, C.c_void_p) else int(defer)
        lst = ReduceList.from_address(addr)
        if lst.n + len(writes) > lst.cap:
            self._err = b"acm_reduce: deferral list full"
            return 5
        for dst, val in writes:
            dst[...] = np.nan
        self._pending.setdefault(addr, []).extend((dst, np.array(val, dtype=np.float64)) for dst, val in writes)
        lst.n += len(writes)
        return 0

    def acm_reduce_flush(self, lst_ref, stream):
        from acm_gnn_amd._lib import ReduceList
        lst = lst_ref._obj if hasattr(lst_ref, "_obj") else ReduceList.from_address(int(lst_ref))
        addr = C.addressof(lst)
        for dst, val in self._pending.pop(addr, []):
            dst[...] = val
        lst.n = 0
        return 0

    def acm_last_error(self):
        return self._err

    def _new(self, obj, out):
        h = self._next
        self._next += 1
        self._handles[h] = obj
        out._obj.value = h
        return 0

    def _get(self, h):
        return self._handles[h.value if isinstance(h, C.c_void_p) else int(h)]

    def acm_csr_create(self, n_rows, n_cols, nnz, indptr, indices, vals, chunk, out):
        ip = _vec(indptr, n_rows + 1, np.int32)
        ix = _vec(indices, nnz, np.int32) if nnz else np.zeros(0, np.int32)
        v = (_vec(vals, nnz, np.float32) if vals else np.ones(nnz, np.float32)) if nnz else np.zeros(0, np.float32)
        if ip[0] != 0 or ip[-1] != nnz or np.any(np.diff(ip) < 0) or (nnz and (ix.min() < 0 or ix.max() >= n_cols)):
            self._err = b"acm_csr_create: bad CSR"
            return 2
        obj = _Csr(ip, ix, v, n_cols, chunk)
        obj.unit = bool(nnz) and not vals                   # pattern-only: implicit ones
        return self._new(obj, out)

    def acm_csr_transpose(self, h, chunk, out):
        import scipy.sparse as sp
        a = self._get(h)
        m = sp.csr_matrix((a.vals, a.indices, a.indptr), shape=(a.n_rows, a.n_cols)).T.tocsr()
        m.sort_indices()
        t = _Csr(m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float32),
                 a.n_rows, chunk)
        pos = sp.csr_matrix((np.arange(1, len(a.vals) + 1, dtype=np.float64), a.indices, a.indptr),
                            shape=(a.n_rows, a.n_cols)).T.tocsr()
        pos.sort_indices()
        t.src_pos = (pos.data - 1).astype(np.int32)
        t.unit = getattr(a, "unit", False)
        return self._new(t, out)

    def acm_csr_slice_rows(self, h, b, e, chunk, out):
        a = self._get(h)
        ip = a.indptr[b:e + 1] - a.indptr[b]
        obj = _Csr(ip, a.indices[a.indptr[b]:a.indptr[e]], a.vals[a.indptr[b]:a.indptr[e]], a.n_cols, chunk)
        obj.unit = getattr(a, "unit", False)
        return self._new(obj, out)

    def acm_csr_destroy(self, h):
        self._handles.pop(h.value if isinstance(h, C.c_void_p) else int(h), None)

    def acm_csr_build_item_streams(self, h, n_waves):
        a = self._get(h)
        if np.any(a.vals != 1) or len(a.indices) == 0:
            self._err = b"acm_csr_build_item_streams: pattern-only operators only"
            return 4
        a.item_stream_waves = int(n_waves) if n_waves > 0 else 2048
        return 0

    def acm_csr_build_streams(self, h, n_waves, lmax):
        a = self._get(h)
        if not getattr(a, "unit", False):
            self._err = b"acm_csr_build_streams: pattern-only operators only"
            return 4
        lmax = lmax if lmax > 0 else 512
        deg = np.diff(a.indptr)
        pieces = np.maximum(1, -(-deg // lmax))
        a.stream_slices = int(-(-pieces.sum() // 4))
        a.stream_steps = max(a.stream_slices, int(-(-deg.sum() // 128)))      # a double: only "built" matters
        a.stream_waves = n_waves if n_waves > 0 else min(5120, -(-a.stream_slices // 4) * 4)
        a.stream_long = int((deg > lmax).sum())
        return 0

    def acm_csr_info(self, h, info):
        a, i = self._get(h), info._obj
        i.stream_steps, i.stream_slices = getattr(a, "stream_steps", 0), getattr(a, "stream_slices", 0)
        i.stream_waves, i.stream_long_rows = getattr(a, "stream_waves", 0), getattr(a, "stream_long", 0)
        i.item_stream_waves = getattr(a, "item_stream_waves", 0)
        deg = np.diff(a.indptr)
        longs = deg[deg > a.chunk]
        i.n_rows, i.n_cols, i.nnz = a.n_rows, a.n_cols, len(a.indices)
        i.n_long_rows = len(longs)
        # the library's work list (acm_csr.cpp, build_items): min(16, ceil(deg / chunk)) pieces per long row, packed
        # into windows of 16 items that no row straddles (gaps and the last window filled with empty pieces)
        # (a row that sixteen pieces of 4 x chunk do not cover takes ceil(deg / (32 chunk)) <= 16 WHOLE windows)
        used = 0
        for deg_r in longs:
            p, multi = int(-(-deg_r // a.chunk)), False
            if p > 16 and -(-deg_r // 16) > 4 * a.chunk:
                p, multi = 16 * min(16, int(-(-deg_r // (32 * a.chunk)))), True
            p = p if multi else min(p, 16)
            room = 16 - used % 16
            if (room < p or multi) and room < 16:
                used += room
            used += p
        used += (-used) % 16
        i.n_partial_slots = used
        i.n_items = a.n_rows - len(longs) + i.n_partial_slots
        i.chunk, i.max_degree = a.chunk, int(deg.max()) if len(deg) else 0
        i.indptr, i.indices = a.indptr.ctypes.data, a.indices.ctypes.data
        i.vals = None if getattr(a, "unit", False) else a.vals.ctypes.data
        sp_ = getattr(a, "src_pos", None)
        i.src_pos = sp_.ctypes.data if sp_ is not None else None
        return 0

    def acm_spmm_workspace_bytes(self, h, width, out):
        out._obj.value = 0
        return 0

    def acm_gemm_workspace_bytes(self, ta, tb, m, n, k, out):
        out._obj.value = 0
        return 0

    def acm_proj_bwd_workspace_bytes(self, n, f_in, q, out):
        if q not in (3, 6, 9, 12, 15):
            self._err = b"acm_proj_bwd: unsupported n_out"
            return 4
        out._obj.value = 4
        return 0

    def acm_proj_fwd(self, n, f_in, f, x, ldx, wl, wh, wm, ldw, relu, zlh, ld_lh, zi, ld_i, stream):
        return self.acm_proj_fwd_at(n, f_in, f, x, ldx, wl, wh, wm, ldw, relu, zlh, ld_lh, f, zi, ld_i, stream)

    def acm_proj_fwd_at(self, n, f_in, f, x, ldx, wl, wh, wm, ldw, relu, zlh, ld_lh, h_col, zi, ld_i, stream):
        X = _view(x, n, f_in, ldx).astype(np.float64)
        W = np.concatenate([_view(w, f_in, f, ldw).astype(np.float64) for w in (wl, wh, wm)], 1)
        out = X @ W
        if relu:
            out = np.maximum(out, 0)
        zv = _view(zlh, n, h_col + f, ld_lh)
        if h_col > f:
            zv[:, f:h_col] = np.nan                       # the pad columns are undefined: nothing may depend on them
        zv[:, :f] = out[:, :f]
        zv[:, h_col:h_col + f] = out[:, f:2 * f]
        _view(zi, n, f, ld_i)[...] = out[:, 2 * f:]
        return 0

    def acm_proj_bwd(self, n, f_in, q, x, ldx, dz, lddz, wl, wh, wm, ldw, dx, lddx, dw, lddw, cb, cbs, ws, wsb, defer, stream):
        X, DZ = _view(x, n, f_in, ldx).astype(np.float64), _view(dz, n, q, lddz).astype(np.float64)
        W = np.concatenate([_view(w, f_in, q // 3, ldw).astype(np.float64) for w in (wl, wh, wm)], 1)
        _view(dx, n, f_in, lddx)[...] = DZ @ W.T
        full = X.T @ DZ
        base = dw.value if isinstance(dw, C.c_void_p) else int(dw)
        writes = []
        if not cb:
            writes.append((_view(base, f_in, q, lddw), full))
        else:
            for j, q0 in enumerate(range(0, q, cb)):
                wd = min(cb, q - q0)
                writes.append((_view(base + 4 * j * cbs, f_in, wd, lddw), full[:, q0:q0 + wd]))
        return self._emit(defer, writes)

    def acm_conv_bwd_local_workspace_bytes(self, n, f, k, out):
        out._obj.value = 4
        return 0

    def acm_conv_agg_bwd_workspace_bytes(self, n, f_in, f, out):
        out._obj.value = 4
        return 0

    def acm_conv_fwd_tail_workspace_bytes(self, n, f, k, out):
        if f > 8 or k != 3:
            self._err = b"acm_conv_fwd_tail: needs f_out <= 8, three channels"
            return 4
        out._obj.value = 4
        return 0

    def acm_conv_fwd_tail(self, h, pp, ll, bb, ws, wsb, wt, wtb, stream):
        """The three calls it fuses, in order (same arguments, same outputs)."""
        p, l, b = pp._obj, ll._obj, bb._obj
        a = self._get(h)
        if p.f_out > 8 or p.n_channels != 3 or l.n_classes != p.f_out or p.post_relu or p.post_scale or p.post_drop.p > 0:
            self._err = b"acm_conv_fwd_tail: layer does not qualify"
            return 4
        if b.grad_out != l.dlogits or b.pre != p.pre:
            self._err = b"acm_conv_fwd_tail: bwd must read what fwd / loss write"
            return 1
        st = self.acm_conv_fwd(h, pp, ws, wsb, stream)
        if st:
            return st
        st = self.acm_nll_loss(a.n_rows, l.n_classes, p.out, p.ld_out, l.labels, l.row_weight, l.loss, l.dlogits,
                               l.ld_dlogits, wt, wtb, b.defer, stream)
        if st:
            return st
        return self.acm_conv_bwd_local(a.n_rows, bb, wt, wtb, stream)

    def acm_nll_loss_workspace_bytes(self, n, out):
        out._obj.value = 4
        return 0

    def acm_nll_loss(self, n, c, z, ldz, y, w, loss, dz, ldd, ws, wsb, defer, stream):
        Z = _view(z, n, c, ldz).astype(np.float64)
        Y = _vec(y, n, np.int64)
        W = _vec(w, n).astype(np.float64)
        m = Z.max(1, keepdims=True)
        e = np.exp(Z - m)
        s = e.sum(1, keepdims=True)
        lse = (m + np.log(s))[:, 0]
        onehot = np.zeros_like(Z)
        onehot[np.arange(n), Y] = 1.0
        _view(dz, n, c, ldd)[...] = W[:, None] * (e / s - onehot)
        return self._emit(defer, [(_vec(loss, 1), float((W * (lse - Z[np.arange(n), Y])).sum()))])

    # ---- evaluation metrics (ABI 28): accuracy per index set + NLL on one of them, from eval-mode logits
    def acm_eval_metrics_workspace_bytes(self, n, n_sets, out):
        out._obj.value = 4 * (int(n_sets) + 2)
        return 0

    def acm_eval_metrics(self, n, c, z, ldz, y, w, ldw, n_sets, loss_set, out, ws, wsb, stream):
        if not (1 <= n_sets <= 8 and 0 <= loss_set < n_sets and c <= 64):
            self._err = b"acm_eval_metrics: bad sizes"
            return 2
        self.eval_metrics_calls = getattr(self, "eval_metrics_calls", 0) + 1
        Z = _view(z, n, c, ldz).astype(np.float64)
        Y = _vec(y, n, np.int64)
        W = _view(w, n_sets, n, ldw).astype(np.float64)
        used = (W != 0).any(0)
        Ys = np.where(used, Y, 0)
        hit = (Z.argmax(1) == Ys) & used                 # (numpy's argmax takes the first maximum too)
        m = Z.max(1)
        nll = np.where(used, m + np.log(np.exp(Z - m[:, None]).sum(1)) - Z[np.arange(n), Ys], 0.0)
        res = _vec(out, n_sets + 1)
        res[:n_sets] = W @ hit.astype(np.float64)
        res[n_sets] = float((W[loss_set] * nll).sum())
        return 0

    # ---- compute ----------------------------------------------------------
    def acm_gemm(self, ta, tb, m, n, k, a, lda, b, ldb, c, ldc, relu, ws, wsb, stream):
        A = _view(a, k, m, lda).T if ta else _view(a, m, k, lda)
        B = _view(b, n, k, ldb).T if tb else _view(b, k, n, ldb)
        out = A.astype(np.float64) @ B.astype(np.float64)
        if relu:
            out = np.maximum(out, 0)
        _view(c, m, n, ldc)[...] = out
        return 0

    # ---- ACMII first layer, recompute on gather: K1 + K2 through the doubles of those two calls --------------------
    def acm_conv_acmii_fwd_workspace_bytes(self, handle, out):
        out._obj.value = 64
        return 0

    def acm_conv_acmii_fwd(self, handle, pp, ws, wsb, stream):
        from acm_gnn_amd import _lib
        p = pp._obj
        g = self._handles[handle.value if isinstance(handle, C.c_void_p) else int(handle)]
        n, F, fi = g.n_rows, p.f_out, p.f_in
        if F != 64 or p.f_pad != 8 or fi > 8:
            self._err = b"acm_conv_acmii_fwd: unsupported shape"
            return 4
        w = np.concatenate([_view(ptr, fi, F, p.ld_w) for ptr in (p.w_low, p.w_high, p.w_mlp)], 1).astype(np.float64)
        zg = np.maximum(_view(p.xg, g.n_cols, fi, p.ld_xg).astype(np.float64) @ w, 0)          # relu(X Wcat), all nodes
        zs = np.maximum(_view(p.xs, n, fi, p.ld_xs).astype(np.float64) @ w, 0)
        _view(p.zlh, n, 2 * F, p.ld_zlh)[...] = zs[:, :2 * F]
        _view(p.zi, n, F, p.ld_zi)[...] = zs[:, 2 * F:]
        self._keep_acmii = zg32 = np.ascontiguousarray(zg[:, :2 * F].astype(np.float32))
        q = _lib.ConvFwd()
        k = p.n_channels
        q.f_out, q.n_channels, q.relu_after, q.relu_mlp, q.layernorm, q.scale = F, k, 0, 1, p.layernorm, p.scale
        if k == 4:        # the call receives ps = A_low S already aggregated, acm_conv_fwd wants the parameter itself:
            return self._acmii_four(handle, p, zg, zs)      # restate the four-channel math directly
        q.g_low, q.ld_g_low = zg32.ctypes.data, 2 * F
        q.g_high, q.ld_g_high = zg32.ctypes.data + 4 * F, 2 * F
        q.s_high, q.ld_s_high = p.zlh + 4 * F, p.ld_zlh
        q.s_mlp, q.ld_s_mlp = p.zi, p.ld_zi
        for c in range(4):
            q.att_vec[c], q.ln_weight[c], q.ln_bias[c] = p.att_vec[c], p.ln_weight[c], p.ln_bias[c]
        q.att_mix, q.out, q.ld_out, q.pre, q.ld_pre, q.att = p.att_mix, p.out, p.ld_out, p.pre, p.ld_pre, p.att
        q.post_scale, q.ld_post_scale, q.post_relu, q.row_scale, q.post_drop = (p.post_scale, p.ld_post_scale, p.post_relu,
                                                                                p.row_scale, p.post_drop)
        return self.acm_conv_fwd(handle, C.byref(q), ws, wsb, stream)

    def _acmii_four(self, handle, p, zg, zs):
        """ACMII layer with the structure channel from the precomputed ps = A_low S (numpy, float64)."""
        g = self._handles[handle.value if isinstance(handle, C.c_void_p) else int(handle)]
        n, F = g.n_rows, p.f_out
        f64 = np.float64
        rs = _vec(p.row_scale, n).astype(f64)[:, None] if p.row_scale else 1.0
        pl = rs * g.dense_mul(zg[:, :F].astype(np.float32))
        ph = zs[:, F:2 * F] - rs * g.dense_mul(zg[:, F:2 * F].astype(np.float32))
        pS = _vec(p.deg, n).astype(f64)[:, None] * _view(p.ps, n, F, p.ld_ps).astype(f64) - _view(p.ss, n, F, p.ld_ss).astype(f64)
        H = [pl, ph, zs[:, 2 * F:], np.maximum(pS, 0)]
        vecs = [_vec(p.att_vec[c], F).astype(f64) for c in range(4)]
        lnw = [_vec(p.ln_weight[c], F).astype(f64) for c in range(4)] if p.layernorm else None
        lnb = [_vec(p.ln_bias[c], F).astype(f64) for c in range(4)] if p.layernorm else None
        hd = _head(H, 4, p.layernorm, vecs, lnw, lnb, _view(p.att_mix, 4, 4, 4).astype(f64))
        out = p.scale * sum(hd["alpha"][:, c:c + 1] * H[c] for c in range(4))
        _view(p.out, n, F, p.ld_out)[...] = _post_fwd(p, out, n, F)
        pre = _view(p.pre, n, 3 * F, p.ld_pre)
        pre[:, :F], pre[:, F:2 * F], pre[:, 2 * F:] = pl, ph, pS
        _view(p.att, n, 4, 4)[...] = hd["alpha"]
        return 0

    # ---- ACMII first layer, the mask form (acm_conv_acmii_v.hip): the table keeps what the kernels would read from it ----
    def acm_acmii_table_bytes(self, n_rows, out):
        out._obj.value = (int(n_rows) + 1) * 64
        return 0

    def acm_acmii_table(self, n_rows, f_in, x, ldx, w_low, w_high, ldw, table, table_bytes, stream):
        if f_in > 8 or table_bytes < (n_rows + 1) * 64:
            self._err = b"acm_acmii_table: unsupported"
            return 4
        key = table.value if isinstance(table, C.c_void_p) else int(table)
        if not hasattr(self, "_acmii_tables"):
            self._acmii_tables = {}
        self._acmii_tables[key] = dict(x=_view(x, n_rows, f_in, ldx).astype(np.float32).copy(),
                                       w_low=_view(w_low, f_in, 64, ldw).astype(np.float64).copy(),
                                       w_high=_view(w_high, f_in, 64, ldw).astype(np.float64).copy())
        return 0

    def _acmii_table(self, table):
        key = table.value if isinstance(table, C.c_void_p) else int(table)
        return getattr(self, "_acmii_tables", {}).get(key)

    def acm_conv_acmii_v_fwd(self, handle, pp, table, ws, wsb, stream):
        from acm_gnn_amd import _lib
        p, t = pp._obj, self._acmii_table(table)
        g = self._handles[handle.value if isinstance(handle, C.c_void_p) else int(handle)]
        if t is None or np.any(g.vals != 1) or t["x"].shape[0] != g.n_cols or not p.row_scale or not getattr(g, "item_stream_waves", 0):
            self._err = b"acm_conv_acmii_v_fwd: unsupported operator / unknown table / no item streams"
            return 4
        q = _lib.ConvAcmiiFwd.from_buffer_copy(p)
        xg = np.zeros((g.n_cols, 8), np.float32)
        xg[:, :t["x"].shape[1]] = t["x"]
        q.xg, q.ld_xg = xg.ctypes.data, 8
        zlh = None
        if not p.zlh:
            zlh = np.zeros((g.n_rows, 128), np.float32)
            q.zlh, q.ld_zlh = zlh.ctypes.data, 128
        return self.acm_conv_acmii_fwd(handle, C.byref(q), ws, wsb, stream)

    def acm_conv_acmii_v_bwd_workspace_bytes(self, handle, out):
        out._obj.value = 64
        return 0

    def acm_conv_acmii_v_bwd(self, handle, pp, ws, wsb, stream):
        p = pp._obj
        t = self._acmii_table(p.table)
        g = self._handles[handle.value if isinstance(handle, C.c_void_p) else int(handle)]
        if t is None or np.any(g.vals != 1) or t["x"].shape[0] != g.n_cols or not getattr(g, "item_stream_waves", 0):
            self._err = b"acm_conv_acmii_v_bwd: unsupported operator / unknown table / no item streams"
            return 4
        import scipy.sparse as sp
        n, nc, fi, off = g.n_rows, g.n_cols, p.f_in, int(p.self_offset)
        xg = t["x"].astype(np.float64)                                       # every column of the operator
        xs = _view(p.x, n, fi, p.ld_x).astype(np.float64)                     # its own rows
        a = sp.csr_matrix((g.vals.astype(np.float64), g.indices, g.indptr), shape=(n, nc))
        rs = _vec(p.row_scale, n).astype(np.float64)[:, None]
        gl, gh = _view(p.g_low, n, 64, p.ld_g_low).astype(np.float64), _view(p.g_high, n, 64, p.ld_g_high).astype(np.float64)
        ml, mh = (xg @ t["w_low"] > 0), (xg @ t["w_high"] > 0)
        d_l = xg.T @ (ml * (a.T @ (rs * gl)))                     # X^T (m_L o A_low^T G_L),  A_low = diag(rs) P
        d_h = xs.T @ (mh[off:off + n] * gh) - xg.T @ (mh * (a.T @ (rs * gh)))
        d_i = _view(p.x, n, fi, p.ld_x).astype(np.float64).T @ _view(p.g_mlp, n, 64, p.ld_g_mlp).astype(np.float64)
        return self._emit(p.defer, [(_view(p.d_w_low, fi, 64, p.ld_dw), d_l), (_view(p.d_w_high, fi, 64, p.ld_dw), d_h),
                                    (_view(p.d_w_mlp, fi, 64, p.ld_dw), d_i)])

    # ---- ACM-GCN++ residual branch ------------------------------------------------
    @staticmethod
    def _drop_obj(d):
        return None if d is None else getattr(d, "_obj", d)

    def acm_linear_fwd(self, n, f_in, f_out, x, ldx, w, ldw, bias, relu, drop, y, ldy, ws, wsb, stream):
        out = _view(x, n, f_in, ldx).astype(np.float64) @ _view(w, f_out, f_in, ldw).astype(np.float64).T
        if bias:
            out = out + _vec(bias, f_out).astype(np.float64)[None, :]
        if relu:
            out = np.maximum(out, 0)
        out = out * dropout_factors(self._drop_obj(drop), n, f_out)
        _view(y, n, f_out, ldy)[...] = out
        return 0

    def acm_bias_act(self, n, f, y, ldy, bias, relu, drop, stream):
        Y = _view(y, n, f, ldy)
        out = Y.astype(np.float64)
        if bias:
            out = out + _vec(bias, f).astype(np.float64)[None, :]
        if relu:
            out = np.maximum(out, 0)
        Y[...] = out * dropout_factors(self._drop_obj(drop), n, f)
        return 0

    def _lin_pre_and_factor(self, n, f_in, f_out, x, ldx, w, ldw, bias, drop):
        pre = _view(x, n, f_in, ldx).astype(np.float64) @ _view(w, f_out, f_in, ldw).astype(np.float64).T
        if bias:
            pre = pre + _vec(bias, f_out).astype(np.float64)
        d = self._drop_obj(drop)
        fac = dropout_factors(d, n, f_out) if (d is not None and d.p > 0) else np.ones((n, f_out))
        return pre, fac

    def acm_linear_fwd_add(self, n, f_in, f_out, x, ldx, w, ldw, bias, relu, drop, add, ld_add, y, ldy, stream):
        if f_in > 16 or f_out > 256:
            self._err = b"acm_linear_fwd_add: unsupported shape"
            return 4
        pre, fac = self._lin_pre_and_factor(n, f_in, f_out, x, ldx, w, ldw, bias, drop)
        if relu:
            pre = np.maximum(pre, 0)
        _view(y, n, f_out, ldy)[...] = _view(add, n, f_out, ld_add).astype(np.float64) + pre * fac
        return 0

    def acm_linear_bwd_recompute(self, n, f_in, f_out, x, ldx, w, ldw, bias, relu, drop, dy, lddy, dw, lddw, db, ws, wsb, defer,
                                 stream):
        pre, fac = self._lin_pre_and_factor(n, f_in, f_out, x, ldx, w, ldw, bias, drop)
        G = _view(dy, n, f_out, lddy).astype(np.float64) * fac
        if relu:
            G = np.where(pre > 0, G, 0.0)
        X = _view(x, n, f_in, ldx).astype(np.float64)
        return self._emit(defer, [(_view(dw, f_out, f_in, lddw), G.T @ X), (_vec(db, f_out), G.sum(0))])

    def acm_linear_bwd_workspace_bytes(self, n, f_in, f_out, out):
        if f_in > 16 or f_out > 256:
            self._err = b"acm_linear_bwd: unsupported shape"
            return 4
        out._obj.value = 64
        return 0

    def acm_linear_bwd(self, n, f_in, f_out, x, ldx, y, ldy, dy, lddy, keep, relu, dw, lddw, db, ws, wsb, defer, stream):
        Y, G = _view(y, n, f_out, ldy).astype(np.float64), _view(dy, n, f_out, lddy).astype(np.float64)
        if relu:
            G = np.where(Y > 0, G * keep, 0.0)
        elif keep != 1.0:
            G = np.where(Y != 0, G * keep, 0.0)
        X = _view(x, n, f_in, ldx).astype(np.float64)
        return self._emit(defer, [(_view(dw, f_out, f_in, lddw), G.T @ X), (_vec(db, f_out), G.sum(0))])

    def acm_bias_act_bwd_workspace_bytes(self, n, f, out):
        out._obj.value = 4 * f * min(max(n, 1), 1024)
        return 0

    def acm_bias_act_bwd(self, n, f, y, ldy, dy, lddy, keep_scale, relu, g, ldg, d_bias, ws, wsb, defer, stream):
        Y, dY = _view(y, n, f, ldy).astype(np.float64), _view(dy, n, f, lddy).astype(np.float64)
        if relu:
            G = np.where(Y > 0, dY * keep_scale, 0.0)
        elif keep_scale != 1.0:
            G = np.where(Y != 0, dY * keep_scale, 0.0)
        else:
            G = dY
        _view(g, n, f, ldg)[...] = G
        return self._emit(defer, [(_vec(d_bias, f), G.sum(0))])

    def acm_gemm_split(self, ta, tb, m, n, k, a, lda, b, ldb, c, ldc, split, c2, ldc2, relu, ws, wsb, stream):
        A = _view(a, k, m, lda).T if ta else _view(a, m, k, lda)
        B = _view(b, n, k, ldb).T if tb else _view(b, k, n, ldb)
        out = A.astype(np.float64) @ B.astype(np.float64)
        if relu:
            out = np.maximum(out, 0)
        _view(c, m, split, ldc)[...] = out[:, :split]
        _view(c2, m, n - split, ldc2)[...] = out[:, split:]
        return 0

    def acm_gemm_drop(self, ta, tb, m, n, k, a, lda, b, ldb, c, ldc, cb, cbs, relu, drop, ws, wsb, stream):
        """acm_gemm_blocks with the dropout applied to the stored matrix A (ABI 20); the shape envelope of the row-panel
        kernels is enforced like the library does."""
        d = self._drop_obj(drop)
        if d is None or d.p <= 0:
            return self.acm_gemm_blocks(ta, tb, m, n, k, a, lda, b, ldb, c, ldc, cb, cbs, relu, ws, wsb, stream)
        rows, cols = (k, m) if ta else (m, k)
        ok = (not tb and 1 <= n <= 192 and 16 <= cols and (self._tuning["gemm_forms"] & 9) != 0 and
              ((not ta and rows >= 4096 and cols <= 4096) or (ta and rows >= 8192 and cols <= 128)))
        if not ok:
            self._err = b"acm_gemm_drop: the dropout in the tile load exists in the row-panel kernels only"
            return 4
        A = _view(a, rows, cols, lda).astype(np.float64) * dropout_factors(d, rows, cols)
        A = A.T if ta else A
        out = A @ _view(b, k, n, ldb).astype(np.float64)
        if relu:
            out = np.maximum(out, 0)
        if not cb:
            _view(c, m, n, ldc)[...] = out
            return 0
        for j, n0 in enumerate(range(0, n, cb)):
            w = min(cb, n - n0)
            base = c.value if isinstance(c, C.c_void_p) else int(c)
            _view(base + 4 * j * cbs, m, w, ldc)[...] = out[:, n0:n0 + w]
        return 0

    def acm_proj3(self, n_rows, k, x, ldx, wl, wh, wm, ld_w, f, fb, c, ldc, split, c2, ldc2, relu, drop, stream):
        """[Z_L 0 | Z_H 0 | Z_I] = relu?(drop?(X) [W_L 0 | W_H 0 | W_I]) from the three weight matrices in place (ABI 21); the
        shape envelope of the split-bf16 kernel is enforced like the library does."""
        n_cols = 2 * fb + f
        ptr = x.value if isinstance(x, C.c_void_p) else int(x)
        if n_rows == 0:
            return 0
        if not (n_rows >= 8192 and 32 <= k <= 128 and k % 4 == 0 and ldx % 4 == 0 and ptr % 16 == 0 and n_cols <= 192
                and (self._tuning["gemm_forms"] & 2)):
            self._err = b"acm_proj3: the split-bf16 row-panel kernel takes >= 8192 rows of 32..128 features"
            return 4
        X = _view(x, n_rows, k, ldx).astype(np.float64)
        d = self._drop_obj(drop)
        if d is not None and d.p > 0:
            X = X * dropout_factors(d, n_rows, k)
        W = np.zeros((k, n_cols))
        for blk, w in enumerate((wl, wh, wm)):
            W[:, blk * fb: blk * fb + f] = _view(w, k, f, ld_w)
        out = X @ W
        if relu:
            out = np.maximum(out, 0)
        if split:
            _view(c, n_rows, split, ldc)[...] = out[:, :split]
            _view(c2, n_rows, n_cols - split, ldc2)[...] = out[:, split:]
        else:
            _view(c, n_rows, n_cols, ldc)[...] = out
        return 0

    def acm_gemm_blocks(self, ta, tb, m, n, k, a, lda, b, ldb, c, ldc, cb, cbs, relu, ws, wsb, stream):
        if not cb:
            return self.acm_gemm(ta, tb, m, n, k, a, lda, b, ldb, c, ldc, relu, ws, wsb, stream)
        A = _view(a, k, m, lda).T if ta else _view(a, m, k, lda)
        B = _view(b, n, k, ldb).T if tb else _view(b, k, n, ldb)
        out = A.astype(np.float64) @ B.astype(np.float64)
        if relu:
            out = np.maximum(out, 0)
        for j, n0 in enumerate(range(0, n, cb)):
            w = min(cb, n - n0)
            base = c.value if isinstance(c, C.c_void_p) else int(c)
            _view(base + 4 * j * cbs, m, w, ldc)[...] = out[:, n0:n0 + w]
        return 0

    def acm_cast_bf16(self, n, c, src, lds, dst, ldd, stream):
        b = _view(src, n, c, lds).copy().view(np.uint32)
        b = b + 0x7FFF + ((b >> 16) & 1)
        _view(dst, n, c, ldd, np.uint16)[...] = (b >> 16).astype(np.uint16)
        return 0

    def acm_spmm(self, h, g, ldg, width, y, ldy, ws, wsb, stream):
        return self.acm_spmm_v(h, None, g, ldg, width, y, ldy, 0, ws, wsb, stream)

    def acm_spmm_v(self, h, vals, g, ldg, width, y, ldy, relu, ws, wsb, stream):
        return self._spmm(h, g, ldg, width, y, ldy, vals=vals, relu=relu)

    def acm_spmm_ex(self, h, g, ldg, width, y, ldy, opts, ws, wsb, stream):
        o = opts._obj if opts else None
        if o is None:
            return self._spmm(h, g, ldg, width, y, ldy)
        return self._spmm(h, g, ldg, width, y, ldy, vals=o.vals, relu=o.relu, row_scale=o.row_scale, sub=o.sub,
                          ld_sub=o.ld_sub, sub_scale=o.sub_scale, bf16=o.g_bf16)

    def _spmm(self, h, g, ldg, width, y, ldy, vals=None, relu=0, row_scale=None, sub=None, ld_sub=0, sub_scale=None,
              bf16=0):
        import scipy.sparse as sp
        a = self._get(h)
        v = _vec(vals, len(a.vals)).astype(np.float64) if vals else a.vals.astype(np.float64)
        m = sp.csr_matrix((v, a.indices, a.indptr), shape=(a.n_rows, a.n_cols))
        if bf16:
            dense = (_view(g, a.n_cols, width, ldg, np.uint16).astype(np.uint32) << 16).view(np.float32)
        else:
            dense = _view(g, a.n_cols, width, ldg)
        out = m @ dense.astype(np.float64)
        if row_scale:
            out = _vec(row_scale, a.n_rows).astype(np.float64)[:, None] * out
        if sub:
            ss = _vec(sub_scale, a.n_rows).astype(np.float64)[:, None] if sub_scale else 1.0
            out = out - ss * _view(sub, a.n_rows, width, ld_sub)
        _view(y, a.n_rows, width, ldy)[...] = np.maximum(out, 0) if relu else out
        return 0

    @staticmethod
    def _params(p, k, F, layernorm):
        vecs = [_vec(p.att_vec[c], F).astype(np.float64) for c in range(k)]
        lnw = [_vec(p.ln_weight[c], F).astype(np.float64) for c in range(k)] if layernorm else None
        lnb = [_vec(p.ln_bias[c], F).astype(np.float64) for c in range(k)] if layernorm else None
        return vecs, lnw, lnb, _view(p.att_mix, k, k, k).astype(np.float64)

    def acm_conv_fwd(self, h, pp, ws, wsb, stream):
        a, p = self._get(h), pp._obj
        n, F, k = a.n_rows, p.f_out, p.n_channels
        f64 = np.float64
        if p.gather_bf16:
            def gat(ptr, ld):
                return (_view(ptr, a.n_cols, F, ld, np.uint16).astype(np.uint32) << 16).view(np.float32)
        else:
            def gat(ptr, ld):
                return _view(ptr, a.n_cols, F, ld)
        rs = _vec(p.row_scale, n).astype(f64)[:, None] if p.row_scale else 1.0
        pl = rs * a.dense_mul(gat(p.g_low, p.ld_g_low))
        ph = _view(p.s_high, n, F, p.ld_s_high).astype(f64) - rs * a.dense_mul(gat(p.g_high, p.ld_g_high))
        zi = _view(p.s_mlp, n, F, p.ld_s_mlp).astype(f64)
        act = (lambda t: np.maximum(t, 0)) if p.relu_after else (lambda t: t)
        H = [act(pl), act(ph), np.maximum(zi, 0) if p.relu_mlp else zi]
        pre = [pl, ph]
        if k == 4:
            deg = _vec(p.deg, n).astype(f64)[:, None]
            ps = deg * (rs * a.dense_mul(gat(p.g_struc, p.ld_g_struc))) - _view(p.s_struc, n, F, p.ld_s_struc)
            H.append(np.maximum(ps, 0))
            pre.append(ps)
        vecs, lnw, lnb, mix = self._params(p, k, F, p.layernorm)
        hd = _head(H, k, p.layernorm, vecs, lnw, lnb, mix)
        _view(p.out, n, F, p.ld_out)[...] = _post_fwd(p, p.scale * sum(hd["alpha"][:, c:c + 1] * H[c] for c in range(k)), n, F)
        _view(p.pre, n, (k - 1) * F, p.ld_pre)[...] = np.concatenate(pre, 1)
        att = _view(p.att, n, 4, 4)
        att[...] = 0
        att[:, :k] = hd["alpha"]
        return 0

    def acm_conv_head_fwd(self, n, pp, stream):
        """acm_conv_fwd's epilogue without a gather = acm_conv_fwd over the identity operator."""
        eye = _Csr(np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), np.ones(n, np.float32), n, 0)
        h = self._next
        self._next += 1
        self._handles[h] = eye
        try:
            return self.acm_conv_fwd(C.c_void_p(h), pp, None, 0, stream)
        finally:
            self._handles.pop(h, None)

    def acm_conv_aggw_fwd(self, n, f_in, f_pad, agg, ld_agg, xs, ld_xs, wl, wh, wm, ld_w, zi, ld_zi, pp, stream):
        """pre_L = P W_L, pre_H = (Xd - P) W_H, Z_I = Xd W_I (float64 products, stored fp32), then acm_conv_fwd's epilogue."""
        p = pp._obj
        F = p.f_out
        if F != 64 or p.n_channels != 3 or not 0 < f_in <= f_pad <= 128 or f_pad % 4:
            self._err = b"acm_conv_aggw_fwd: unsupported shape"
            return 4
        P = _view(agg, n, f_in, ld_agg).astype(np.float64)
        X = _view(xs, n, f_in, ld_xs).astype(np.float64)
        W = [_view(w, f_in, F, ld_w).astype(np.float64) for w in (wl, wh, wm)]
        zl = (P @ W[0]).astype(np.float32)
        zh = ((X - P) @ W[1]).astype(np.float32)
        z_i = (X @ W[2]).astype(np.float32)
        _view(zi, n, F, ld_zi)[...] = z_i
        zero = np.zeros((n, F), np.float32)
        q = type(p).from_buffer_copy(p)                       # the head over the identity: pre_L = zl, pre_H = zh - 0, Z_I
        q.g_low, q.ld_g_low = zl.ctypes.data, F
        q.g_high, q.ld_g_high = zero.ctypes.data, F
        q.s_high, q.ld_s_high = zh.ctypes.data, F
        q.s_mlp, q.ld_s_mlp = z_i.ctypes.data, F
        q.row_scale = None
        return FakeLib.acm_conv_head_fwd(self, n, C.byref(q), stream)      # (the class's own: tests wrap the instance's entry points)

    def acm_conv_bwd_local(self, n, qq, ws, wsb, stream):
        q = qq._obj
        F, k = q.f_out, q.n_channels
        f64 = np.float64
        pre = _view(q.pre, n, (k - 1) * F, q.ld_pre).astype(f64)
        zi = _view(q.s_mlp, n, F, q.ld_s_mlp).astype(f64)
        raw = [pre[:, :F], pre[:, F:2 * F], zi] + ([pre[:, 2 * F:3 * F]] if k == 4 else [])
        relu = [q.relu_after, q.relu_after, q.relu_mlp, 1][:k]
        pos = [(r > 0) if f else np.ones_like(r, bool) for r, f in zip(raw, relu)]
        H = [np.where(m, r, 0.0) for r, m in zip(raw, pos)]
        vecs, lnw, lnb, mix = self._params(q, k, F, q.layernorm)
        hd = _head(H, k, q.layernorm, vecs, lnw, lnb, mix)
        dO = _view(q.grad_out, n, F, q.ld_grad_out).astype(f64)
        dO = _post_bwd(q, dO, q.scale * sum(hd["alpha"][:, c:c + 1] * H[c] for c in range(k)), n, F)
        dH, d_vec, d_lnw, d_lnb, d_mix = _head_backward(H, hd, dO, k, q.layernorm, vecs, lnw, mix, q.scale)
        G = [np.where(m, d, 0.0) for d, m in zip(dH, pos)]
        gsc = _vec(q.g_scale, n).astype(f64)[:, None] if q.g_scale else 1.0
        _view(q.g_low, n, F, q.ld_g_low)[...] = gsc * G[0]
        _view(q.g_high, n, F, q.ld_g_high)[...] = gsc * G[1]
        _view(q.g_mlp, n, F, q.ld_g_mlp)[...] = G[2]
        if k == 4:
            _view(q.g_struc, n, F, q.ld_g_struc)[...] = (_vec(q.deg, n).astype(f64)[:, None] if q.deg else 1.0) * G[3]
        writes = []
        for c in range(k):
            writes.append((_vec(q.d_att_vec[c], F), d_vec[c]))
            if q.layernorm:
                writes.append((_vec(q.d_ln_weight[c], F), d_lnw[c]))
                writes.append((_vec(q.d_ln_bias[c], F), d_lnb[c]))
        writes.append((_view(q.d_att_mix, k, k, k), d_mix))
        return self._emit(q.defer, writes)

    def acm_conv_aggw_bwd_workspace_bytes(self, n, f_pad, out):
        out._obj.value = 1024
        return 0

    def acm_conv_aggw_bwd(self, n, f_in, f_pad, agg, ld_agg, xs, ld_xs, qq, dwl, dwh, dwm, ld_dw, ws, wsb, stream):
        """acm_conv_bwd_local's G tables (kept in host arrays) contracted with P, Xd - P, Xd: dW_L, dW_H, dW_I."""
        q = qq._obj
        F = q.f_out
        if F != 64 or q.n_channels != 3 or q.post_scale or q.g_scale or not 0 < f_in <= f_pad <= 128:
            self._err = b"acm_conv_aggw_bwd: unsupported configuration"
            return 4
        G = [np.zeros((n, F), np.float32) for _ in range(3)]
        r = type(q).from_buffer_copy(q)
        r.g_low, r.g_high, r.g_mlp = (t.ctypes.data for t in G)
        r.ld_g_low = r.ld_g_high = r.ld_g_mlp = F
        st = FakeLib.acm_conv_bwd_local(self, n, C.byref(r), ws, wsb, stream)
        if st:
            return st
        P = _view(agg, n, f_in, ld_agg).astype(np.float64)
        X = _view(xs, n, f_in, ld_xs).astype(np.float64)
        prods = (P.T @ G[0].astype(np.float64), (X - P).T @ G[1].astype(np.float64), X.T @ G[2].astype(np.float64))
        return self._emit(q.defer, [(_view(dst, f_in, F, ld_dw), val) for dst, val in zip((dwl, dwh, dwm), prods)])

    def acm_conv_bwd_spmm(self, h, rr, ws, wsb, stream):
        at, r = self._get(h), rr._obj
        n, F = at.n_rows, r.f_out
        f64 = np.float64

        def table(ptr, ld):                       # the gathered operand: fp32, or bf16 widened exactly (ABI 20)
            if r.gather_bf16:
                return (_view(ptr, at.n_cols, F, ld, np.uint16).astype(np.uint32) << 16).view(np.float32)
            return _view(ptr, at.n_cols, F, ld)
        if r.gather_bf16 and not (8 < F <= 64 and F % 2 == 0):
            self._err = b"acm_conv_bwd_spmm: bf16 gathered operands are implemented for even 8 < F <= 64"
            return 4
        dl = at.dense_mul(table(r.g_low, r.ld_g_low))
        ssc = _vec(r.self_scale, n).astype(f64)[:, None] if r.self_scale else 1.0
        dh = ssc * _view(r.s_high, n, F, r.ld_s_high).astype(f64) - at.dense_mul(table(r.g_high, r.ld_g_high))
        if r.mask_low:
            dl = np.where(_view(r.mask_low, n, F, r.ld_mask_low) > 0, dl, 0.0)
        if r.mask_high:
            dh = np.where(_view(r.mask_high, n, F, r.ld_mask_high) > 0, dh, 0.0)
        _view(r.dz_low, n, F, r.ld_dz_low)[...] = dl
        _view(r.dz_high, n, F, r.ld_dz_high)[...] = dh
        if r.g_struc:
            ds = at.dense_mul(table(r.g_struc, r.ld_g_struc)) - \
                _view(r.s_struc, n, F, r.ld_s_struc).astype(f64) * (_vec(r.inv_deg, n).astype(f64)[:, None] if r.inv_deg else 1.0)
            _view(r.d_struc, n, F, r.ld_d_struc)[...] = ds
        return 0

    def _agg_common(self, p, n):
        F, fi, fp = p.f_out, p.f_in, p.f_pad
        f64 = np.float64
        x = _view(p.xs, n, fp, p.ld_xs).astype(f64)[:, :fi]
        W = [_view(w, fi, F, p.ld_w).astype(f64) for w in (p.w_low, p.w_high, p.w_mlp)]
        return F, fi, fp, x, W

    def acm_conv_agg_fwd(self, h, pp, ws, wsb, stream):
        a, p = self._get(h), pp._obj
        n, k = a.n_rows, p.n_channels
        F, fi, fp, x, W = self._agg_common(p, n)
        rs = _vec(p.row_scale, n).astype(np.float64)[:, None] if p.row_scale else 1.0
        if p.agg_given:                                    # P = A_low X from an earlier call
            P = _view(p.agg, n, fp, p.ld_agg).astype(np.float64)[:, :fi].copy()
        else:
            P = (rs * a.dense_mul(_view(p.xg, a.n_cols, fp, p.ld_xg)))[:, :fi]
        raw = [P @ W[0], (x - P) @ W[1], x @ W[2]]
        relu = [p.relu_after, p.relu_after, p.relu_mlp]
        if k == 4:
            if p.sg_bf16:
                sg = (_view(p.sg, a.n_cols, F, p.ld_sg, np.uint16).astype(np.uint32) << 16).view(np.float32)
            else:
                sg = _view(p.sg, a.n_cols, F, p.ld_sg)
            PS = rs * a.dense_mul(sg)
            _view(p.ps, n, F, p.ld_ps)[...] = PS
            PS = _view(p.ps, n, F, p.ld_ps).astype(np.float64)
            raw.append(_vec(p.deg, n).astype(np.float64)[:, None] * PS - _view(p.ss, n, F, p.ld_ss))
            relu.append(1)
        H = [np.maximum(r, 0) if f else r for r, f in zip(raw, relu)]
        vecs, lnw, lnb, mix = self._params(p, k, F, p.layernorm)
        hd = _head(H, k, p.layernorm, vecs, lnw, lnb, mix)
        _view(p.out, n, F, p.ld_out)[...] = _post_fwd(p, p.scale * sum(hd["alpha"][:, c:c + 1] * H[c] for c in range(k)), n, F)
        if p.next_f > 0:                                   # the following layer's narrow projection of `out`
            f2 = p.next_f
            o = _view(p.out, n, F, p.ld_out).astype(np.float64)
            z = [o @ _view(w, F, f2, p.next_ld_w).astype(np.float64) for w in (p.next_w_low, p.next_w_high, p.next_w_mlp)]
            if p.next_relu:
                z = [np.maximum(t, 0) for t in z]
            _view(p.next_zlh, n, 2 * f2, p.ld_next_zlh)[...] = np.concatenate(z[:2], 1)
            _view(p.next_zi, n, f2, p.ld_next_zi)[...] = z[2]
        agg = _view(p.agg, n, fp, p.ld_agg)
        agg[...] = 0
        agg[:, :fi] = P
        if p.agg_copy:                                      # the backward's operands, left by the row-local stage
            if not p.agg_given or not p.xs_copy:
                self._err = b"acm_conv_agg_fwd: agg_copy needs agg_given and xs_copy"
                return 1
            _view(p.agg_copy, n, fp, p.ld_agg_copy)[...] = agg
            _view(p.xs_copy, n, fp, p.ld_xs_copy)[...] = _view(p.xs, n, fp, p.ld_xs)
            if p.next_x:                                    # ... and refills xs for the next step (ABI 22)
                nxt = np.zeros((n, fp))
                nxt[:, :fi] = _view(p.next_x, n, fi, p.ld_next_x).astype(np.float64) * dropout_factors(p.next_drop, n, fi)
                _view(p.xs, n, fp, p.ld_xs)[...] = nxt
        att = _view(p.att, n, 4, 4)
        att[...] = 0
        att[:, :k] = hd["alpha"]
        if p.head_stats:                                    # mean | rstd | sigmoid | alpha per channel
            st = _view(p.head_stats, n, 4 * k, p.ld_head_stats)
            for c in range(k):
                st[:, c] = H[c].mean(1) if p.layernorm else 0.0
                st[:, k + c] = np.asarray(hd["rstd"][c]).reshape(-1) if p.layernorm else 1.0
                st[:, 2 * k + c] = hd["g"][:, c]
                st[:, 3 * k + c] = hd["alpha"][:, c]
        return 0

    def acm_conv_agg_bwd(self, n, qq, ws, wsb, stream):
        q = qq._obj
        k = q.n_channels
        F, fi, fp, x, W = self._agg_common(q, n)
        P = _view(q.agg, n, fp, q.ld_agg).astype(np.float64)[:, :fi]
        A = [P, x - P, x]
        raw = [A[c] @ W[c] for c in range(3)]
        relu = [q.relu_after, q.relu_after, q.relu_mlp]
        if k == 4:
            deg = _vec(q.deg, n).astype(np.float64)[:, None]
            raw.append(deg * _view(q.ps, n, F, q.ld_ps).astype(np.float64) - _view(q.ss, n, F, q.ld_ss))
            relu.append(1)
        pos = [(r > 0) if f else np.ones_like(r, bool) for r, f in zip(raw, relu)]
        H = [np.where(m, r, 0.0) for r, m in zip(raw, pos)]
        vecs, lnw, lnb, mix = self._params(q, k, F, q.layernorm)
        hd = _head(H, k, q.layernorm, vecs, lnw, lnb, mix)
        extra = []
        if q.proj_dz:                             # the following layer's projection backward rides along (ABI 20)
            f2 = q.proj_f
            if not self._bwd16_ok(q, k, fp, F) or not q.out or not 1 <= f2 <= 2:
                self._err = b"acm_conv_agg_bwd: proj_dz: unsupported configuration"
                return 4
            dz2 = _view(q.proj_dz, n, 3 * f2, q.ld_proj_dz).astype(np.float64)
            wcat = np.concatenate([_view(ptr, F, f2, q.proj_ld_w) for ptr in (q.proj_w_low, q.proj_w_high, q.proj_w_mlp)], 1)
            dO = dz2 @ wcat.astype(np.float64).T
            outf = _view(q.out, n, F, q.ld_out).astype(np.float64)
            dw2 = outf.T @ dz2                                              # [F, 3 f2] -> three contiguous F x f2 blocks
            extra = [(_vec(q.proj_d_w, 3 * F * f2), np.concatenate([dw2[:, c * f2:(c + 1) * f2].reshape(-1) for c in range(3)]))]
        else:
            dO = _view(q.grad_out, n, F, q.ld_grad_out).astype(np.float64)
        dO = _post_bwd(q, dO, q.scale * sum(hd["alpha"][:, c:c + 1] * H[c] for c in range(k)), n, F)
        dH, d_vec, d_lnw, d_lnb, d_mix = _head_backward(H, hd, dO, k, q.layernorm, vecs, lnw, mix, q.scale)
        G = [np.where(m, d, 0.0) for d, m in zip(dH, pos)]
        if k == 4:
            _view(q.g_struc, n, F, q.ld_g_struc)[...] = \
                (_vec(q.g_struc_scale, n).astype(np.float64)[:, None] if q.g_struc_scale else 1.0) * G[3]
        npg = 3 * fi * F + 3 * k * F + k * k
        dst = _vec(q.d_params, npg)
        out = np.zeros(npg)
        for c in range(3):
            out[c * fi * F:(c + 1) * fi * F] = (A[c].T @ G[c]).reshape(-1)
        base = 3 * fi * F
        for c in range(k):
            out[base + c * F: base + (c + 1) * F] = d_vec[c]
            if q.layernorm:
                out[base + (k + c) * F: base + (k + 1 + c) * F] = d_lnw[c]
                out[base + (2 * k + c) * F: base + (2 * k + 1 + c) * F] = d_lnb[c]
        out[base + 3 * k * F:] = d_mix.reshape(-1)
        if q.next_agg:                            # the next step's input aggregation rides along
            a = self._get(q.next_a)
            if (not self._bwd16_ok(q, k, fp, F) or not getattr(a, "stream_waves", 0) or a.stream_waves % 4
                    or a.stream_waves > 1024):
                self._err = b"acm_conv_agg_bwd: carried gather: unsupported configuration"
                return 4
            if a.stream_waves // 4 > min((n + 15) // 16, 768):
                self._err = b"acm_conv_agg_bwd: too many stream waves for the workspace"
                return 4
            xg = _view(q.next_xg, a.n_cols, 8, q.ld_next_xg).astype(np.float64)
            import scipy.sparse as sp
            pn = sp.csr_matrix((a.vals.astype(np.float64), a.indices, a.indptr), shape=(a.n_rows, a.n_cols)) @ xg
            if q.next_row_scale:
                pn = pn * _vec(q.next_row_scale, a.n_rows).astype(np.float64)[:, None]
            _view(q.next_agg, a.n_rows, 8, q.ld_next_agg)[...] = pn
        return self._emit(q.defer, [(dst, out)] + extra)

    def _bwd16_ok(self, q, k, fp, F):
        """The envelope of the sixteen-rows-per-wave backward (acm_conv_agg16.hip: acm_agg_bwd16), the only carrier of
        proj_* and next_agg."""
        out_mask = bool(q.out) and bool(q.post_relu) and not q.post_scale
        no_post = not q.post_relu and not q.post_scale and not q.post_drop.p > 0
        return (fp == 8 and F == 64 and bool(q.head_stats) and out_mask and (self._tuning["rows16"] & 2) != 0)

    # ---- the fused small-graph step (ABI 25): stated with the oracle's layer + torch autograd (float64) -------------
    def acm_small_step_workspace_bytes(self, a, x, xt, out):
        out._obj.value = 4096
        return 0

    def acm_small_step(self, a, x, xt, pp, stream):
        import scipy.sparse as sp
        import torch
        from acm_gnn_amd import _lib
        from oracle import acm_oracle as O
        p = pp._obj
        g = self._get(a)
        n, C_, k = g.n_rows, int(p.n_classes), int(p.n_channels)
        if np.any(g.vals != 1) or g.n_rows != g.n_cols or C_ > 8 or n > 16384:
            self._err = b"acm_small_step: outside the envelope"
            return 4
        self.small_calls = getattr(self, "small_calls", 0) + 1
        pat = sp.csr_matrix((np.ones(len(g.indices)), g.indices, g.indptr), shape=(n, n))
        rs = _vec(p.row_scale, n).astype(np.float64)
        low = torch.from_numpy((sp.diags(rs) @ pat).toarray())
        high = torch.eye(n, dtype=torch.float64) - low
        raw = torch.from_numpy((pat - sp.identity(n)).toarray())
        fx = self._get(x)
        f_in = int(p.f_in)
        vals = _vec(p.x_vals, len(fx.indices)).astype(np.float64)
        train = bool(p.train)
        if train:
            vals = vals * dropout_factors(p.drop_in, len(vals), 1)[:, 0]
        xd = torch.from_numpy(sp.csr_matrix((vals, fx.indices, fx.indptr), shape=(n, f_in)).toarray())
        names = {_lib.SR_W_LOW: "weight_low", _lib.SR_W_HIGH: "weight_high", _lib.SR_W_MLP: "weight_mlp", _lib.SR_V_LOW: "att_vec_low",
                 _lib.SR_V_HIGH: "att_vec_high", _lib.SR_V_MLP: "att_vec_mlp", _lib.SR_V_STRUC: "att_struc_low",
                 _lib.SR_LNW_LOW: "layer_norm_low.weight", _lib.SR_LNW_HIGH: "layer_norm_high.weight", _lib.SR_LNW_MLP: "layer_norm_mlp.weight",
                 _lib.SR_LNW_STRUC: "layer_norm_struc_low.weight", _lib.SR_LNB_LOW: "layer_norm_low.bias",
                 _lib.SR_LNB_HIGH: "layer_norm_high.bias", _lib.SR_LNB_MLP: "layer_norm_mlp.bias", _lib.SR_LNB_STRUC: "layer_norm_struc_low.bias",
                 _lib.SR_MIX: "att_vec", _lib.SR_STRUC: "struc_low"}
        dims = [(f_in, 64), (64, C_)]

        def shape(li, role):
            fi, fo = dims[li]
            if role <= _lib.SR_W_MLP:
                return (fi, fo)
            if role == _lib.SR_MIX:
                return (k, k)
            if role == _lib.SR_STRUC:
                return (n, fo)
            return (fo, 1) if role <= _lib.SR_V_STRUC else (fo,)

        views, leaves = {}, [{}, {}]
        for li in range(2):
            for role, nm in names.items():
                t = p.t[li][role]
                if not t.param:
                    continue
                sh = shape(li, role)
                v = _vec(t.param, int(np.prod(sh)))
                views[(li, role)] = (v, t)
                leaves[li][nm] = torch.from_numpy(v.astype(np.float64).reshape(sh)).requires_grad_(train)
        kw = dict(model_type="acmgcnp", variant=bool(p.relu_before), structure_info=int(k == 4), attn_layernorm=bool(p.layernorm))
        un = raw if k == 4 else None
        fea, att1 = O.layer_forward(leaves[0], xd, low, high, un, return_att=True, **kw)
        fea = torch.relu(fea)
        if train:
            fea = fea * torch.from_numpy(dropout_factors(p.drop_hidden, n, 64))
        out, att2 = O.layer_forward(leaves[1], fea, low, high, un, return_att=True, **kw)
        _view(p.logits, n, C_, C_)[...] = out.detach().numpy()
        _view(p.att1, n, 4, 4)[:, :k] = att1.detach().numpy()
        _view(p.att2, n, 4, 4)[:, :k] = att2.detach().numpy()
        if not train:
            return 0
        y = torch.from_numpy(_vec(p.labels, n, np.int64).copy())
        w = torch.from_numpy(_vec(p.row_weight, n).astype(np.float64))
        loss = -(w * torch.log_softmax(out, 1).gather(1, y.view(-1, 1)).view(-1)).sum()
        loss.backward()
        _vec(p.loss, 1)[0] = float(loss.detach())
        f32 = np.float32
        stepped = set()
        for (li, role), (v, t) in views.items():
            gr = leaves[li][names[role]].grad
            gr = (torch.zeros_like(leaves[li][names[role]]) if gr is None else gr).numpy().reshape(-1).astype(np.float32)
            if t.grad:
                _vec(t.grad, len(v))[...] = gr
            if not p.update:
                continue
            m, vv, step = _vec(t.exp_avg, len(v)), _vec(t.exp_avg_sq, len(v)), _vec(t.step, 1)
            kk = float(step[0]) + 1.0
            step_size = f32(p.lr / (1.0 - p.beta1 ** kk))
            bc2_sqrt = f32((1.0 - p.beta2 ** kk) ** 0.5)
            gg = gr.copy()
            if p.weight_decay != 0:
                if p.decoupled:
                    v *= f32(1.0 - p.lr * p.weight_decay)
                else:
                    gg = gg + f32(p.weight_decay) * v
            m += f32(1.0 - p.beta1) * (gg - m)
            vv[...] = vv * f32(p.beta2) + f32(1.0 - p.beta2) * gg * gg
            v -= step_size * (m / (np.sqrt(vv) / bc2_sqrt + f32(p.eps)))
            stepped.add(int(t.step))
        if p.update:
            for addr in stepped:
                _vec(addr, 1)[0] += 1.0
            if p.also_advance:
                _vec(p.also_advance, 1, np.int64)[0] += 1
        return 0

    def acm_dropout(self, n, c, src, lds, dst, ldd, dst_cols, d, stream):
        out = np.zeros((n, dst_cols))
        out[:, :c] = _view(src, n, c, lds).astype(np.float64) * dropout_factors(d._obj, n, c)
        _view(dst, n, dst_cols, ldd)[...] = out
        return 0

    def acm_adam_step(self, n, tensors, cfg, stream):
        import ctypes as C
        from acm_gnn_amd import _lib
        c = cfg._obj
        if c.pending:                                   # ABI 23: the call flushes the step's deferred second phases itself
            self.adam_flushed = getattr(self, "adam_flushed", 0) + 1
            self.acm_reduce_flush(c.pending, stream)
        if c.also_advance:
            _vec(c.also_advance, 1, np.int64)[0] += 1
        arr = C.cast(tensors, C.POINTER(_lib.AdamTensor * n)).contents if n else []
        f32 = np.float32
        for t in arr:
            k = int(t.numel)
            p, g, m, v = (_vec(ptr, k) for ptr in (t.param, t.grad, t.exp_avg, t.exp_avg_sq))
            step = _vec(t.step, 1)
            kk = float(step[0]) + 1.0
            step_size = f32(c.lr / (1.0 - c.beta1 ** kk))
            bc2_sqrt = f32((1.0 - c.beta2 ** kk) ** 0.5)
            gg = g.copy()
            if c.weight_decay != 0:
                if c.decoupled:
                    p *= f32(1.0 - c.lr * c.weight_decay)
                else:
                    gg = gg + f32(c.weight_decay) * p
            m += f32(1.0 - c.beta1) * (gg - m)
            v[...] = v * f32(c.beta2) + f32(1.0 - c.beta2) * gg * gg
            p -= step_size * (m / (np.sqrt(v) / bc2_sqrt + f32(c.eps)))
            step[0] = kk
        return 0


def install(monkeypatch):
    """Route acm_gnn_amd through the test double and lift its GPU-only guards (CPU tests only)."""
    from acm_gnn_amd import _lib, functional, graph, optim, tuning
    fake = FakeLib()
    monkeypatch.setattr(tuning, "_kernel_cache", None)        # the cached acm_tuning_t belongs to whichever library was loaded
    monkeypatch.setattr(_lib, "load", lambda build_if_missing=True: fake)
    monkeypatch.setattr(_lib, "check", lambda st, what="": (_ for _ in ()).throw(
        RuntimeError(f"{what}: {fake.acm_last_error().decode()} ({_lib.STATUS_NAMES.get(st, st)})")) if st else None)
    from acm_gnn_amd import small
    for mod in (graph, functional, optim):
        monkeypatch.setattr(mod, "_require_cuda", lambda t, name: None)
        monkeypatch.setattr(mod, "_stream", lambda: None)
    monkeypatch.setattr(small, "_stream", lambda: None)
    monkeypatch.setattr(small, "_device_ctx", lambda dev: contextlib.nullcontext())
    monkeypatch.setattr(small, "_ON_DEVICE", lambda t: True)
    monkeypatch.setattr(functional, "_device_ctx", lambda dev: contextlib.nullcontext())
    monkeypatch.setattr(optim, "_device_ctx", lambda dev: contextlib.nullcontext())
    monkeypatch.setattr(graph, "_device_ctx", lambda dev: contextlib.nullcontext())
    monkeypatch.setattr(graph, "_sync", lambda dev: None)

    def copy_from_ptr(ptr, n, dtype, device):
        import torch
        npdt = {torch.int32: np.int32, torch.float32: np.float32}[dtype]
        return torch.from_numpy(_vec(ptr, n, npdt).copy())

    monkeypatch.setattr(graph, "_copy_from_ptr", copy_from_ptr)
    import torch
    from acm_gnn_amd import layers
    monkeypatch.setattr(layers, "_default_device", lambda: torch.device("cpu"))
    graph.clear_cache()
    return fake
