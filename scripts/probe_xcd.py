#!/usr/bin/env python3
"""Would a column partition of the COLD neighbours over the XCDs pay?  (the layer-1 gather is bound by
the 22 % of its row requests that miss the XCD's L2 -- the table is 5.4 MB against 4 MB of L2 and every XCD needs all of
it.)  Split P = A X into
    hot part : neighbours with id < H (degree order: the hubs), gathered as today -- the table is H x 32 B, L2-resident;
    cold part: neighbours with id >= H, column ranges dealt to the eight XCDs (block b serves range b % 8), one lane pair
               per (row, range) item, partial sums written to partial[row][range] (scripts/micro/sell_gather.hip, (d)).
Prints us for the hot gather (streamed and CSR kernel), the cold pass, and checks hot + sum of partials against scipy."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from acm_gnn_amd import data as D, functional as AF  # noqa: E402
from acm_gnn_amd.graph import CsrGraph  # noqa: E402
from probe_sell import build_streams, SENT8  # noqa: E402

DEV = torch.device("cuda:0")


def build_cold(indptr, indices, hot, n, n_waves, pad_steps=8):
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
    m = indices >= hot
    r, c = rows[m], indices[m].astype(np.int64)
    cnt = np.bincount(c - hot, minlength=n - hot)
    cum = np.cumsum(cnt)
    cuts = np.searchsorted(cum, np.arange(1, 8) * cum[-1] / 8.0)          # equal edge counts per range
    owner = np.searchsorted(cuts, c - hot, side="right")
    key = r * 8 + owner
    start = np.concatenate([[0], np.flatnonzero(np.diff(key)) + 1])
    item_key, item_begin = key[start], start
    item_len = np.diff(np.concatenate([start, [key.size]]))
    item_owner = item_key % 8
    wpx = n_waves // 8                                                     # local waves per XCD
    slices = []                                                            # (global wave, round, steps, item ids[<=32])
    for k in range(8):
        sel = np.flatnonzero(item_owner == k)
        sel = sel[np.argsort(-item_len[sel], kind="stable")]
        ns = -(-sel.size // 32)
        for j in range(ns):
            its = sel[j * 32:(j + 1) * 32]
            lw = j % wpx
            gw = ((lw // 4) * 8 + k) * 4 + lw % 4
            slices.append((gw, j // wpx, int(item_len[its[0]]), its))
    slices.sort(key=lambda t: (t[0], t[1]))
    n_sl = len(slices)
    steps = np.array([t[2] for t in slices], np.int64)
    step_base = np.cumsum(steps) - steps
    wave_of = np.array([t[0] for t in slices], np.int64)
    wave_ptr = np.zeros(n_waves + 1, np.int64)
    np.add.at(wave_ptr, wave_of + 1, 1)
    wave_ptr = np.cumsum(wave_ptr)
    total = int(steps.sum())
    wave_step = np.full(n_waves, total, np.int64)
    has = wave_ptr[1:] > wave_ptr[:-1]
    wave_step[has] = step_base[wave_ptr[:-1][has]]
    desc = np.full((n_sl, 40), -1, np.int32)
    desc[:, 0] = steps
    stream = np.full((total + pad_steps) * 32, SENT8, np.int32)
    it_all = np.concatenate([t[3] for t in slices])
    sl_of = np.repeat(np.arange(n_sl), [t[3].size for t in slices])
    q_of = np.concatenate([np.arange(t[3].size) for t in slices])
    desc[sl_of, 8 + q_of] = item_key[it_all]
    ln = item_len[it_all]
    kk = np.arange(int(ln.sum())) - np.repeat(np.cumsum(ln) - ln, ln)
    pos = (np.repeat(step_base[sl_of], ln) + kk) * 32 + np.repeat(q_of, ln)
    stream[pos] = c[np.repeat(item_begin[it_all], ln) + kk]
    per_xcd = np.bincount(owner, minlength=8)
    return stream, wave_ptr.astype(np.int32), wave_step.astype(np.int32), desc, total, item_key.size, per_xcd


def main():
    so = "/tmp/sell_gather.so"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-w", "-o", so,
                           os.path.join(ROOT, "scripts", "micro", "sell_gather.hip")])
    lib = C.CDLL(so)
    lib.sell_gather.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_uint, C.c_void_p, C.c_int, C.c_void_p]
    lib.cold_gather.argtypes = [C.c_void_p] * 5 + [C.c_uint, C.c_void_p, C.c_int, C.c_void_p]
    adj, x_np, y_np, (tr, _, _), n = D.synthetic_dataset("twitch-gamer")
    perm = D.degree_order(adj)
    adj, *_ = D.permute_dataset(adj, x_np, y_np, (tr, tr, tr), perm)
    low, deg = D.build_filters(adj)
    low = low.tocsr()
    low.sort_indices()
    pat = low.copy()
    pat.data[:] = 1.0
    indptr, indices = low.indptr.astype(np.int64), low.indices.astype(np.int32)
    nnz = low.nnz
    sh = torch.cuda.current_stream().cuda_stream
    x = torch.randn(n, 8, device=DEV)
    ref = pat @ x.cpu().numpy().astype(np.float64)

    def timeit(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    g_full = CsrGraph.from_csr(torch.from_numpy(low.indptr.astype(np.int32)).to(DEV), torch.from_numpy(indices).to(DEV), None, n)
    y = torch.empty(n, 8, device=DEV)
    print(f"n {n} nnz {nnz}; full gather, CSR kernel: {timeit(lambda: AF.spmm(g_full, x, out=y)):.1f} us", flush=True)
    for hot in (8192, 16384, 32768, 65536):
        keep = indices < hot
        rows = np.repeat(np.arange(n), np.diff(indptr))
        hot_csr = sp.csr_matrix((np.ones(int(keep.sum()), np.float32), (rows[keep], indices[keep])), shape=(n, n))
        hot_csr.sort_indices()
        hip, hix = hot_csr.indptr.astype(np.int64), hot_csr.indices.astype(np.int32)
        g_hot = CsrGraph.from_csr(torch.from_numpy(hot_csr.indptr.astype(np.int32)).to(DEV), torch.from_numpy(hix).to(DEV), None, n)
        t_hot_csr = timeit(lambda: AF.spmm(g_hot, x, out=y))
        y_hot = y.cpu().double().numpy()
        line = [f"H {hot:6d}: hot edges {keep.mean():.3f}  hot gather CSR kernel {t_hot_csr:6.1f} us"]
        for n_waves in (4096, 8192):
            stream, wptr, wstep, desc, item_row, total = build_streams(hip, hix, n_waves, max_steps=8)
            d = [torch.from_numpy(a).to(DEV) for a in (stream, wptr, wstep, desc)]
            out = torch.zeros(item_row.size, 8, device=DEV)

            def run_hot():
                st = lib.sell_gather(8, 2, 1, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), x.data_ptr(),
                                     n * 32, out.data_ptr(), n_waves, sh)
                assert st == 0
            line.append(f"streamed/{n_waves} {timeit(run_hot):6.1f}")
        for n_waves in (4096, 8192, 16384):
            stream, wptr, wstep, desc, total, n_items, per_xcd = build_cold(indptr, indices, hot, n, n_waves)
            d = [torch.from_numpy(a).to(DEV) for a in (stream, wptr, wstep, desc)]
            part = torch.zeros(n * 8, 8, device=DEV)

            def run_cold():
                st = lib.cold_gather(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), x.data_ptr(), n * 32,
                                     part.data_ptr(), n_waves, sh)
                assert st == 0
            run_cold()
            torch.cuda.synchronize()
            got = y_hot + part.cpu().double().numpy().reshape(n, 8, 8).sum(1)
            err = float(np.abs(got - ref).max())
            assert err < 1e-3, err
            line.append(f"| cold/{n_waves} {timeit(run_cold):6.1f} us ({n_items} items, {total * 32 / max(1, int((~keep).sum())):.2f} slots/edge, "
                        f"xcd max/mean {per_xcd.max() / per_xcd.mean():.2f})")
        print(" ".join(line), flush=True)


if __name__ == "__main__":
    main()
